#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own modules on CPU.

Run once in the build container (the reference never travels to the GPU box):

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference:/root/repo \
        python3 tests/golden/make_golden.py

Only data is written (inputs are regenerated from seeds by tests/synth.py; outputs are stored by
value).  The one reference data file taken along is the smallest zoo checkpoint
(model/segmentation/model_zoo/espnetv2/espnetv2_s_0.5_city_512x256.pth, weights only), stored as
an .npz so the real-weights case can run without /root/reference.
"""
import argparse
import ast
import json
import os
import sys

import numpy as np
import torch

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from tests.cases import ASPP_CASES, ESPDNET_CASES, EVAL_CASES, IMAGEIO_CASES, LABEL_LOOP_CASES, LAYER_CASES, LR_CASES, NID_CASES, SUPERVISED_CASE, MODEL_CASES, RGBD_CASES, TRAIN_CASE, TRAIN_CASES  # noqa: E402
from tests.synth import grad_sample_index, synth_adversarial_logits, synth_eval_batches, synth_image_u8, synth_input, synth_label_loop_images, synth_nid_inputs, synth_labels, synth_state_dict  # noqa: E402

# reference imports (torch-only modules)
from nn_layers.eesp import EESP, DownSampler  # noqa: E402
from nn_layers.efficient_pyramid_pool import EfficientPyrPool  # noqa: E402
from nn_layers.efficient_pt import EfficientPWConv  # noqa: E402
from nn_layers import aspp as ref_aspp  # noqa: E402
from model.segmentation.espdnet_ue import ESPDNetwithUncertaintyEstimation  # noqa: E402
from model.segmentation.espnetv2 import ESPNetv2Segmentation  # noqa: E402
from model.segmentation.espdnet import ESPDNetSegmentation  # noqa: E402
from loss_fns.segmentation_loss import NIDLoss, PixelwiseKLD, SegmentationLoss, UncertaintyWeightedSegmentationLoss  # noqa: E402

torch.set_num_threads(8)


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **{k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                                 for k, v in arrays.items()})
    print('wrote %s (%.1f KiB)' % (path, os.path.getsize(path) / 1024))


def build_layer(kind, kw):
    if kind == 'eesp':
        return EESP(**kw)
    if kind == 'down':
        return DownSampler(**kw)
    if kind == 'pyr':
        return EfficientPyrPool(scales=[2.0, 1.5, 1.0, 0.5, 0.1], **kw)
    if kind == 'pw':
        return EfficientPWConv(**kw)
    raise KeyError(kind)


def build_model(kind, s, classes, dataset):
    a = argparse.Namespace(s=s, channels=3, num_classes=1000)
    if kind == 'espdnetue':
        return ESPDNetwithUncertaintyEstimation(a, classes=classes, dataset=dataset, fix_pyr_plane_proj=True)
    return ESPNetv2Segmentation(a, classes=classes, dataset=dataset)


def gen_layers():
    out = {}
    keys = {}
    for i, (name, (kind, kw, shp, shp2)) in enumerate(sorted(LAYER_CASES.items())):
        m = build_layer(kind, kw).eval()
        keys[name] = {k: list(v.shape) for k, v in m.state_dict().items()}
        m.load_state_dict(synth_state_dict(m.state_dict(), 100 + i))
        x = synth_input(shp, 200 + i)
        with torch.no_grad():
            y = m(x, synth_input(shp2, 300 + i)) if shp2 is not None else m(x)
        out[name] = y
    save('layers', **out)
    with open(os.path.join(HERE, 'layer_keys.json'), 'w') as f:
        json.dump(keys, f, sort_keys=True)


def gen_models():
    keysets = {}
    for name, (kind, s, classes, dataset, shp, sd_seed, in_seed) in sorted(MODEL_CASES.items()):
        m = build_model(kind, s, classes, dataset).eval()
        sd = m.state_dict()
        keysets['%s_s%s_c%d' % (kind, s, classes)] = {k: list(v.shape) for k, v in sd.items()}
        m.load_state_dict(synth_state_dict(sd, sd_seed))
        x = synth_input(shp, in_seed)
        with torch.no_grad():
            y = m(x)
        if kind == 'espdnetue':
            main, aux = y
            kld = PixelwiseKLD()(main, aux)
            prob = torch.softmax(main + 0.5 * aux, 1)
            amax = np.argmax(prob.numpy().transpose(0, 2, 3, 1), axis=3).astype(np.uint8)
            srt = torch.sort(prob, dim=1, descending=True)[0]
            margin = (srt[:, 0] - srt[:, 1])
            if shp[2] * shp[3] > 64 * 64:
                st = 8
                save('model_' + name, main=main[:, :, ::st, ::st], aux=aux[:, :, ::st, ::st],
                     kld=kld[:, ::st, ::st], amax=amax, margin=margin.half(), stride=st)
            else:
                save('model_' + name, main=main, aux=aux, kld=kld, amax=amax, margin=margin.half(), stride=1)
        else:
            save('model_' + name, main=y, stride=1)
    with open(os.path.join(HERE, 'state_dict_keys.json'), 'w') as f:
        json.dump(keysets, f, sort_keys=True)
    print('param counts:', {k: sum(int(np.prod(s)) for kk, s in v.items()
                                   if not kk.endswith(('running_mean', 'running_var', 'num_batches_tracked')))
                            for k, v in keysets.items()})


class _cuda_means_here(object):
    """nn_layers/fusion_gate.py:38 builds its ones tensor with a hard-coded `.to('cuda')`, which cannot run in this
    GPU-less container: inside this context Tensor.to treats a 'cuda' target as "stay where you are" -- the reference's
    arithmetic is untouched."""

    def __enter__(self):
        real_to = self.real_to = torch.Tensor.to

        def cpu_to(t, *a, **k):
            if a and isinstance(a[0], str) and a[0].startswith('cuda'):
                return t
            return real_to(t, *a, **k)
        torch.Tensor.to = cpu_to

    def __exit__(self, *exc):
        torch.Tensor.to = self.real_to


def gen_rgbd():
    """ESPDNet-UE with a depth image (x_d), and the single-head ESPDNetSegmentation with and without one."""
    out = {}
    for name, (classes, dataset, shp, sd_seed, in_seed, d_seed, dense, trainable) in sorted(RGBD_CASES.items()):
        a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
        m = ESPDNetwithUncertaintyEstimation(a, classes=classes, dataset=dataset, dense_fuse=dense,
                                             trainable_fusion=trainable, fix_pyr_plane_proj=True).eval()
        m.load_state_dict(synth_state_dict(m.state_dict(), sd_seed))
        x = synth_input(shp, in_seed)
        x_d = synth_input((shp[0], 1) + tuple(shp[2:]), d_seed)
        with _cuda_means_here(), torch.no_grad():
            main, aux = m(x, x_d)
        out[name + '.main'], out[name + '.aux'] = main, aux
    keys = {}
    for name, (classes, dataset, shp, sd_seed, in_seed, d_seed, dense, trainable) in sorted(ESPDNET_CASES.items()):
        a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
        m = ESPDNetSegmentation(a, classes=classes, dataset=dataset, dense_fuse=dense, trainable_fusion=trainable).eval()
        keys[name] = {k: list(v.shape) for k, v in m.state_dict().items()}
        m.load_state_dict(synth_state_dict(m.state_dict(), sd_seed))
        x = synth_input(shp, in_seed)
        x_d = None if d_seed is None else synth_input((shp[0], 1) + tuple(shp[2:]), d_seed)
        with _cuda_means_here(), torch.no_grad():
            out[name] = m(x, x_d)
    save('rgbd', **out)
    with open(os.path.join(HERE, 'espdnet_keys.json'), 'w') as f:
        json.dump(keys, f, sort_keys=True)


def gen_zoo():
    """Real weights: ESPNetv2 s=0.5, Cityscapes 512x256 checkpoint (strict load), BASELINE config 1 shape."""
    src = os.path.join(REF, 'model/segmentation/model_zoo/espnetv2/espnetv2_s_0.5_city_512x256.pth')
    sd = torch.load(src, map_location='cpu')
    m = build_model('espnetv2', 0.5, 20, 'city').eval()
    m.load_state_dict(sd, strict=True)
    np.savez_compressed(os.path.join(HERE, 'zoo_espnetv2_s0.5_city_512x256.npz'),
                        **{k: v.numpy() for k, v in sd.items()})
    x = synth_input((2, 3, 288, 480), 40)
    with torch.no_grad():
        y = m(x)
    amax = y.argmax(1).to(torch.uint8)
    srt = torch.sort(y, dim=1, descending=True)[0]
    save('model_v2_zoo_288x480', main=y[:, :, ::8, ::8], amax=amax, margin=(srt[:, 0] - srt[:, 1]).half(),
         class_sum=y.double().sum((0, 2, 3)), abs_sum=y.double().abs().sum(), stride=8)


def extract_functions(path, names, ns):
    """AST-extract pure functions from a script that cannot be imported (argparse at import time)."""
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module([node], []), path, 'exec'), ns)
    return ns


def gen_labels():
    ns = {'np': np, 'args': argparse.Namespace(classes=5)}
    extract_functions(os.path.join(REF, 'uest_seg_multi_os.py'), {'merge_outputs'}, ns)
    merge = ns['merge_outputs']
    out = {}
    # exhaustive truth tables: S sources x 5 classes, every policy the CLI can produce (strings) + None
    for S in (1, 2, 3, 4):
        grids = np.stack(np.meshgrid(*[np.arange(5)] * S, indexing='ij')).reshape(S, -1)
        out['tt_in_S%d' % S] = grids.astype(np.uint8)
        for pol in ('all', 'half', 'none'):
            out['tt_S%d_%s' % (S, pol)] = merge(grids, 5, None if pol == 'none' else pol).astype(np.uint8)
    rng = np.random.RandomState(7)
    rnd = rng.randint(0, 5, size=(3, 64, 96)).astype(np.uint8)
    out['rnd_in'] = rnd
    out['rnd_all'] = merge(rnd, 5, 'all').astype(np.uint8)
    out['rnd_half'] = merge(rnd, 5, 'half').astype(np.uint8)
    # LUTs as data (data_loader/segmentation/greenhouse.py:15-58 are literal arrays; file not importable)
    tree = ast.parse(open(os.path.join(REF, 'data_loader/segmentation/greenhouse.py')).read())
    lns = {'np': np}
    for node in tree.body:
        if isinstance(node, ast.Assign) and getattr(node.targets[0], 'id', '').startswith('id_'):
            exec(compile(ast.Module([node], []), 'greenhouse', 'exec'), lns)
    for k in ('id_camvid_to_greenhouse', 'id_cityscapes_to_greenhouse', 'id_forest_to_greenhouse'):
        out['lut_' + k] = lns[k].astype(np.int64)
    # uncertainty estimator on small logits
    for C in (5, 13, 20):
        d1 = synth_input((2, C, 12, 20), 50 + C) * 3
        d2 = synth_input((2, C, 12, 20), 70 + C) * 3
        out['kld_C%d' % C] = PixelwiseKLD()(d1, d2)
        out['prob_C%d' % C] = torch.nn.Softmax2d()(d1 + 0.5 * d2)
    save('labels', **out)


def gen_loss():
    pred = (synth_input((2, 5, 32, 48), 90) * 2).requires_grad_(True)
    aux = (synth_input((2, 5, 32, 48), 91) * 2).requires_grad_(True)
    tgt = synth_labels((2, 32, 48), 5, 92)
    cw = torch.tensor([0.0, 6.31, 3.78, 3.18, 7.64])
    crit = UncertaintyWeightedSegmentationLoss(5, class_weights=cw.clone(), ignore_idx=4, device='cpu')
    kld = PixelwiseKLD()(pred, aux)
    loss = crit(pred + 0.5 * aux, tgt, kld) * 20 + kld.mean()
    loss.backward()
    torch.autograd.set_detect_anomaly(False)
    save('loss', loss=loss.detach(), dpred=pred.grad, daux=aux.grad, cw=cw)


def gen_train():
    for gname, c in sorted(TRAIN_CASES.items()):
        _gen_train_case(gname, c)


def _gen_train_case(gname, c):
    m = build_model('espdnetue', c['s'], c['classes'], c['dataset']).eval()  # eval: uest default (Appendix B-3)
    m.load_state_dict(synth_state_dict(m.state_dict(), c['sd_seed']))
    x = synth_input(c['shape'], c['in_seed'])
    labels = synth_labels((c['shape'][0],) + c['shape'][2:], c['classes'], c['in_seed'])
    opt = torch.optim.Adam(m.parameters(), lr=c['lr'], weight_decay=c['weight_decay'])
    crit = UncertaintyWeightedSegmentationLoss(c['classes'], class_weights=torch.ones(c['classes']),
                                               ignore_idx=c['ignore_idx'], device='cpu')
    opt.zero_grad()
    pred, aux = m(x)
    kld = PixelwiseKLD()(pred, aux)
    loss = crit(pred + 0.5 * aux, labels, kld) * 20 + kld.mean()
    loss.backward()
    torch.autograd.set_detect_anomaly(False)
    names = [n for n, _ in m.named_parameters()]
    gnorm = np.array([float(p.grad.double().norm()) if p.grad is not None else -1.0 for _, p in m.named_parameters()])
    gsum = np.array([float(p.grad.double().sum()) if p.grad is not None else 0.0 for _, p in m.named_parameters()])
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    opt.step()
    delta = np.array([float((p.detach() - before[n]).double().norm()) for n, p in m.named_parameters()])
    keep = ['base_net.level1.conv.weight', 'base_net.level4.6.spp_dw.3.conv.weight', 'bu_dec_l4.merge_layer.3.bias',
            'depth_base_net.level3.1.proj_1x1.act.weight', 'aux_decoder.stages.0.weight',
            'merge_enc_dec_l3.wt_layer.1.weight', 'bu_br_l3.0.weight', 'base_net.level2_0.inp_reinf.1.bn.bias']
    pd = dict(m.named_parameters())
    # a strided sample of EVERY gradient (up to ~64 elements per tensor, first and last element included): norms alone would pass a
    # sign or permutation error inside a tensor
    gs_val, gs_off = [], [0]
    for _, p_ in m.named_parameters():
        if p_.grad is not None:
            flat = p_.grad.detach().reshape(-1)
            idx = grad_sample_index(flat.numel())
            gs_val.append(flat[idx].numpy())
        gs_off.append(gs_off[-1] + (len(gs_val[-1]) if p_.grad is not None else 0))
    save(gname, loss=loss.detach(), names=np.array(names), gnorm=gnorm, gsum=gsum, delta=delta,
         gsample=np.concatenate(gs_val), gsample_off=np.array(gs_off, dtype=np.int64),
         keep=np.array(keep), **{'after_%d' % i: pd[k].detach() for i, k in enumerate(keep)})


def gen_aspp():
    """ASPP / ASPP_Bottleneck logits (eval mode, randomised BN statistics and biases) + their state-dict key tables."""
    out, keys = {}, {}
    for name, (cls, ncls, shp, sd_seed, x_seed) in sorted(ASPP_CASES.items()):
        m = getattr(ref_aspp, cls)(num_classes=ncls).eval()
        keys[name] = {k: list(v.shape) for k, v in m.state_dict().items()}
        m.load_state_dict(synth_state_dict(m.state_dict(), sd_seed))
        with torch.no_grad():
            out[name] = m(synth_input(shp, x_seed))
    save('aspp', **out)
    with open(os.path.join(HERE, 'aspp_keys.json'), 'w') as f:
        json.dump(keys, f, sort_keys=True)


def gen_supervised():
    """One iteration of the supervised loop with the reference's modules: the body of train_seg_ue
    (utilities/train_eval_seg.py:179-225; the file itself imports tensorboard-era helpers and is restated here line by
    line) on ESPDNet-UE in train() mode, SegmentationLoss (CrossEntropy), flooding, torch.optim.SGD over the two
    learning-rate groups of train_segmentation.py:248-253.  Also the epoch-wise schedules of utilities/lr_scheduler.py."""
    import copy
    from utilities import lr_scheduler as ref_lr
    c = SUPERVISED_CASE
    m = build_model('espdnetue', c['s'], c['classes'], c['dataset'])
    m.load_state_dict(synth_state_dict(m.state_dict(), c['sd_seed']))
    m.train()
    x = synth_input(c['shape'], c['in_seed'])
    target = synth_labels((c['shape'][0],) + c['shape'][2:], c['classes'], c['in_seed'])
    crit = SegmentationLoss(n_classes=c['classes'], device='cpu', ignore_idx=c['ignore_idx'], class_weights=None)
    groups = [{'params': m.get_basenet_params(), 'lr': c['lr']},
              {'params': m.get_segment_params(), 'lr': c['lr'] * c['lr_mult']}]
    opt = torch.optim.SGD(groups, lr=c['lr'] * c['lr_mult'], momentum=c['momentum'], weight_decay=c['weight_decay'])
    outputs = m(x)
    kld = PixelwiseKLD()(outputs[0], outputs[1])  # noqa: F841  (computed and unused, :197)
    out = outputs[0] + 0.5 * outputs[1]
    loss = crit(out, target).mean()
    b = c['flood']
    loss = (loss - b).abs() + b
    opt.zero_grad()
    loss.backward()
    names = [n for n, _ in m.named_parameters()]
    gnorm = np.array([float(p.grad.double().norm()) if p.grad is not None else -1.0 for _, p in m.named_parameters()])
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    opt.step()
    moved = np.array([bool((p.detach() != before[n]).any()) for n, p in m.named_parameters()])
    keep = ['base_net.level1.conv.weight', 'base_net.level1.bn.weight', 'base_net.level4.6.spp_dw.3.conv.weight',
            'bu_dec_l4.merge_layer.3.bias', 'bu_dec_l2.stages.1.weight', 'merge_enc_dec_l3.wt_layer.1.weight', 'bu_br_l3.0.bias',
            'base_net.level3_0.eesp.conv_1x1_exp.bn.weight']
    stats = ['base_net.level1.bn.running_mean', 'base_net.level1.bn.running_var', 'base_net.level4.3.br_after_cat.bn.running_var',
             'bu_br_l2.0.running_mean', 'bu_dec_l3.projection_layer.cbr.1.running_var', 'depth_base_net.level3.1.proj_1x1.bn.running_mean',
             'aux_decoder.merge_layer.0.br.0.running_var', 'base_net.level1.bn.num_batches_tracked']
    pd, sd = dict(m.named_parameters()), m.state_dict()
    save('supervised_step', loss=loss.detach(), logits=out.detach()[:, :, ::4, ::4], names=np.array(names), gnorm=gnorm, moved=moved,
         keep=np.array(keep), stats=np.array(stats), **{'after_%d' % i: pd[k].detach() for i, k in enumerate(keep)},
         **{'stat_%d' % i: sd[k] for i, k in enumerate(stats)})
    tables = []
    for name, kw, epochs in LR_CASES:
        sch = getattr(ref_lr, name)(**copy.deepcopy(kw))
        tables.append([sch.step(e) for e in range(epochs)])
    with open(os.path.join(HERE, 'lr_schedules.json'), 'w') as f:
        json.dump(tables, f)


def gen_nid():
    """NIDLoss value and gradient w.r.t. the label logits (its `.to('cuda')` calls mapped to "stay here", see
    _cuda_means_here)."""
    out = {}
    for name, (shape, classes, K, seed) in sorted(NID_CASES.items()):
        cam, lab = synth_nid_inputs(shape, classes, seed)
        lab.requires_grad_()
        with _cuda_means_here():
            loss = NIDLoss(image_bin=K, label_bin=classes)(cam, lab)
            loss.backward()
        out[name + '.loss'] = loss.detach()
        out[name + '.grad'] = lab.grad
    save('nid', **out)


def gen_imageio():
    """Loader transforms: Pillow's own resize (PIL is importable here) followed by what torchvision's to_tensor / normalize
    do (torchvision itself is absent: `pic.permute(2,0,1).float().div(255)`, then `sub_(mean).div_(std)` on CPU) -- the
    val_transforms of data_loader/segmentation/greenhouse.py:216-222.  Outputs are stored as SHA-256 of the exact bytes
    (bit-exact contract) plus a strided sample for diagnosis."""
    import hashlib
    from PIL import Image
    mean = torch.tensor([0.485, 0.456, 0.406])[:, None, None]
    std = torch.tensor([0.229, 0.224, 0.225])[:, None, None]
    out = {}
    for name, (hs, ws, size, seed, norm, flip, with_depth) in sorted(IMAGEIO_CASES.items()):
        rgb, label, depth = synth_image_u8(hs, ws, seed)
        r = Image.fromarray(rgb).resize(size, Image.BILINEAR)
        lab = Image.fromarray(label).resize(size, Image.NEAREST)
        d = Image.fromarray(depth).resize(size, Image.BILINEAR)
        if flip:
            r, lab, d = (im.transpose(Image.FLIP_LEFT_RIGHT) for im in (r, lab, d))
        t = torch.from_numpy(np.asarray(r).copy()).permute(2, 0, 1).float().div(255)
        if norm:
            t = t.sub_(mean).div_(std)
        lt = torch.LongTensor(np.array(lab).astype(np.int64))
        dt = torch.from_numpy(np.asarray(d).copy())[None].float().div(255)
        out[name + '.rgb_sha'] = np.frombuffer(hashlib.sha256(t.contiguous().numpy().tobytes()).digest(), np.uint8)
        out[name + '.label_sha'] = np.frombuffer(hashlib.sha256(lt.numpy().tobytes()).digest(), np.uint8)
        out[name + '.rgb_s'] = t[:, ::7, ::5]
        out[name + '.label_s'] = lt[::7, ::5].to(torch.uint8)
        if with_depth:
            out[name + '.depth_sha'] = np.frombuffer(hashlib.sha256(dt.contiguous().numpy().tobytes()).digest(), np.uint8)
            out[name + '.depth_s'] = dt[:, ::7, ::5]
    save('imageio', **out)


def gen_eval():
    """val_seg_ue (utilities/train_eval_seg.py:249-324), AST-extracted and run with the reference's own model, loss and MIOU classes on
    a seeded loader; and the body of test() (uest_seg_multi_os.py:1150-1200): the main head alone through the same criterion / MIOU."""
    from utilities.metrics.segmentation_miou import MIOU
    from utilities import print_utils
    from collections import OrderedDict
    import time
    ns = {'torch': torch, 'np': np, 'time': time, 'MIOU': MIOU, 'OrderedDict': OrderedDict, 'gather': None}
    ns.update({k: getattr(print_utils, k) for k in dir(print_utils) if k.startswith('print_')})
    tree = ast.parse(open(os.path.join(REF, 'utilities/utils.py')).read())
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == 'AverageMeter':
            exec(compile(ast.Module([node], []), 'utils', 'exec'), ns)
    extract_functions(os.path.join(REF, 'utilities/train_eval_seg.py'), {'val_seg_ue'}, ns)
    out = {}
    for name, case in sorted(EVAL_CASES.items()):
        C, ds, shape, nb, sd_seed, in_seed, ign, cw_seed, with_void = case
        m = build_model('espdnetue', 2.0, C, ds).eval()
        m.load_state_dict(synth_state_dict(m.state_dict(), sd_seed))
        cw = torch.rand(C, generator=torch.Generator().manual_seed(cw_seed)) + 0.5
        crit = SegmentationLoss(n_classes=C, device='cpu', ignore_idx=ign, class_weights=cw)
        loader = synth_eval_batches(case)
        iou, loss = ns['val_seg_ue'](m, loader, criterion=crit, num_classes=C, device='cpu')
        out[name + '.iou'] = np.asarray(iou, dtype=np.float64)
        out[name + '.loss'] = np.float64(loss)
        out[name + '.cw'] = cw
        # test() of the uest script: pred alone (":1196-1198": criterion(pred, labels); get_iou(pred, labels))
        AverageMeter = ns['AverageMeter']
        inter_m, union_m, losses = AverageMeter(), AverageMeter(), AverageMeter()
        miou_class = MIOU(num_classes=C - 1)
        with torch.no_grad():
            for x, y in loader:
                pred, _ = m(x)
                l = crit(pred, y)
                inter, union = miou_class.get_iou(pred, y)
                inter_m.update(inter); union_m.update(union); losses.update(l.item(), x.size(0))
        out[name + '.test_iou'] = np.asarray(inter_m.sum / (union_m.sum + 1e-10), dtype=np.float64)
        out[name + '.test_loss'] = np.float64(losses.avg)
        out[name + '.inter'] = np.asarray(inter_m.sum, dtype=np.float64)
        out[name + '.union'] = np.asarray(union_m.sum, dtype=np.float64)
    save('eval', **out)


def gen_label_loops():
    """The two relabelling loops as WHOLE functions -- generate_pseudo_label (uest_seg_multi_os.py:730-830) and
    generate_pseudo_label_multi_model (:832-956) -- AST-extracted together with the script functions they call (get_output :669-693,
    merge_outputs :695-718, update_image_list :720-728, ScoreUpdater :1257-) and run with the reference's own model class and
    PixelwiseKLD on a stub dataset that serves seeded tensors (the real GreenhouseRGBDSegmentation needs image files and cv2).  What
    is real: the loop, torch's DataLoader (batch size 1, string collation), PIL's PNG writer, the list writer, the class-weight rule.
    Stored: the list file's lines (save path as {SAVE}), the DECODED label files, the class weights, and per pixel the smallest top-2
    probability margin over the models (so a test knows where two fp32 forwards may legitimately disagree)."""
    import tempfile
    import time
    import types
    from collections import OrderedDict
    import os.path as osp
    from packaging import version
    from PIL import Image
    from torch.utils import data
    from torch import nn

    class Quiet(object):                       # tqdm(total=...) as a context manager and tqdm(iterable)
        def __init__(self, it=None, total=None):
            self.it = it
        def __iter__(self):
            return iter(self.it)
        def __enter__(self):
            return self
        def __exit__(self, *exc):
            return False
        def close(self):
            pass

    class Log(object):
        def info(self, *a):
            pass

    lns = {'np': np}
    tree = ast.parse(open(os.path.join(REF, 'data_loader/segmentation/greenhouse.py')).read())
    for node in tree.body:
        if isinstance(node, ast.Assign) and getattr(node.targets[0], 'id', '').startswith('id_'):
            exec(compile(ast.Module([node], []), 'greenhouse', 'exec'), lns)
    ns = {'np': np, 'torch': torch, 'nn': nn, 'data': data, 'version': version, 'time': time, 'osp': osp, 'Image': Image,
          'tqdm': Quiet, 'OrderedDict': OrderedDict, 'PixelwiseKLD': PixelwiseKLD}
    ns.update({k: v for k, v in lns.items() if k.startswith('id_')})
    path = os.path.join(REF, 'uest_seg_multi_os.py')
    extract_functions(path, {'get_output', 'merge_outputs', 'update_image_list', 'generate_pseudo_label',
                             'generate_pseudo_label_multi_model'}, ns)
    for node in ast.parse(open(path).read()).body:
        if isinstance(node, ast.ClassDef) and node.name == 'ScoreUpdater':
            exec(compile(ast.Module([node], []), path, 'exec'), ns)
    out, meta = {}, {}
    real_mod = sys.modules.get('data_loader.segmentation.greenhouse')
    for name, case in sorted(LABEL_LOOP_CASES.items()):
        specs, (H, W), n, in_seed, policy, weighting = case[:6]
        eval_training = len(case) > 6 and case[6]
        items = synth_label_loop_images(case)

        class StubDataset(data.Dataset):      # train=False item of GreenhouseRGBDSegmentation without depth (greenhouse.py:270)
            def __init__(self, **kw):
                assert kw.get('train') is False and not kw.get('use_depth')
            def __len__(self):
                return len(items)
            def __getitem__(self, i):
                return items[i][0], torch.zeros(H, W, dtype=torch.int64), items[i][1], 1.0

        stub = types.ModuleType('data_loader.segmentation.greenhouse')
        stub.GreenhouseRGBDSegmentation = StubDataset
        sys.modules['data_loader.segmentation.greenhouse'] = stub
        ms = []
        for C, ds, os_data, sd_seed in specs:
            m = build_model('espdnetue', 2.0, C, ds).eval()
            m.load_state_dict(synth_state_dict(m.state_dict(), sd_seed))
            ms.append(m)
        args = argparse.Namespace(classes=5, test_image_size='%d,%d' % (H, W), eval_scale=1.0, dataset='greenhouse',
                                  data_tgt_train_list='unused.lst', use_traversable=False, use_depth=False, pin_memory=False,
                                  eval_training=eval_training, merge_label_policy=policy, class_weighting=weighting)
        ns['args'] = args
        save_path = tempfile.mkdtemp()
        os.makedirs(os.path.join(save_path, 'pred'))
        try:
            with _cuda_means_here(), torch.no_grad():        # get_output's default device is the literal 'cuda' (:669)
                if specs[0][2] is None:
                    lst, cw = ns['generate_pseudo_label'](ms[0], 'cpu', save_path, 0, n, None, None, args, Log(), None, None)
                else:
                    lst, cw = ns['generate_pseudo_label_multi_model'](ms, [s[2] for s in specs], 'cpu', save_path, 0, n, None, None,
                                                                       args, Log(), None, None)
                margin = np.full((n, H, W), np.inf, dtype=np.float32)
                for i, (x, _) in enumerate(items):
                    for m in ms:                      # (still in the mode the function left them in: train() for eval_training)
                        prob, _ = ns['get_output'](m, x[None])
                        top = np.sort(prob, axis=0)
                        margin[i] = np.minimum(margin[i], top[-1] - top[-2])
        finally:
            if real_mod is None:
                del sys.modules['data_loader.segmentation.greenhouse']
            else:
                sys.modules['data_loader.segmentation.greenhouse'] = real_mod
        lines = open(lst).read().replace(save_path, '{SAVE}').splitlines()
        maps = np.stack([np.asarray(Image.open(ln.split(',')[1].replace('{SAVE}', save_path))) for ln in lines])
        assert maps.dtype == np.uint8 and maps.shape == (n, H, W) and osp.basename(lst) == 'tgt_train.lst'
        out[name + '.maps'] = maps
        out[name + '.margin'] = margin
        out[name + '.class_weights'] = cw.numpy()
        meta[name] = lines
        print(name, 'weights', cw.numpy(), 'pixels with margin < 1e-4:', int((margin < 1e-4).sum()))
    save('label_loops', **out)
    with open(os.path.join(HERE, 'label_loops.json'), 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)


def gen_argmax_adversarial():
    """The reference's label rule -- get_output (uest_seg_multi_os.py:669-693, AST-extracted, fed through a stub "model" that returns
    the prepared heads) followed by np.argmax over the class axis (:797-798) -- on logits whose two largest entries are 0..4 ulp apart
    (tests.synth.synth_adversarial_logits).  Stored: the reference's class per pixel and the probabilities it gave the two candidates."""
    from collections import OrderedDict
    from torch import nn
    ns = {'np': np, 'torch': torch, 'nn': nn, 'OrderedDict': OrderedDict, 'PixelwiseKLD': PixelwiseKLD,
          'args': argparse.Namespace(use_depth=False)}
    extract_functions(os.path.join(REF, 'uest_seg_multi_os.py'), {'get_output'}, ns)
    out = {}
    for C in (5, 13, 20):
        pred, aux, a, b, k = synth_adversarial_logits(C, C)
        amax = np.zeros(a.shape, dtype=np.uint8)
        pa = np.zeros(a.shape, dtype=np.float32)
        pb = np.zeros(a.shape, dtype=np.float32)
        with _cuda_means_here(), torch.no_grad():
            for i in range(pred.shape[0]):
                output, _ = ns['get_output'](lambda image, i=i: (pred[i:i + 1], aux[i:i + 1]), torch.zeros(1))
                amax[i] = np.asarray(np.argmax(output.transpose(1, 2, 0), axis=2), dtype=np.uint8)
                pa[i] = np.take_along_axis(output, a[i][None], 0)[0]
                pb[i] = np.take_along_axis(output, b[i][None], 0)[0]
        z = (pred + 0.5 * aux).numpy()
        print('C=%d: reference class differs from argmax of the logits on %d of %d adversarial pixels; candidates tied in probability: %d'
              % (C, int((amax != z.argmax(1)).sum()), amax.size, int((pa == pb).sum())))
        out['C%d.amax' % C] = amax
        out['C%d.pa' % C] = pa
        out['C%d.pb' % C] = pb
    save('argmax_adversarial', **out)


if __name__ == '__main__':
    which = sys.argv[1:] or ['layers', 'models', 'zoo', 'labels', 'loss', 'train', 'aspp', 'rgbd', 'imageio', 'supervised', 'nid', 'eval', 'label_loops', 'argmax_adversarial']
    for w in which:
        globals()['gen_' + w]()
