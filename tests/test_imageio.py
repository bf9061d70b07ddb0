"""Loader transforms, label PNG writer and list round trip.

CPU: the oracle's restatement of Pillow's resampling / torchvision's to_tensor+normalize against vectors produced by Pillow
itself (tests/golden/make_golden.py::gen_imageio), the PNG writer against PIL's decoder and the oracle's own decoder, the list
file round trip.  GPU: the batched device transforms (through the C ABI) bit-exact against the same vectors and the oracle;
the asynchronous writer end to end.
"""
import hashlib
import io
import os

import numpy as np
import pytest
import torch

from oracle import imageio as oio
from tests.cases import IMAGEIO_CASES
from tests.synth import synth_image_u8


def _sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


@pytest.mark.parametrize('name', sorted(IMAGEIO_CASES))
def test_oracle_transforms_vs_pillow_golden(name, golden):
    hs, ws, size, seed, norm, flip, with_depth = IMAGEIO_CASES[name]
    rgb, label, depth = synth_image_u8(hs, ws, seed)
    t, lt, dt = oio.val_transform(rgb, label, depth if with_depth else None, size=size, normalise=norm, flip=flip)
    g = golden('imageio')
    assert np.array_equal(t[:, ::7, ::5], g[name + '.rgb_s'])
    assert np.array_equal(_sha(t), g[name + '.rgb_sha'])
    assert np.array_equal(_sha(lt), g[name + '.label_sha'])
    if with_depth:
        assert np.array_equal(_sha(dt), g[name + '.depth_sha'])


def test_oracle_resample_vs_live_pillow():
    """Extra sizes, checked against the Pillow installed here (skipped where PIL is missing)."""
    Image = pytest.importorskip('PIL.Image')
    rng = np.random.default_rng(3)
    for hs, ws, size in [(61, 47, (31, 90)), (300, 400, (133, 77)), (90, 160, (160, 90)), (256, 480, (480, 256))]:
        img = rng.integers(0, 256, (hs, ws, 3), dtype=np.uint8)
        lab = rng.integers(0, 6, (hs, ws), dtype=np.uint8)
        assert np.array_equal(oio.resize_bilinear_u8(img, size), np.asarray(Image.fromarray(img).resize(size, Image.BILINEAR)))
        assert np.array_equal(oio.resize_nearest_u8(lab, size), np.asarray(Image.fromarray(lab).resize(size, Image.NEAREST)))


def test_host_tables_match_oracle():
    """The C ABI's host-side coefficient builders (the only part of the transforms that runs without a GPU)."""
    from mspl_amd._native import check, lib
    for n_in, n_out in [(360, 256), (360, 288), (720, 256), (100, 256), (37, 48), (53, 64), (256, 256), (2048, 1024), (5, 7)]:
        k = lib.mspl_resample_ksize(n_in, n_out)
        bounds, kk = np.zeros((n_out, 2), np.int32), np.zeros((n_out, k), np.int32)
        check(lib.mspl_resample_coeffs(n_in, n_out, bounds.ctypes.data, kk.ctypes.data))
        ob, ok = oio.precompute_coeffs(n_in, n_out)
        assert k == ok.shape[1] and np.array_equal(bounds, ob) and np.array_equal(kk, ok), (n_in, n_out)
        idx = np.zeros(n_out, np.int32)
        check(lib.mspl_nearest_index(n_in, n_out, idx.ctypes.data))
        assert np.array_equal(idx, oio.nearest_index(n_in, n_out)), (n_in, n_out)
    assert lib.mspl_resample_ksize(0, 4) < 0


def test_png_writer_round_trip():
    from mspl_amd.io import encode_png_gray8
    rng = np.random.default_rng(7)
    cases = [np.repeat(np.repeat(rng.integers(0, 5, (16, 30), dtype=np.uint8), 16, 0), 16, 1),      # blocky label map
             rng.integers(0, 256, (37, 53), dtype=np.uint8),                                         # noise, odd size
             np.full((1, 1), 4, np.uint8), np.zeros((3, 500), np.uint8)]
    for a in cases:
        data = encode_png_gray8(a)
        assert np.array_equal(oio.png_decode_gray8(data), a)
        try:
            from PIL import Image
        except ImportError:
            continue
        im = Image.open(io.BytesIO(data))
        assert im.mode == 'L' and im.size == (a.shape[1], a.shape[0]) and np.array_equal(np.asarray(im), a)
    with pytest.raises(ValueError):
        encode_png_gray8(np.zeros((4, 4), np.int64))


def test_native_label_writer_host_tensors(tmp_path):
    """The native writer (C++ threads + zlib inside the library) with host tensors: no GPU involved; files decode to the maps,
    back-pressure with a single staging slot, error reporting for an unwritable directory."""
    from mspl_amd.io import LabelWriter
    rng = np.random.default_rng(5)
    maps = torch.from_numpy(np.repeat(np.repeat(rng.integers(0, 5, (4, 9, 13), dtype=np.uint8), 5, 1), 5, 2))
    with LabelWriter(str(tmp_path / 'pred'), workers=3, max_inflight=1) as w:
        for b in range(6):
            w.submit(['/data/color/f%d_%d.png' % (b, i) for i in range(4)], maps)
    assert len(w.label_paths) == 24 and w.label_paths[5] == '%s/f1_1.png' % (tmp_path / 'pred')
    for i, p in enumerate(w.label_paths):
        assert np.array_equal(oio.png_decode_gray8(open(p, 'rb').read()), maps[i % 4].numpy())
    w2 = LabelWriter(str(tmp_path / 'gone'), workers=1)
    os.rmdir(str(tmp_path / 'gone'))
    w2.submit(['a.png'], maps[:1])
    with pytest.raises(RuntimeError, match='file write'):
        w2.close()
    with pytest.raises(RuntimeError, match='closed'):
        w2.submit(['b.png'], maps[:1])


def test_image_list_round_trip(tmp_path):
    from mspl_amd.io import read_image_list, update_image_list
    files = []
    for n in ('a_color.png', 'a_label.png', 'a_depth.png', 'b_color.png', 'b_label.png', 'b_depth.png'):
        p = tmp_path / n
        p.write_bytes(b'x')
        files.append(str(p))
    lst = str(tmp_path / 'tgt_train.lst')
    update_image_list(lst, files[0::3], files[1::3])
    assert open(lst).read() == '%s,%s\n%s,%s\n' % (files[0], files[1], files[3], files[4])       # uest_seg_multi_os.py:726
    assert read_image_list(lst) == (files[0::3], files[1::3], [])
    update_image_list(lst, files[0::3], files[1::3], files[2::3])
    assert read_image_list(lst, use_depth=True) == (files[0::3], files[1::3], files[2::3])
    os.remove(files[4])
    with pytest.raises(AssertionError):
        read_image_list(lst)


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(IMAGEIO_CASES))
def test_device_transforms_bit_exact(name, golden):
    from mspl_amd.io import Preprocessor
    hs, ws, size, seed, norm, flip, with_depth = IMAGEIO_CASES[name]
    N = 3
    imgs = [synth_image_u8(hs, ws, seed + 100 * i) for i in range(N)]
    imgs[0] = synth_image_u8(hs, ws, seed)                                 # image 0 = the golden's image
    flips = [flip, not flip, flip]                                         # per-image flags
    rgb = torch.from_numpy(np.stack([im[0] for im in imgs]))
    lab = torch.from_numpy(np.stack([im[1] for im in imgs]))
    dep = torch.from_numpy(np.stack([im[2] for im in imgs])) if with_depth else None
    pre = Preprocessor(size=size, normalize=norm)
    x, y, d = pre(rgb, lab, dep, flip=torch.tensor(flips))
    assert x.shape == (N, 3, size[1], size[0]) and x.dtype == torch.float32 and y.dtype == torch.int64
    g = golden('imageio')
    assert np.array_equal(_sha(x[0].cpu().numpy()), g[name + '.rgb_sha'])
    assert np.array_equal(_sha(y[0].cpu().numpy()), g[name + '.label_sha'])
    if with_depth:
        assert np.array_equal(_sha(d[0].cpu().numpy()), g[name + '.depth_sha'])
    for i in range(N):                                                     # every image against the oracle
        t, lt, dt = oio.val_transform(imgs[i][0], imgs[i][1], imgs[i][2] if with_depth else None, size=size, normalise=norm,
                                      flip=flips[i])
        assert np.array_equal(x[i].cpu().numpy(), t) and np.array_equal(y[i].cpu().numpy(), lt)
        if with_depth:
            assert np.array_equal(d[i].cpu().numpy(), dt)
    with pytest.raises(RuntimeError, match='uint8'):
        pre(rgb.float())


@pytest.mark.gpu
def test_label_writer_async(tmp_path):
    from mspl_amd.io import LabelWriter, read_image_list, update_image_list
    rng = np.random.default_rng(11)
    maps = [torch.from_numpy(np.repeat(np.repeat(rng.integers(0, 5, (4, 32, 60), dtype=np.uint8), 8, 1), 8, 2)) for _ in range(3)]
    names = [['/data/greenhouse/color/img_%d_%d.jpg' % (b, i) for i in range(4)] for b in range(3)]
    with LabelWriter(str(tmp_path / 'pred'), workers=3) as w:
        for n, m in zip(names, maps):
            dev = m.cuda()
            w.submit(n, dev)
            dev.zero_()                                  # submit() snapshots: the caller may reuse its buffer at once
    images, labels = w.image_paths, w.label_paths
    assert images == sum(names, []) and labels[0] == '%s/img_0_0.png' % (tmp_path / 'pred')
    for b in range(3):
        for i in range(4):
            got = oio.png_decode_gray8(open(labels[b * 4 + i], 'rb').read())
            assert np.array_equal(got, maps[b][i].numpy())
    lst = str(tmp_path / 'tgt_train.lst')
    update_image_list(lst, images, labels)
    assert read_image_list(lst, check_files=False) == (images, labels, [])


@pytest.mark.gpu
@pytest.mark.parametrize('use_graph', [False, True])
def test_generate_pseudo_label_multi_model_end_to_end(tmp_path, use_graph):
    """The whole reference function (uest_seg_multi_os.py:832-956): uint8 frames -> device Resize+Normalize -> three source
    models -> LUT -> merge -> PNG files + tgt_train.lst + class weights; every stage checked against the oracle."""
    import argparse
    from mspl_amd import models, uest
    from mspl_amd.io import Preprocessor, read_image_list
    from oracle import labels as olab
    from oracle import net as onet
    from tests.synth import synth_state_dict
    a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
    specs = [(13, 'camvid', 'camvid', 61), (20, 'city', 'cityscapes', 62), (5, 'greenhouse', 'forest', 63)]
    ms, sds = [], []
    for C_, ds, _, seed in specs:
        m = models.ESPDNetwithUncertaintyEstimation(a, classes=C_, dataset=ds, fix_pyr_plane_proj=True)
        sd = synth_state_dict(m.state_dict(), seed)
        m.load_state_dict(sd)
        ms.append(m)
        sds.append(sd)
    pre = Preprocessor(size=(64, 48))
    frames = [np.stack([synth_image_u8(72, 96, 70 + 4 * b + i)[0] for i in range(2)]) for b in range(5)]     # 5 batches: the 3 lanes wrap
    names = [['/d/color/f_%d_%d.jpg' % (b, i) for i in range(2)] for b in range(5)]
    loader = [(pre(torch.from_numpy(f))[0], None, n, 0.0) for f, n in zip(frames, names)]
    lst, cw = uest.generate_pseudo_label_multi_model(ms, [s[2] for s in specs], loader, str(tmp_path), use_graph=use_graph)
    images, labels, _ = read_image_list(lst, check_files=False)
    assert images == sum(names, []) and all(os.path.isfile(p) for p in labels)
    hist = np.zeros(5)
    for b in range(5):
        for i in range(2):
            x = torch.from_numpy(oio.val_transform(frames[b][i], size=(64, 48))[0])[None]
            srcs = []
            for sd, (_, _, os_data, _) in zip(sds, specs):
                with torch.no_grad():
                    main, aux = onet.espdnet_ue_forward(sd, x)
                prob, _ = olab.get_output(main, aux)
                srcs.append(olab.to_greenhouse(olab.argmax_labels(prob)[0], os_data))
            want = olab.merge_outputs(np.array(srcs), 5, 'all')
            got = oio.png_decode_gray8(open(labels[b * 2 + i], 'rb').read())
            # argmax ties between near-equal logits may flip a pixel between fp32 summation orders: allow a handful
            assert (got != want).mean() < 2e-3
            hist += np.bincount(got.ravel(), minlength=5)[:5]
    ref_w = olab.class_weights_from_histogram(hist)
    assert np.allclose(cw.cpu().numpy(), ref_w.astype(np.float32), rtol=1e-6)


@pytest.mark.gpu
def test_generate_pseudo_label_batches_per_launch(tmp_path):
    """Two consecutive loader batches per launch (PipelinedLabelPass group) with a ragged loader -- five batches of two and one
    of one image -- writes byte-identical label files, the same list and the same class weights as one batch per launch."""
    import argparse
    from mspl_amd import models, uest
    from mspl_amd.io import Preprocessor
    from tests.synth import synth_state_dict
    a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
    ms = []
    for C_, ds, seed in [(13, 'camvid', 61), (5, 'greenhouse', 63)]:
        m = models.ESPDNetwithUncertaintyEstimation(a, classes=C_, dataset=ds, fix_pyr_plane_proj=True)
        m.load_state_dict(synth_state_dict(m.state_dict(), seed))
        ms.append(m)
    pre = Preprocessor(size=(64, 48))
    sizes = [2, 2, 2, 2, 2, 1]
    frames = [np.stack([synth_image_u8(72, 96, 90 + 4 * b + i)[0] for i in range(n)]) for b, n in enumerate(sizes)]
    names = [['/d/color/g_%d_%d.jpg' % (b, i) for i in range(n)] for b, n in enumerate(sizes)]
    outs = []
    for grp in (1, 2):
        d = tmp_path / ('g%d' % grp)
        d.mkdir()
        loader = [(pre(torch.from_numpy(f))[0], None, n, 0.0) for f, n in zip(frames, names)]
        lst, cw = uest.generate_pseudo_label_multi_model(ms, ['camvid', 'forest'], loader, str(d), use_graph=True, in_flight=2,
                                                         batches_per_launch=grp)
        files = sorted(os.listdir(str(d / 'pred')))
        outs.append((open(lst).read().replace(str(d), ''), cw.cpu().numpy(), {f: open(str(d / 'pred' / f), 'rb').read() for f in files}))
    assert outs[0][0] == outs[1][0] and len(outs[0][2]) == sum(sizes)
    assert np.array_equal(outs[0][1], outs[1][1])
    assert outs[0][2] == outs[1][2]


def _self_label_case(a_seed=71, C=5):
    import argparse
    from mspl_amd import models
    from tests.synth import synth_state_dict
    a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
    m = models.ESPDNetwithUncertaintyEstimation(a, classes=C, dataset='greenhouse', fix_pyr_plane_proj=True)
    sd = synth_state_dict(m.state_dict(), a_seed)
    m.load_state_dict(sd)
    return m, sd


def _check_against_oracle_loop(sd, frames, names, size, lst, cw, save, use_depth, classes=5):
    """Files, list and weights of a generate_pseudo_label run against the oracle's statement of uest_seg_multi_os.py:783-828."""
    from mspl_amd.io import read_image_list
    from oracle import labels as olab
    from oracle import net as onet

    def fwd(x):
        with torch.no_grad():
            return onet.espdnet_ue_forward(sd, x)
    ref_loader = [(torch.stack([torch.from_numpy(oio.val_transform(f, size=size)[0]) for f in fb]), None, n, 0.0)
                  for fb, n in zip(frames, names)]
    ri, rl, rd, rmaps, rw = olab.generate_pseudo_label(fwd, ref_loader, classes, '%s/pred' % save, 'normal', use_depth)
    images, labels, depths = read_image_list(lst, use_depth=use_depth, check_files=False)
    assert images == ri and labels == rl and depths == rd          # same lines, same order as the reference's list file
    hist = np.zeros(classes)
    for path, want in zip(labels, rmaps):
        got = oio.png_decode_gray8(open(path, 'rb').read())
        assert got.shape == want.shape and (got != want).mean() < 2e-3      # fp32 near-ties between two forwards; the rest equal
        hist += np.bincount(got.ravel(), minlength=classes)[:classes]
    # the integer / float64 stage is exact given the maps that were written
    assert np.allclose(cw.cpu().numpy(), olab.class_weights_from_histogram(hist).astype(np.float32), rtol=1e-6)
    assert np.allclose(cw.cpu().numpy()[1:], rw.astype(np.float32)[1:], rtol=2e-2) and float(cw[0]) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize('use_graph,use_depth', [(False, False), (True, True)])
def test_generate_pseudo_label_end_to_end(tmp_path, use_graph, use_depth):
    """The single-model relabelling function (uest_seg_multi_os.py:730-830) end to end: PNG bytes, list lines and order, class
    weights against the oracle's statement of :783-828; ragged loader (the lanes wrap and the last launch is partly filled)."""
    from mspl_amd import uest
    from mspl_amd.io import Preprocessor
    m, sd = _self_label_case()
    pre = Preprocessor(size=(64, 48))
    sizes = [2, 2, 2, 2, 2, 2, 1]
    frames = [np.stack([synth_image_u8(72, 96, 170 + 4 * b + i)[0] for i in range(n)]) for b, n in enumerate(sizes)]
    names = [['/d/color/s_%d_%d.jpg' % (b, i) for i in range(n)] for b, n in enumerate(sizes)]
    tup = (lambda x, n: (x, None, None, n, 0.0)) if use_depth else (lambda x, n: (x, None, n, 0.0))
    loader = [tup(pre(torch.from_numpy(f))[0], n) for f, n in zip(frames, names)]
    lst, cw = uest.generate_pseudo_label(m, loader, str(tmp_path), use_graph=use_graph, use_depth=use_depth)
    assert lst == os.path.join(str(tmp_path), 'tgt_train.lst') and cw.is_cuda and cw.dtype == torch.float32
    _check_against_oracle_loop(sd, frames, names, (64, 48), lst, cw, str(tmp_path), use_depth)
    if use_depth:
        assert open(lst).readline().rstrip().split(',')[2] == '/d/depth/s_0_0.jpg'


@pytest.mark.gpu
def test_generate_pseudo_label_transform_into_lane_slots(tmp_path):
    """Loader of decoded uint8 frames + transform=Preprocessor: the network input is written straight into the static input slot of
    the lane that labels it.  Same files as the tensor loader, and after the lanes' first launches no batch is copied."""
    from mspl_amd import uest
    from mspl_amd.io import Preprocessor
    m, sd = _self_label_case(72)
    pre = Preprocessor(size=(64, 48))
    frames = [np.stack([synth_image_u8(72, 96, 270 + 4 * b + i)[0] for i in range(2)]) for b in range(13)]
    names = [['/d/color/t_%d_%d.jpg' % (b, i) for i in range(2)] for b in range(13)]
    outs = []
    for k, kw in enumerate([dict(), dict(transform=pre)]):
        d = tmp_path / ('v%d' % k)
        d.mkdir()
        if k == 0:
            loader = [(pre(torch.from_numpy(f))[0], None, n, 0.0) for f, n in zip(frames, names)]
        else:
            loader = [(torch.from_numpy(f), None, n, 0.0) for f, n in zip(frames, names)]
        lst, cw = uest.generate_pseudo_label(m, loader, str(d), in_flight=2, batches_per_launch=2, **kw)
        files = sorted(os.listdir(str(d / 'pred')))
        outs.append((open(lst).read().replace(str(d), ''), cw.cpu().numpy(), {f: open(str(d / 'pred' / f), 'rb').read() for f in files}))
    assert outs[0][0] == outs[1][0] and len(outs[0][2]) == 26
    assert np.array_equal(outs[0][1], outs[1][1]) and outs[0][2] == outs[1][2]
    _check_against_oracle_loop(sd, frames, names, (64, 48), str(tmp_path / 'v1' / 'tgt_train.lst'),
                               torch.from_numpy(outs[1][1]), str(tmp_path / 'v1'), False)


@pytest.mark.gpu
def test_script_level_generate_pseudo_label_reference_signature(tmp_path, monkeypatch):
    """patch_script(): the script's own call `generate_pseudo_label(model, device, save_path, round_idx, tgt_num, label_2_id,
    valid_labels, args, logger, class_encoding, writer)` (uest_seg_multi_os.py:527) with a loader built from args through the
    reference's dataset class (a stand-in package on sys.path plays data_loader.segmentation.greenhouse)."""
    import argparse
    import sys
    import mspl_amd
    from mspl_amd import script
    from tests.test_host import _purge_reference_names
    m, sd = _self_label_case(73)
    frames = [synth_image_u8(48, 64, 370 + i)[0] for i in range(5)]
    np.save(str(tmp_path / 'frames.npy'), np.stack(frames))
    (tmp_path / 'data_loader' / 'segmentation').mkdir(parents=True)
    (tmp_path / 'data_loader' / '__init__.py').write_text('')
    (tmp_path / 'data_loader' / 'segmentation' / '__init__.py').write_text('')
    (tmp_path / 'data_loader' / 'segmentation' / 'greenhouse.py').write_text(
        'import numpy as np, torch\n'
        'from oracle import imageio as oio\n'
        'class GreenhouseRGBDSegmentation(torch.utils.data.Dataset):\n'
        '    def __init__(self, list_name, train=True, use_traversable=False, use_depth=False):\n'
        '        assert train is False and use_depth is False\n'
        '        self.frames = np.load(list_name)\n'
        '    def __len__(self):\n'
        '        return len(self.frames)\n'
        '    def __getitem__(self, i):\n'
        '        x = torch.from_numpy(oio.val_transform(self.frames[i], size=(64, 48))[0])\n'
        '        return x, torch.zeros(48, 64, dtype=torch.int64), "/t/color/img_%02d.png" % i, 0.0\n')
    monkeypatch.syspath_prepend(str(tmp_path))
    _purge_reference_names()
    try:
        ns = {'generate_pseudo_label': None, 'merge_outputs': None}
        mspl_amd.install_dropin(script=ns)
        assert ns['generate_pseudo_label'] is script.generate_pseudo_label and ns['merge_outputs'] is mspl_amd.uest.merge_outputs
        args = argparse.Namespace(classes=5, dataset='greenhouse', data_tgt_train_list=str(tmp_path / 'frames.npy'),
                                  use_traversable=False, use_depth=False, pin_memory=False, class_weighting='normal',
                                  eval_training=False, label_batch_size=2)
        logged = []

        class Log:
            def info(self, s):
                logged.append(s)
        save = str(tmp_path / 'run')
        lst, cw = ns['generate_pseudo_label'](m, 'cuda', save, 3, 5, None, None, args, Log(), None, None)
        assert 'round 3' in logged[0]
        names = [['/t/color/img_%02d.png' % i for i in range(b, min(b + 2, 5))] for b in range(0, 5, 2)]
        fb = [np.stack(frames[b:b + 2]) for b in range(0, 5, 2)]
        _check_against_oracle_loop(sd, fb, names, (64, 48), lst, cw, save, False)
        # --eval-training (:749-752): the adapter hands the flag on; the model is left in train() mode like the reference leaves it
        args.eval_training = True
        lst2, cw2 = ns['generate_pseudo_label'](m, 'cuda', save, 4, 5, None, None, args, Log(), None, None)
        assert m.training and lst2 == lst and cw2.shape == cw.shape and not torch.equal(cw2, cw)
    finally:
        _purge_reference_names()


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['self_c5', 'self_c5_unweighted', 'multi3_all', 'multi2_half', 'self_c5_evaltrain', 'multi2_all_evaltrain'])
def test_label_loop_functions_vs_reference_golden(tmp_path, name, golden):
    """generate_pseudo_label / generate_pseudo_label_multi_model against the REFERENCE's own functions (uest_seg_multi_os.py:730-830,
    :832-956, AST-extracted and run on CPU by tests/golden/make_golden.py gen_label_loops): list lines and order, decoded label files,
    class weights.  Two fp32 forwards cannot agree on the argmax where the top-2 probabilities are within rounding of each other; the
    golden carries the reference's top-2 margin per pixel, and every pixel whose margin exceeds 2e-3 (10x the logit tolerance) must
    be IDENTICAL."""
    import argparse
    import json
    from mspl_amd import models, uest
    from tests.cases import LABEL_LOOP_CASES
    from tests.conftest import GOLDEN
    from tests.synth import synth_label_loop_images, synth_state_dict
    g = golden('label_loops')
    lines = json.load(open(os.path.join(GOLDEN, 'label_loops.json')))[name]
    specs, (H, W), n, in_seed, policy, weighting = LABEL_LOOP_CASES[name][:6]
    # the `--eval-training` cases (:749-752, :871-876): models in train() mode, BatchNorm with the statistics of each single image
    eval_training = len(LABEL_LOOP_CASES[name]) > 6 and LABEL_LOOP_CASES[name][6]
    items = synth_label_loop_images(LABEL_LOOP_CASES[name])
    a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
    ms = []
    for C_, ds, os_data, seed in specs:
        m = models.ESPDNetwithUncertaintyEstimation(a, classes=C_, dataset=ds, fix_pyr_plane_proj=True)
        m.load_state_dict(synth_state_dict(m.state_dict(), seed))
        ms.append(m)
    loader = [(torch.stack([x for x, _ in items[i:i + 2]]), None, [nm for _, nm in items[i:i + 2]], 1.0) for i in range(0, n, 2)]
    if specs[0][2] is None:
        lst, cw = uest.generate_pseudo_label(ms[0], loader, str(tmp_path), class_weighting=weighting, in_flight=2, batches_per_launch=2,
                                             eval_training=eval_training)
    else:
        lst, cw = uest.generate_pseudo_label_multi_model(ms, [s[2] for s in specs], loader, str(tmp_path), merge_label_policy=policy,
                                                         class_weighting=weighting, in_flight=2, eval_training=eval_training)
    assert all(m.training == bool(eval_training) for m in ms)
    got_lines = open(lst).read().replace(str(tmp_path), '{SAVE}').splitlines()
    assert got_lines == lines
    want, margin = g[name + '.maps'], g[name + '.margin']
    got = np.stack([oio.png_decode_gray8(open(ln.split(',')[1].replace('{SAVE}', str(tmp_path)), 'rb').read()) for ln in got_lines])
    assert got.shape == want.shape and got.dtype == np.uint8
    sure = margin > 2e-3
    assert sure.mean() > 0.5 and np.array_equal(got[sure], want[sure])       # (three models: 75 % of the pixels have all margins > 2e-3)
    ndiff = int((got != want).sum())
    assert ndiff <= int((~sure).sum())
    ref_w = g[name + '.class_weights']
    if ndiff == 0:
        assert np.array_equal(cw.cpu().numpy(), ref_w)
    else:       # counts differ by at most ndiff pixels per class
        hist = np.array([(got == c).sum() for c in range(5)], dtype=np.float64)
        from oracle import labels as olab
        assert np.array_equal(cw.cpu().numpy(), olab.class_weights_from_histogram(hist, weighting).astype(np.float32))
        big = ref_w < 1e9
        assert np.allclose(cw.cpu().numpy()[big], ref_w[big], rtol=4.0 * ndiff / max(1.0, hist[big & (ref_w > 0)].min()) + 1e-6)
