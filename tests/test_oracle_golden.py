"""Pin the CPU oracle (oracle/) against vectors produced by the reference's own modules.

The golden files were written by tests/golden/make_golden.py, which imports /root/reference on CPU.
These tests run anywhere (no reference, no GPU).
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import labels as olab
from oracle import net as onet
from tests.cases import ASPP_CASES, ESPDNET_CASES, EVAL_CASES, LABEL_LOOP_CASES, LAYER_CASES, MODEL_CASES, RGBD_CASES, TRAIN_CASE, TRAIN_CASES
from tests.conftest import GOLDEN
from tests.synth import synth_adversarial_logits, synth_eval_batches, synth_input, synth_label_loop_images, synth_labels, synth_state_dict

KEYS = json.load(open(os.path.join(GOLDEN, 'state_dict_keys.json')))


LAYER_KEYS = json.load(open(os.path.join(GOLDEN, 'layer_keys.json')))


@pytest.mark.parametrize('name', sorted(LAYER_CASES))
def test_layer(name, golden):
    kind, kw, shp, shp2 = LAYER_CASES[name]
    i = sorted(LAYER_CASES).index(name)
    sd = synth_state_dict(LAYER_KEYS[name], 100 + i)
    sd = {'m.' + k: v for k, v in sd.items()}
    x = synth_input(shp, 200 + i)
    with torch.no_grad():
        if kind == 'eesp':
            y = onet.eesp(x, sd, 'm', kw['stride'], kw['r_lim'], k=kw['k'])
        elif kind == 'down':
            img = synth_input(shp2, 300 + i) if shp2 is not None else None
            y = onet.downsampler(x, sd, 'm', kw['r_lim'], img, k=kw['k'])
        elif kind == 'pyr':
            y = onet.pyr_pool(x, sd, 'm', kw['last_layer_br'])
        else:
            y = onet.pw_conv(x, sd, 'm')
    ref = torch.from_numpy(golden('layers')[name])
    assert y.shape == ref.shape
    torch.testing.assert_close(y, ref, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize('name', sorted(MODEL_CASES))
def test_model(name, golden):
    kind, s, classes, dataset, shp, sd_seed, in_seed = MODEL_CASES[name]
    sd = synth_state_dict(KEYS['%s_s%s_c%d' % (kind, s, classes)], sd_seed)
    x = synth_input(shp, in_seed)
    g = golden('model_' + name)
    st = int(g['stride'])
    with torch.no_grad():
        if kind == 'espdnetue':
            main, aux = onet.espdnet_ue_forward(sd, x)
            torch.testing.assert_close(main[:, :, ::st, ::st], torch.from_numpy(g['main']), rtol=1e-4, atol=1e-4)
            torch.testing.assert_close(aux[:, :, ::st, ::st], torch.from_numpy(g['aux']), rtol=1e-4, atol=1e-4)
            prob, kld = olab.get_output(main, aux)
            torch.testing.assert_close(kld[:, ::st, ::st], torch.from_numpy(g['kld']), rtol=1e-3, atol=1e-4)
            amax = olab.argmax_labels(prob)
            # two independent fp32 forwards may flip argmax where the top-2 margin is at rounding level
            diff = amax != g['amax']
            assert not np.any(diff & (g['margin'].astype(np.float32) > 1e-4))
            assert diff.mean() < 1e-3
        else:
            main = onet.espnetv2_forward(sd, x)
            torch.testing.assert_close(main[:, :, ::st, ::st], torch.from_numpy(g['main']), rtol=1e-4, atol=1e-4)


def test_param_counts_match_published_tables():
    """model/segmentation/model_zoo/README.md:51-87: 0.79 M (s=2.0) / 0.08 M (s=0.5) parameters."""
    def count(keys):
        return sum(int(np.prod(v)) for k, v in keys.items()
                   if not k.endswith(('running_mean', 'running_var', 'num_batches_tracked')))
    zoo = np.load(os.path.join(GOLDEN, 'zoo_espnetv2_s0.5_city_512x256.npz'))
    n = sum(int(np.prod(zoo[k].shape)) for k in zoo.files
            if not k.endswith(('running_mean', 'running_var', 'num_batches_tracked')))
    assert round(n / 1e6, 2) == 0.08
    assert count(KEYS['espdnetue_s2.0_c5']) == 2234230
    assert count(KEYS['espdnetue_s2.0_c13']) == 2234502
    assert count(KEYS['espdnetue_s2.0_c20']) == 2234740
    assert len(KEYS['espdnetue_s2.0_c20']) == 918


def test_zoo_real_weights(golden):
    """BASELINE config 1 shape with the real Cityscapes checkpoint (strict key match)."""
    zoo = np.load(os.path.join(GOLDEN, 'zoo_espnetv2_s0.5_city_512x256.npz'))
    sd = {k: torch.from_numpy(zoo[k]) for k in zoo.files}
    x = synth_input((2, 3, 288, 480), 40)
    g = golden('model_v2_zoo_288x480')
    with torch.no_grad():
        y = onet.espnetv2_forward(sd, x)
    torch.testing.assert_close(y[:, :, ::8, ::8], torch.from_numpy(g['main']), rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(y.double().sum((0, 2, 3)).numpy(), g['class_sum'], rtol=1e-4)
    amax = y.argmax(1).to(torch.uint8).numpy()
    diff = amax != g['amax']
    assert not np.any(diff & (g['margin'].astype(np.float32) > 1e-3))
    assert diff.mean() < 1e-3


@pytest.mark.parametrize('S', [1, 2, 3, 4])
@pytest.mark.parametrize('pol', ['all', 'half', 'none'])
def test_merge_truth_table(S, pol, golden):
    g = golden('labels')
    got = olab.merge_outputs(g['tt_in_S%d' % S], 5, None if pol == 'none' else pol)
    np.testing.assert_array_equal(got.astype(np.uint8), g['tt_S%d_%s' % (S, pol)])


def test_merge_random_and_luts(golden):
    g = golden('labels')
    np.testing.assert_array_equal(olab.merge_outputs(g['rnd_in'], 5, 'all'), g['rnd_all'])
    np.testing.assert_array_equal(olab.merge_outputs(g['rnd_in'], 5, 'half'), g['rnd_half'])
    np.testing.assert_array_equal(olab.ID_CAMVID_TO_GREENHOUSE, g['lut_id_camvid_to_greenhouse'])
    np.testing.assert_array_equal(olab.ID_CITYSCAPES_TO_GREENHOUSE, g['lut_id_cityscapes_to_greenhouse'])
    np.testing.assert_array_equal(olab.ID_FOREST_TO_GREENHOUSE, g['lut_id_forest_to_greenhouse'])


@pytest.mark.parametrize('C', [5, 13, 20])
def test_uncertainty_estimator(C, golden):
    g = golden('labels')
    d1 = synth_input((2, C, 12, 20), 50 + C) * 3
    d2 = synth_input((2, C, 12, 20), 70 + C) * 3
    prob, kld = olab.get_output(d1, d2)
    torch.testing.assert_close(kld, torch.from_numpy(g['kld_C%d' % C]), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(prob, torch.from_numpy(g['prob_C%d' % C]), rtol=1e-5, atol=1e-7)


def test_uw_loss_value_and_grads(golden):
    g = golden('loss')
    pred = (synth_input((2, 5, 32, 48), 90) * 2).requires_grad_(True)
    aux = (synth_input((2, 5, 32, 48), 91) * 2).requires_grad_(True)
    tgt = synth_labels((2, 32, 48), 5, 92)
    loss = olab.uest_train_loss(pred, aux, tgt, torch.from_numpy(g['cw']), ignore_idx=4)
    loss.backward()
    torch.testing.assert_close(loss.detach(), torch.from_numpy(g['loss']), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(pred.grad, torch.from_numpy(g['dpred']), rtol=1e-4, atol=1e-8)
    torch.testing.assert_close(aux.grad, torch.from_numpy(g['daux']), rtol=1e-4, atol=1e-8)


@pytest.mark.parametrize('gname', sorted(TRAIN_CASES))
def test_train_step(gname, golden):
    """One uest self-training step (frozen BN, Adam + weight decay, unused params skipped)."""
    from oracle.train import train_step
    c = TRAIN_CASES[gname]
    g = golden(gname)
    sd = synth_state_dict(KEYS['espdnetue_s%s_c%d' % (c['s'], c['classes'])], c['sd_seed'])
    x = synth_input(c['shape'], c['in_seed'])
    labels = synth_labels((c['shape'][0],) + c['shape'][2:], c['classes'], c['in_seed'])
    names = [str(n) for n in g['names']]
    loss, grads, new = train_step(sd, names, x, labels, torch.ones(c['classes']), c['ignore_idx'],
                                  lr=c['lr'], weight_decay=c['weight_decay'])
    torch.testing.assert_close(loss, torch.from_numpy(g['loss']), rtol=1e-5, atol=1e-6)
    for n, gn, gs in zip(names, g['gnorm'], g['gsum']):
        if gn < 0:
            assert grads[n] is None, n
        else:
            assert grads[n] is not None, n
            assert abs(float(grads[n].double().norm()) - gn) <= 2e-3 * gn + 1e-7, n
    assert sum(1 for v in g['gnorm'] if v >= 0) == 340
    for i, k in enumerate(str(s) for s in g['keep']):
        torch.testing.assert_close(new[k], torch.from_numpy(g['after_%d' % i]), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('name', sorted(ASPP_CASES))
def test_aspp(name, golden):
    """Oracle restatement of nn_layers/aspp.py vs the reference's own output (BASELINE configs[4] head)."""
    cls, ncls, shp, sd_seed, x_seed = ASPP_CASES[name]
    keys = json.load(open(os.path.join(GOLDEN, 'aspp_keys.json')))[name]
    sd = synth_state_dict(keys, sd_seed)
    with torch.no_grad():
        y = onet.aspp_forward(sd, synth_input(shp, x_seed))
    ref = torch.from_numpy(golden('aspp')[name])
    assert y.shape == ref.shape
    torch.testing.assert_close(y, ref, rtol=1e-5, atol=5e-5)


@pytest.mark.parametrize('name', sorted(RGBD_CASES))
def test_rgbd_forward(name, golden):
    """The x_d path (depth encoder + fusion gates, espdnet_ue.py:186-270) vs the reference's own (main, aux)."""
    classes, dataset, shp, sd_seed, in_seed, d_seed, dense, trainable = RGBD_CASES[name]
    sd = synth_state_dict(KEYS['espdnetue_s2.0_c%d' % classes], sd_seed)
    x, x_d = synth_input(shp, in_seed), synth_input((shp[0], 1) + tuple(shp[2:]), d_seed)
    with torch.no_grad():
        main, aux = onet.espdnet_ue_forward(sd, x, x_d, dense_fuse=dense, trainable_fusion=trainable)
    g = golden('rgbd')
    torch.testing.assert_close(main, torch.from_numpy(g[name + '.main']), rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(aux, torch.from_numpy(g[name + '.aux']), rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize('name', sorted(ESPDNET_CASES))
def test_espdnet_forward(name, golden):
    """Single-head ESPDNetSegmentation (model/segmentation/espdnet.py), with and without a depth image."""
    classes, dataset, shp, sd_seed, in_seed, d_seed, dense, trainable = ESPDNET_CASES[name]
    keys = json.load(open(os.path.join(GOLDEN, 'espdnet_keys.json')))[name]
    sd = synth_state_dict(keys, sd_seed)
    x = synth_input(shp, in_seed)
    x_d = None if d_seed is None else synth_input((shp[0], 1) + tuple(shp[2:]), d_seed)
    with torch.no_grad():
        y = onet.espdnet_forward(sd, x, x_d, dense_fuse=dense, trainable_fusion=trainable)
    torch.testing.assert_close(y, torch.from_numpy(golden('rgbd')[name]), rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize('name', sorted(EVAL_CASES))
def test_eval_step(name, golden):
    """oracle.labels.val_seg_ue against the reference's own val_seg_ue (AST-extracted, tests/golden/make_golden.py gen_eval) and
    against the body of the uest script's test() (main head alone)."""
    g = golden('eval')
    C, ds, shape, nb, sd_seed, in_seed, ign, cw_seed, with_void = EVAL_CASES[name]
    sd = synth_state_dict(KEYS['espdnetue_s2.0_c%d' % C], sd_seed)
    cw = torch.from_numpy(g[name + '.cw'])
    loader = synth_eval_batches(EVAL_CASES[name])
    fwd = lambda x: onet.espdnet_ue_forward(sd, x)
    iou, loss = olab.val_seg_ue(fwd, loader, cw, ign, C, aux_weight=0.5)
    np.testing.assert_allclose(iou, g[name + '.iou'], rtol=0, atol=1e-6)
    assert abs(loss - float(g[name + '.loss'])) < 2e-5
    iou0, loss0 = olab.val_seg_ue(fwd, loader, cw, ign, C, aux_weight=0.0)
    np.testing.assert_allclose(iou0, g[name + '.test_iou'], rtol=0, atol=1e-6)
    assert abs(loss0 - float(g[name + '.test_loss'])) < 2e-5


LOOP_LINES = json.load(open(os.path.join(GOLDEN, 'label_loops.json')))


@pytest.mark.parametrize('name', sorted(LABEL_LOOP_CASES))
def test_label_loops_vs_reference_functions(name, golden):
    """oracle.labels.generate_pseudo_label / generate_pseudo_label_multi_model against the reference's own functions
    (uest_seg_multi_os.py:730-830, :832-956; AST-extracted and run by tests/golden/make_golden.py gen_label_loops on a stub dataset):
    the list file's lines and order (file-name rule), the label maps the reference wrote as PNG files, the class weights."""
    g = golden('label_loops')
    specs, (H, W), n, in_seed, policy, weighting = LABEL_LOOP_CASES[name][:6]
    eval_training = len(LABEL_LOOP_CASES[name]) > 6 and LABEL_LOOP_CASES[name][6]
    items = synth_label_loop_images(LABEL_LOOP_CASES[name])
    # ragged batches on purpose: the reference's loader has batch size 1, the restatement walks a batch element by element
    loader, i = [], 0
    for bs in [2, 1, 3, 2, 2]:
        if i < n:
            part = items[i:i + bs]
            loader.append((torch.stack([x for x, _ in part]), None, [nm for _, nm in part], 1.0))
            i += len(part)
    fwds = []
    for C, ds, os_data, sd_seed in specs:
        sd = synth_state_dict(KEYS['espdnetue_s2.0_c%d' % C], sd_seed)
        # eval_training: BatchNorm with the statistics of the (single-image) batch -- the restatement walks a batch element by element
        fwds.append(lambda x, sd=sd: onet.espdnet_ue_forward(sd, x))
    import contextlib
    with torch.no_grad(), (onet.bn_training() if eval_training else contextlib.nullcontext()):
        if specs[0][2] is None:
            ri, rl, rd, maps, cw = olab.generate_pseudo_label(fwds[0], loader, 5, '{SAVE}/pred', weighting)
        else:
            ri, rl, rd, maps, cw = olab.generate_pseudo_label_multi_model(fwds, [s[2] for s in specs], loader, 5, '{SAVE}/pred',
                                                                          policy, weighting)
    assert ['%s,%s' % (a, b) for a, b in zip(ri, rl)] == LOOP_LINES[name] and rd == []
    want, margin = g[name + '.maps'], g[name + '.margin']
    got = np.stack(maps)
    assert got.dtype == np.uint8 and got.shape == want.shape
    # the restatement and the reference module agree to ~1e-6 on the probabilities: away from exact ties the maps are IDENTICAL
    assert np.array_equal(got[margin > 1e-5], want[margin > 1e-5])
    ndiff = int((got != want).sum())
    assert ndiff <= int((margin <= 1e-5).sum())
    if ndiff == 0:
        np.testing.assert_array_equal(cw.astype(np.float32), g[name + '.class_weights'])
    # the integer / float64 stage alone, on the reference's own maps: exact
    hist = np.array([(want == c).sum() for c in range(5)], dtype=np.float64)
    np.testing.assert_array_equal(olab.class_weights_from_histogram(hist, weighting).astype(np.float32), g[name + '.class_weights'])


@pytest.mark.parametrize('C', [5, 13, 20])
def test_argmax_rule_on_adversarial_logits(C, golden):
    """The oracle's label rule (softmax THEN first-max argmax, uest_seg_multi_os.py:687-691,798) on logits whose top two entries are
    0..4 ulp apart, against the reference's own get_output + np.argmax on the same logits (gen_argmax_adversarial).  The restatement
    runs the same torch softmax, so it reproduces the reference's choice wherever that choice does not hinge on the last bit of a
    probability: allowed to differ only where the reference itself gave both candidates probabilities at most 1 ulp apart."""
    g = golden('argmax_adversarial')
    pred, aux, a, b, k = synth_adversarial_logits(C, C)
    prob, _ = olab.get_output(pred, aux)
    got = olab.argmax_labels(prob)
    want, pa, pb = g['C%d.amax' % C], g['C%d.pa' % C], g['C%d.pb' % C]
    assert np.all((got == a) | (got == b)) and np.all((want == a) | (want == b))
    one_ulp = np.abs(pa - pb) <= np.spacing(np.maximum(pa, pb))
    assert np.array_equal(got[~one_ulp], want[~one_ulp])
    # the rule itself on the stored probabilities: larger probability wins, equal probabilities -> the LOWER class id
    rule = np.where(pa > pb, a, np.where(pb > pa, b, np.minimum(a, b)))
    assert np.array_equal(rule, want)
