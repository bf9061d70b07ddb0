"""Parity of the HIP path (through the C ABI) against the golden vectors produced by the reference and
against the CPU oracle on the same seeded inputs.  Needs a real MI355X: run with `-m gpu`.

Tolerances: logits within 1e-3 absolute (BASELINE.json north_star) -- tested much tighter (2e-4) --;
integer label work (LUT, merge, histogram) bit-exact; argmax equal wherever the reference's own top-2
probability margin exceeds rounding level.
"""
import argparse
import json
import os

import numpy as np
import pytest
import torch

from oracle import labels as olab
from oracle import net as onet
from tests.cases import ASPP_CASES, ESPDNET_CASES, LAYER_CASES, MODEL_CASES, RGBD_CASES
from tests.conftest import GOLDEN
from tests.synth import synth_input, synth_state_dict

pytestmark = pytest.mark.gpu
KEYS = json.load(open(os.path.join(GOLDEN, 'state_dict_keys.json')))
DEV = 'cuda'
LOGIT_ATOL = 2e-4


def _args(s):
    return argparse.Namespace(s=s, channels=3, num_classes=1000)


def _build_layer(kind, kw):
    from mspl_amd import layers as L
    cls = {'eesp': L.EESP, 'down': L.DownSampler, 'pyr': L.EfficientPyrPool, 'pw': L.EfficientPWConv}[kind]
    return cls(**kw)


def _build_model(kind, s, classes, dataset):
    from mspl_amd import models as M
    if kind == 'espdnetue':
        return M.ESPDNetwithUncertaintyEstimation(_args(s), classes=classes, dataset=dataset, fix_pyr_plane_proj=True)
    return M.ESPNetv2Segmentation(_args(s), classes=classes, dataset=dataset)


def test_native_library_is_loaded():
    import mspl_amd
    from mspl_amd import _native
    assert os.path.isfile(_native.LIB_PATH)
    with open('/proc/self/maps') as f:
        assert 'libmspl_hip.so' in f.read()


@pytest.mark.parametrize('name', sorted(LAYER_CASES))
def test_layer_vs_reference_golden(name, golden):
    kind, kw, shp, shp2 = LAYER_CASES[name]
    i = sorted(LAYER_CASES).index(name)
    m = _build_layer(kind, kw)
    m.load_state_dict(synth_state_dict(m.state_dict(), 100 + i))
    m = m.to(DEV).eval()
    x = synth_input(shp, 200 + i).to(DEV)
    with torch.no_grad():
        y = m(x, synth_input(shp2, 300 + i).to(DEV)) if shp2 is not None else m(x)
    ref = torch.from_numpy(golden('layers')[name])
    assert y.shape == ref.shape
    torch.testing.assert_close(y.cpu(), ref, rtol=1e-4, atol=5e-5)


@pytest.mark.parametrize('name', sorted(MODEL_CASES))
def test_model_vs_reference_golden(name, golden):
    kind, s, classes, dataset, shp, sd_seed, in_seed = MODEL_CASES[name]
    m = _build_model(kind, s, classes, dataset)
    m.load_state_dict(synth_state_dict(KEYS['%s_s%s_c%d' % (kind, s, classes)], sd_seed))
    m = m.to(DEV).eval()
    x = synth_input(shp, in_seed).to(DEV)
    g = golden('model_' + name)
    st = int(g['stride'])
    from mspl_amd import ops
    with torch.no_grad():
        if kind == 'espdnetue':
            main, aux = m(x)
            torch.testing.assert_close(main[:, :, ::st, ::st].cpu(), torch.from_numpy(g['main']), rtol=1e-4, atol=LOGIT_ATOL)
            torch.testing.assert_close(aux[:, :, ::st, ::st].cpu(), torch.from_numpy(g['aux']), rtol=1e-4, atol=LOGIT_ATOL)
            lo_main, lo_aux = m.forward_lowres(x)
            r = ops.label_epilogue(lo_main, lo_aux, shp[2:], want_kld=True, want_prob=True)
            torch.testing.assert_close(r['kld'][:, ::st, ::st].cpu(), torch.from_numpy(g['kld']), rtol=2e-3, atol=2e-4)
            amax = r['labels'].cpu().numpy()
            diff = amax != g['amax']
            assert not np.any(diff & (g['margin'].astype(np.float32) > 1e-3)), 'argmax differs at a confident pixel'
            assert diff.mean() < 2e-3
            # the fused kernel's own argmax is consistent with its own probabilities (first max)
            p = r['prob'].cpu().numpy()
            assert np.array_equal(np.argmax(p, axis=1).astype(np.uint8)[~diff], amax[~diff])
        else:
            y = m(x)
            torch.testing.assert_close(y[:, :, ::st, ::st].cpu(), torch.from_numpy(g['main']), rtol=1e-4, atol=LOGIT_ATOL)


def test_zoo_real_weights_config1(golden):
    """BASELINE config 1: ESPNetv2 s=0.5 with the real Cityscapes checkpoint on 288x480 inputs."""
    zoo = np.load(os.path.join(GOLDEN, 'zoo_espnetv2_s0.5_city_512x256.npz'))
    m = _build_model('espnetv2', 0.5, 20, 'city')
    m.load_state_dict({k: torch.from_numpy(zoo[k]) for k in zoo.files}, strict=True)
    m = m.to(DEV).eval()
    x = synth_input((2, 3, 288, 480), 40).to(DEV)
    g = golden('model_v2_zoo_288x480')
    with torch.no_grad():
        y = m(x)
    torch.testing.assert_close(y[:, :, ::8, ::8].cpu(), torch.from_numpy(g['main']), rtol=1e-4, atol=5e-4)
    np.testing.assert_allclose(y.double().sum((0, 2, 3)).cpu().numpy(), g['class_sum'], rtol=2e-4)
    amax = y.argmax(1).to(torch.uint8).cpu().numpy()
    diff = amax != g['amax']
    assert not np.any(diff & (g['margin'].astype(np.float32) > 2e-3))
    assert diff.mean() < 1e-3


@pytest.mark.parametrize('shape', [(1, 3, 64, 1024), (2, 3, 48, 512), (1, 3, 32, 352)])
def test_model_wide_inputs_fused_and_fallback(shape, monkeypatch):
    """Cityscapes-shaped inputs (train_espdnetue_city.sh:6: 1024x512 and 512x256): the stride-1 EESP blocks take the fused K2 + K3 launch at
    64 / 128 columns too (level 4: one row per band; level 3: two half-row bands per row) -- bit-identical to the three-launch form
    when the next block's projection is left out (that stage differs in summation order), and equal to the CPU oracle.  352 columns
    (22 / 44 at levels 4 / 3) are NOT covered by the fused launch: the blocks fall back to the three launches, same check."""
    from mspl_amd import layers, ops
    m = _build_model('espdnetue', 2.0, 20, 'city')
    sd = synth_state_dict(KEYS['espdnetue_s2.0_c20'], 21)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    x = synth_input(shape, 77)
    H, W = shape[2] // 16, shape[3] // 16
    covered = ops.eesp_dw_exp_fits((shape[0], 128, H, W), [1, 1, 2, 3]) and ops.eesp_dw_exp_fits((shape[0], 64, 2 * H, 2 * W), [1, 2, 3, 4])
    assert covered == (shape[3] in (512, 1024))
    with torch.no_grad():
        main, aux = m(x.to(DEV))
        monkeypatch.setattr(layers, '_FUSED_NEXT_PROJ', False)
        main1, aux1 = m(x.to(DEV))
        monkeypatch.setattr(layers, '_FUSED_DW_EXP', False)
        main0, aux0 = m(x.to(DEV))
        rmain, raux = onet.espdnet_ue_forward(sd, x)
    assert torch.equal(main1, main0) and torch.equal(aux1, aux0)          # K2 + K3 in one launch == the two launches, bit for bit
    torch.testing.assert_close(main.cpu(), rmain, rtol=1e-4, atol=2e-4)
    torch.testing.assert_close(aux.cpu(), raux, rtol=1e-4, atol=2e-4)
    torch.testing.assert_close(main0.cpu(), rmain, rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize('S', [1, 2, 3, 4])
@pytest.mark.parametrize('pol', ['all', 'half', 'none'])
def test_merge_truth_table_bit_exact(S, pol, golden):
    from mspl_amd import uest
    g = golden('labels')
    got = uest.merge_outputs(g['tt_in_S%d' % S], 5, None if pol == 'none' else pol)
    assert got.dtype == np.int64
    np.testing.assert_array_equal(got.astype(np.uint8), g['tt_S%d_%s' % (S, pol)])


def test_merge_random_maps_and_histogram(golden):
    from mspl_amd import ops, uest
    g = golden('labels')
    np.testing.assert_array_equal(uest.merge_outputs(g['rnd_in'], 5, 'all'), g['rnd_all'])
    np.testing.assert_array_equal(uest.merge_outputs(g['rnd_in'], 5, 'half'), g['rnd_half'])
    # ragged sizes (tails of the 16-pixel vector path), unaligned views, histogram
    rng = np.random.RandomState(3)
    for npix in (1, 15, 16, 17, 4099, 256 * 480 * 3 + 5):
        src = rng.randint(0, 5, size=(3, npix)).astype(np.uint8)
        t = torch.from_numpy(src).to(DEV)
        hist = torch.zeros(5, dtype=torch.int64, device=DEV)
        out = ops.merge_labels([t[0], t[1], t[2]], 5, 2, 4, hist).cpu().numpy()
        ref = olab.merge_outputs(src, 5, 'half')
        np.testing.assert_array_equal(out, ref)
        np.testing.assert_array_equal(hist.cpu().numpy(), olab.class_histogram(ref).astype(np.int64))
    # labels outside the class range (e.g. 255) never win a vote
    src = np.array([[255, 1, 0], [255, 1, 3], [2, 7, 3]], dtype=np.uint8)
    out = uest.merge_outputs(src, 5, 'half')
    np.testing.assert_array_equal(out, olab.merge_outputs(src, 5, 'half'))
    # empty input
    assert uest.merge_outputs(np.zeros((3, 0), dtype=np.uint8), 5, 'all').shape == (0,)
    # 20 / 21-class label spaces (a Cityscapes / Pascal source model relabelling its own domain): identity vote + histogram
    for ncls in (20, 21, 32):
        src = rng.randint(0, ncls, size=(2, 5000)).astype(np.uint8)
        t = torch.from_numpy(src).to(DEV)
        hist = torch.zeros(ncls, dtype=torch.int64, device=DEV)
        out = ops.merge_labels([t[0]], ncls, 1, 4, hist).cpu().numpy()
        np.testing.assert_array_equal(out, src[0])
        np.testing.assert_array_equal(hist.cpu().numpy(), np.bincount(src[0], minlength=ncls))
        out2 = ops.merge_labels([t[0], t[1]], ncls, 2, 4).cpu().numpy()
        np.testing.assert_array_equal(out2, olab.merge_outputs(src, ncls, 'all'))


@pytest.mark.parametrize('C', [5, 13, 20])
def test_uncertainty_estimator_vs_reference(C, golden):
    """get_output's softmax + PixelwiseKLD on full-resolution logits (identity upsample)."""
    from mspl_amd import ops
    g = golden('labels')
    d1 = (synth_input((2, C, 12, 20), 50 + C) * 3).to(DEV)
    d2 = (synth_input((2, C, 12, 20), 70 + C) * 3).to(DEV)
    r = ops.label_epilogue(d1, d2, (12, 20), want_prob=True, want_kld=True, want_logits=True)
    torch.testing.assert_close(r['kld'].cpu(), torch.from_numpy(g['kld_C%d' % C]), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(r['prob'].cpu(), torch.from_numpy(g['prob_C%d' % C]), rtol=1e-4, atol=1e-6)
    assert torch.equal(r['main_up'], d1) and torch.equal(r['aux_up'], d2)      # same-size bilinear is the identity
    ref = np.argmax(g['prob_C%d' % C].transpose(0, 2, 3, 1), axis=3).astype(np.uint8)
    np.testing.assert_array_equal(r['labels'].cpu().numpy(), ref)


@pytest.mark.parametrize('C', [5, 13, 20])
def test_argmax_rule_bound_on_adversarial_logits(C, golden):
    """The label kernel takes the first maximum of the LOGITS pred + 0.5 aux (softmax is monotone); the reference takes
    np.argmax(softmax(...)) in fp32 (uest_seg_multi_os.py:687-691,798), where two logits a few ulp apart can receive the SAME
    probability and the first maximum then picks the lower class id.  On logits built to sit there (top two 0..4 ulp apart, four
    magnitude bands; reference outcome in tests/golden/argmax_adversarial.npz) the two rules must agree wherever the reference's two
    probabilities differ, and where they are equal the kernel's class must be the one with the LARGER (or equal and lower-id) logit --
    i.e. the divergence is confined to exact fp32 probability ties, and its size on this adversarial set is bounded here (DESIGN
    section 2; on random logits it is 0 in 2 M pixels)."""
    from mspl_amd import ops
    from tests.synth import synth_adversarial_logits
    g = golden('argmax_adversarial')
    pred, aux, a, b, k = synth_adversarial_logits(C, C)
    r = ops.label_epilogue(pred.to(DEV), aux.to(DEV), (8, 8), want_logits=True)
    assert torch.equal(r['main_up'].cpu(), pred) and torch.equal(r['aux_up'].cpu(), aux)       # same-size bilinear = identity
    got = r['labels'].cpu().numpy()
    want, pa, pb = g['C%d.amax' % C], g['C%d.pa' % C], g['C%d.pb' % C]
    z = (pred + 0.5 * aux).numpy()
    assert np.array_equal(got, z.argmax(axis=1).astype(np.uint8))           # first maximum of the logits, bit for bit
    tied = pa == pb
    assert np.array_equal(got[~tied], want[~tied])
    diverged = got != want
    assert not np.any(diverged & ~tied)
    assert np.all(np.abs(k[diverged]) >= 1) and np.all(got[diverged] > want[diverged])      # only "higher id, larger logit, same probability"
    assert diverged.mean() < 0.2           # measured 0.13-0.15 on this set


def test_argmax_first_max_tie_rule_and_lut():
    from mspl_amd import ops
    main = torch.zeros((1, 13, 4, 8), device=DEV)
    main[0, 5, :, :4] = 1.0
    main[0, 9, :, :4] = 1.0           # tie between 5 and 9 -> 5 ; all-zero pixels -> class 0
    lut = torch.from_numpy(olab.ID_CAMVID_TO_GREENHOUSE.astype(np.uint8)).to(DEV)
    r = ops.label_epilogue(main, None, (4, 8), lut=lut)
    exp = np.zeros((1, 4, 8), dtype=np.uint8)
    exp[..., :4] = olab.ID_CAMVID_TO_GREENHOUSE[5]
    exp[..., 4:] = olab.ID_CAMVID_TO_GREENHOUSE[0]
    np.testing.assert_array_equal(r['labels'].cpu().numpy(), exp)


def test_pseudo_label_pass_three_sources_vs_oracle():
    """BASELINE config 3 (small shape): CamVid(13)+Cityscapes(20)+Forest(5) models -> LUT -> merge -> histogram."""
    from mspl_amd import uest
    x = synth_input((2, 3, 64, 96), 9)
    specs = [(13, 'camvid', 'camvid'), (20, 'city', 'cityscapes'), (5, 'forest', 'forest')]
    nets, sds = [], []
    for i, (C, ds, _) in enumerate(specs):
        m = _build_model('espdnetue', 2.0, C, ds)
        sd = synth_state_dict(KEYS['espdnetue_s2.0_c%d' % C], 60 + i)
        m.load_state_dict(sd)
        nets.append(m)
        sds.append(sd)
    for policy in ('all', 'half'):
        for use_graph in (False, True):
            p = uest.PseudoLabelPass(nets, [s[2] for s in specs], merge_label_policy=policy, device=DEV,
                                     use_graph=use_graph)
            merged = p(x).clone().cpu().numpy()
            maps = [t.cpu().numpy() for t in p.source_maps(x)]
            p.reset()
            merged2 = p(x).clone().cpu().numpy()          # graph replay gives the same answer
            np.testing.assert_array_equal(merged, merged2)
            hist = p.hist.cpu().numpy()
            with torch.no_grad():
                for (C, ds, od), sd, got in zip(specs, sds, maps):
                    main, aux = onet.espdnet_ue_forward(sd, x)
                    prob, _ = olab.get_output(main, aux)
                    ref = olab.to_greenhouse(olab.argmax_labels(prob), od)
                    assert (got == ref).mean() > 0.999
            ref_merged = olab.merge_outputs(np.stack(maps), 5, policy)          # integer stage: bit-exact
            np.testing.assert_array_equal(merged, ref_merged.astype(np.uint8))
            np.testing.assert_array_equal(hist, olab.class_histogram(ref_merged).astype(np.int64))


def test_get_output_dropin_signature():
    from mspl_amd import uest
    m = _build_model('espdnetue', 2.0, 5, 'greenhouse')
    sd = synth_state_dict(KEYS['espdnetue_s2.0_c5'], 70)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    x = synth_input((2, 3, 32, 48), 7)
    out, kld = uest.get_output(m, x, device=DEV)
    assert out.shape == (5, 32, 48) and kld.shape == (32, 48)            # batch element 0 only, like the reference
    with torch.no_grad():
        main, aux = onet.espdnet_ue_forward(sd, x)
        prob, k = olab.get_output(main, aux)
    np.testing.assert_allclose(out, prob[0].numpy(), rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose(kld, k[0].numpy(), rtol=2e-3, atol=2e-4)


def test_input_not_multiple_of_16_raises_like_reference():
    m = _build_model('espdnetue', 2.0, 5, 'greenhouse').to(DEV).eval()
    with torch.no_grad(), pytest.raises(RuntimeError, match='must match'):
        m(torch.randn(1, 3, 360, 480, device=DEV))


@pytest.mark.parametrize('name', sorted(ASPP_CASES))
def test_aspp_heads_vs_reference_golden(name, golden):
    """BASELINE configs[4]: the DeepLabv3 ASPP heads (dense dilated 3x3 on the matrix cores, K13) against the reference's own
    output and the oracle; same state_dict keys as nn_layers/aspp.py."""
    from mspl_amd import aspp
    cls, ncls, shp, sd_seed, x_seed = ASPP_CASES[name]
    keys = json.load(open(os.path.join(GOLDEN, 'aspp_keys.json')))[name]
    m = getattr(aspp, cls)(num_classes=ncls)
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == keys
    sd = synth_state_dict(keys, sd_seed)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    x = synth_input(shp, x_seed)
    with torch.no_grad():
        y = m(x.to(DEV)).cpu()
        ref_o = onet.aspp_forward(sd, x)
    ref = torch.from_numpy(golden('aspp')[name])
    assert y.shape == ref.shape
    # 2048 (x9) products per output: fp32 summation order differs from ATen's; logits stay well inside north_star's 1e-3
    torch.testing.assert_close(y, ref, rtol=1e-4, atol=5e-4)
    torch.testing.assert_close(y, ref_o, rtol=1e-4, atol=5e-4)
    with pytest.raises(RuntimeError, match='inference-only'):
        m(x.to(DEV))


def test_dense_conv_shapes():
    """K13 alone on awkward shapes: pixel counts that are not tile multiples, fewer than 128 output channels, a channel-slice
    destination, dilation larger than the map."""
    import torch.nn.functional as F
    from mspl_amd import ops
    from mspl_amd.ops import Epi
    g = torch.Generator().manual_seed(3)
    for (N, Cin, Cout, H, W, k, d) in [(1, 64, 40, 7, 9, 3, 2), (2, 96, 130, 5, 13, 1, 1), (1, 32, 256, 11, 6, 3, 12), (3, 128, 16, 9, 9, 3, 3),
                                       (1, 64, 130, 250, 263, 3, 2)]:          # >= 512 tiles of 128 pixels: the wide-tile form
        x = torch.randn(N, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, k, k, generator=g) * (Cin * k * k) ** -0.5
        sc, sh = torch.rand(Cout + 7, generator=g) + 0.5, torch.randn(Cout + 7, generator=g) * 0.1
        ref = F.relu(F.conv2d(x, w, None, 1, d * (k // 2), d) * sc[3:3 + Cout].view(1, -1, 1, 1) + sh[3:3 + Cout].view(1, -1, 1, 1))
        dst = torch.full((N, Cout + 7, H, W), -3.0, device=DEV)
        ops.dense_conv(x.to(DEV), ops.pack_dense_weight(w.to(DEV)), k, d, Epi(sc.to(DEV), sh.to(DEV), torch.zeros(Cout + 7, device=DEV)),
                       out=(dst, 3))
        torch.testing.assert_close(dst[:, 3:3 + Cout].cpu(), ref, rtol=1e-4, atol=2e-4)
        assert torch.all(dst[:, :3] == -3.0) and torch.all(dst[:, 3 + Cout:] == -3.0)
    with pytest.raises(RuntimeError, match='multiple of 32'):
        ops.dense_conv(torch.zeros(1, 24, 4, 4, device=DEV), torch.zeros(1, 8, 24, device=DEV), 1)


@pytest.mark.parametrize('name', sorted(RGBD_CASES))
def test_rgbd_forward_vs_reference_golden(name, golden):
    """SURVEY 8f-3 / 8a-8 `model(x, x_d)`: depth encoder + fusion gates (matrix-core 1x1 over the two weight halves + blend
    kernel) against the reference's own (main, aux) and the oracle."""
    from mspl_amd import models
    classes, dataset, shp, sd_seed, in_seed, d_seed, dense, trainable = RGBD_CASES[name]
    a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
    m = models.ESPDNetwithUncertaintyEstimation(a, classes=classes, dataset=dataset, dense_fuse=dense,
                                                trainable_fusion=trainable, fix_pyr_plane_proj=True)
    sd = synth_state_dict(KEYS['espdnetue_s2.0_c%d' % classes], sd_seed)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    x, x_d = synth_input(shp, in_seed), synth_input((shp[0], 1) + tuple(shp[2:]), d_seed)
    with torch.no_grad():
        main, aux = m(x.to(DEV), x_d.to(DEV))
        o_main, o_aux = onet.espdnet_ue_forward(sd, x, x_d, dense_fuse=dense, trainable_fusion=trainable)
    g = golden('rgbd')
    for got, ref, ora in ((main, g[name + '.main'], o_main), (aux, g[name + '.aux'], o_aux)):
        torch.testing.assert_close(got.cpu(), torch.from_numpy(ref), rtol=1e-4, atol=1e-3)     # north_star: logits within 1e-3
        torch.testing.assert_close(got.cpu(), ora, rtol=1e-4, atol=1e-3)
    with torch.no_grad(), pytest.raises(RuntimeError, match='depth input'):
        m(x.to(DEV), x_d[:, :, :16].to(DEV))


@pytest.mark.parametrize('trainable', [True, False])
def test_fusion_gate_gradients(trainable):
    """FusionGate forward + backward kernels against torch autograd over the oracle's formula."""
    from mspl_amd import models
    g = torch.Generator().manual_seed(17)
    C_, shp = 8, (2, 8, 5, 7)                                   # 280 elements per image: exercises the scalar tail too
    gate = models.FusionGate(C_, is_trainable=trainable).to(DEV)
    w = torch.randn(C_, 2 * C_, 1, 1, generator=g) * 0.3
    gate.conv_1x1.conv.weight.data.copy_(w)
    rgb, dep, gy = (torch.randn(shp, generator=g) for _ in range(3))
    r1, d1 = rgb.to(DEV).requires_grad_(), dep.to(DEV).requires_grad_()
    out = gate(r1, d1)
    out.backward(gy.to(DEV))
    r0, d0, w0 = rgb.clone().requires_grad_(), dep.clone().requires_grad_(), w.clone().requires_grad_()
    ref = onet.fusion_gate(r0, d0, {'g.conv_1x1.conv.weight': w0}, 'g', trainable)
    ref.backward(gy)
    torch.testing.assert_close(out.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(r1.grad.cpu(), r0.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(d1.grad.cpu(), d0.grad, rtol=1e-4, atol=1e-5)
    if trainable:
        torch.testing.assert_close(gate.conv_1x1.conv.weight.grad.cpu(), w0.grad, rtol=1e-4, atol=1e-5)
    with torch.no_grad():
        torch.testing.assert_close(gate(r1, d1).cpu(), ref.detach(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('name', sorted(ESPDNET_CASES))
def test_espdnet_vs_reference_golden(name, golden):
    """the single-head `espdnet` variant (model/segmentation/espdnet.py), same state_dict keys, with and
    without a depth image, against the reference's own logits and the oracle."""
    from mspl_amd import models
    classes, dataset, shp, sd_seed, in_seed, d_seed, dense, trainable = ESPDNET_CASES[name]
    keys = json.load(open(os.path.join(GOLDEN, 'espdnet_keys.json')))[name]
    m = models.ESPDNetSegmentation(argparse.Namespace(s=2.0, channels=3, num_classes=1000), classes=classes,
                                   dataset=dataset, dense_fuse=dense, trainable_fusion=trainable)
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == keys
    sd = synth_state_dict(keys, sd_seed)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    x = synth_input(shp, in_seed)
    x_d = None if d_seed is None else synth_input((shp[0], 1) + tuple(shp[2:]), d_seed)
    with torch.no_grad():
        y = m(x.to(DEV), None if x_d is None else x_d.to(DEV)).cpu()
        ora = onet.espdnet_forward(sd, x, x_d, dense_fuse=dense, trainable_fusion=trainable)
    torch.testing.assert_close(y, torch.from_numpy(golden('rgbd')[name]), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(y, ora, rtol=1e-4, atol=1e-3)
