"""Training-step parity on the GPU: autograd Functions (HIP forward + HIP backward) against torch-CPU autograd of the
oracle, and one full uest self-training step against the reference's own golden (loss, per-parameter gradient norms,
which parameters receive no gradient, parameters after one Adam step)."""
import argparse
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import labels as olab
from tests.cases import TRAIN_CASE, TRAIN_CASES
from tests.synth import assert_weights_close_after_adam
from tests.conftest import GOLDEN
from tests.synth import synth_input, synth_labels, synth_state_dict

pytestmark = pytest.mark.gpu
DEV = 'cuda'
KEYS = json.load(open(os.path.join(GOLDEN, 'state_dict_keys.json')))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def close(a, b, atol=2e-5, rtol=1e-4):
    torch.testing.assert_close(a.detach().cpu(), b.detach().cpu(), rtol=rtol, atol=atol)


def grads_of(fn, inputs, seed=99):
    outs = fn(*inputs)
    go = rnd(*outs.shape, seed=seed).to(outs.device)
    return outs, torch.autograd.grad(outs, [i for i in inputs if i.requires_grad], go, allow_unused=True)


def check_op(gpu_fn, cpu_fn, tensors, atol=2e-4, rtol=2e-3):
    cpu_in = [t.clone().requires_grad_(t.is_floating_point()) for t in tensors]
    gpu_in = [t.clone().to(DEV).requires_grad_(t.is_floating_point()) for t in tensors]
    yc, gc = grads_of(cpu_fn, cpu_in)
    yg, gg = grads_of(gpu_fn, gpu_in)
    torch.testing.assert_close(yg.detach().cpu(), yc.detach(), rtol=1e-4, atol=5e-5)
    for a, b in zip(gg, gc):
        torch.testing.assert_close(a.cpu(), b, rtol=rtol, atol=atol)


@pytest.mark.parametrize('cfg', [(2, 32, 24, 4, 9, 13, 1, 1), (1, 64, 64, 4, 6, 10, 1, 1), (2, 3, 8, 1, 12, 16, 3, 2),
                                 (1, 16, 16, 16, 11, 9, 3, 1), (1, 20, 4, 4, 7, 7, 3, 1), (1, 24, 12, 4, 8, 8, 3, 2),
                                 # 1x1 weight gradient on the matrix cores: several 32-tiles, partial tiles, pixel tails, slices
                                 (2, 128, 96, 1, 8, 12, 1, 1), (3, 512, 512, 4, 18, 30, 1, 1), (2, 48, 40, 1, 6, 10, 1, 1),
                                 (1, 64, 256, 2, 36, 60, 1, 1),
                                 # 3x3 weight gradient, strip form (stride 1, rows of whole 16-byte strips): cin_g = 1, 8, 5, 4, 3, 2
                                 (2, 16, 16, 16, 12, 16, 3, 1), (1, 16, 8, 2, 9, 20, 3, 1), (2, 10, 4, 2, 6, 8, 3, 1), (1, 12, 6, 3, 7, 12, 3, 1),
                                 (1, 6, 4, 2, 5, 4, 3, 1), (2, 8, 8, 4, 10, 24, 3, 1), (3, 32, 12, 4, 33, 68, 3, 1),
                                 # the stem's form: stride 2, three input channels, rows of whole 32-byte blocks
                                 (2, 3, 16, 1, 12, 16, 3, 2), (1, 3, 32, 1, 30, 40, 3, 2)])
def test_conv_fn(cfg):
    from mspl_amd import autograd as ag
    N, ci, co, g, h, w, k, s = cfg
    x, wt = rnd(N, ci, h, w, seed=1), rnd(co, ci // g, k, k, seed=2, scale=0.3)
    check_op(lambda a, b: ag.conv(a, b, s, g), lambda a, b: F.conv2d(a, b, None, s, (k - 1) // 2, 1, g), [x, wt])


@pytest.mark.parametrize('dil,stride,shape', [([1, 2, 3, 4], 1, (2, 8, 12, 20)), ([1, 1, 2, 3], 1, (1, 16, 9, 15)),
                                              ([1, 2, 3, 4], 2, (2, 4, 16, 24)), ([1, 2, 3, 4], 2, (1, 8, 15, 21)),
                                              ([1, 1, 2, 3], 2, (2, 6, 10, 36)), ([1, 1, 1, 2], 2, (1, 3, 6, 4))])
def test_eesp_dw_fn(dil, stride, shape):
    from mspl_amd import autograd as ag
    N, n, h, w = shape
    x = rnd(*shape, seed=1)
    ws = [rnd(n, 1, 3, 3, seed=10 + k, scale=0.3) for k in range(4)]

    def cpu(a, w0, w1, w2, w3):
        outs = []
        for k, wk in enumerate((w0, w1, w2, w3)):
            o = F.conv2d(a, wk, None, stride, dil[k], dil[k], n)
            outs.append(o if k == 0 else o + outs[-1])
        return torch.cat(outs, 1)
    check_op(lambda a, w0, w1, w2, w3: ag.eesp_dw(a, [w0, w1, w2, w3], dil, stride), cpu, [x] + ws)


@pytest.mark.parametrize('cfg', [(2, 64, 16, 4, 9, 12), (1, 128, 32, 4, 18, 30), (3, 32, 8, 1, 7, 5)])
def test_conv_skip_fn(cfg):
    """autograd.ConvSkipFn: projection + skip alias on one node (the data gradient adds the skip gradient in its epilogue) against torch."""
    from mspl_amd import autograd as ag
    N, ci, co, g, h, w = cfg
    x, wt, m = rnd(N, ci, h, w, seed=1), rnd(co, ci // g, 1, 1, seed=2, scale=0.3), rnd(N, ci, h, w, seed=3)

    def gpu(a, b, mm):
        y, skip = ag.conv_skip(a, b, g)
        return torch.cat([y, skip * mm], 1)

    check_op(gpu, lambda a, b, mm: torch.cat([F.conv2d(a, b, None, 1, 0, 1, g), a * mm], 1), [x, wt, m])


def test_affine_prelu_fn():
    from mspl_amd import autograd as ag
    c, pre, res = rnd(2, 6, 9, 11, seed=1), rnd(2, 6, 9, 11, seed=2), rnd(2, 6, 9, 11, seed=3)
    sc, sh, al = rnd(6, seed=4).abs() + 0.5, rnd(6, seed=5), rnd(6, seed=6).abs() * 0.3
    cpu = lambda c_, sc_, sh_, al_, pre_, res_: F.prelu((c_ + pre_) * sc_.view(1, -1, 1, 1) + sh_.view(1, -1, 1, 1) + res_, al_)
    check_op(lambda *a: ag.affine_prelu(*a), cpu, [c, sc, sh, al, pre, res])
    # no activation / no scale
    check_op(lambda c_, sh_: ag.affine_prelu(c_, None, sh_, None), lambda c_, sh_: c_ + sh_.view(1, -1, 1, 1), [c, sh])


@pytest.mark.parametrize('shape', [(2, 6, 10, 8, 12), (1, 32, 96, 64, 120), (3, 5, 3, 2, 2), (2, 4, 4, 7, 9)])
@pytest.mark.parametrize('with_reinf', [True, False])
def test_down_tail_fn(shape, with_reinf):
    """PReLU(cat[a, b] + reinf), the DownSampler tail without the concatenation (autograd.DownTailFn), forward and every gradient
    against torch autograd; the last shape (63 pixels per plane) takes the cat + affine fallback."""
    from mspl_amd import autograd as ag
    N, nin, nb, h, w = shape
    a, b = rnd(N, nin, h, w, seed=1), rnd(N, nb, h, w, seed=2)
    al = rnd(nin + nb, seed=3).abs() * 0.3
    if with_reinf:
        r = rnd(N, nin + nb, h, w, seed=4)
        check_op(lambda a_, b_, al_, r_: ag.down_tail(a_, b_, al_, r_), lambda a_, b_, al_, r_: F.prelu(torch.cat([a_, b_], 1) + r_, al_),
                 [a, b, al, r])
    else:
        check_op(lambda a_, b_, al_: ag.down_tail(a_, b_, al_), lambda a_, b_, al_: F.prelu(torch.cat([a_, b_], 1), al_), [a, b, al])


@pytest.mark.parametrize('cfg', [(2, 32, 24, 4, 9, 13, 1, 1, True), (1, 64, 64, 4, 6, 10, 1, 1, False), (2, 3, 8, 1, 12, 16, 3, 2, False),
                                 (1, 16, 16, 16, 11, 9, 3, 1, True), (1, 24, 12, 4, 8, 8, 3, 2, False), (2, 8, 13, 1, 10, 12, 1, 1, False),
                                 (2, 512, 16, 1, 18, 30, 1, 1, False),     # split-K form of the 1x1
                                 (1, 16, 16, 16, 12, 64, 3, 1, True),      # streaming depthwise 3x3
                                 (3, 512, 512, 4, 18, 30, 1, 1, True),
                                 (2, 40, 24, 1, 9, 11, 1, 1, True), (1, 80, 48, 2, 8, 12, 1, 1, False),   # ring-buffer MFMA form (K % 8 != 0 or unlisted K)
                                 (2, 36, 20, 1, 7, 9, 1, 1, True)])
def test_conv_affine_prelu_fn(cfg):
    """One-launch convolution + affine + PReLU of the training forward (the kernel also stores its bare result for the backward):
    forward, every gradient, and the stored bare result against torch; and bit-equal to the two-launch form it replaces."""
    from mspl_amd import autograd as ag
    N, ci, co, g, h, w, k, s, extras = cfg
    x, wt = rnd(N, ci, h, w, seed=1), rnd(co, ci // g, k, k, seed=2, scale=0.3)
    ho, wo = (h - 1) // s + 1, (w - 1) // s + 1
    sc, sh, al = rnd(co, seed=4).abs() + 0.5, rnd(co, seed=5), rnd(co, seed=6).abs() * 0.3
    conv = lambda a, b: F.conv2d(a, b, None, s, (k - 1) // 2, 1, g)
    if extras:
        pre, res = rnd(N, co, ho, wo, seed=7), rnd(N, co, ho, wo, seed=8)
        cpu = lambda a, b, sc_, sh_, al_, pre_, res_: F.prelu((conv(a, b) + pre_) * sc_.view(1, -1, 1, 1) + sh_.view(1, -1, 1, 1) + res_, al_)
        gpu = lambda a, b, sc_, sh_, al_, pre_, res_: ag.conv_affine_prelu(a, b, s, g, sc_, sh_, al_, pre_, res_)
        two = lambda a, b, sc_, sh_, al_, pre_, res_: ag.affine_prelu(ag.conv(a, b, s, g), sc_, sh_, al_, pre_, res_)
        tensors = [x, wt, sc, sh, al, pre, res]
    else:
        cpu = lambda a, b, sh_: conv(a, b) + sh_.view(1, -1, 1, 1)
        gpu = lambda a, b, sh_: ag.conv_affine_prelu(a, b, s, g, None, sh_, None)
        two = lambda a, b, sh_: ag.affine_prelu(ag.conv(a, b, s, g), None, sh_, None)
        tensors = [x, wt, sh]
    check_op(gpu, cpu, tensors, rtol=2e-3, atol=2e-3 if ci >= 256 else 5e-4)
    ins1 = [t.clone().to(DEV).requires_grad_(True) for t in tensors]
    ins2 = [t.clone().to(DEV).requires_grad_(True) for t in tensors]
    y1, g1 = grads_of(gpu, ins1)
    y2, g2 = grads_of(two, ins2)
    assert torch.equal(y1, y2)
    for a, b in zip(g1, g2):
        # same kernels on the same operands; the weight gradients combine ~1e3 partial products with atomics in varying order
        torch.testing.assert_close(a, b, rtol=1e-4, atol=2e-3 if ci >= 256 else 2e-4)


def test_resample_fns():
    from mspl_amd import autograd as ag
    x = rnd(2, 3, 15, 21, seed=1)
    check_op(lambda a: ag.avgpool(a), lambda a: F.avg_pool2d(a, 3, 2, 1), [x])
    for shp in ((2, 3, 16, 24), (1, 2, 6, 8), (2, 5, 34, 40)):          # even planes, rows of whole 16-byte strips: the vectorised backward
        check_op(lambda a: ag.avgpool(a), lambda a: F.avg_pool2d(a, 3, 2, 1), [rnd(*shp, seed=3)])
    check_op(lambda a: ag.bilinear(a, (30, 42)), lambda a: F.interpolate(a, (30, 42), mode='bilinear', align_corners=True), [x])
    check_op(lambda a: ag.bilinear(a, (7, 9)), lambda a: F.interpolate(a, (7, 9), mode='bilinear', align_corners=True), [x])
    check_op(lambda a: ag.adaptive_avgpool(a, (5, 5)), lambda a: F.adaptive_avg_pool2d(a, (5, 5)), [x])
    check_op(lambda a: ag.adaptive_avgpool(a, (10, 14)), lambda a: F.adaptive_avg_pool2d(a, (10, 14)), [x])
    # the pyramid's enlarging "pools" (scales 2.0 / 1.5), the x4 logits up-sampling and the x10 up-sampling of the 0.1 branch
    check_op(lambda a: ag.adaptive_avgpool(a, (30, 42)), lambda a: F.adaptive_avg_pool2d(a, (30, 42)), [x])
    check_op(lambda a: ag.adaptive_avgpool(a, (22, 31)), lambda a: F.adaptive_avg_pool2d(a, (22, 31)), [x])
    check_op(lambda a: ag.adaptive_avgpool(a, (1, 1)), lambda a: F.adaptive_avg_pool2d(a, (1, 1)), [x])
    check_op(lambda a: ag.bilinear(a, (60, 84)), lambda a: F.interpolate(a, (60, 84), mode='bilinear', align_corners=True), [x])
    y = rnd(1, 2, 5, 6, seed=2)
    check_op(lambda a: ag.bilinear(a, (64, 120)), lambda a: F.interpolate(a, (64, 120), mode='bilinear', align_corners=True), [y])
    check_op(lambda a: ag.bilinear(a, (5, 6)), lambda a: F.interpolate(a, (5, 6), mode='bilinear', align_corners=True), [y])
    # tiled separable backward (x2 / x4 and in between): several 64-column tiles, ragged tiles, a non-integer ratio, down-scaling rows
    for shp, size in (((2, 3, 20, 150), (40, 300)), ((1, 2, 13, 70), (52, 280)), ((1, 2, 17, 33), (29, 57)), ((2, 2, 9, 130), (36, 260)),
                      ((1, 3, 24, 96), (48, 191)), ((1, 2, 30, 40), (45, 100))):
        check_op(lambda a: ag.bilinear(a, size), lambda a: F.interpolate(a, size, mode='bilinear', align_corners=True), [rnd(*shp, seed=4)])
    # rows form of the wide backward: more output columns than threads, more input columns than one candidate window
    z = rnd(2, 3, 13, 24, seed=3)
    check_op(lambda a: ag.bilinear(a, (128, 300)), lambda a: F.interpolate(a, (128, 300), mode='bilinear', align_corners=True), [z])


def test_two_head_sum_fn():
    """autograd.TwoHeadSumFn = bilinear(main) + 0.5 * bilinear(aux) of the supervised loop: forward bit-identical to the three-step form,
    gradients against torch."""
    from mspl_amd import autograd as ag
    main, aux = rnd(2, 5, 16, 24, seed=1), rnd(2, 5, 8, 12, seed=2)
    size = (32, 48)
    up = lambda t: F.interpolate(t, size, mode='bilinear', align_corners=True)
    check_op(lambda a, b: ag.two_head_sum(a, b, size), lambda a, b: up(a) + 0.5 * up(b), [main, aux])
    m, a = main.to(DEV), aux.to(DEV)
    assert torch.equal(ag.two_head_sum(m, a, size), ag.bilinear(m, size) + 0.5 * ag.bilinear(a, size))


def test_gate_fns():
    from mspl_amd import autograd as ag
    x, w = rnd(2, 8, 7, 9, seed=1), rnd(6, 8, 1, 1, seed=2)
    y = rnd(2, 6, 7, 9, seed=3)

    def cpu(a, b, c):
        return c * torch.sigmoid(F.conv2d(F.adaptive_avg_pool2d(a, 1), b))
    check_op(lambda a, b, c: ag.channel_scale(c, ag.gap_gate(a, b)), cpu, [x, w, y])


def test_uw_loss_vs_reference_golden(golden):
    from mspl_amd import training
    g = golden('loss')
    pred = (synth_input((2, 5, 32, 48), 90) * 2).to(DEV).requires_grad_(True)
    aux = (synth_input((2, 5, 32, 48), 91) * 2).to(DEV).requires_grad_(True)
    tgt = synth_labels((2, 32, 48), 5, 92).to(DEV)
    loss = training.uest_loss(pred, aux, tgt, torch.from_numpy(g['cw']), ignore_idx=4)
    loss.backward()
    torch.testing.assert_close(loss.detach().cpu(), torch.from_numpy(g['loss']), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(pred.grad.cpu(), torch.from_numpy(g['dpred']), rtol=1e-3, atol=1e-8)
    torch.testing.assert_close(aux.grad.cpu(), torch.from_numpy(g['daux']), rtol=1e-3, atol=1e-8)


# (N, C, main size, aux size, label size, ignore class): the model's x2 / x4 heads, odd sizes with a ragged last band, more than one
# 256-column tile, 13 and 20 classes (exponentials recomputed), a class count on the predicated kernel, labels outside 0..C-1
HEADS_CASES = [(2, 5, (16, 24), (8, 12), (32, 48), 4), (1, 13, (9, 11), (5, 6), (18, 22), None), (1, 5, (20, 150), (10, 75), (40, 300), 0),
               (2, 20, (7, 40), (4, 20), (14, 80), 19), (1, 3, (12, 136), (6, 68), (23, 271), None), (3, 5, (5, 5), (5, 5), (5, 5), 4)]


@pytest.mark.parametrize('case', HEADS_CASES, ids=lambda c: 'n%dc%d_%dx%d' % (c[0], c[1], c[4][0], c[4][1]))
def test_uw_loss_at_head_resolution(case):
    """mspl_uw_loss_heads_fwd_bwd (loss + gradients w.r.t. the two low-resolution decoder outputs, one launch) against the oracle's
    loss on F.interpolate(..., align_corners=True) of both heads with torch autograd on the CPU, and against the three-step HIP form."""
    from mspl_amd import autograd as ag, training
    from oracle import labels as olab
    N, C, ms, as_, size, ign = case
    main = synth_input((N, C) + ms, 300 + C) * 2
    aux = synth_input((N, C) + as_, 301 + C) * 2
    tgt = synth_labels((N,) + size, C, 302 + C)
    wild = ign is None                        # these cases carry labels outside 0..C-1: weight 0, as in the three-step kernel (the
    if wild:                                  # oracle's loss is not defined on them: they are compared with the three-step form only)
        tgt[0, 0, :2] = 255
        tgt[-1, -1, -1] = -1
    cw = torch.linspace(0.5, 1.5, C)
    assert ag.uw_loss_heads_supported(C)
    cwd = training._device_class_weights(cw, torch.device(DEV), ign)
    mg, agd = main.to(DEV).requires_grad_(True), aux.to(DEV).requires_grad_(True)
    loss = ag.uw_loss_heads(mg, agd, tgt.to(DEV), cwd)
    loss.backward()
    m3, a3 = main.to(DEV).requires_grad_(True), aux.to(DEV).requires_grad_(True)
    l3 = ag.uw_loss(ag.bilinear(m3, size), ag.bilinear(a3, size), tgt.to(DEV), cwd)
    l3.backward()
    torch.testing.assert_close(loss.detach(), l3.detach(), rtol=1e-5, atol=1e-6)       # (v_exp_f32 / v_log_f32 softmax terms here, expf / logf there)
    sm, sa = float(m3.grad.abs().max()), float(a3.grad.abs().max())
    torch.testing.assert_close(mg.grad, m3.grad, rtol=1e-4, atol=2e-6 * sm)
    torch.testing.assert_close(agd.grad, a3.grad, rtol=1e-4, atol=2e-6 * sa)
    if not wild:
        mr, ar = main.clone().requires_grad_(True), aux.clone().requires_grad_(True)
        up = lambda t: F.interpolate(t, size=size, mode='bilinear', align_corners=True)      # noqa: E731
        ref = olab.uest_train_loss(up(mr), up(ar), tgt, cw, ignore_idx=ign)
        ref.backward()
        torch.testing.assert_close(loss.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(mg.grad.cpu(), mr.grad, rtol=1e-3, atol=1e-5 * sm)
        torch.testing.assert_close(agd.grad.cpu(), ar.grad, rtol=1e-3, atol=1e-5 * sa)


def test_uw_loss_at_head_resolution_scaled_and_not_root():
    """out_scale (a micro-batch lane's 1 / lanes) and an incoming gradient other than one; unsupported class counts are refused."""
    from mspl_amd import autograd as ag, training
    from oracle import labels as olab
    N, C, ms, as_, size, ign = 2, 5, (18, 30), (9, 15), (36, 60), 4
    main, aux = synth_input((N, C) + ms, 310) * 2, synth_input((N, C) + as_, 311) * 2
    tgt = synth_labels((N,) + size, C, 312)
    cw = torch.linspace(0.5, 1.5, C)
    mr, ar = main.clone().requires_grad_(True), aux.clone().requires_grad_(True)
    up = lambda t: F.interpolate(t, size=size, mode='bilinear', align_corners=True)      # noqa: E731
    ref = olab.uest_train_loss(up(mr), up(ar), tgt, cw, ignore_idx=ign)
    ref.backward()
    mg, agd = main.to(DEV).requires_grad_(True), aux.to(DEV).requires_grad_(True)
    loss = ag.uw_loss_heads(mg, agd, tgt.to(DEV), training._device_class_weights(cw, torch.device(DEV), ign), out_scale=0.25)
    (loss * 4).backward()                     # not the root: the backward multiplies by the incoming 4
    torch.testing.assert_close(loss.detach().cpu() * 4, ref.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(mg.grad.cpu(), mr.grad, rtol=1e-3, atol=1e-5 * float(mr.grad.abs().max()))
    torch.testing.assert_close(agd.grad.cpu(), ar.grad, rtol=1e-3, atol=1e-5 * float(ar.grad.abs().max()))
    assert not ag.uw_loss_heads_supported(11) and not ag.uw_loss_heads_supported(21)
    with pytest.raises(RuntimeError, match='uw_loss_heads'):
        ag.uw_loss_heads(torch.zeros(1, 11, 4, 4, device=DEV), torch.zeros(1, 11, 2, 2, device=DEV),
                         torch.zeros(1, 8, 8, dtype=torch.int64, device=DEV), torch.ones(11, device=DEV))


@pytest.mark.parametrize('loss_form', ['heads', 'upsampled'])
@pytest.mark.parametrize('gname', sorted(TRAIN_CASES))
def test_train_step_vs_reference_golden(gname, loss_form, golden):
    """ESPDNet-UE C=5, frozen BN, one Adam step: loss, gradient norms, the 230 gradient-less tensors, updated weights -- at 32x48 and
    at 64x96 (the streaming pyramid kernels, the matrix-core weight gradients and the fused EESP backward are reached by the second)."""
    from mspl_amd import models, training
    c = TRAIN_CASES[gname]
    g = golden(gname)
    a = argparse.Namespace(s=c['s'], channels=3, num_classes=1000)
    m = models.ESPDNetwithUncertaintyEstimation(a, classes=c['classes'], dataset=c['dataset'], fix_pyr_plane_proj=True)
    m.load_state_dict(synth_state_dict(KEYS['espdnetue_s%s_c%d' % (c['s'], c['classes'])], c['sd_seed']))
    m = m.to(DEV).eval()
    x = synth_input(c['shape'], c['in_seed']).to(DEV)
    labels = synth_labels((c['shape'][0],) + c['shape'][2:], c['classes'], c['in_seed']).to(DEV)
    with torch.enable_grad():
        if loss_form == 'heads':              # what train_step does: the loss at head resolution, up-sampling inside the kernel
            cwd = training._device_class_weights(torch.ones(c['classes']), torch.device(DEV), c['ignore_idx'])
            loss = training.forward_loss(m, x, labels, cwd)
            assert type(loss.grad_fn).__name__.startswith('UWLossHeadsFn')
        else:                                 # the reference's own call sequence: model(x), then the loss on the full-size logits
            pred, aux = m(x)
            loss = training.uest_loss(pred, aux, labels, torch.ones(c['classes']), ignore_idx=c['ignore_idx'])
        loss.backward()
    torch.testing.assert_close(loss.detach().cpu(), torch.from_numpy(g['loss']), rtol=2e-5, atol=1e-5)
    params = dict(m.named_parameters())
    names = [str(n) for n in g['names']]
    assert names == list(params.keys())
    n_with = 0
    for n, gn in zip(names, g['gnorm']):
        p = params[n]
        if gn < 0:
            assert p.grad is None, '%s should not receive a gradient' % n
        else:
            n_with += 1
            assert p.grad is not None, n
            got = float(p.grad.double().norm())
            assert abs(got - gn) <= 5e-3 * gn + 1e-6, (n, got, gn)
    assert n_with == 340
    # element by element on a strided sample of every gradient (norms alone would pass a sign / permutation error inside a tensor)
    from tests.synth import grad_sample_index
    off = g['gsample_off']
    for i, n in enumerate(names):
        if g['gnorm'][i] < 0:
            continue
        p = params[n]
        ref = torch.from_numpy(g['gsample'][off[i]:off[i + 1]])
        got = p.grad.detach().reshape(-1).cpu()[grad_sample_index(p.numel())]
        scale = float(g['gnorm'][i]) / max(1.0, p.numel() ** 0.5)          # typical element size of this gradient
        assert float((got - ref).abs().max()) <= 2e-3 * float(ref.abs().max()) + 5e-3 * scale + 1e-7, (n, got[:4], ref[:4])
    opt = training.FlatAdam(m.parameters(), lr=c['lr'], weight_decay=c['weight_decay'])
    assert len(opt.params) == 340 and opt.bucket.numel_params == sum(p.numel() for p in opt.params) <= opt.flat_p.numel()
    assert all(p.data_ptr() % 16 == 0 and p.grad.data_ptr() % 16 == 0 for p in opt.params)      # 16-byte vector loads of weights
    opt.step()
    torch.cuda.synchronize()
    for i, k in enumerate(str(s) for s in g['keep']):
        torch.testing.assert_close(params[k].detach().cpu(), torch.from_numpy(g['after_%d' % i]), rtol=1e-4, atol=2e-6)
    # the step went through raw pointers: inference caches must notice
    with torch.no_grad():
        y1 = m(x)[0]
    opt.step()
    with torch.no_grad():
        y2 = m(x)[0]
    assert not torch.equal(y1, y2)


def test_second_step_and_train_step_helper():
    from mspl_amd import models, training
    a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
    m = models.ESPDNetwithUncertaintyEstimation(a, classes=5, dataset='greenhouse', fix_pyr_plane_proj=True)
    m.load_state_dict(synth_state_dict(KEYS['espdnetue_s2.0_c5'], 3))
    m = m.to(DEV).eval()
    x = synth_input((2, 3, 32, 48), 8).to(DEV)
    y = synth_labels((2, 32, 48), 5, 8).to(DEV)
    cw = torch.ones(5)
    l0, opt = training.train_step(m, x, y, cw, None, ignore_idx=4, lr=2e-3)
    losses = [float(l0)]
    for _ in range(5):
        l, opt = training.train_step(m, x, y, cw, opt, ignore_idx=4)
        losses.append(float(l))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses      # same batch: the loss must go down


def test_graphed_train_step_equals_eager():
    """zero_grad + forward + loss + backward replayed as one hipGraph gives the same losses and weights as eager steps."""
    from mspl_amd import models, training
    a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
    x = synth_input((2, 3, 32, 48), 8).to(DEV)
    y = synth_labels((2, 32, 48), 5, 8).to(DEV)
    cw = torch.ones(5)
    nets = []
    for _ in range(2):
        m = models.ESPDNetwithUncertaintyEstimation(a, classes=5, dataset='greenhouse', fix_pyr_plane_proj=True)
        m.load_state_dict(synth_state_dict(KEYS['espdnetue_s2.0_c5'], 3))
        nets.append(m.to(DEV).eval())
    l, opt = training.train_step(nets[0], x, y, cw, None, ignore_idx=4)
    eager = [float(l)]
    for _ in range(3):
        l, opt = training.train_step(nets[0], x, y, cw, opt, ignore_idx=4)
        eager.append(float(l))
    gs = training.GraphedTrainStep(nets[1], x, y, cw, ignore_idx=4)      # eager step 1 + captured step 2
    graphed = [float(gs(x, y)) for _ in range(2)]
    np.testing.assert_allclose(graphed, eager[2:], rtol=2e-4, atol=1e-6)
    assert_weights_close_after_adam(nets[1].state_dict(), nets[0].state_dict(), lr=5e-4, steps=4)


@pytest.mark.parametrize('lanes', [2, 4])
def test_micro_batch_lanes_equal_one_graph(lanes):
    """The step cut into micro-batches that run as concurrent graphs on separate streams (every lane back-propagates loss / L into
    the shared gradient buffer with atomic adds) = the one-graph step: same losses, same weights after three steps, and the
    gradient buffer itself right after a step."""
    from mspl_amd import models, training
    a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
    x = synth_input((4, 3, 32, 48), 8).to(DEV)
    y = synth_labels((4, 32, 48), 5, 8).to(DEV)
    cw = torch.tensor([1.0, 0.5, 2.0, 1.0, 1.0])
    nets = []
    for _ in range(2):
        m = models.ESPDNetwithUncertaintyEstimation(a, classes=5, dataset='greenhouse', fix_pyr_plane_proj=True)
        m.load_state_dict(synth_state_dict(KEYS['espdnetue_s2.0_c5'], 3))
        nets.append(m.to(DEV).eval())
    one = training.GraphedTrainStep(nets[0], x, y, cw, ignore_idx=4)
    many = training.GraphedTrainStep(nets[1], x, y, cw, ignore_idx=4, lanes=lanes)
    assert many.lanes == lanes and len(many.lane_graphs) == lanes
    np.testing.assert_allclose(many.optimizer.flat_g.cpu().numpy(), one.optimizer.flat_g.cpu().numpy(), rtol=1e-4, atol=1e-4)
    l1 = [float(one(x, y)) for _ in range(2)]
    l2 = [float(many(x, y)) for _ in range(2)]
    np.testing.assert_allclose(l2, l1, rtol=2e-4, atol=1e-6)
    assert_weights_close_after_adam(nets[1].state_dict(), nets[0].state_dict(), lr=5e-4, steps=4)
    # a batch the lanes cannot split evenly falls back to one graph
    assert training.GraphedTrainStep(nets[1], x[:3], y[:3], cw, ignore_idx=4, lanes=2).lanes == 1


@pytest.mark.parametrize('hw', [(32, 48), (64, 96)])
def test_direct_gradient_sinks_equal_autograd_accumulation(hw, monkeypatch):
    """Steps 2-3 with the parameter-gradient kernels writing straight into the flat gradient buffer (grad_sinks, the default
    of train_step) against the same steps with autograd's own AccumulateGrad (MSPL_GRAD_SINKS=0)."""
    from mspl_amd import models, training
    a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
    # (64x96: the queued 1x1 weight gradients take the matrix-core batch kernel, the pyramids their streaming kernels)
    x = synth_input((2, 3) + hw, 18).to(DEV)
    y = synth_labels((2,) + hw, 5, 18).to(DEV)
    cw = torch.ones(5)
    outs = []
    for flag in ('1', '0'):
        monkeypatch.setenv('MSPL_GRAD_SINKS', flag)
        m = models.ESPDNetwithUncertaintyEstimation(a, classes=5, dataset='greenhouse', fix_pyr_plane_proj=True)
        m.load_state_dict(synth_state_dict(KEYS['espdnetue_s2.0_c5'], 4))
        m = m.to(DEV).eval()
        opt, losses, grads, weights = None, [], [], []
        for _ in range(3):
            l, opt = training.train_step(m, x, y, cw, opt, ignore_idx=4)
            losses.append(float(l))
            grads.append(opt.flat_g.clone())
            weights.append(opt.flat_p.clone())
        outs.append((losses, grads, weights))
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=1e-5)
    # Step 2 is the first one through the sinks and starts from weights that differ by float atomics' order only: tight.
    # By step 3 an ulp of difference in a weight can put one PReLU input of a 2x3 level-4 map on the other side of zero
    # (measured: either 5e-7 or one block at 1e-2 of ITS OWN gradient, 6e-5 of the largest, run to run with the same
    # settings on either side - tools/diag_sinks.py), so the third comparison is looser.
    for step, tol in ((1, 1e-5), (2, 5e-4)):
        g1, g0 = outs[0][1][step], outs[1][1][step]
        assert float((g1 - g0).abs().max()) <= tol * float(g0.abs().max()) + 1e-7, step
    # Weights after the second Adam step: see assert_weights_close_after_adam for what two runs of the same steps can differ by.
    assert_weights_close_after_adam(outs[0][2][1], outs[1][2][1], lr=5e-4, steps=2, tight=2e-5)


def _ref_losses():
    """Torch-CPU statements of the three loss modules (loss_fns/segmentation_loss.py:11-52,146-189)."""
    import torch.nn.functional as F

    def kld(d1, d2):
        p1 = F.softmax(d1, 1)
        return (p1 * F.log_softmax(d1, 1) - p1 * F.log_softmax(d2, 1)).sum(1)

    def uw(pred, target, u, cw):
        n, c, h, w = pred.shape
        lp = -F.log_softmax(pred, 1) * cw.view(1, c, 1, 1)
        return (lp.gather(1, target.view(n, 1, h, w)) * torch.exp(-u.view(n, 1, h, w))).mean()
    return kld, uw


def test_loss_modules_forward_backward():
    """PixelwiseKLD / UncertaintyWeightedSegmentationLoss / SegmentationLoss drop-ins: values and gradients vs torch CPU,
    composed the way uest_seg_multi_os.py:1020-1023 composes them, and equal to the fused K11 form."""
    from mspl_amd import losses, training
    kld_ref, uw_ref = _ref_losses()
    N, C, H, W = 2, 5, 24, 40
    pred = rnd(N, C, H, W, seed=1) * 2
    aux = rnd(N, C, H, W, seed=2) * 2
    y = synth_labels((N, H, W), C, 3)
    cw = torch.tensor([1.0, 0.5, 2.0, 1.5, 1.0])
    # reference composition on CPU
    pr, ar = pred.clone().requires_grad_(), aux.clone().requires_grad_()
    cwr = cw.clone(); cwr[4] = 0.0
    k = kld_ref(pr, ar)
    ref = uw_ref(pr + 0.5 * ar, y, k, cwr) * 20 + k.mean()
    ref.backward()
    # drop-in modules on the GPU
    pg, agd = pred.to(DEV).requires_grad_(), aux.to(DEV).requires_grad_()
    crit = losses.UncertaintyWeightedSegmentationLoss(C, class_wts=cw.clone(), ignore_idx=4, device=DEV)
    kg = losses.PixelwiseKLD()(pg, agd)
    close(kg.detach(), k.detach(), atol=2e-6)
    out = crit(pg + 0.5 * agd, y.to(DEV), kg) * 20 + kg.mean()
    out.backward()
    close(out.detach(), ref.detach(), atol=1e-5)
    close(pg.grad, pr.grad, atol=2e-7, rtol=2e-4)
    close(agd.grad, ar.grad, atol=2e-7, rtol=2e-4)
    fused = training.uest_loss(pred.to(DEV), aux.to(DEV), y.to(DEV), cw, ignore_idx=4)
    close(fused, ref.detach(), atol=1e-5)
    # SegmentationLoss == nn.CrossEntropyLoss(weight, ignore_index), single tensor and (main, aux) tuple
    y255 = y.clone(); y255[0, :3] = 255
    ce = torch.nn.CrossEntropyLoss(ignore_index=255, weight=cw)
    pr2, ar2 = pred.clone().requires_grad_(), aux.clone().requires_grad_()
    r2 = ce(pr2, y255) + ce(ar2, y255)
    r2.backward()
    pg2, ag2 = pred.to(DEV).requires_grad_(), aux.to(DEV).requires_grad_()
    seg = losses.SegmentationLoss(n_classes=C, device=DEV, ignore_idx=255, class_weights=cw)
    o2 = seg((pg2, ag2), y255.to(DEV))
    o2.backward()
    close(o2.detach(), r2.detach(), atol=1e-5)
    close(pg2.grad, pr2.grad, atol=2e-7, rtol=2e-4)
    close(ag2.grad, ar2.grad, atol=2e-7, rtol=2e-4)
    with pytest.raises(RuntimeError, match='no CPU path'):
        losses.PixelwiseKLD()(pred, aux)
    with pytest.raises(RuntimeError, match="only 'ce'"):
        losses.SegmentationLoss(loss_type='bce')


def test_weight_transposer_equals_on_the_spot_copies():
    """One-launch transposed weights (autograd.WeightTransposer, the data-gradient convolutions of a step) against the
    permute / flip copies ConvFn.backward makes without it; and the table is only read inside `active()`."""
    from mspl_amd import autograd as ag
    torch.manual_seed(3)
    ws = [(torch.randn(64, 16, 1, 1, device='cuda'), 4, 1), (torch.randn(48, 8, 3, 3, device='cuda'), 16, 3),
          (torch.randn(16, 1, 3, 3, device='cuda'), 16, 3), (torch.randn(13, 16, 1, 1, device='cuda'), 1, 1),
          (torch.randn(300, 7, 3, 3, device='cuda'), 2, 3)]
    want = [ag._transposed_weights(w, g, k).clone() for w, g, k in ws]
    tr = ag.WeightTransposer(ws + ws[:2])                  # duplicates collapse
    assert len(tr.items) == len(ws)
    with tr.active():
        for (w, g, k), ref in zip(ws, want):
            got = ag._transposed_weights(w, g, k)
            assert got.data_ptr() != w.data_ptr() and got.data_ptr() >= tr.flat.data_ptr()
            assert torch.equal(got, ref)
        ws[0][0].mul_(2.0)                                  # weights change: the next activation refreshes the copies
    assert ag._WT_ACTIVE[0] is None
    with tr.active():
        assert torch.equal(ag._transposed_weights(ws[0][0], 4, 1), want[0] * 2.0)
    fresh = ag._transposed_weights(ws[1][0], 16, 3)        # outside the scope: made on the spot
    assert fresh.data_ptr() < tr.flat.data_ptr() or fresh.data_ptr() >= tr.flat.data_ptr() + tr.flat.numel() * 4
    assert torch.equal(fresh, want[1])


def test_fan_out_fn_sums_branch_gradients_in_one_launch():
    """autograd.fan_out: n aliases of a tensor, their gradients summed by mspl_sum_n; equal to autograd's own pairwise accumulation
    (same order of additions), incl. an unused alias and the pass-through cases."""
    from mspl_amd import autograd as ag
    x = rnd(2, 6, 9, 12, seed=1).to(DEV).requires_grad_(True)
    ws = [rnd(2, 6, 9, 12, seed=10 + i).to(DEV) for i in range(5)]
    a = ag.fan_out(x, 5)
    assert len(a) == 5 and all(t.data_ptr() == x.data_ptr() for t in a)
    (a[0] * ws[0] + a[1] * ws[1] + a[2] * ws[2] + a[4] * ws[4]).sum().backward()      # a[3] unused
    got = x.grad.clone()
    want = ((ws[0] + ws[1]) + ws[2]) + ws[4]
    assert torch.equal(got, want)
    y = torch.zeros(3, device=DEV)                                                     # no gradient wanted: plain aliases
    assert ag.fan_out(y, 3)[1] is y
    z = rnd(1, 1, 3, 3, seed=2).to(DEV).requires_grad_(True)                           # 9 elements: not a multiple of 4 -> ATen adds
    b = ag.fan_out(z, 2)
    (b[0] * 2 + b[1] * 3).sum().backward()
    assert torch.equal(z.grad, torch.full_like(z, 5.0))


def _pyr_module(in_planes, out_planes, last_layer_br, seed):
    from mspl_amd import layers
    m = layers.EfficientPyrPool(in_planes, 16, out_planes, last_layer_br=last_layer_br)
    m.load_state_dict(synth_state_dict(m.state_dict(), seed))
    return m.to(DEV).eval()


@pytest.mark.parametrize('cfg', [(2, 32, 24, 16, 30, True), (2, 24, 5, 32, 60, False), (1, 16, 13, 64, 120, False),
                                 (2, 16, 20, 18, 34, True), (1, 48, 32, 40, 72, True), (3, 16, 5, 17, 33, False),
                                 (1, 16, 8, 8, 12, True)])
def test_fused_pyramid_training_equals_node_per_op(cfg):
    """autograd.PyrBodyFn (one fused forward launch, mspl_pyrpool_merge_bwd + mspl_pyrpool_branch_bwd in the backward) against the
    node-per-op training path it replaces: output, input gradient and the gradient of every parameter of the module."""
    from mspl_amd import autograd as ag, layers
    N, cin, cout, h, w, last_br = cfg
    m = _pyr_module(cin, cout, last_br, 77)
    assert ag.pyr_body_fits((N, 16, h, w), m.branch_sizes(h, w))
    x = rnd(N, cin, h, w, seed=5).to(DEV)
    go = rnd(N, cout, h, w, seed=6).to(DEV)
    res = {}
    for fused in (False, True):
        prev = layers._FUSED_PYR_TRAIN
        layers._FUSED_PYR_TRAIN = fused
        try:
            xi = x.clone().requires_grad_(True)
            for p in m.parameters():
                p.grad = None
            with torch.enable_grad():
                y = m(xi)
                y.backward(go)
            res[fused] = (y.detach().clone(), xi.grad.clone(), {k: p.grad.clone() for k, p in m.named_parameters()})
        finally:
            layers._FUSED_PYR_TRAIN = prev
    close(res[True][0], res[False][0], atol=2e-5, rtol=1e-4)
    close(res[True][1], res[False][1], atol=5e-5, rtol=1e-3)
    for k in res[False][2]:
        a, b = res[True][2][k], res[False][2][k]
        scale = float(b.abs().max()) + 1e-6
        assert float((a - b).abs().max()) <= 2e-3 * scale + 1e-5, (k, float((a - b).abs().max()), scale)


@pytest.mark.parametrize('cfg', [(2, 32, 24, 16, 30, True), (2, 24, 5, 32, 60, False), (1, 16, 13, 64, 120, False),
                                 (2, 16, 20, 18, 34, True), (4, 48, 32, 40, 72, True), (3, 16, 8, 8, 12, True)])
def test_fused_pyramid_batch_statistics_equals_node_per_op(cfg):
    """autograd.PyrBodyBNFn (BatchNorms in train(): the supervised loop) against the node-per-op training path: output, input gradient,
    every parameter gradient, and the BatchNorms' buffers (running statistics, num_batches_tracked) after the step."""
    from mspl_amd import autograd as ag, layers
    N, cin, cout, h, w, last_br = cfg
    m = _pyr_module(cin, cout, last_br, 79).train()
    assert ag.pyr_body_fits((N, 16, h, w), m.branch_sizes(h, w)) and (h * w) % 4 == 0
    x = rnd(N, cin, h, w, seed=15).to(DEV)
    go = rnd(N, cout, h, w, seed=16).to(DEV)
    start = {k: v.clone() for k, v in m.state_dict().items()}
    res, calls = {}, []
    orig = ag.PyrBodyBNFn.apply
    for fused in (False, True):
        prev = layers._FUSED_PYR_TRAIN
        layers._FUSED_PYR_TRAIN = fused
        m.load_state_dict(start)
        ag.PyrBodyBNFn.apply = lambda *a: (calls.append(fused), orig(*a))[1]
        try:
            xi = x.clone().requires_grad_(True)
            for p in m.parameters():
                p.grad = None
            with torch.enable_grad():
                y = m(xi)
                y.backward(go)
            res[fused] = (y.detach().clone(), xi.grad.clone(), {k: p.grad.clone() for k, p in m.named_parameters()},
                          {k: v.clone() for k, v in m.named_buffers()})
        finally:
            layers._FUSED_PYR_TRAIN = prev
            ag.PyrBodyBNFn.apply = orig
    assert calls == [True]
    close(res[True][0], res[False][0], atol=5e-5, rtol=2e-4)
    close(res[True][1], res[False][1], atol=1e-4, rtol=2e-3)
    for k in res[False][2]:
        a, b = res[True][2][k], res[False][2][k]
        scale = float(b.abs().max()) + 1e-6
        assert float((a - b).abs().max()) <= 3e-3 * scale + 2e-5, (k, float((a - b).abs().max()), scale)
    for k, b in res[False][3].items():
        a = res[True][3][k]
        if b.dtype == torch.int64:
            assert torch.equal(a, b), k
        else:
            close(a, b, atol=1e-6, rtol=1e-5)


@pytest.mark.parametrize('cfg', [(2, 64, 64, 7, 12, 20, 1), (1, 128, 128, 9, 9, 15, 1), (3, 256, 256, 9, 18, 30, 1), (16, 64, 64, 9, 36, 60, 1),
                                 (2, 32, 96, 13, 16, 24, 2), (1, 64, 64, 9, 15, 21, 2), (1, 16, 16, 9, 40, 300, 1)])
def test_eesp_block_batch_statistics_fused_k2_node(cfg):
    """autograd.EespDwBNFn (K2 + br_after_cat in train(): BatchNorm sums launch + mspl_eesp_bwd_fused_bnstat) against the node-per-op
    form of the same block: output, input gradient, every parameter gradient, BatchNorm buffers."""
    from mspl_amd import layers
    N, cin, cout, r_lim, h, w, stride = cfg
    m = layers.EESP(cin, cout, stride=stride, r_lim=r_lim, down_method='avg' if stride == 2 else 'esp')
    m.load_state_dict(synth_state_dict(m.state_dict(), 43))
    m = m.to(DEV).train()
    x = rnd(N, cin, h, w, seed=3).to(DEV)
    go = rnd(N, cout, (h - 1) // stride + 1, (w - 1) // stride + 1, seed=4).to(DEV)
    start = {k: v.clone() for k, v in m.state_dict().items()}
    res = {}
    for fused in (False, True):
        prev = layers._EESP_DW_BN
        layers._EESP_DW_BN = fused
        m.load_state_dict(start)
        try:
            xi = x.clone().requires_grad_(True)
            for p in m.parameters():
                p.grad = None
            with torch.enable_grad():
                y = m(xi)
                y.backward(go)
            res[fused] = (y.detach().clone(), xi.grad.clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None},
                          {k: v.clone() for k, v in m.named_buffers()})
        finally:
            layers._EESP_DW_BN = prev
    close(res[True][0], res[False][0], atol=2e-5, rtol=1e-4)
    close(res[True][1], res[False][1], atol=1e-4, rtol=2e-3)
    for k in res[False][2]:
        a, b = res[True][2][k], res[False][2][k]
        scale = float(b.abs().max()) + 1e-6
        assert float((a - b).abs().max()) <= 3e-3 * scale + 2e-5, (k, float((a - b).abs().max()), scale)
    for k, b in res[False][3].items():
        a = res[True][3][k]
        assert torch.equal(a, b) if b.dtype == torch.int64 else bool(torch.allclose(a, b, rtol=1e-5, atol=1e-6)), k


def test_fused_pyramid_training_with_gradient_sinks():
    """Inside grad_sinks() the fused node adds its parameter gradients straight into existing .grad buffers (the flat optimizer
    buffers of the train step): same values as the returned-gradient form, accumulated on top of what the buffers held."""
    from mspl_amd import autograd as ag
    m = _pyr_module(32, 24, True, 78)
    x = rnd(2, 32, 16, 30, seed=8).to(DEV)
    go = rnd(2, 24, 16, 30, seed=9).to(DEV)
    with torch.enable_grad():
        m(x).backward(go)
    ref = {k: p.grad.clone() for k, p in m.named_parameters()}
    for p in m.parameters():
        p.grad = torch.full_like(p, 0.5)
    with torch.enable_grad(), ag.grad_sinks():
        m(x).backward(go)
    for k, p in m.named_parameters():
        close(p.grad - 0.5, ref[k], atol=2e-4, rtol=2e-3)


@pytest.mark.parametrize('cfg', [(2, 64, 64, 1, 7, 12, 20), (1, 128, 128, 1, 9, 9, 15), (2, 32, 96, 2, 13, 16, 24), (1, 64, 64, 2, 9, 15, 21),
                                 (3, 256, 256, 1, 9, 18, 30)])
def test_fused_eesp_block_training_equals_node_per_op(cfg):
    """autograd.EESPFn (whole block as one node: residual gradient in the projection's data-gradient epilogue, BatchNorm/PReLU backward
    fused with the HFF suffix sum, branch weights as one view) against the node-per-op path: output, input gradient, every parameter
    gradient.  Stride 2 is the DownSampler's EESP (down_method='avg': no residual, no module_act)."""
    from mspl_amd import layers
    N, cin, cout, stride, r_lim, h, w = cfg
    m = layers.EESP(cin, cout, stride=stride, r_lim=r_lim, down_method='avg' if stride == 2 else 'esp')
    m.load_state_dict(synth_state_dict(m.state_dict(), 41))
    m = m.to(DEV).eval()
    x = rnd(N, cin, h, w, seed=3).to(DEV)
    res = {}
    for fused in (False, True):
        prev = layers._FUSED_EESP_TRAIN
        layers._FUSED_EESP_TRAIN = fused
        try:
            xi = x.clone().requires_grad_(True)
            for p in m.parameters():
                p.grad = None
            with torch.enable_grad():
                y = m(xi)
                go = rnd(*y.shape, seed=4).to(DEV)
                y.backward(go)
            res[fused] = (y.detach().clone(), xi.grad.clone(), {k: (None if p.grad is None else p.grad.clone()) for k, p in m.named_parameters()})
        finally:
            layers._FUSED_EESP_TRAIN = prev
    assert torch.equal(res[True][0], res[False][0])
    close(res[True][1], res[False][1], atol=5e-5, rtol=1e-3)
    for k, b in res[False][2].items():
        a = res[True][2][k]
        assert (a is None) == (b is None), k
        if b is not None:
            scale = float(b.abs().max()) + 1e-6
            assert float((a - b).abs().max()) <= 2e-3 * scale + 1e-5, (k, float((a - b).abs().max()), scale)


@pytest.mark.parametrize('cfg', [(2, 16, 16, 30), (1, 16, 144, 240), (3, 8, 17, 33), (2, 16, 36, 60), (1, 4, 9, 7)])
def test_pyramid_low_resolution_branches_one_launch_each_way(cfg):
    """The scale < 1 pyramid branches (nn_layers/efficient_pyramid_pool.py:44-47) in the training step: ops.pyr_down_prep(keep_pooled=True)
    (one forward launch: dw3x3(adaptive_avg_pool2d(x)) AND the pooled maps) and mspl_pyr_down_mid_bwd (one backward launch: depthwise
    data + weight gradient + the adaptive pool's transpose of both branches) against torch autograd of the same expression."""
    import ctypes
    import math
    from mspl_amd import ops
    from mspl_amd._native import check, lib
    N, P, h, w = cfg
    sizes = [(max(math.ceil(h * s), 5), max(math.ceil(w * s), 5)) for s in (0.5, 0.1)]
    sizes = [(min(a, h), min(b, w)) for a, b in sizes]
    x = rnd(N, P, h, w, seed=1)
    ws = [rnd(P, 1, 3, 3, seed=2 + i, scale=0.4) for i in range(2)]
    g_es = [rnd(N, P, sz[0], sz[1], seed=10 + i) for i, sz in enumerate(sizes)]
    xr = x.clone().requires_grad_(True)
    wr = [t.clone().requires_grad_(True) for t in ws]
    pooled_ref = [F.adaptive_avg_pool2d(xr, sz) for sz in sizes]
    es_ref = [F.conv2d(p, t, None, 1, 1, 1, P) for p, t in zip(pooled_ref, wr)]
    grads = [torch.autograd.grad(e, [xr, t], g, retain_graph=True) for e, t, g in zip(es_ref, wr, g_es)]
    d = lambda t: t.to(DEV)
    if ops.pyr_down_prep_fits((N, P, h, w), sizes):
        outs, pools = ops.pyr_down_prep(d(x), sizes, [d(t) for t in ws], keep_pooled=True)
        for i in range(2):
            close(pools[i], pooled_ref[i], atol=1e-6, rtol=1e-5)
            close(outs[i], es_ref[i], atol=2e-6, rtol=1e-5)
    pools = [d(p.detach()) for p in pooled_ref]
    gw = [torch.full((P, 1, 3, 3), 0.5, device=DEV) for _ in range(2)]                    # accumulated INTO (a gradient sink)
    gx = [torch.full((N, P, h, w), 7.0, device=DEV) for _ in range(2)]
    keep = [d(g) for g in g_es], [d(t) for t in ws]
    arr = lambda ts: (ctypes.c_void_p * 2)(*[t.data_ptr() for t in ts])
    hsa, wsa = (ctypes.c_int32 * 2)(*[s_[0] for s_ in sizes]), (ctypes.c_int32 * 2)(*[s_[1] for s_ in sizes])
    check(lib.mspl_pyr_down_mid_bwd(arr(keep[0]), arr(pools), arr(keep[1]), N, P, h, w, 2, hsa, wsa, arr(gw), arr(gx),
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    for i in range(2):
        close(gx[i], grads[i][0], atol=2e-6, rtol=1e-5)
        close(gw[i] - 0.5, grads[i][1], atol=2e-4 * float(grads[i][1].abs().max() + 1), rtol=1e-4)


def test_conv1x1_weight_gradients_batched_in_one_launch():
    """mspl_conv1x1_wgrad_batch: the weight gradients of several grouped 1x1 convolutions accumulated into their buffers by one
    launch per run of matrix-core problems (a problem with fewer than 8 channels per group in between goes through the generic kernel),
    against torch; and autograd.WgradQueue: inside grad_sinks() the nodes queue them, the exit of the context sends them out."""
    import ctypes
    from mspl_amd import autograd as ag
    from mspl_amd._native import check, lib
    probs = [(2, 128, 512, 4, 16, 30), (3, 512, 128, 4, 16, 30), (1, 12, 24, 4, 8, 12), (2, 64, 48, 1, 36, 60), (1, 256, 256, 4, 5, 8),
             (2, 32, 16, 1, 20, 36), (2, 512, 512, 4, 9, 12), (1, 96, 80, 1, 10, 14)]
    # (N, Cin, Cout, groups, H, W); the third one has 3 input channels per group; 2x2 or more 32x32 tiles per group (the fourth, fifth
    # and the last two: 4x4 tiles, and 3x3 with a ragged last tile) sit four tiles to a workgroup
    gys, xs, gws, refs = [], [], [], []
    for i, (N, Cin, Cout, G, H, W) in enumerate(probs):
        x = rnd(N, Cin, H, W, seed=40 + i)
        gy = rnd(N, Cout, H, W, seed=60 + i)
        w = torch.zeros(Cout, Cin // G, 1, 1, requires_grad=True)
        refs.append(torch.autograd.grad(F.conv2d(x, w, None, 1, 0, 1, G), w, gy)[0])
        gys.append(gy.to(DEV)); xs.append(x.to(DEV)); gws.append(torch.full((Cout, Cin // G, 1, 1), 0.25, device=DEV))
    n = len(probs)
    arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
    col = lambda k: (ctypes.c_int32 * n)(*[p[k] for p in probs])
    hw = (ctypes.c_int32 * n)(*[p[4] * p[5] for p in probs])
    # rowscale on the first and fourth problem: gw[co, :] += s[co] * sum (the gradient operand is the one BEFORE a per-channel scale)
    rsc = [rnd(probs[0][2], seed=90).abs().to(DEV) + 0.5, None, None, rnd(probs[3][2], seed=91).abs().to(DEV) + 0.5, None, None, None,
           rnd(probs[7][2], seed=92).abs().to(DEV) + 0.5]
    rs = (ctypes.c_void_p * n)(*[None if t is None else t.data_ptr() for t in rsc])
    check(lib.mspl_conv1x1_wgrad_batch(arr(gys), arr(xs), arr(gws), rs, col(0), col(1), col(2), col(3), hw, n,
                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    for g, r, sc in zip(gws, refs, rsc):
        if sc is not None:
            r = r * sc.cpu().view(-1, 1, 1, 1)
        assert float((g.cpu() - 0.25 - r).abs().max()) <= 2e-4 * float(r.abs().max()) + 1e-5
    # the queue: two convolutions' weight gradients are still pending inside the context, in their buffers after it
    w1 = rnd(48, 64, 1, 1, seed=80, scale=0.1).to(DEV).requires_grad_(True)
    w2 = rnd(512, 128, 1, 1, seed=81, scale=0.1).to(DEV).requires_grad_(True)
    x1, x2 = rnd(2, 64, 36, 60, seed=82).to(DEV).requires_grad_(True), rnd(2, 512, 16, 30, seed=83).to(DEV).requires_grad_(True)
    ref = torch.autograd.grad((ag.conv(x1, w1, 1, 1).sum() + ag.conv(x2, w2, 1, 4).square().sum()), [w1, w2])
    w1.grad, w2.grad = torch.zeros_like(w1), torch.zeros_like(w2)
    with torch.enable_grad(), ag.grad_sinks():
        (ag.conv(x1, w1, 1, 1).sum() + ag.conv(x2, w2, 1, 4).square().sum()).backward()
        pending = len(ag.WGRADS.items)
        torch.cuda.synchronize()
        assert float(w1.grad.abs().max()) == 0.0 and float(w2.grad.abs().max()) == 0.0
    assert pending == 2 and len(ag.WGRADS.items) == 0
    for g, r in zip((w1.grad, w2.grad), ref):
        assert float((g - r).abs().max()) <= 2e-4 * float(r.abs().max()) + 1e-5


def test_fan_out_takes_the_gate_gradient_as_plane_constants():
    """An encoder output feeds the EfficientPWConv gate (global average pool -> 1x1 -> sigmoid, nn_layers/efficient_pt.py:25-29) and
    other consumers: the pool's gradient is constant over every plane; GapGateFn hands it out as an expanded (N,C,1,1) tensor and
    FanOutFn adds it inside its one summation launch (mspl_sum_n_planes) -- against torch autograd on the CPU."""
    from mspl_amd import autograd as ag
    x, w = rnd(2, 8, 6, 10, seed=1), rnd(6, 8, 1, 1, seed=2)
    y, m1, m2 = rnd(2, 6, 6, 10, seed=3), rnd(2, 8, 6, 10, seed=4), rnd(2, 8, 6, 10, seed=5)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = (y * torch.sigmoid(F.conv2d(F.adaptive_avg_pool2d(xr, 1), wr))).sum() + (xr * m1).sum() + (xr * xr * m2).sum()
    gx_ref, gw_ref = torch.autograd.grad(ref, [xr, wr])
    xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    a = ag.fan_out(xd, 3)
    out = ag.channel_scale(y.to(DEV), ag.gap_gate(a[0], wd)).sum() + (a[1] * m1.to(DEV)).sum() + (a[2] * a[2] * m2.to(DEV)).sum()
    gx, gw = torch.autograd.grad(out, [xd, wd])
    close(gx, gx_ref, atol=1e-5, rtol=1e-5)
    close(gw, gw_ref, atol=1e-5, rtol=1e-4)
