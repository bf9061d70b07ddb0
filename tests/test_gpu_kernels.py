"""Per-kernel parity against plain torch-CPU fp32 statements of the same op (the oracle's primitives),
over the awkward shapes: odd sizes, widths not divisible by 4, channel slices, every epilogue term."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def close(a, b, atol=2e-5, rtol=1e-4):
    torch.testing.assert_close(a.cpu(), b, rtol=rtol, atol=atol)


@pytest.mark.parametrize('dil', [[1, 2, 3, 4], [1, 1, 2, 3], [1, 1, 1, 2]])
@pytest.mark.parametrize('stride', [1, 2])
@pytest.mark.parametrize('shape', [(2, 8, 16, 30), (1, 24, 33, 61), (2, 4, 128, 240), (1, 6, 5, 7), (3, 16, 32, 60)])
def test_eesp_dw_hff(dil, stride, shape):
    from mspl_amd import ops
    from mspl_amd.ops import Epi
    N, n, H, W = shape
    x = rnd(*shape, seed=1)
    w = rnd(4, n, 3, 3, seed=2, scale=0.3)
    scale, shift, alpha = rnd(4 * n, seed=3).abs() + 0.5, rnd(4 * n, seed=4) * 0.1, rnd(4 * n, seed=5).abs() * 0.3
    outs = []
    for k in range(4):
        o = F.conv2d(x, w[k].unsqueeze(1), None, stride, dil[k], dil[k], n)
        outs.append(o if k == 0 else o + outs[-1])
    ref = torch.cat(outs, 1)
    ref = F.prelu(ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1), alpha)
    got = ops.eesp_dw_hff(x.to(DEV), w.to(DEV), dil, stride, Epi(scale.to(DEV), shift.to(DEV), alpha.to(DEV)))
    close(got, ref)
    raw = ops.eesp_dw_hff(x.to(DEV), w.to(DEV), dil, stride)       # no epilogue
    close(raw, torch.cat(outs, 1))
    # training forward: activated output and the bare sums from ONE launch (raw_out)
    keep = torch.full_like(raw, 7.0)
    both = ops.eesp_dw_hff(x.to(DEV), w.to(DEV), dil, stride, Epi(scale.to(DEV), shift.to(DEV), alpha.to(DEV), raw_out=keep))
    assert torch.equal(both, got) and torch.equal(keep, raw)


@pytest.mark.parametrize('dil', [[1, 2, 3, 4], [1, 1, 2, 3]])
@pytest.mark.parametrize('shape', [(2, 3, 144, 240), (3, 5, 70, 248), (5, 2, 37, 120), (1, 2, 9, 504), (2, 4, 64, 8)])
def test_eesp_dw_hff_stride2_streaming_form(dil, shape):
    """The register-streaming stride-2 kernel (forced: these planes are too few for its automatic choice) against the same
    reference, and bit-identical to the direct / tiled forms (same summation order per accumulator)."""
    from mspl_amd import ops
    from mspl_amd.ops import Epi
    N, n, H, W = shape
    x = rnd(*shape, seed=1)
    w = rnd(4, n, 3, 3, seed=2, scale=0.3)
    scale, shift, alpha = rnd(4 * n, seed=3).abs() + 0.5, rnd(4 * n, seed=4) * 0.1, rnd(4 * n, seed=5).abs() * 0.3
    outs = []
    for k in range(4):
        o = F.conv2d(x, w[k].unsqueeze(1), None, 2, dil[k], dil[k], n)
        outs.append(o if k == 0 else o + outs[-1])
    ref = F.prelu(torch.cat(outs, 1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1), alpha)
    ep = Epi(scale.to(DEV), shift.to(DEV), alpha.to(DEV))
    with ops.launch_flags(k2_stream=False):
        base = ops.eesp_dw_hff(x.to(DEV), w.to(DEV), dil, 2, ep)
    with ops.launch_flags(k2_stream=True):
        got = ops.eesp_dw_hff(x.to(DEV), w.to(DEV), dil, 2, ep)
        raw = ops.eesp_dw_hff(x.to(DEV), w.to(DEV), dil, 2)
        # into a channel slice of a wider destination
        dst = torch.full((N, 4 * n + 3, got.shape[2], got.shape[3]), -5.0, device=DEV)
        sc2, sh2, al2 = torch.ones(4 * n + 3), torch.zeros(4 * n + 3), torch.ones(4 * n + 3)
        sc2[2:2 + 4 * n], sh2[2:2 + 4 * n], al2[2:2 + 4 * n] = scale, shift, alpha
        ops.eesp_dw_hff(x.to(DEV), w.to(DEV), dil, 2, Epi(sc2.to(DEV), sh2.to(DEV), al2.to(DEV)), out=(dst, 2))
    close(got, ref)
    close(raw, torch.cat(outs, 1))
    assert torch.equal(got, base)
    assert torch.equal(dst[:, 2:2 + 4 * n], got) and torch.all(dst[:, :2] == -5.0) and torch.all(dst[:, 2 + 4 * n:] == -5.0)


@pytest.mark.parametrize('cfg', [(2, 128, 18, 30, [1, 1, 2, 3]), (1, 128, 16, 30, [1, 1, 2, 3]), (3, 128, 5, 30, [1, 1, 2, 3]), (17, 128, 18, 30, [1, 1, 2, 3]),
                                 (2, 64, 36, 60, [1, 2, 3, 4]), (1, 64, 32, 60, [1, 2, 3, 4]), (3, 64, 3, 60, [1, 2, 3, 4]), (1, 64, 1, 60, [1, 2, 3, 4]),
                                 (2, 128, 16, 32, [1, 1, 2, 3]), (1, 128, 7, 32, [1, 1, 2, 3]), (2, 64, 32, 64, [1, 2, 3, 4]), (1, 64, 5, 64, [1, 2, 3, 4]),
                                 # 1024-pixel-wide inputs: 64 columns at level 4 (one row per band), 128 at level 3 (two half-row bands per row)
                                 (2, 128, 9, 64, [1, 1, 2, 3]), (1, 128, 1, 64, [1, 1, 2, 3]), (2, 64, 11, 128, [1, 2, 3, 4]), (1, 64, 2, 128, [1, 2, 3, 4])])
def test_eesp_dw_exp(cfg):
    """K2 + K3 of a stride-1 EESP block in one launch (nn_layers/eesp.py:68-93) against torch fp32, and bit-identical to the
    two-launch form (same operation order in the branch arithmetic and in the matrix-core sums)."""
    from mspl_amd import ops
    from mspl_amd.ops import Epi
    N, n, H, W, dil = cfg
    assert ops.eesp_dw_exp_fits((N, n, H, W), dil)
    r = rnd(N, n, H, W, seed=1)
    x_in = rnd(N, 4 * n, H, W, seed=6)
    w = rnd(4, n, 3, 3, seed=2, scale=0.3)
    bs, bb, ba = rnd(4 * n, seed=3).abs() + 0.5, rnd(4 * n, seed=4) * 0.1, rnd(4 * n, seed=5).abs() * 0.3
    wexp = rnd(4 * n, n, 1, 1, seed=7, scale=0.1)
    es, eb, ea = rnd(4 * n, seed=8).abs() + 0.5, rnd(4 * n, seed=9) * 0.1, rnd(4 * n, seed=10).abs() * 0.3
    outs = []
    for k in range(4):
        o = F.conv2d(r, w[k].unsqueeze(1), None, 1, dil[k], dil[k], n)
        outs.append(o if k == 0 else o + outs[-1])
    cat = F.prelu(torch.cat(outs, 1) * bs.view(1, -1, 1, 1) + bb.view(1, -1, 1, 1), ba)
    ref = F.conv2d(cat, wexp, None, 1, 0, 1, 4) * es.view(1, -1, 1, 1) + eb.view(1, -1, 1, 1) + x_in
    ref = F.prelu(ref, ea)
    d = lambda t: t.to(DEV)
    packed = ops.eesp_dw_exp_pack(d(w), d(bs), d(bb), d(ba), d(wexp), H, W, dil)
    got = ops.eesp_dw_exp(d(r), packed, dil, Epi(d(es), d(eb), d(ea), residual=d(x_in)))
    close(got, ref)
    cat2 = ops.eesp_dw_hff(d(r), d(w), dil, 1, Epi(d(bs), d(bb), d(ba)))
    two = ops.conv1x1(cat2, d(wexp), 4, Epi(d(es), d(eb), d(ea), residual=d(x_in)))
    assert torch.equal(got, two)
    # not covered: another width, another dilation set
    assert not ops.eesp_dw_exp_fits((N, n, H, W + 6), dil)
    assert not ops.eesp_dw_exp_fits((N, n, H, W), [1, 1, 1, 2])


@pytest.mark.parametrize('cfg', [(2, 128, 18, 30, [1, 1, 2, 3]), (3, 128, 5, 30, [1, 1, 2, 3]), (17, 128, 18, 30, [1, 1, 2, 3]),
                                 (2, 64, 36, 60, [1, 2, 3, 4]), (1, 64, 3, 60, [1, 2, 3, 4]), (2, 128, 16, 32, [1, 1, 2, 3]), (2, 64, 32, 64, [1, 2, 3, 4]),
                                 (2, 128, 9, 64, [1, 1, 2, 3]), (2, 64, 11, 128, [1, 2, 3, 4]), (1, 64, 1, 128, [1, 2, 3, 4])])
def test_eesp_dw_exp_next_projection(cfg):
    """The fused K2 + K3 launch that also computes the FOLLOWING block's proj_1x1 (grouped 1x1 + BN + PReLU over its own output):
    the block output stays bit-identical to the launch without it, the reduced tensor equals conv1x1 on that output up to the
    summation order of the K = n products (torch fp32 as the reference)."""
    from mspl_amd import ops
    from mspl_amd.ops import Epi
    N, n, H, W, dil = cfg
    r = rnd(N, n, H, W, seed=1)
    x_in = rnd(N, 4 * n, H, W, seed=6)
    w = rnd(4, n, 3, 3, seed=2, scale=0.3)
    bs, bb, ba = rnd(4 * n, seed=3).abs() + 0.5, rnd(4 * n, seed=4) * 0.1, rnd(4 * n, seed=5).abs() * 0.3
    wexp = rnd(4 * n, n, 1, 1, seed=7, scale=0.1)
    es, eb, ea = rnd(4 * n, seed=8).abs() + 0.5, rnd(4 * n, seed=9) * 0.1, rnd(4 * n, seed=10).abs() * 0.3
    w1 = rnd(n, n, 1, 1, seed=11, scale=0.1)
    ns, nb, na = rnd(n, seed=12).abs() + 0.5, rnd(n, seed=13) * 0.1, rnd(n, seed=14).abs() * 0.3
    d = lambda t: t.to(DEV)
    packed = ops.eesp_dw_exp_pack(d(w), d(bs), d(bb), d(ba), d(wexp), H, W, dil)
    ep = Epi(d(es), d(eb), d(ea), residual=d(x_in))
    alone = ops.eesp_dw_exp(d(r), packed, dil, ep)
    y, rn = ops.eesp_dw_exp(d(r), packed, dil, ep, next_proj=(ops.eesp_dw_exp_next_pack(d(w1)), d(ns), d(nb), d(na)))
    assert torch.equal(y, alone)
    ref = F.prelu(F.conv2d(alone.cpu(), w1, None, 1, 0, 1, 4) * ns.view(1, -1, 1, 1) + nb.view(1, -1, 1, 1), na)
    close(rn, ref)
    two = ops.conv1x1(alone, d(w1), 4, Epi(d(ns), d(nb), d(na)))
    close(rn, two.cpu(), atol=1e-5)


@pytest.mark.parametrize('cfg', [(2, 512, 128, 4, 18, 30, [1, 1, 2, 3]), (1, 512, 128, 4, 16, 30, [1, 1, 2, 3]), (2, 256, 64, 4, 18, 30, [1, 2, 3, 4]),
                                 (1, 256, 64, 4, 8, 12, [1, 2, 3, 4]), (3, 512, 128, 4, 6, 10, [1, 1, 2, 3]), (1, 256, 128, 4, 10, 44, [1, 2, 3, 4]),
                                 (32, 512, 128, 4, 18, 30, [1, 1, 2, 3]), (1, 512, 64, 4, 20, 36, [1, 1, 2, 3])])
def test_eesp_proj_dw_hff(cfg):
    """K1 + K2 in one launch (projection on the matrix cores straight into K2's LDS tile) against torch, and against the two
    launches it replaces: the projection differs from the 32x32x2 kernel only in the last bits (k order inside an MFMA), K2's
    arithmetic on it is the same code."""
    from mspl_amd import ops
    from mspl_amd.ops import Epi
    N, Cin, n, G, H, W, dil = cfg
    x = rnd(N, Cin, H, W, seed=1)
    wp = rnd(n, Cin // G, 1, 1, seed=2, scale=0.1)
    ps, pb, pa = rnd(n, seed=3).abs() + 0.5, rnd(n, seed=4) * 0.1, rnd(n, seed=5).abs() * 0.3
    w = rnd(4, n, 3, 3, seed=6, scale=0.3)
    scale, shift, alpha = rnd(4 * n, seed=7).abs() + 0.5, rnd(4 * n, seed=8) * 0.1, rnd(4 * n, seed=9).abs() * 0.3
    o1 = F.prelu(F.conv2d(x, wp, None, 1, 0, 1, G) * ps.view(1, -1, 1, 1) + pb.view(1, -1, 1, 1), pa)
    outs = []
    for k in range(4):
        o = F.conv2d(o1, w[k].unsqueeze(1), None, 1, dil[k], dil[k], n)
        outs.append(o if k == 0 else o + outs[-1])
    ref = F.prelu(torch.cat(outs, 1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1), alpha)
    assert ops.eesp_proj_dw_hff_fits(x.shape, n, G, dil, 1)
    d = lambda t: t.to(DEV)
    ep = Epi(d(scale), d(shift), d(alpha))
    got = ops.eesp_proj_dw_hff(d(x), d(wp), d(ps), d(pb), d(pa), d(w), dil, G, ep)
    close(got, ref, atol=2e-4, rtol=2e-4)
    two = ops.eesp_dw_hff(ops.conv1x1(d(x), d(wp), G, Epi(d(ps), d(pb), d(pa))), d(w), dil, 1, ep)
    close(got, two.cpu(), atol=2e-5, rtol=1e-5)
    # into a channel slice of a wider destination
    wide = torch.zeros(N, 4 * n + 8, H, W, device=DEV)
    ep2 = Epi(F.pad(d(scale), (8, 0)), F.pad(d(shift), (8, 0)), F.pad(d(alpha), (8, 0)))
    ops.eesp_proj_dw_hff(d(x), d(wp), d(ps), d(pb), d(pa), d(w), dil, G, ep2, out=(wide, 8))
    assert torch.equal(wide[:, 8:], got) and float(wide[:, :8].abs().sum()) == 0.0


def test_eesp_proj_dw_hff_shapes_left_to_two_launches():
    from mspl_amd import ops
    assert not ops.eesp_proj_dw_hff_fits((16, 256, 72, 120), 64, 4, [1, 2, 3, 4], 1)      # planes wider than 64 columns
    assert not ops.eesp_proj_dw_hff_fits((16, 512, 18, 30), 128, 4, [1, 1, 2, 3], 2)      # stride 2
    assert not ops.eesp_proj_dw_hff_fits((16, 96, 18, 30), 24, 4, [1, 1, 2, 3], 1)        # K = 24 per group
    assert not ops.eesp_proj_dw_hff_fits((1, 512, 18, 31), 128, 4, [1, 1, 2, 3], 1)       # odd row length
    with pytest.raises(RuntimeError, match='not covered'):
        ops.eesp_proj_dw_hff(torch.zeros(1, 256, 72, 120, device=DEV), torch.zeros(64, 64, 1, 1, device=DEV), None, None, None,
                             torch.zeros(4, 64, 3, 3, device=DEV), [1, 2, 3, 4], 4)


def test_eesp_dw_unsupported_dilation_raises():
    from mspl_amd import ops
    with pytest.raises(RuntimeError, match='unsupported dilation'):
        ops.eesp_dw_hff(torch.zeros(1, 4, 8, 8, device=DEV), torch.zeros(4, 4, 3, 3, device=DEV), [1, 2, 4, 8], 1)


@pytest.mark.parametrize('cfg', [
    # (N, Cin, Cout, groups, H, W)
    (2, 32, 24, 4, 16, 30), (1, 512, 512, 4, 16, 30), (2, 256, 64, 4, 9, 13), (1, 512, 16, 1, 16, 30),
    (2, 96, 96, 4, 8, 12), (1, 16, 13, 1, 17, 23), (1, 48, 16, 1, 20, 20), (2, 16, 4, 4, 10, 10),
    (1, 128, 512, 4, 6, 6), (1, 160, 640, 4, 5, 9), (1, 3, 128, 1, 12, 16), (1, 512, 64, 1, 4, 8),
    # many pixel tiles: several tiles per wave (the ring runs across tiles), 48 rows in a 64-row weight tile, odd pixel count
    (1, 32, 16, 1, 288, 720), (1, 64, 96, 2, 160, 320), (3, 24, 8, 1, 191, 201),
    # few output channels per group on small maps: the split-K form (EESP reduce projections at levels 3 / 4, pyramid projections)
    (2, 512, 128, 4, 18, 30), (3, 256, 64, 4, 36, 60), (2, 128, 20, 1, 7, 11),
    # thin projections on large maps: the streaming vector-unit form (<= 16 output channels per group, >= 400 workgroups)
    (12, 32, 16, 1, 144, 240), (8, 32, 24, 4, 144, 240), (12, 16, 13, 1, 144, 240)])
def test_conv1x1_epilogues(cfg):
    from mspl_amd import ops
    from mspl_amd.ops import Epi
    N, Cin, Cout, G, H, W = cfg
    x = rnd(N, Cin, H, W, seed=1)
    w = rnd(Cout, Cin // G, 1, 1, seed=2, scale=(Cin // G) ** -0.5)
    scale, shift, alpha = rnd(Cout, seed=3).abs() + 0.5, rnd(Cout, seed=4) * 0.1, rnd(Cout, seed=5).abs() * 0.3
    res = rnd(N, Cout, H, W, seed=6)
    base = F.conv2d(x, w, None, 1, 0, 1, G)
    close(ops.conv1x1(x.to(DEV), w.to(DEV), G), base)
    # BN + PReLU only, into a channel slice (the epilogue every kernel form supports)
    dst0 = torch.full((N, Cout + 4, H, W), -3.0, device=DEV)
    sc0, sh0, al0 = rnd(Cout + 4, seed=21).abs() + 0.5, rnd(Cout + 4, seed=22) * 0.1, rnd(Cout + 4, seed=23).abs() * 0.3
    ops.conv1x1(x.to(DEV), w.to(DEV), G, Epi(sc0.to(DEV), sh0.to(DEV), al0.to(DEV)), out=(dst0, 2))
    s0 = slice(2, 2 + Cout)
    close(dst0[:, s0], F.prelu(base * sc0[s0].view(1, -1, 1, 1) + sh0[s0].view(1, -1, 1, 1), al0[s0]))
    assert torch.all(dst0[:, :2] == -3.0) and torch.all(dst0[:, 2 + Cout:] == -3.0)
    ref = F.prelu(base * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res, alpha)
    got = ops.conv1x1(x.to(DEV), w.to(DEV), G, Epi(scale.to(DEV), shift.to(DEV), alpha.to(DEV), residual=res.to(DEV)))
    close(got, ref)
    # reinforcement + channel-slice destination + gate + pre_add
    ctot, coff = Cout + 5, 3
    r = rnd(N, 3, H, W, seed=7)
    rw = rnd(ctot, 3, seed=8)
    gate = torch.sigmoid(rnd(N, ctot, seed=9))
    pre = rnd(N, ctot, H, W, seed=10)
    sc, sh, al = rnd(ctot, seed=11).abs() + 0.5, rnd(ctot, seed=12) * 0.1, rnd(ctot, seed=13).abs() * 0.3
    dst = torch.full((N, ctot, H, W), -7.0, device=DEV)
    ops.conv1x1(x.to(DEV), w.to(DEV), G, Epi(sc.to(DEV), sh.to(DEV), al.to(DEV), pre_add=pre.to(DEV), reinf_r=r.to(DEV),
                                             reinf_w=rw.to(DEV), gate=gate.to(DEV)), out=(dst, coff))
    s = slice(coff, coff + Cout)
    v = (base + pre[:, s]) * sc[s].view(1, -1, 1, 1) + sh[s].view(1, -1, 1, 1)
    v = v + torch.einsum('cj,njhw->nchw', rw[s], r)
    v = F.prelu(v, al[s]) * gate[:, s, None, None]
    close(dst[:, s], v)
    assert torch.all(dst[:, :coff] == -7.0) and torch.all(dst[:, coff + Cout:] == -7.0)


@pytest.mark.parametrize('cfg', [
    # (N, Cin, Cout, groups, H, W, stride, shuffle_groups)
    (2, 3, 32, 1, 32, 48, 2, 0), (1, 3, 3, 1, 9, 15, 1, 0), (2, 16, 16, 16, 17, 29, 1, 0), (1, 80, 16, 16, 16, 30, 1, 5),
    (1, 128, 48, 16, 12, 20, 1, 0), (1, 256, 64, 64, 8, 15, 1, 0), (1, 4, 4, 4, 256, 480, 1, 0), (1, 3, 32, 1, 31, 45, 2, 0),
    (1, 20, 4, 4, 5, 5, 1, 5),
    # depthwise maps wide enough for the register-streaming form: odd height, exactly one column block, a 4-column second block
    (2, 6, 6, 6, 37, 64, 1, 0), (1, 5, 5, 5, 9, 248, 1, 0), (1, 2, 2, 2, 40, 252, 1, 0),
    # grouped shapes around the streaming form (row length % 4 == 0, >= 16 groups, (cin, cout) per group (4,1) or (1,4)): several
    # units per wave, a partly filled last wave, Shuffle-aware input channels; the other group shapes stay on the LDS-tiled kernel
    (2, 128, 48, 16, 24, 120, 1, 0), (1, 256, 64, 64, 12, 60, 1, 0), (2, 3, 3, 1, 18, 32, 1, 0), (1, 48, 128, 16, 10, 40, 1, 0),
    (1, 64, 256, 64, 9, 28, 1, 0), (1, 80, 30, 10, 16, 32, 1, 5), (3, 8, 3, 1, 29, 256, 1, 0), (1, 3, 3, 1, 50, 16, 1, 0),
    (3, 64, 16, 16, 29, 256, 1, 0), (2, 80, 20, 20, 16, 32, 1, 5), (1, 32, 128, 32, 7, 20, 1, 0),
    # the LDS-tiled kernel's lean staging (stride 1, one tile per row, W % 4 == 0, no Shuffle): narrowest row, a single row,
    # a row count that leaves a short last tile, a last chunk shared by the spare threads
    (1, 8, 3, 1, 3, 8, 1, 0), (2, 16, 6, 2, 1, 12, 1, 0), (2, 16, 6, 2, 41, 124, 1, 0), (1, 24, 9, 3, 19, 128, 1, 0)])
def test_conv3x3(cfg):
    from mspl_amd import ops
    from mspl_amd.ops import Epi
    N, Cin, Cout, G, H, W, stride, sg = cfg
    x = rnd(N, Cin, H, W, seed=1)
    w = rnd(Cout, Cin // G, 3, 3, seed=2, scale=0.2)
    xin = x
    if sg:
        xin = x.view(N, sg, Cin // sg, H, W).transpose(1, 2).contiguous().view(N, Cin, H, W)
    base = F.conv2d(xin, w, None, stride, 1, 1, G)
    close(ops.conv3x3(x.to(DEV), w.to(DEV), G, stride, sg), base)
    scale, shift, alpha = rnd(Cout, seed=3).abs() + 0.5, rnd(Cout, seed=4) * 0.1, rnd(Cout, seed=5).abs() * 0.3
    gate = torch.sigmoid(rnd(N, Cout, seed=6))
    ref = F.prelu(base * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1), alpha) * gate[:, :, None, None]
    got = ops.conv3x3(x.to(DEV), w.to(DEV), G, stride, sg, Epi(scale.to(DEV), shift.to(DEV), alpha.to(DEV), gate=gate.to(DEV)))
    close(got, ref)


@pytest.mark.parametrize('shape', [(2, 5, 16, 30), (1, 3, 15, 21), (1, 2, 1, 1), (1, 3, 256, 480), (2, 7, 9, 10)])
def test_avgpool3x3s2(shape):
    from mspl_amd import ops
    x = rnd(*shape, seed=1)
    close(ops.avgpool3x3s2(x.to(DEV)), F.avg_pool2d(x, 3, 2, 1))


@pytest.mark.parametrize('cfg', [((2, 4, 16, 30), (32, 60)), ((1, 3, 5, 5), (16, 30)), ((1, 2, 8, 15), (24, 45)),
                                 ((1, 5, 13, 24), (128, 240)), ((2, 3, 7, 9), (7, 9)), ((1, 2, 1, 1), (4, 6)),
                                 ((1, 2, 64, 120), (256, 480))])
def test_bilinear_align_corners(cfg):
    from mspl_amd import ops
    from mspl_amd.ops import Epi
    shape, size = cfg
    x = rnd(*shape, seed=1)
    ref = F.interpolate(x, size, mode='bilinear', align_corners=True)
    close(ops.bilinear(x.to(DEV), size), ref, atol=1e-5)
    C = shape[1]
    pre = rnd(shape[0], C, *size, seed=2)
    scale, shift, alpha = rnd(C, seed=3).abs() + 0.5, rnd(C, seed=4) * 0.1, rnd(C, seed=5).abs() * 0.3
    got = ops.bilinear(x.to(DEV), size, Epi(scale.to(DEV), shift.to(DEV), alpha.to(DEV), pre_add=pre.to(DEV)))
    close(got, F.prelu((ref + pre) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1), alpha))


@pytest.mark.parametrize('align', [True, False])
@pytest.mark.parametrize('cfg', [((2, 3, 40, 64), (16, 32)), ((1, 5, 9, 16), (27, 16)), ((3, 2, 17, 20), (17, 40)), ((1, 2, 3, 8), (64, 8)),
                                 ((2, 2, 72, 120), (144, 240))])
def test_bilinear_streaming_walks(cfg, align):
    """The register-streaming bilinear kernel keeps two horizontally interpolated source rows and walks down the output rows:
    walks that jump more than one source row per output row (down-scaling), that repeat a row many
    times (x21), unchanged heights, several planes per wave -- with pre_add AND residual operands -- against torch-CPU fp32."""
    from mspl_amd import ops
    from mspl_amd.ops import Epi
    shape, size = cfg
    x = rnd(*shape, seed=11)
    ref = F.interpolate(x, size, mode='bilinear', align_corners=align)
    C = shape[1]
    pre, res = rnd(shape[0], C, *size, seed=12), rnd(shape[0], C, *size, seed=13)
    scale, shift, alpha = rnd(C, seed=3).abs() + 0.5, rnd(C, seed=4) * 0.1, rnd(C, seed=5).abs() * 0.3
    got = ops.bilinear(x.to(DEV), size, Epi(scale.to(DEV), shift.to(DEV), alpha.to(DEV), pre_add=pre.to(DEV), residual=res.to(DEV)),
                       align_corners=align)
    close(got, F.prelu((ref + pre) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res, alpha))
    close(ops.bilinear(x.to(DEV), size, align_corners=align), ref, atol=1e-5)


@pytest.mark.parametrize('cfg', [((2, 4, 32, 60), (16, 30)), ((1, 3, 24, 45), (16, 30)), ((1, 2, 16, 30), (5, 5)),
                                 ((1, 2, 6, 9), (5, 5)), ((1, 3, 128, 240), (13, 24)), ((1, 2, 5, 7), (5, 7)),
                                 ((1, 2, 192, 360), (128, 240))])
def test_adaptive_avgpool(cfg):
    from mspl_amd import ops
    shape, size = cfg
    x = rnd(*shape, seed=1)
    close(ops.adaptive_avgpool(x.to(DEV), size), F.adaptive_avg_pool2d(x, size))


def test_pointwise_and_gap_gate():
    from mspl_amd import ops
    from mspl_amd.ops import Epi
    for shape in [(2, 6, 16, 30), (1, 5, 7, 9)]:
        x = rnd(*shape, seed=1)
        C = shape[1]
        scale, shift, alpha = rnd(C, seed=3).abs() + 0.5, rnd(C, seed=4) * 0.1, rnd(C, seed=5).abs() * 0.3
        got = ops.pointwise(x.to(DEV), Epi(scale.to(DEV), shift.to(DEV), alpha.to(DEV)))
        close(got, F.prelu(x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1), alpha))
        w = rnd(9, C, 1, 1, seed=6)
        ref = torch.sigmoid(F.conv2d(F.adaptive_avg_pool2d(x, 1), w)).flatten(1)
        close(ops.gap_gate(x.to(DEV), w.to(DEV)), ref, atol=1e-6)


def test_label_epilogue_fused_upsample_vs_unfused():
    """Fused (low-res in) label epilogue == the reference order: upsample both heads, then combine."""
    from mspl_amd import ops
    N, C, H, W = 2, 13, 64, 96
    main = rnd(N, C, H // 2, W // 2, seed=1, scale=2.0)
    aux = rnd(N, C, H // 4, W // 4, seed=2, scale=2.0)
    mu = F.interpolate(main, (H, W), mode='bilinear', align_corners=True)
    au = F.interpolate(aux, (H, W), mode='bilinear', align_corners=True)
    r = ops.label_epilogue(main.to(DEV), aux.to(DEV), (H, W), want_prob=True, want_kld=True, want_logits=True)
    close(r['main_up'], mu, atol=1e-5)
    close(r['aux_up'], au, atol=1e-5)
    p1, lp1, lp2 = F.softmax(mu, 1), F.log_softmax(mu, 1), F.log_softmax(au, 1)
    close(r['kld'], (p1 * lp1 - p1 * lp2).sum(1), atol=2e-5, rtol=1e-3)
    prob = F.softmax(mu + 0.5 * au, 1)
    close(r['prob'], prob, atol=1e-6)
    ref = np.argmax(prob.numpy().transpose(0, 2, 3, 1), axis=3).astype(np.uint8)
    srt = torch.sort(prob, 1, descending=True)[0]
    sure = (srt[:, 0] - srt[:, 1]).numpy() > 1e-5
    assert np.array_equal(r['labels'].cpu().numpy()[sure], ref[sure])


@pytest.mark.parametrize('cfg', [(2, 13, 64, 96, 13, False), (1, 5, 48, 80, 5, True), (3, 20, 32, 272, 32, False), (1, 13, 16, 16, 16, True),
                                 (1, 13, 36, 52, 13, False), (2, 7, 20, 26, 8, True), (1, 24, 288, 480, 24, False)])
def test_label_epilogue_hist_equals_epilogue_plus_merge(cfg):
    """The single-source pass's fused form (labels + KL map + class histogram in one launch) against the two-launch form it
    replaces, bit for bit: label_epilogue, then merge_labels(S=1, thresh=1) as identity + histogram; and against np.bincount."""
    from mspl_amd import ops
    N, C, H, W, ncls, use_lut = cfg
    main = rnd(N, C, H // 2, W // 2, seed=11, scale=2.0).to(DEV)
    aux = rnd(N, C, H // 4, W // 4, seed=12, scale=2.0).to(DEV)
    lut = (torch.arange(C) % ncls).to(torch.uint8).to(DEV) if use_lut else None
    two = ops.label_epilogue(main, aux, (H, W), lut=lut, want_kld=True)
    h2 = torch.zeros(ncls, dtype=torch.int64, device=DEV)
    lab2 = ops.merge_labels([two['labels']], ncls, 1, 4, h2)
    h1 = torch.full((ncls,), 3, dtype=torch.int64, device=DEV)             # accumulates into what is there
    one = ops.label_epilogue_hist(main, aux, (H, W), h1, ncls, lut=lut, want_kld=True)
    assert torch.equal(one['labels'], lab2) and torch.equal(one['labels'], two['labels'])
    assert torch.equal(one['kld'], two['kld'])
    # the staged form against the definition: upsample both heads (ATen), then argmax / KL on the full-resolution logits
    mu = F.interpolate(main.cpu(), (H, W), mode='bilinear', align_corners=True)
    au = F.interpolate(aux.cpu(), (H, W), mode='bilinear', align_corners=True)
    o = mu + 0.5 * au
    srt = torch.sort(o, 1, descending=True)[0]
    sure = ((srt[:, 0] - srt[:, 1]) > 1e-4).numpy()
    ref = o.argmax(1).numpy()
    if lut is not None:
        ref = lut.cpu().numpy()[ref]
    assert np.array_equal(one['labels'].cpu().numpy()[sure], ref.astype(np.uint8)[sure]) and sure.mean() > 0.99
    p1, lp1, lp2 = F.softmax(mu, 1), F.log_softmax(mu, 1), F.log_softmax(au, 1)
    close(one['kld'], (p1 * lp1 - p1 * lp2).sum(1), atol=3e-5, rtol=1e-3)
    assert torch.equal(h1 - 3, h2) and int(h2.sum()) == N * H * W
    np.testing.assert_array_equal(h2.cpu().numpy(), np.bincount(lab2.cpu().numpy().ravel(), minlength=ncls))
    only = ops.label_epilogue_hist(main, None, (H, W), h1, ncls, lut=lut)                  # single-head nets, no KL map
    assert torch.equal(only['labels'], ops.label_epilogue(main, None, (H, W), lut=lut)['labels']) and 'kld' not in only
    with pytest.raises(RuntimeError, match='logit channels'):
        ops.label_epilogue_hist(torch.zeros(1, 25, 4, 4, device=DEV), None, (8, 8), torch.zeros(32, dtype=torch.int64, device=DEV), 32)


def test_bad_arguments_raise():
    from mspl_amd import ops
    with pytest.raises(RuntimeError, match='exceeds the LDS weight tile'):     # K per group > ~700 is not on the path
        ops.conv1x1(torch.zeros(1, 1024, 4, 8, device=DEV), torch.zeros(64, 1024, 1, 1, device=DEV), 1)
    with pytest.raises(RuntimeError, match='no CPU path'):
        ops.avgpool3x3s2(torch.zeros(1, 1, 4, 4))
    with pytest.raises(RuntimeError, match='float32'):
        ops.avgpool3x3s2(torch.zeros(1, 1, 4, 4, device=DEV, dtype=torch.float16))
    with pytest.raises(RuntimeError, match='does not match'):
        ops.conv1x1(torch.zeros(1, 8, 4, 4, device=DEV), torch.zeros(4, 3, 1, 1, device=DEV), 1)


@pytest.mark.parametrize('cfg', [(2, 24, 8, 12, 14, 22, True), (1, 12, 16, 5, 16, 30, False), (1, 8, 4, 6, 6, 9, True),
                                 (1, 32, 16, 13, 72, 120, False), (2, 16, 16, 7, 33, 47, True), (1, 8, 16, 5, 4, 4, True),
                                 (1, 16, 16, 20, 144, 240, False)])
def test_pyrpool_fused_equals_unfused_and_oracle(cfg):
    """Fused K6 kernel vs the branch-by-branch kernels vs the torch-CPU oracle, incl. odd sizes, tiles that do not
    divide the map, the clamp-to-5 branch sizes and a map too small for the fused kernel (falls back)."""
    from mspl_amd import layers as L
    from oracle import net as onet
    from tests.synth import synth_state_dict
    N, cin, P, cout, h, w, lbr = cfg
    m = L.EfficientPyrPool(cin, P, cout, last_layer_br=lbr)
    sd = synth_state_dict(m.state_dict(), 5)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    x = rnd(N, cin, h, w, seed=3)
    with torch.no_grad():
        yf = m(x.to(DEV))
        yu = m(x.to(DEV), fused=False)
        ref = onet.pyr_pool(x, {'m.' + k: v for k, v in sd.items()}, 'm', lbr)
    close(yu, ref, atol=5e-5)
    close(yf, ref, atol=5e-5)


@pytest.mark.parametrize('cfg', [(2, 16, 144, 240, [(72, 120), (15, 24)]), (1, 8, 18, 30, [(9, 15), (5, 5)]),
                                 (2, 4, 33, 47, [(17, 24), (5, 5)]), (1, 3, 7, 9, [(5, 5)]), (1, 2, 36, 60, [(18, 30), (5, 6), (36, 60)]),
                                 (2, 16, 72, 120, [(36, 60), (8, 12)]), (1, 4, 36, 60, [(18, 30), (5, 6)]), (1, 3, 40, 64, [(7, 9)]),
                                 (1, 2, 20, 36, [(10, 18)])])
def test_pyr_down_prep(cfg):
    """One-launch low-resolution pyramid maps == adaptive_avg_pool2d + depthwise 3x3 (torch CPU), incl. overlapping and
    ragged pooling windows, a branch as large as the map, and the LDS-size query.  Even row lengths with exact 2x2 and / or
    ~10x10 windows take the streaming form (all four decoder stages), everything else the band form."""
    from mspl_amd import ops
    N, P, h, w, sizes = cfg
    x = rnd(N, P, h, w, seed=4)
    ws = [rnd(P, 1, 3, 3, seed=10 + i) for i in range(len(sizes))]
    assert ops.pyr_down_prep_fits(x.shape, sizes)
    outs = ops.pyr_down_prep(x.to(DEV), sizes, [t.to(DEV) for t in ws])
    for o, sz, wt in zip(outs, sizes, ws):
        ref = F.conv2d(F.adaptive_avg_pool2d(x, sz), wt, padding=1, groups=P)
        close(o, ref, atol=2e-5)
    assert not ops.pyr_down_prep_fits((1, 1, 4096, 8192), [(5, 5)])     # 800-row windows: left to the per-branch kernels


@pytest.mark.parametrize('K,C', [(4, 5), (21, 21), (5, 13)])
def test_miou_areas(K, C):
    """Device MIOU.get_iou == the reference's uint8 + torch.histc arithmetic (oracle), incl. the 255 -> ignored wrap, classes
    beyond num_classes, logits and ready-made argmax maps."""
    from mspl_amd.metrics import MIOU
    from oracle import labels as olab
    N, H, W = 3, 37, 53
    logits = rnd(N, C, H, W, seed=11)
    tgt = torch.randint(0, C, (N, H, W), generator=torch.Generator().manual_seed(5))
    tgt[0, :5] = 255
    ri, ru = olab.miou_areas(logits, tgt, K)
    m = MIOU(K)
    gi, gu = m.get_iou(logits.to(DEV), tgt.to(DEV))
    assert np.array_equal(gi, ri) and np.allclose(gu, ru, rtol=0, atol=1e-3)
    gi2, gu2 = m.get_iou(logits.argmax(1).to(DEV), tgt.to(DEV))
    assert np.array_equal(gi2, ri) and np.allclose(gu2, ru, rtol=0, atol=1e-3)
    gi3, _ = m.get_iou((logits.to(DEV), logits.to(DEV)), tgt.to(DEV))       # (main, aux) tuple: first element
    assert np.array_equal(gi3, ri)


@pytest.mark.parametrize('cfg', [(2, 3, 8, 16, 128, 256), (1, 5, 7, 9, 20, 31), (1, 2, 12, 10, 5, 4), (1, 1, 1, 1, 6, 3)])
def test_bilinear_align_corners_false(cfg):
    """The default F.interpolate rule (half-pixel centres) of the DeepLab heads, incl. down-scaling and a 1x1 source."""
    from mspl_amd import ops
    N, C, Hi, Wi, Ho, Wo = cfg
    x = rnd(N, C, Hi, Wi, seed=8)
    close(ops.bilinear(x.to(DEV), (Ho, Wo), align_corners=False), F.interpolate(x, size=(Ho, Wo), mode='bilinear', align_corners=False),
          atol=1e-5)


@pytest.mark.parametrize('shape', [(2, 5, 16, 24), (1, 3, 15, 21), (2, 4, 9, 8), (1, 2, 1, 7), (16, 8, 144, 240)])
def test_avgpool_plane_sums_and_gate_from_sums(shape):
    """The pool that also leaves its input's plane sums, and the gate computed from them (= gap_gate without the second read)."""
    from mspl_amd import ops
    x = rnd(*shape, seed=31)
    xd = x.to(DEV)
    y, sums = ops.avgpool3x3s2(xd, plane_sums=True)
    close(y, F.avg_pool2d(x, 3, 2, 1))
    assert sums.shape[0] == shape[0] * shape[1]
    torch.testing.assert_close(sums.sum(1).cpu().view(shape[0], shape[1]), x.double().sum((2, 3)).float(), rtol=1e-5, atol=1e-3)
    w = rnd(6, shape[1], 1, 1, seed=32)
    g1 = ops.gate_from_sums(sums, w.to(DEV), shape[2] * shape[3])
    g0 = ops.gap_gate(xd, w.to(DEV))
    torch.testing.assert_close(g1, g0, rtol=1e-5, atol=1e-6)
    y2, sums2 = ops.avgpool3x3s2(xd, plane_sums=True)                      # deterministic: same bits every time
    assert torch.equal(sums2, sums) and torch.equal(ops.gate_from_sums(sums2, w.to(DEV), shape[2] * shape[3]), g1)


@pytest.mark.parametrize('shape', [(2, 6, 36, 60), (1, 3, 4, 4), (2, 5, 6, 20), (2, 4, 8, 24), (1, 2, 2, 8), (16, 128, 36, 60)])
def test_downsampler_pool_lean_forms_equal_the_strip_kernel(shape):
    """The DownSampler's pool call (nn_layers/eesp.py:131-144: avg_pool into channels [0, nin) + BatchNorm fold + image
    reinforcement + PReLU, with the input's plane sums) takes a lean kernel when W % 4 == 0: whole 16-byte strips (Wo % 4 == 0)
    or rows that end in a two-output strip (Wo % 4 == 2: 60 -> 30 columns at level 4 of a 480-wide input).  The same call into a
    destination slice at channel 1 takes the generic strip kernel: identical bits out, plane sums equal up to the order of the
    additions; both against the definition in torch-CPU fp32."""
    from mspl_amd import ops
    N, C, H, W = shape
    Ho, Wo, ctot = H // 2, W // 2, C + 2
    g = torch.Generator().manual_seed(77)
    x = torch.randn(N, C, H, W, generator=g)
    sc, sh, al = torch.rand(ctot, generator=g) + 0.5, torch.randn(ctot, generator=g), torch.rand(ctot, generator=g) * 0.3
    rw = torch.randn(ctot, 3, generator=g) * 0.5
    r = torch.randn(N, 3, Ho, Wo, generator=g)
    xd = x.to(DEV)
    lean = torch.zeros(N, ctot, Ho, Wo, device=DEV)
    _, s_lean = ops.avgpool3x3s2(xd, ops.Epi(sc.to(DEV), sh.to(DEV), al.to(DEV), reinf_r=r.to(DEV), reinf_w=rw.to(DEV)),
                                 out=(lean, 0), plane_sums=True)
    strip = torch.zeros(N, ctot, Ho, Wo, device=DEV)
    roll = lambda v: torch.roll(v, 1, 0).contiguous().to(DEV)
    _, s_strip = ops.avgpool3x3s2(xd, ops.Epi(roll(sc), roll(sh), roll(al), reinf_r=r.to(DEV), reinf_w=roll(rw)),
                                  out=(strip, 1), plane_sums=True)
    assert torch.equal(lean[:, :C], strip[:, 1:1 + C])
    assert torch.count_nonzero(lean[:, C:]) == 0 and torch.count_nonzero(strip[:, 0]) == 0      # nothing outside the slice
    torch.testing.assert_close(s_lean.sum(1), s_strip.sum(1), rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(s_lean.sum(1).cpu().view(N, C), x.double().sum((2, 3)).float(), rtol=1e-5, atol=1e-3)
    want = F.avg_pool2d(x, 3, 2, 1) * sc[:C].view(1, -1, 1, 1) + sh[:C].view(1, -1, 1, 1) + torch.einsum('ck,nkhw->nchw', rw[:C], r)
    want = torch.where(want > 0, want, want * al[:C].view(1, -1, 1, 1))
    close(lean[:, :C], want, atol=1e-5)


def test_label_epilogue_hist_fits_and_fallback():
    """Shapes outside the LDS-staged form (heads at or near the output resolution: staged rows wider than 256 columns) are reported
    by label_epilogue_hist_fits; SelfLabelPass then takes label_epilogue + merge_labels(S=1), the two-launch form, instead of
    raising.  Covered shapes of every class-count instantiation launch (the dynamic-LDS opt-in is set for all of them)."""
    from mspl_amd import ops
    main = rnd(2, 5, 64, 300, seed=21, scale=2.0).to(DEV)
    aux = rnd(2, 5, 32, 150, seed=22, scale=2.0).to(DEV)
    assert not ops.label_epilogue_hist_fits(main, aux, (64, 300))            # main head at the output resolution: 300 > 256 staged columns
    h = torch.zeros(5, dtype=torch.int64, device=DEV)
    with pytest.raises(RuntimeError):
        ops.label_epilogue_hist(main, aux, (64, 300), h, 5)
    two = ops.label_epilogue(main, aux, (64, 300))
    ops.merge_labels([two['labels']], 5, 1, 4, h)
    assert int(h.sum()) == 2 * 64 * 300
    for C in (3, 8, 11, 16, 19, 24):          # the <8>, <16>, <24> instantiations, with tiles above and below 64 KB
        m2 = rnd(1, C, 96, 240, seed=23 + C, scale=2.0).to(DEV)
        a2 = rnd(1, C, 48, 120, seed=24 + C, scale=2.0).to(DEV)
        assert ops.label_epilogue_hist_fits(m2, a2, (192, 480))
        hh = torch.zeros(C, dtype=torch.int64, device=DEV)
        one = ops.label_epilogue_hist(m2, a2, (192, 480), hh, C)
        assert torch.equal(one['labels'], ops.label_epilogue(m2, a2, (192, 480))['labels']) and int(hh.sum()) == 192 * 480


@pytest.mark.parametrize('cfg', [(2, 6, 5, 16, 30), (1, 16, 5, 40, 72), (3, 4, 3, 33, 65), (2, 8, 5, 7, 9), (1, 3, 1, 18, 130), (2, 5, 4, 35, 32)])
def test_pyrpool_merge_from_kept_branches(cfg):
    """mspl_pyrpool_merge_fwd against the definition (BatchNorm fold + PReLU, channel shuffle, grouped 3x3 with zero padding after the
    activation): both tile shapes, ragged tiles, widths that are not a multiple of 4, fewer than five branches."""
    import torch.nn.functional as F
    from mspl_amd import ops
    N, P, nb, h, w = cfg
    g = torch.Generator().manual_seed(3)
    z = torch.randn(N, nb * P, h, w, generator=g)
    sc, sh, al = torch.rand(nb * P, generator=g) + 0.5, torch.randn(nb * P, generator=g), torch.rand(nb * P, generator=g) * 0.3
    mw = torch.randn(P, nb, 3, 3, generator=g) * 0.3
    a = F.prelu(z * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1), al)
    a = a.view(N, nb, P, h, w).transpose(1, 2).reshape(N, nb * P, h, w)           # Shuffle(groups=nb)
    want = F.conv2d(a, mw, None, 1, 1, 1, P)
    got = ops.pyrpool_merge(z.to(DEV), sc.to(DEV), sh.to(DEV), al.to(DEV), mw.to(DEV))
    torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=1e-5)
