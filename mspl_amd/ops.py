"""Tensor-level wrappers of the C ABI (forward ops).

PyTorch is plumbing only here: it owns device memory (caching allocator) and the stream.  Every wrapper
checks device/dtype/contiguity on the host before a pointer reaches a kernel, allocates the output with
torch.empty, and launches on torch's current stream so the calls are capturable by torch.cuda.graph().
"""
import ctypes

import torch

from . import _native as nat
from ._native import Epilogue, check, lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError('mspl_amd: %s must be a CUDA (ROCm) tensor; there is no CPU path' % name)
    if t.dtype != torch.float32:
        raise RuntimeError('mspl_amd: %s must be float32, got %s' % (name, t.dtype))
    if not t.is_contiguous():
        t = t.contiguous()
    return t


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _vec(t, n, name):
    if t is None:
        return None
    t = _f32(t, name)
    if t.numel() != n:
        raise RuntimeError('mspl_amd: %s has %d elements, expected %d' % (name, t.numel(), n))
    return t


class Epi:
    """Python mirror of mspl_epilogue_t (see include/mspl_hip.h for the exact order of operations)."""
    __slots__ = ('scale', 'shift', 'alpha', 'pre_add', 'residual', 'reinf_r', 'reinf_w', 'gate', 'raw_out')

    def __init__(self, scale=None, shift=None, alpha=None, pre_add=None, residual=None, reinf_r=None,
                 reinf_w=None, gate=None, raw_out=None):
        self.scale, self.shift, self.alpha = scale, shift, alpha
        self.pre_add, self.residual = pre_add, residual
        self.reinf_r, self.reinf_w, self.gate = reinf_r, reinf_w, gate
        self.raw_out = raw_out          # convolutions only: a tensor of the destination's shape that also receives the bare result


# Launch-shape preference of the calls issued by this THREAD (mspl_epilogue_t.flags / MSPL_LAUNCH_*): per call, carried in every
# epilogue struct -- the library itself holds no mutable state (rounds 1-2 flipped a process-wide switch around every lane launch).
import threading
_LAUNCH = threading.local()


class launch_flags(object):
    """with launch_flags(throughput=True): every op issued by this thread asks for launch shapes that favour steady-state
    efficiency over the latency of a lone launch (PipelinedLabelPass captures its lanes like this).  Results never change."""

    def __init__(self, throughput=False, k2_stream=None):
        self.value = nat.LAUNCH_THROUGHPUT if throughput else 0
        if k2_stream is not None:       # False: never the streaming form of the stride-2 depthwise launch; True: whenever the shape allows
            self.value |= nat.LAUNCH_K2_STREAM_FORCE if k2_stream else nat.LAUNCH_K2_STREAM_OFF

    def __enter__(self):
        self.prev = getattr(_LAUNCH, 'flags', 0)
        _LAUNCH.flags = self.value
        return self

    def __exit__(self, *exc):
        _LAUNCH.flags = self.prev


def current_launch_flags():
    return getattr(_LAUNCH, 'flags', 0)


def _build(ep, out, coff, N, C, hw):
    """Validate an Epi against the destination tensor `out` (N, ctot, ...) and build the C struct."""
    ctot = out.shape[1]
    if coff < 0 or coff + C > ctot:
        raise RuntimeError('mspl_amd: channel slice [%d,%d) outside destination with %d channels' % (coff, coff + C, ctot))
    s = Epilogue()
    s.struct_size = ctypes.sizeof(Epilogue)
    s.flags = current_launch_flags()
    s.out_ctot, s.out_coff = ctot, coff
    keep = []
    if ep is not None:
        for f in ('scale', 'shift', 'alpha'):
            v = _vec(getattr(ep, f), ctot, f)
            keep.append(v)
            setattr(s, f, None if v is None else v.data_ptr())
        for f in ('pre_add', 'residual'):
            v = getattr(ep, f)
            if v is not None:
                v = _f32(v, f)
                if v.numel() != N * ctot * hw:
                    raise RuntimeError('mspl_amd: %s shape %s does not match destination %s' % (f, tuple(v.shape), tuple(out.shape)))
                keep.append(v)
                setattr(s, f, v.data_ptr())
        if ep.reinf_r is not None:
            r = _vec(ep.reinf_r, N * 3 * hw, 'reinf_r')
            w = _vec(ep.reinf_w, ctot * 3, 'reinf_w')
            keep += [r, w]
            s.reinf_r, s.reinf_w = r.data_ptr(), w.data_ptr()
        if ep.gate is not None:
            g = _vec(ep.gate, N * ctot, 'gate')
            keep.append(g)
            s.gate = g.data_ptr()
        if ep.raw_out is not None:
            r = ep.raw_out
            if not r.is_cuda or r.dtype != torch.float32 or not r.is_contiguous() or r.numel() != N * C * hw or coff != 0 or ctot != C:
                raise RuntimeError('mspl_amd: raw_out must be a contiguous float32 CUDA tensor of the (un-sliced) destination shape')
            keep.append(r)
            s.raw_out = r.data_ptr()
    return s, keep


def _dest(out, shape, like):
    if out is None:
        return torch.empty(shape, device=like.device, dtype=torch.float32), 0
    dst, coff = out
    if (dst.shape[0],) + tuple(dst.shape[2:]) != (shape[0],) + tuple(shape[2:]) or not dst.is_contiguous() \
            or dst.dtype != torch.float32 or not dst.is_cuda:
        raise RuntimeError('mspl_amd: destination %s incompatible with result %s' % (tuple(dst.shape), tuple(shape)))
    return dst, coff


def eesp_dw_hff(x, w4, dil, stride, ep=None, out=None):
    """K2.  x (N,n,H,W), w4 (4,n,3,3) -> (N,4n,Ho,Wo); `out=(tensor, channel_offset)` writes a slice."""
    x, w4 = _f32(x, 'x'), _f32(w4, 'w4')
    N, n, H, W = x.shape
    if tuple(w4.shape) != (4, n, 3, 3):
        raise RuntimeError('mspl_amd: eesp_dw_hff weight %s, expected (4,%d,3,3)' % (tuple(w4.shape), n))
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    dst, coff = _dest(out, (N, 4 * n, Ho, Wo), x)
    s, keep = _build(ep, dst, coff, N, 4 * n, Ho * Wo)
    d = (ctypes.c_int32 * 4)(*[int(v) for v in dil])
    check(lib.mspl_eesp_dw_hff_fwd(_p(x), _p(w4), d, stride, N, n, H, W, ctypes.byref(s), _p(dst), _stream()))
    return dst


def eesp_proj_dw_hff_fits(shape, n, groups, dilations, stride):
    """True when K1 + K2 of an EESP block run as one launch for an input of `shape` (N,Cin,H,W)."""
    if stride != 1:
        return False
    N, Cin, H, W = [int(v) for v in shape]
    d = (ctypes.c_int32 * 4)(*[int(v) for v in dilations])
    return bool(lib.mspl_eesp_proj_dw_hff_fits(N, Cin, int(n), int(groups), H, W, d, current_launch_flags()))


def eesp_proj_dw_hff(x, wproj, pscale, pshift, palpha, w4, dilations, groups, ep=None, out=None):
    """K1 + K2 (stride 1): PReLU(BN(grouped 1x1(x))) -> four dilated depthwise 3x3 + HFF + cat + epilogue, one launch."""
    x = _f32(x, 'x')
    N, Cin, H, W = x.shape
    wproj = _f32(wproj, 'projection weight')
    n = wproj.shape[0]
    w4 = _vec(w4, 4 * n * 9, 'w')
    ps, pb, pa = _vec(pscale, n, 'pscale'), _vec(pshift, n, 'pshift'), _vec(palpha, n, 'palpha')
    dst, coff = _dest(out, (N, 4 * n, H, W), x)
    s, keep = _build(ep, dst, coff, N, 4 * n, H * W)
    d = (ctypes.c_int32 * 4)(*[int(v) for v in dilations])
    check(lib.mspl_eesp_proj_dw_hff_fwd(_p(x), _p(wproj), _p(ps), _p(pb), _p(pa), _p(w4), d, N, Cin, n, int(groups), H, W,
                                        ctypes.byref(s), _p(dst), _stream()))
    return dst


def eesp_dw_exp_fits(shape, dilations):
    """True when K2 + K3 of a stride-1 EESP block run as one launch for a reduced tensor of `shape` (N,n,H,W)."""
    N, n, H, W = [int(v) for v in shape]
    d = (ctypes.c_int32 * 4)(*[int(v) for v in dilations])
    return bool(lib.mspl_eesp_dw_exp_fits(N, n, H, W, d, current_launch_flags()))


def eesp_dw_exp_pack(w4, bscale, bshift, balpha, wexp, H, W, dilations):
    """The packed parameter block of mspl_eesp_dw_exp_fwd (depends on the weights only: cache it per weight version)."""
    w4 = _f32(w4, 'w4')
    n = w4.shape[1]
    if tuple(w4.shape) != (4, n, 3, 3):
        raise RuntimeError('mspl_amd: eesp_dw_exp weight %s, expected (4,%d,3,3)' % (tuple(w4.shape), n))
    bs, bb, ba = _vec(bscale, 4 * n, 'bscale'), _vec(bshift, 4 * n, 'bshift'), _vec(balpha, 4 * n, 'balpha')
    wexp = _vec(wexp, 4 * n * n, 'expansion weight')
    packed = torch.empty(int(lib.mspl_eesp_dw_exp_pack_floats(n)), device=w4.device, dtype=torch.float32)
    d = (ctypes.c_int32 * 4)(*[int(v) for v in dilations])
    check(lib.mspl_eesp_dw_exp_pack(_p(w4), _p(bs), _p(bb), _p(ba), _p(wexp), n, int(H), int(W), d, _p(packed), _stream()))
    return packed


def eesp_dw_exp_next_pack(w1):
    """The next block's proj_1x1 weight (n, n[, 1, 1]) in the operand order of mspl_eesp_dw_exp_next_fwd (cache it per weight version)."""
    w1 = _f32(w1, 'next projection weight')
    n = w1.shape[0]
    if w1.numel() != n * n:
        raise RuntimeError('mspl_amd: next projection weight %s, expected (%d,%d)' % (tuple(w1.shape), n, n))
    packed = torch.empty(int(lib.mspl_eesp_dw_exp_next_pack_floats(n)), device=w1.device, dtype=torch.float32)
    check(lib.mspl_eesp_dw_exp_next_pack(_p(w1), n, _p(packed), _stream()))
    return packed


def eesp_dw_exp(r, packed, dilations, ep, next_proj=None):
    """K2 + K3 of a stride-1 EESP block: r (N,n,H,W) -> (N,4n,H,W); ep carries conv_1x1_exp's folded BN, module_act's slope
    and the residual (the block's input).  next_proj = (packed weight, scale, shift, alpha) of the FOLLOWING block's proj_1x1: the
    launch then also returns that block's reduced tensor (N,n,H,W): (y, r_next)."""
    r = _f32(r, 'r')
    N, n, H, W = r.shape
    packed = _vec(packed, int(lib.mspl_eesp_dw_exp_pack_floats(n)), 'packed')
    dst = torch.empty((N, 4 * n, H, W), device=r.device, dtype=torch.float32)
    s, keep = _build(ep, dst, 0, N, 4 * n, H * W)
    d = (ctypes.c_int32 * 4)(*[int(v) for v in dilations])
    if next_proj is None:
        check(lib.mspl_eesp_dw_exp_fwd(_p(r), _p(packed), d, N, n, H, W, ctypes.byref(s), _p(dst), _stream()))
        return dst
    npk = _vec(next_proj[0], int(lib.mspl_eesp_dw_exp_next_pack_floats(n)), 'next_packed')
    ns, nb, na = _vec(next_proj[1], n, 'nscale'), _vec(next_proj[2], n, 'nshift'), _vec(next_proj[3], n, 'nalpha')
    rn = torch.empty((N, n, H, W), device=r.device, dtype=torch.float32)
    check(lib.mspl_eesp_dw_exp_next_fwd(_p(r), _p(packed), d, N, n, H, W, ctypes.byref(s), _p(dst), _p(npk), _p(ns), _p(nb), _p(na), _p(rn),
                                        _stream()))
    return dst, rn


def conv1x1(x, w, groups=1, ep=None, out=None):
    """K1/K3.  x (N,Cin,H,W), w (Cout,Cin/groups[,1,1])."""
    x, w = _f32(x, 'x'), _f32(w, 'w')
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    if w.numel() != Cout * (Cin // groups) or Cin % groups or Cout % groups:
        raise RuntimeError('mspl_amd: conv1x1 weight %s does not match Cin=%d groups=%d' % (tuple(w.shape), Cin, groups))
    dst, coff = _dest(out, (N, Cout, H, W), x)
    s, keep = _build(ep, dst, coff, N, Cout, H * W)
    check(lib.mspl_conv1x1_fwd(_p(x), _p(w), N, Cin, Cout, groups, H * W, ctypes.byref(s), _p(dst), _stream()))
    return dst


def conv3x3(x, w, groups=1, stride=1, shuffle_groups=0, ep=None, out=None):
    x, w = _f32(x, 'x'), _f32(w, 'w')
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    if tuple(w.shape[2:]) != (3, 3) or Cin % groups or Cout % groups or w.shape[1] != Cin // groups:
        raise RuntimeError('mspl_amd: conv3x3 weight %s does not match Cin=%d groups=%d' % (tuple(w.shape), Cin, groups))
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    dst, coff = _dest(out, (N, Cout, Ho, Wo), x)
    s, keep = _build(ep, dst, coff, N, Cout, Ho * Wo)
    check(lib.mspl_conv3x3_fwd(_p(x), _p(w), N, Cin, Cout, groups, H, W, stride, shuffle_groups, ctypes.byref(s),
                               _p(dst), _stream()))
    return dst


def pack_dense_weight(w):
    """(Cout, Cin, k, k) conv weight -> the tap-major (k*k, Cout, Cin) layout dense_conv reads (a permute copy)."""
    w = _f32(w, 'w')
    Cout, Cin, kh, kw = w.shape
    return w.permute(2, 3, 0, 1).reshape(kh * kw, Cout, Cin).contiguous()


def dense_conv(x, w_packed, ksize, dilation=1, ep=None, out=None):
    """K13.  Dense 1x1 / dilated 3x3 convolution (padding = dilation) on the matrix cores; w_packed from pack_dense_weight."""
    x, w_packed = _f32(x, 'x'), _f32(w_packed, 'w_packed')
    N, Cin, H, W = x.shape
    taps = ksize * ksize
    if w_packed.dim() != 3 or w_packed.shape[0] != taps or w_packed.shape[2] != Cin:
        raise RuntimeError('mspl_amd: dense_conv packed weight %s does not match k=%d Cin=%d' % (tuple(w_packed.shape), ksize, Cin))
    Cout = w_packed.shape[1]
    dst, coff = _dest(out, (N, Cout, H, W), x)
    s, keep = _build(ep, dst, coff, N, Cout, H * W)
    check(lib.mspl_dense_conv_fwd(_p(x), _p(w_packed), N, Cin, Cout, H, W, ksize, int(dilation), ctypes.byref(s), _p(dst),
                                  _stream()))
    return dst


def avgpool3x3s2(x, ep=None, out=None, plane_sums=False):
    """3x3 / stride 2 / pad 1 average pool (+ epilogue).  plane_sums=True: returns (pooled, partial sums (N*C, nblk) of x per
    plane) -- for a gate over the same tensor (gate_from_sums) without reading it again."""
    x = _f32(x, 'x')
    N, C, H, W = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    dst, coff = _dest(out, (N, C, Ho, Wo), x)
    s, keep = _build(ep, dst, coff, N, C, Ho * Wo)
    if not plane_sums:
        check(lib.mspl_avgpool3x3s2_fwd(_p(x), N, C, H, W, ctypes.byref(s), _p(dst), _stream()))
        return dst
    nblk = lib.mspl_avgpool3x3s2_psum_blocks(H, W)
    if nblk <= 0:
        check(nblk)
    sums = torch.empty((N * C, nblk), device=x.device, dtype=torch.float32)
    check(lib.mspl_avgpool3x3s2_psum_fwd(_p(x), N, C, H, W, ctypes.byref(s), _p(dst), _p(sums), _stream()))
    return dst, sums


def gate_from_sums(plane_sums, w, hw):
    """sigmoid(W . sum_j plane_sums[:, j] / hw) -> (N, Cout): gap_gate without re-reading the tensor."""
    w = _f32(w, 'w')
    Cout = w.shape[0]
    Cin = w.numel() // Cout
    planes, nblk = plane_sums.shape
    N = planes // Cin
    if planes != N * Cin or plane_sums.dtype != torch.float32 or not plane_sums.is_contiguous():
        raise RuntimeError('mspl_amd: gate_from_sums: sums %s do not match Cin=%d' % (tuple(plane_sums.shape), Cin))
    gate = torch.empty((N, Cout), device=w.device, dtype=torch.float32)
    check(lib.mspl_gate_from_sums_fwd(_p(plane_sums), _p(w), N, Cin, Cout, nblk, int(hw), _p(gate), _stream()))
    return gate


def bilinear(x, size, ep=None, out=None, align_corners=True):
    """Bilinear resize to size=(Ho,Wo); align_corners=True is the ESPNet decoders' rule, False F.interpolate's default."""
    x = _f32(x, 'x')
    N, C, H, W = x.shape
    Ho, Wo = int(size[0]), int(size[1])
    dst, coff = _dest(out, (N, C, Ho, Wo), x)
    s, keep = _build(ep, dst, coff, N, C, Ho * Wo)
    check(lib.mspl_bilinear_fwd(_p(x), N, C, H, W, Ho, Wo, 1 if align_corners else 0, ctypes.byref(s), _p(dst), _stream()))
    return dst


def adaptive_avgpool(x, size, ep=None, out=None):
    x = _f32(x, 'x')
    N, C, H, W = x.shape
    Ho, Wo = int(size[0]), int(size[1])
    dst, coff = _dest(out, (N, C, Ho, Wo), x)
    s, keep = _build(ep, dst, coff, N, C, Ho * Wo)
    check(lib.mspl_adaptive_avgpool_fwd(_p(x), N, C, H, W, Ho, Wo, ctypes.byref(s), _p(dst), _stream()))
    return dst


def pointwise(x, ep, out=None):
    """Apply the epilogue elementwise (BatchNorm+PReLU blocks)."""
    x = _f32(x, 'x')
    N, C = x.shape[:2]
    hw = x[0, 0].numel()
    dst = torch.empty_like(x) if out is None else out
    s, keep = _build(ep, dst, 0, N, C, hw)
    check(lib.mspl_pointwise_fwd(_p(x), N, C, hw, ctypes.byref(s), _p(dst), _stream()))
    return dst


def pyrpool_merge(zcat, scale, shift, alpha, merge_w, out=None):
    """merge_layer.0's fold + PReLU, Shuffle(groups=nb) and merge_layer.2's grouped 3x3 over kept branch values zcat (N, nb*P, h, w);
    merge_w (P, nb, 3, 3).  Returns the bare convolution result (N, P, h, w)."""
    zcat, merge_w = _f32(zcat, 'zcat'), _f32(merge_w, 'merge_w')
    N, C, h, w = zcat.shape
    P, nb = merge_w.shape[:2]
    if C != P * nb or tuple(merge_w.shape[2:]) != (3, 3) or any(t.numel() != C for t in (scale, shift, alpha)):
        raise RuntimeError('mspl_amd: pyrpool_merge operands do not match: zcat %s merge_w %s' % (tuple(zcat.shape), tuple(merge_w.shape)))
    dst = torch.empty((N, P, h, w), device=zcat.device, dtype=torch.float32) if out is None else out
    check(lib.mspl_pyrpool_merge_fwd(_p(zcat), N, P, h, w, nb, _p(_f32(scale, 'scale')), _p(_f32(shift, 'shift')), _p(_f32(alpha, 'alpha')),
                                     _p(merge_w), _p(dst), _stream()))
    return dst


def gap_gate(x, w):
    """sigmoid(W . mean_hw(x)) -> (N, Cout)."""
    x, w = _f32(x, 'x'), _f32(w, 'w')
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    if w.numel() != Cout * Cin:
        raise RuntimeError('mspl_amd: gap_gate weight %s does not match Cin=%d' % (tuple(w.shape), Cin))
    ws = torch.empty((N, Cin), device=x.device, dtype=torch.float32)
    gate = torch.empty((N, Cout), device=x.device, dtype=torch.float32)
    check(lib.mspl_gap_gate_fwd(_p(x), _p(w), N, Cin, Cout, H * W, _p(ws), _p(gate), _stream()))
    return gate


def fusion_gate(z, rgb, depth):
    """FusionGate blend (nn_layers/fusion_gate.py:36-44): sigmoid(z)-weighted mix, or rgb + depth when z is None."""
    rgb, depth = _f32(rgb, 'rgb'), _f32(depth, 'depth')
    if rgb.shape != depth.shape or (z is not None and z.shape != rgb.shape):
        raise RuntimeError('mspl_amd: fusion_gate operands differ in shape: %s %s %s'
                           % (tuple(rgb.shape), tuple(depth.shape), None if z is None else tuple(z.shape)))
    out = torch.empty_like(rgb)
    check(lib.mspl_fusion_gate_fwd(None if z is None else _p(_f32(z, 'z')), _p(rgb), _p(depth), rgb.numel(), _p(out),
                                   _stream()))
    return out


def pyr_down_prep_fits(shape, sizes):
    """True when pyr_down_prep can stage a row band of an (N,P,h,w) map with these branch sizes in LDS
    (else: adaptive_avgpool + conv3x3 per branch)."""
    nb = len(sizes)
    N, P, h, w = [int(v) for v in shape]
    if nb < 1 or nb > 4 or any(int(s[0]) > h or int(s[1]) > w for s in sizes):
        return False
    hs = (ctypes.c_int32 * nb)(*[int(s[0]) for s in sizes])
    ws = (ctypes.c_int32 * nb)(*[int(s[1]) for s in sizes])
    return lib.mspl_pyr_down_prep_lds_bytes(N, P, h, w, nb, hs, ws) > 0


def pyr_down_prep(x, sizes, stage_ws, keep_pooled=False):
    """K6 prologue: [dw3x3(adaptive_avg_pool2d(x, size_i)) for each low-resolution branch] in one launch.  keep_pooled=True (the
    training forward): returns (those maps, the pooled maps themselves)."""
    x = _f32(x, 'x')
    N, P, h, w = x.shape
    nb = len(sizes)
    hs = (ctypes.c_int32 * nb)(*[int(s[0]) for s in sizes])
    ws = (ctypes.c_int32 * nb)(*[int(s[1]) for s in sizes])
    sw, op, pp = (ctypes.c_void_p * nb)(), (ctypes.c_void_p * nb)(), (ctypes.c_void_p * nb)()
    outs, pools, keep = [], [], []
    for i in range(nb):
        t = _f32(stage_ws[i], 'stage weight')
        if t.numel() != P * 9:
            raise RuntimeError('mspl_amd: pyramid stage weight %s, expected (%d,1,3,3)' % (tuple(t.shape), P))
        keep.append(t)
        sw[i] = t.data_ptr()
        o = torch.empty((N, P, int(sizes[i][0]), int(sizes[i][1])), device=x.device, dtype=torch.float32)
        outs.append(o)
        op[i] = o.data_ptr()
        if keep_pooled:
            pools.append(torch.empty_like(o))
            pp[i] = pools[-1].data_ptr()
    if keep_pooled:
        check(lib.mspl_pyr_down_prep_train_fwd(_p(x), N, P, h, w, nb, hs, ws, sw, op, pp, _stream()))
        return outs, pools
    check(lib.mspl_pyr_down_prep_fwd(_p(x), N, P, h, w, nb, hs, ws, sw, op, _stream()))
    return outs


def pyrpool_fused(x, sizes, stage_ws, down_es, br_scale, br_shift, br_alpha, merge_w, ep=None, out=None):
    """K6.  x (N,P,h,w); sizes: [(hs,ws)] per branch; stage_ws / down_es: per-branch tensors or None."""
    x = _f32(x, 'x')
    N, P, h, w = x.shape
    nb = len(sizes)
    merge_w = _f32(merge_w, 'merge_w')
    if tuple(merge_w.shape) != (P, nb, 3, 3):
        raise RuntimeError('mspl_amd: pyrpool merge weight %s, expected (%d,%d,3,3)' % (tuple(merge_w.shape), P, nb))
    hs = (ctypes.c_int32 * nb)(*[int(s[0]) for s in sizes])
    ws = (ctypes.c_int32 * nb)(*[int(s[1]) for s in sizes])
    keep = []
    sw, de = (ctypes.c_void_p * nb)(), (ctypes.c_void_p * nb)()
    for i in range(nb):
        if stage_ws[i] is not None:
            t = _f32(stage_ws[i], 'stage weight')
            if t.numel() != P * 9:
                raise RuntimeError('mspl_amd: pyrpool stage weight %s, expected (%d,1,3,3)' % (tuple(t.shape), P))
            keep.append(t)
            sw[i] = t.data_ptr()
        if down_es[i] is not None:
            t = _f32(down_es[i], 'down map')
            if tuple(t.shape) != (N, P, int(sizes[i][0]), int(sizes[i][1])):
                raise RuntimeError('mspl_amd: pyrpool low-res map %s does not match branch size %s' % (tuple(t.shape), sizes[i]))
            keep.append(t)
            de[i] = t.data_ptr()
    bs, bh, ba = _vec(br_scale, nb * P, 'br_scale'), _vec(br_shift, nb * P, 'br_shift'), _vec(br_alpha, nb * P, 'br_alpha')
    dst, coff = _dest(out, (N, P, h, w), x)
    s, k2 = _build(ep, dst, coff, N, P, h * w)
    check(lib.mspl_pyrpool_fused_fwd(_p(x), N, P, h, w, nb, hs, ws, sw, de, _p(bs), _p(bh), _p(ba), _p(merge_w),
                                     ctypes.byref(s), _p(dst), _stream()))
    return dst


def label_epilogue(main, aux, size, lut=None, want_labels=True, want_prob=False, want_kld=False,
                   want_logits=False):
    """K8+K9.  Returns dict with any of labels (uint8 N,H,W), prob, kld, main_up, aux_up."""
    main = _f32(main, 'main')
    N, C, Hm, Wm = main.shape
    Ha = Wa = 0
    if aux is not None:
        aux = _f32(aux, 'aux')
        if aux.shape[0] != N or aux.shape[1] != C:
            raise RuntimeError('mspl_amd: aux logits %s do not match main %s' % (tuple(aux.shape), tuple(main.shape)))
        Ha, Wa = aux.shape[2:]
    H, W = int(size[0]), int(size[1])
    dev = main.device
    res = {}
    if lut is not None:
        if lut.dtype != torch.uint8 or not lut.is_cuda or lut.numel() < C:
            raise RuntimeError('mspl_amd: lut must be a CUDA uint8 tensor with >= %d entries' % C)
    if want_labels:
        res['labels'] = torch.empty((N, H, W), device=dev, dtype=torch.uint8)
    if want_prob:
        res['prob'] = torch.empty((N, C, H, W), device=dev, dtype=torch.float32)
    if want_kld:
        res['kld'] = torch.empty((N, H, W), device=dev, dtype=torch.float32)
    if want_logits:
        res['main_up'] = torch.empty((N, C, H, W), device=dev, dtype=torch.float32)
        if aux is not None:
            res['aux_up'] = torch.empty((N, C, H, W), device=dev, dtype=torch.float32)
    check(lib.mspl_label_epilogue_fwd(_p(main), _p(aux), N, C, Hm, Wm, Ha, Wa, H, W, _p(lut), _p(res.get('labels')),
                                      _p(res.get('prob')), _p(res.get('kld')), _p(res.get('main_up')),
                                      _p(res.get('aux_up')), _stream()))
    return res


def label_epilogue_hist_fits(main, aux, size):
    """True when label_epilogue_hist covers these head shapes (else: label_epilogue + merge_labels(S=1))."""
    N, C, Hm, Wm = [int(v) for v in main.shape]
    Ha, Wa = ([int(v) for v in aux.shape[2:]] if aux is not None else (0, 0))
    return bool(lib.mspl_label_epilogue_hist_fits(N, C, Hm, Wm, Ha, Wa, int(size[0]), int(size[1])))


def label_epilogue_hist(main, aux, size, hist, num_classes, lut=None, want_kld=False):
    """K8+K9 + class histogram in one launch (single-source pass).  Requires C <= min(24, num_classes) so that every label is a
    counted class -- then it equals label_epilogue followed by merge_labels(S=1, thresh=1).  Returns dict(labels[, kld])."""
    main = _f32(main, 'main')
    N, C, Hm, Wm = main.shape
    Ha = Wa = 0
    if aux is not None:
        aux = _f32(aux, 'aux')
        if aux.shape[0] != N or aux.shape[1] != C:
            raise RuntimeError('mspl_amd: aux logits %s do not match main %s' % (tuple(aux.shape), tuple(main.shape)))
        Ha, Wa = aux.shape[2:]
    H, W = int(size[0]), int(size[1])
    if hist.dtype != torch.int64 or not hist.is_cuda or hist.numel() < num_classes:
        raise RuntimeError('mspl_amd: hist must be a CUDA int64 tensor with >= num_classes entries')
    if lut is not None and (lut.dtype != torch.uint8 or not lut.is_cuda or lut.numel() < C):
        raise RuntimeError('mspl_amd: lut must be a CUDA uint8 tensor with >= %d entries' % C)
    res = {'labels': torch.empty((N, H, W), device=main.device, dtype=torch.uint8)}
    if want_kld:
        res['kld'] = torch.empty((N, H, W), device=main.device, dtype=torch.float32)
    nbytes = int(lib.mspl_label_epilogue_hist_workspace_bytes(N, H, W))
    ws = torch.empty(max(nbytes, 4), device=main.device, dtype=torch.uint8)         # per-workgroup partial histograms
    check(lib.mspl_label_epilogue_hist_fwd(_p(main), _p(aux), N, C, Hm, Wm, Ha, Wa, H, W, _p(lut), _p(res['labels']),
                                           _p(res.get('kld')), _p(hist), int(num_classes), _p(ws), nbytes, _stream()))
    return res


def merge_labels(sources, num_classes, thresh, fill=4, hist=None):
    """K10.  sources: list of uint8 CUDA tensors of identical shape.  hist: uint64-as-int64 CUDA tensor of
    num_classes entries, accumulated into (caller zeroes), or None."""
    S = len(sources)
    if S < 1:
        raise RuntimeError('mspl_amd: merge_labels needs at least one source')
    src = []
    for t in sources:
        if t.dtype != torch.uint8 or not t.is_cuda:
            raise RuntimeError('mspl_amd: merge_labels sources must be CUDA uint8 tensors')
        if t.shape != sources[0].shape:
            raise RuntimeError('mspl_amd: merge_labels sources differ in shape')
        src.append(t.contiguous())
    out = torch.empty_like(src[0])
    if hist is not None and (hist.dtype != torch.int64 or not hist.is_cuda or hist.numel() < num_classes):
        raise RuntimeError('mspl_amd: hist must be a CUDA int64 tensor with >= num_classes entries')
    arr = (ctypes.c_void_p * S)(*[t.data_ptr() for t in src])
    check(lib.mspl_merge_labels_fwd(arr, S, out.numel(), num_classes, int(thresh), int(fill), _p(out), _p(hist),
                                    _stream()))
    return out
