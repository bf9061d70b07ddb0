"""torch.autograd.Function wrappers: HIP forward + HIP backward for every op of the training step.

Used when gradients are enabled (the uest self-training loop, uest_seg_multi_os.py:958-1089).  BatchNorm is frozen
(eval mode, as in the reference's default run): it is a per-channel affine whose scale/shift are tiny differentiable
expressions of gamma/beta and the running statistics, so gamma/beta still receive gradients.
torch.cat / view / transpose in this path are pure data movement (no arithmetic).
"""
import ctypes
import os

import torch

from . import ops
from ._native import check, lib
from .ops import Epi, _p, _stream


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# ---- direct gradient sinks ---------------------------------------------------------------------------------------------
# Inside `with grad_sinks():` the parameter-gradient kernels accumulate straight into an existing `param.grad` buffer (the
# flat gradient buffer of FlatAdam / FlatSGD, zeroed once per step) and return None to autograd, instead of producing a
# temporary that AccumulateGrad adds with one tiny kernel per parameter (340 of them per step).  Only valid for
# `loss.backward()` with accumulating .grad semantics, which is why it is opt-in (the training steps opt in).
_SINKS = [False]


class grad_sinks(object):
    def __init__(self, enabled=True):
        self.enabled = enabled and os.environ.get('MSPL_GRAD_SINKS', '1') != '0'

    def __enter__(self):
        self.prev = _SINKS[0]
        _SINKS[0] = self.enabled

    def __exit__(self, *exc):
        _SINKS[0] = self.prev
        if exc and exc[0] is not None:
            WGRADS.clear()
        else:
            WGRADS.flush()        # the queued 1x1 weight gradients of this backward (nothing waited for them)


class WgradQueue(object):
    """Weight gradients of the grouped 1x1 convolutions that go straight into a gradient sink, queued during a backward and sent
    out together (mspl_conv1x1_wgrad_batch: runs of up to 16 problems per launch).  Nothing in a backward chain consumes a weight
    gradient, so the only ordering that matters is "before the optimizer reads the buffer": `grad_sinks().__exit__` flushes, on
    the stream the step runs on (inside a graph capture: the capture's stream).  The queue keeps the two operands of every problem
    alive until then.  A uest train step issued 33 such launches of 8-25 us each; MSPL_WGRAD_BATCH=0 restores them (A/B aid)."""

    def __init__(self):
        self.items = []
        self.enabled = os.environ.get('MSPL_WGRAD_BATCH', '1') != '0'

    def add(self, gy, x, N, Cin, Cout, groups, HW, sink, rowscale=None):
        self.items.append((gy, x, sink, int(N), int(Cin), int(Cout), int(groups), int(HW), rowscale))
        if len(self.items) >= 64:
            self.flush()

    def clear(self):
        self.items = []

    def flush(self):
        items, self.items = self.items, []
        n = len(items)
        if n == 0:
            return
        arr = lambda k: (ctypes.c_void_p * n)(*[it[k].data_ptr() for it in items])      # noqa: E731
        ints = lambda k: (ctypes.c_int32 * n)(*[it[k] for it in items])                 # noqa: E731
        rs = (ctypes.c_void_p * n)(*[None if it[8] is None else it[8].data_ptr() for it in items])
        check(lib.mspl_conv1x1_wgrad_batch(arr(0), arr(1), arr(2), rs, ints(3), ints(4), ints(5), ints(6), ints(7), n, _stream()))


WGRADS = WgradQueue()


def _wgrad_1x1_into_sink(gy, x, N, Cin, Cout, groups, H, W, sink, rowscale=None):
    """gw of a grouped 1x1 convolution, accumulated into `sink`: queued inside grad_sinks(), launched at once otherwise.  rowscale: gy is
    the gradient before a per-output-channel scale (applied at the store)."""
    if (WGRADS.enabled and _SINKS[0]) or rowscale is not None:
        WGRADS.add(gy, x, N, Cin, Cout, groups, H * W, sink, rowscale)
        if not (WGRADS.enabled and _SINKS[0]):
            WGRADS.flush()
    else:
        check(lib.mspl_conv_bwd_weight(_p(gy), _p(x), N, Cin, Cout, groups, H, W, 1, 1, 1, 1, _p(sink), _stream()))


def _sink(p):
    if not _SINKS[0] or p is None or not p.requires_grad or not p.is_leaf:
        return None
    g = p.grad
    return g if (g is not None and g.is_contiguous() and g.dtype == torch.float32) else None


# ---- transposed weights of the data-gradient convolutions, all in one launch ------------------------------------------------
# ConvFn.backward runs the data gradient of a stride-1 convolution on the forward kernels with transposed (3x3: also flipped)
# weights.  Repacking them where they are needed is one ATen permute copy (+ one flip) per convolution on the backward chain.
# A WeightTransposer knows the convolutions of a model's step (collected during the first, eager step), keeps their transposed
# copies in one flat buffer and refreshes all of them with ONE kernel at the start of a step; inside `with tr.active():`
# ConvFn.backward takes its copy from there.  Outside that scope (or for a weight the transposer has not seen) the copy is made
# on the spot as before, so a stale table can never be read by accident.
_WT_COLLECT = [None]             # list collecting (weight, groups, k) during the first step, or None
_WT_ACTIVE = [None]              # {data_ptr: (wt view, groups, k, shape)} while a transposer is active


class collect_conv_weights(object):
    def __enter__(self):
        self.prev = _WT_COLLECT[0]
        _WT_COLLECT[0] = []
        return _WT_COLLECT[0]

    def __exit__(self, *exc):
        _WT_COLLECT[0] = self.prev


def _align4(n):
    return (n + 3) & ~3


class WeightTransposer(object):
    def __init__(self, collected):
        import numpy as np
        seen, items = set(), []
        for w, groups, k in collected:
            if w.data_ptr() in seen or not w.is_contiguous() or w.dtype != torch.float32:
                continue
            seen.add(w.data_ptr())
            items.append((w, groups, k))
        self.items = items
        dev = items[0][0].device if items else torch.device('cuda')
        # every copy starts at a multiple of 4 floats: the 1x1 kernels read weights as 16-byte vectors (an unaligned weight
        # tensor drops the data-gradient convolution to the slower generic kernel)
        self.flat = torch.empty(max(sum(_align4(w.numel()) for w, _, _ in items), 1), device=dev, dtype=torch.float32)
        raw = np.zeros((max(len(items), 1), 10), dtype=np.int32)        # struct WtSeg (train.hip): 2 pointers + 6 int32 = 40 bytes
        blocks, self.lookup, off = [], {}, 0
        for i, (w, groups, k) in enumerate(items):
            cout, cin_g = w.shape[0], w.shape[1]
            cout_g = cout // groups
            dst = self.flat[off:off + w.numel()]
            raw[i, 0:4] = np.array([w.data_ptr(), dst.data_ptr()], dtype=np.int64).view(np.int32)
            raw[i, 4:10] = (groups, cin_g, cout_g, k, w.numel(), 0)
            self.lookup[w.data_ptr()] = (dst.view(groups * cin_g, cout_g, k, k), groups, k, tuple(w.shape))
            blocks.extend((i, b) for b in range(0, w.numel(), 256))
            off += _align4(w.numel())
        self.seg = torch.from_numpy(raw).to(dev)
        self.blk = torch.tensor(blocks if blocks else [(0, 0)], dtype=torch.int32).to(dev)
        self.nblocks = len(blocks)

    def run(self):
        """Refresh every transposed copy from the current weights (one launch on the current stream)."""
        check(lib.mspl_transpose_weights(_p(self.seg), _p(self.blk), self.nblocks, _stream()))

    def active(self, refresh=True):
        """Scope in which ConvFn.backward reads the table.  refresh=False: the caller has already run() this step (micro-batch
        lanes share one refresh)."""
        return _ActiveTransposer(self, refresh)


class _ActiveTransposer(object):
    def __init__(self, tr, refresh=True):
        self.tr, self.refresh = tr, refresh

    def __enter__(self):
        self.prev = _WT_ACTIVE[0]
        if self.refresh:
            self.tr.run()
        _WT_ACTIVE[0] = self.tr.lookup
        return self.tr

    def __exit__(self, *exc):
        _WT_ACTIVE[0] = self.prev


def _transposed_weights(w, groups, k):
    """(Cout, Cin/g, k, k) -> (Cin, Cout/g, k, k), spatially flipped for k = 3: the weights of the data-gradient convolution."""
    if _WT_COLLECT[0] is not None:
        _WT_COLLECT[0].append((w, groups, k))
    lk = _WT_ACTIVE[0]
    if lk is not None:
        hit = lk.get(w.data_ptr())
        if hit is not None and hit[1] == groups and hit[2] == k and hit[3] == tuple(w.shape):
            return hit[0]
    Cout, cg_in = w.shape[0], w.shape[1]
    cg_out = Cout // groups
    wt = w.view(groups, cg_out, cg_in, k, k).transpose(1, 2)
    if k == 3:
        wt = wt.flip(3, 4)
    return wt.reshape(groups * cg_in, cg_out, k, k).contiguous()


class ConvFn(torch.autograd.Function):
    """Bias-free grouped conv, K in {1,3} (dilation 1), stride 1|2."""

    @staticmethod
    def forward(ctx, x, w, stride, groups):
        x, w = _c(x), _c(w)
        k = w.shape[-1]
        y = ops.conv1x1(x, w, groups) if k == 1 else ops.conv3x3(x, w, groups, stride)
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride if k == 3 else 1, groups, k)
        ctx.wsink = _sink(w)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx, gw = _conv_backward(x, w, ctx.cfg, ctx.wsink, gy, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return gx, gw, None, None


class ConvSkipFn(torch.autograd.Function):
    """(conv(x), alias of x): the projection of a residual block together with the block's skip connection.  Both consumers of x hang
    on ONE node, so its backward sees both gradients and the data-gradient convolution adds the skip gradient in its epilogue --
    autograd otherwise sums the two full-size gradients with an ATen add per block (15 launches, 0.24 ms of a supervised iteration).
    The alias is a view of x: consumers must not write to it (none of this package's ops does)."""

    @staticmethod
    def forward(ctx, x, w, groups):
        x, w = _c(x), _c(w)
        ctx.save_for_backward(x, w)
        ctx.groups = groups
        ctx.wsink = _sink(w)
        return ops.conv1x1(x, w, groups), x.view_as(x)

    @staticmethod
    def backward(ctx, gy, gskip):
        x, w = ctx.saved_tensors
        groups = ctx.groups
        gx = gw = None
        if gy is None:
            return gskip, None, None
        gy = _c(gy)
        if ctx.needs_input_grad[0]:
            wt = _transposed_weights(w, groups, 1)
            gx = ops.conv1x1(gy, wt, groups, None if gskip is None else Epi(residual=_c(gskip)))
        if ctx.needs_input_grad[1]:
            N, Cin, H, W = x.shape
            sink = ctx.wsink
            gw = torch.empty_like(w) if sink is None else None
            if sink is not None:
                _wgrad_1x1_into_sink(gy, x, N, Cin, w.shape[0], groups, H, W, sink)
            else:
                check(lib.mspl_conv_bwd_weight(_p(gy), _p(x), N, Cin, w.shape[0], groups, H, W, 1, 1, 1, 0, _p(gw), _stream()))
        return gx, gw, None


def conv_skip(x, w, groups):
    """(1x1 convolution of x, alias of x for the block's residual connection): see ConvSkipFn."""
    return ConvSkipFn.apply(x, w, groups)


class EespDwBNFn(torch.autograd.Function):
    """K2 (the four dilated depthwise 3x3 + HFF + cat) and br_after_cat in train() of an EESP block as ONE node (the supervised
    loop; the strided blocks and shapes the fused backward does not cover take mspl_hff_bn_stat_suffix_bwd + mspl_eesp_dw_bwd): forward = the K2 kernel + the BatchNorm node's forward (one launch on small planes); backward = the BatchNorm node's sums /
    coefficients launch (mspl_bn_train_prelu_bwd without outputs) + mspl_eesp_bwd_fused_bnstat, which applies p z + q + the direct
    gradient, the HFF suffix sum and both gradients of the four branches from LDS: two launches where the node-per-op form ran the
    BatchNorm backward (1-2), the suffix sum, the data and the weight gradient, and wrote / re-read the 4n-channel gradient twice."""

    @staticmethod
    def forward(ctx, x, w0, w1, w2, w3, dil, stride, gamma, beta, alpha, running_mean, running_var, eps, momentum, ws, nbt):
        x = _c(x)
        w4 = torch.stack([w.reshape(-1, 3, 3) for w in (w0, w1, w2, w3)]).contiguous()
        z = ops.eesp_dw_hff(x, w4, dil, stride)
        N, C = z.shape[:2]
        hw = z[0, 0].numel()
        st = torch.empty(4, C, dtype=torch.float32, device=z.device)       # mean, invstd, scale, shift
        gamma_c, beta_c, alpha_c = _c(gamma), _c(beta), _c(alpha)
        if _SMALL_BN and lib.mspl_bn_train_small_fits(N, C, hw):
            y = torch.empty_like(z)
            check(lib.mspl_bn_train_small_fwd(_p(z), None, _p(gamma_c), _p(beta_c), _p(alpha_c), N, C, hw, eps, momentum, _p(running_mean),
                                              _p(running_var), _p(nbt), _p(st[0]), _p(st[1]), _p(st[2]), _p(st[3]), _p(y), _stream()))
        else:
            check(lib.mspl_bn_batch_stats_fused_fwd(_p(z), N, C, hw, eps, momentum, _p(running_mean), _p(running_var), _p(gamma_c),
                                                    _p(beta_c), _p(ws), _p(st[0]), _p(st[1]), _p(st[2]), _p(st[3]), _p(nbt), _stream()))
            y = ops.pointwise(z, Epi(st[2], st[3], alpha_c))
        ctx.save_for_backward(x, w4, z, st, gamma_c, alpha_c, ws)
        ctx.cfg = (tuple(dil), w0.shape, stride)
        ctx.sinks = ([_sink(w) for w in (w0, w1, w2, w3)], (_sink(gamma), _sink(beta), _sink(alpha)))
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w4, z, st, gamma, alpha, ws = ctx.saved_tensors
        dil, wshape, stride = ctx.cfg
        wsinks, (s_g, s_b, s_a) = ctx.sinks
        gy = _c(gy)
        N, n, H, W = x.shape
        C = 4 * n
        dev = x.device
        # sums -> (d gamma, d beta, d alpha) and the statistics-path coefficients (p, q); no tensor is written
        direct = s_g is not None and s_b is not None
        out = torch.empty(4, C, dtype=torch.float32, device=dev)
        gal = s_a if s_a is not None else torch.zeros(C, device=dev)
        check(lib.mspl_bn_train_prelu_bwd(_p(z), None, _p(gy), _p(st[2]), _p(st[3]), _p(alpha), _p(gamma), _p(st[0]), _p(st[1]), N, C,
                                          z.shape[2] * z.shape[3], None, None, _p(ws), 1 if direct else 0, _p(s_g if direct else out[0]),
                                          _p(s_b if direct else out[1]), _p(gal), _p(out[2]), _p(out[3]), _stream()))
        tmp = torch.zeros((4,) + tuple(wshape), device=dev, dtype=torch.float32) if any(t is None for t in wsinks) else None
        dst = [wsinks[k] if wsinks[k] is not None else tmp[k] for k in range(4)]
        ptrs = (ctypes.c_void_p * 4)(*[d.data_ptr() for d in dst])
        dil_c = (ctypes.c_int32 * 4)(*dil)
        gx = torch.empty_like(x)
        if stride == 1 and _FUSED_EESP_BWD and lib.mspl_eesp_bwd_fused_fits(N, n, H, W, dil_c):
            check(lib.mspl_eesp_bwd_fused_bnstat(_p(z), _p(gy), _p(x), _p(w4), dil_c, _p(st[2]), _p(st[3]), _p(alpha), _p(out[2]), _p(out[3]),
                                                 N, n, H, W, _p(gx), ptrs, _stream()))
        else:
            # strided blocks / shapes the one-launch form does not cover: statistics path + direct gradient + suffix sum in one pass,
            # then the two gradient kernels of the four branches
            Ho, Wo = z.shape[2:]
            gs = torch.empty((4, N, n, Ho, Wo), device=dev, dtype=torch.float32)
            check(lib.mspl_hff_bn_stat_suffix_bwd(_p(z), _p(gy), _p(st[2]), _p(st[3]), _p(alpha), _p(out[2]), _p(out[3]), N, n, Ho * Wo,
                                                  _p(gs), _stream()))
            check(lib.mspl_eesp_dw_bwd(_p(gs), _p(x), _p(w4), dil_c, stride, N, n, H, W, _p(gx), ptrs, _stream()))
        gws = [None if wsinks[k] is not None else tmp[k] for k in range(4)]
        return (gx, *gws, None, None, None if direct else out[0], None if direct else out[1], None if s_a is not None else gal,
                None, None, None, None, None, None)


def eesp_dw_bn_fits(shape, dil):
    """True when EespDwBNFn covers a stride-1 K2 on an input of `shape` (N, n, H, W): the fused backward's row band fits LDS."""
    N, n, H, W = (int(v) for v in shape)
    return bool(_FUSED_EESP_BWD and lib.mspl_eesp_bwd_fused_fits(N, n, H, W, (ctypes.c_int32 * 4)(*dil)))


def eesp_dw_bn(x, ws, dil, bn, alpha, stride=1):
    """PReLU(BatchNorm_train(K2(x))) for br_after_cat in train(): see EespDwBNFn."""
    if bn.momentum is None or not bn.track_running_stats or not bn.affine:
        raise RuntimeError('mspl_amd: BatchNorm2d variants without momentum / running statistics / affine parameters are '
                           'not on the path (the reference uses the defaults everywhere)')
    return EespDwBNFn.apply(x, ws[0], ws[1], ws[2], ws[3], tuple(dil), int(stride), bn.weight, bn.bias, alpha, bn.running_mean, bn.running_var,
                            float(bn.eps), float(bn.momentum), _bn_workspace(bn, x.device), _bn_nbt(bn))


def _conv_backward(x, w, cfg, sink, gy, need_gx, need_gw):
    """(gx, gw) of a bias-free grouped convolution; gw is None when it was accumulated into `sink`."""
    stride, groups, k = cfg
    gy = _c(gy)
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    gx = gw = None
    if need_gx:
        if stride == 1:
            # the data gradient of a stride-1 convolution is a convolution with the transposed (and, for 3x3,
            # spatially flipped) weights: run it on the forward kernels (MFMA 1x1 / LDS-tiled 3x3).
            wt = _transposed_weights(w, groups, k)
            gx = ops.conv1x1(gy, wt, groups) if k == 1 else ops.conv3x3(gy, wt, groups, 1)
        else:
            gx = torch.empty_like(x)
            check(lib.mspl_conv_bwd_data(_p(gy), _p(w), N, Cin, Cout, groups, H, W, k, stride, 1, 0, _p(gx), _stream()))
    if need_gw:
        gw = torch.empty_like(w) if sink is None else None
        if sink is not None and k == 1 and stride == 1:
            _wgrad_1x1_into_sink(gy, x, N, Cin, Cout, groups, H, W, sink)
        else:
            check(lib.mspl_conv_bwd_weight(_p(gy), _p(x), N, Cin, Cout, groups, H, W, k, stride, 1, 0 if sink is None else 1,
                                           _p(gw if sink is None else sink), _stream()))
    return gx, gw


class EespDwFn(torch.autograd.Function):
    """K2 without epilogue: 4 dilated depthwise 3x3 + hierarchical add + concat."""

    @staticmethod
    def forward(ctx, x, w0, w1, w2, w3, dil, stride):
        x = _c(x)
        w4 = torch.stack([w.reshape(-1, 3, 3) for w in (w0, w1, w2, w3)]).contiguous()
        y = ops.eesp_dw_hff(x, w4, dil, stride)
        ctx.save_for_backward(x, w4)
        ctx.cfg = (tuple(dil), stride, w0.shape)
        ctx.wsinks = [_sink(w) for w in (w0, w1, w2, w3)]
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w4 = ctx.saved_tensors
        dil, stride, wshape = ctx.cfg
        gy = _c(gy)
        N, n, H, W = x.shape
        Ho, Wo = gy.shape[2:]
        gs = torch.empty((4, N, n, Ho, Wo), device=x.device, dtype=torch.float32)
        check(lib.mspl_hff_suffix_sum(_p(gy), N, n, Ho * Wo, _p(gs), _stream()))
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        # all four branches in one data-gradient launch and one weight-gradient launch (eesp_dw_bwd.hip)
        sinks = ctx.wsinks
        tmp = torch.zeros((4,) + tuple(wshape), device=x.device, dtype=torch.float32) if any(s_ is None for s_ in sinks) else None
        dst = [sinks[k] if sinks[k] is not None else tmp[k] for k in range(4)]
        ptrs = (ctypes.c_void_p * 4)(*[d.data_ptr() for d in dst])
        dil_c = (ctypes.c_int32 * 4)(*dil)
        check(lib.mspl_eesp_dw_bwd(_p(gs), _p(x), _p(w4), dil_c, stride, N, n, H, W, _p(gx), ptrs, _stream()))
        gws = [None if sinks[k] is not None else tmp[k] for k in range(4)]
        return (gx, *gws, None, None)


class AffinePReLUFn(torch.autograd.Function):
    """y = PReLU((c + pre_add) * scale + shift + residual); any of scale/shift/alpha/pre_add/residual may be None.

    Frozen-BatchNorm form (gamma given): scale/shift are the folded, non-differentiable (gamma*inv, beta - mean*gamma*inv);
    the backward kernel turns its per-channel sums into d gamma / d beta on the fly (mspl_bn_prelu_bwd), so no separate
    fold/unfold kernels run per BatchNorm.  Per-channel parameter gradients go to the parameters' .grad buffers directly
    inside `grad_sinks()`."""

    @staticmethod
    def forward(ctx, c, scale, shift, alpha, pre_add, residual, gamma, beta, mean, inv):
        c = _c(c)
        pre_add = None if pre_add is None else _c(pre_add)
        residual = None if residual is None else _c(residual)
        y = ops.pointwise(c, Epi(scale, shift, alpha, pre_add=pre_add, residual=residual))
        ctx.save_for_backward(c, scale, shift, alpha, pre_add, residual, mean, inv)
        ctx.bn = gamma is not None
        ctx.sinks = (_sink(gamma if ctx.bn else scale), _sink(beta if ctx.bn else shift), _sink(alpha))
        return y

    @staticmethod
    def backward(ctx, gy):
        c, scale, shift, alpha, pre_add, residual, mean, inv = ctx.saved_tensors
        gc, gz, r_sc, r_sh, r_al = _affine_backward(c, scale, shift, alpha, pre_add, residual, mean, inv, ctx.bn, ctx.sinks, gy)
        gpre = gc if pre_add is not None else None
        if ctx.bn:
            return gc, None, None, r_al, gpre, gz, r_sc, r_sh, None, None
        return gc, r_sc, r_sh, r_al, gpre, gz, None, None, None, None


def _affine_backward(c, scale, shift, alpha, pre_add, residual, mean, inv, bn, sinks, gy):
    """Backward of y = PReLU((c + pre_add) * scale + shift + residual): (gc, gz = d residual, and the per-channel gradients that
    were NOT accumulated into a sink: d scale | d gamma, d shift | d beta, d alpha)."""
    gy = _c(gy)
    N, C = c.shape[:2]
    hw = c[0, 0].numel()
    dev = c.device
    gz = torch.empty_like(c) if residual is not None else None
    gc = torch.empty_like(c)
    s_sc, s_sh, s_al = sinks
    # (a zeroed accumulator only for a gradient that is wanted and has no sink)
    need = (scale is not None and s_sc is None) or (shift is not None and s_sh is None) or (alpha is not None and s_al is None)
    gacc = torch.zeros(3, C, device=dev) if need else None
    gsc = (s_sc if s_sc is not None else gacc[0]) if scale is not None else None
    gsh = (s_sh if s_sh is not None else gacc[1]) if shift is not None else None
    gal = (s_al if s_al is not None else gacc[2]) if alpha is not None else None
    if bn:
        check(lib.mspl_bn_prelu_bwd(_p(c), _p(pre_add), _p(residual), _p(gy), _p(scale), _p(shift), _p(alpha), _p(mean), _p(inv),
                                    N, C, hw, _p(gz), _p(gc), _p(gsc), _p(gsh), _p(gal), _stream()))
    else:
        check(lib.mspl_affine_prelu_bwd(_p(c), _p(pre_add), _p(residual), _p(gy), _p(scale), _p(shift), _p(alpha), N, C, hw,
                                        _p(gz), _p(gc), _p(gsc), _p(gsh), _p(gal), _stream()))
    return gc, gz, (gsc if s_sc is None else None), (gsh if s_sh is None else None), (gal if s_al is None else None)


class ConvAffinePReLUFn(torch.autograd.Function):
    """y = PReLU((conv(x, w) + pre_add) * scale + shift + residual) as ONE forward launch: the convolution kernel applies the
    epilogue and also stores its bare result (mspl_epilogue_t.raw_out), which the backward of the BatchNorm / PReLU needs (d gamma,
    the PReLU sign).  Backward = AffinePReLUFn's followed by ConvFn's.  Replaces ConvFn + AffinePReLUFn (two launches, the second one
    reading the convolution result back) wherever a frozen BatchNorm / bias / PReLU follows a convolution directly."""

    @staticmethod
    def forward(ctx, x, w, stride, groups, scale, shift, alpha, pre_add, residual, gamma, beta, mean, inv):
        x, w = _c(x), _c(w)
        pre_add = None if pre_add is None else _c(pre_add)
        residual = None if residual is None else _c(residual)
        k = w.shape[-1]
        N, _, H, W = x.shape
        s = stride if k == 3 else 1
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        c = torch.empty((N, w.shape[0], Ho, Wo), device=x.device, dtype=torch.float32)
        ep = Epi(scale, shift, alpha, pre_add=pre_add, residual=residual, raw_out=c)
        y = ops.conv1x1(x, w, groups, ep) if k == 1 else ops.conv3x3(x, w, groups, stride, 0, ep)
        ctx.save_for_backward(x, w, c, scale, shift, alpha, pre_add, residual, mean, inv)
        ctx.cfg = (s, groups, k)
        ctx.wsink = _sink(w)
        ctx.bn = gamma is not None
        ctx.sinks = (_sink(gamma if ctx.bn else scale), _sink(beta if ctx.bn else shift), _sink(alpha))
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, c, scale, shift, alpha, pre_add, residual, mean, inv = ctx.saved_tensors
        gc, gz, r_sc, r_sh, r_al = _affine_backward(c, scale, shift, alpha, pre_add, residual, mean, inv, ctx.bn, ctx.sinks, gy)
        gx, gw = _conv_backward(x, w, ctx.cfg, ctx.wsink, gc, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        gpre = gc if pre_add is not None else None
        if ctx.bn:
            return gx, gw, None, None, None, None, r_al, gpre, gz, r_sc, r_sh, None, None
        return gx, gw, None, None, r_sc, r_sh, r_al, gpre, gz, None, None, None, None


class FanOutFn(torch.autograd.Function):
    """n aliases of x for n consumers; backward sums their gradients with ONE launch (mspl_sum_n) instead of autograd's n - 1
    pairwise ATen adds.  The outputs are views of x: consumers must not write to them (none of this package's ops does)."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.n = n
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        # a gradient that is constant over every plane (GapGateFn hands out an expanded (N,C,1,1) tensor: the global average pool's
        # gradient) joins the sum as N * C values: nobody writes, and the sum does not read, a full-size broadcast of it
        pconst = [g for g in gs if g is not None and g.dim() == 4 and g.stride(2) == 0 and g.stride(3) == 0 and g.shape[2] * g.shape[3] > 1]
        full = [g for g in gs if g is not None and not any(g is p for p in pconst)]
        if len(pconst) == 1 and full and full[0].dim() == 4 and (full[0].shape[2] * full[0].shape[3]) % 4 == 0:
            live = [_c(g) for g in full]
            if not any(t.data_ptr() % 16 for t in live):
                N_, C_, H_, W_ = live[0].shape
                pc = pconst[0][:, :, 0, 0].contiguous()
                out = torch.empty_like(live[0])
                ptrs = (ctypes.c_void_p * len(live))(*[t.data_ptr() for t in live])
                check(lib.mspl_sum_n_planes(ptrs, len(live), _p(pc), 1.0, N_ * C_, H_ * W_, _p(out), _stream()))
                return out, None
        live = [_c(g) for g in gs if g is not None]
        if not live:
            return None, None
        if len(live) == 1:
            return live[0], None
        if live[0].numel() % 4 or any(t.data_ptr() % 16 for t in live):
            out = live[0]
            for t in live[1:]:
                out = out + t
            return out, None
        out = torch.empty_like(live[0])
        ptrs = (ctypes.c_void_p * len(live))(*[t.data_ptr() for t in live])
        check(lib.mspl_sum_n(ptrs, len(live), live[0].numel(), _p(out), _stream()))
        return out, None


_FUSED_EESP_BWD = os.environ.get('MSPL_FUSED_EESP_BWD', '1') != '0'


def _stack4(ws):
    """(4, n, 3, 3) tensor of the four depthwise branch weights: a VIEW when they sit back to back in one storage (the flat
    parameter buffer of FlatAdam / FlatSGD lays consecutive parameters out contiguously), else a torch.stack copy."""
    w0 = ws[0]
    n, step = w0.shape[0], w0.numel() * w0.element_size()
    try:
        base = w0.untyped_storage().data_ptr()
        if all(w.is_contiguous() and w.untyped_storage().data_ptr() == base and w.data_ptr() == w0.data_ptr() + k * step
               for k, w in enumerate(ws)):
            return torch.as_strided(w0.detach(), (4, n, 3, 3), (n * 9, 9, 3, 1))
    except RuntimeError:
        pass
    return torch.stack([w.reshape(-1, 3, 3) for w in ws]).contiguous()


class EESPFn(torch.autograd.Function):
    """A whole EESP block (nn_layers/eesp.py:60-93) with frozen BatchNorms as ONE autograd node: proj_1x1 (grouped 1x1 + BN + PReLU)
    -> four dilated depthwise 3x3 + hierarchical add + concat (K2) -> br_after_cat (BN + PReLU) -> conv_1x1_exp (grouped 1x1 + BN)
    [+ input residual, module_act].  Same forward kernels as the node-per-op form; the backward owns its buffers, so
      * the residual link's gradient is added by the epilogue of the projection's data-gradient convolution (autograd summed the two
        gradients of the block input with an ATen kernel per block),
      * br_after_cat's BatchNorm/PReLU backward and the HFF suffix sum are one pass (mspl_hff_bn_prelu_suffix_bwd),
      * the four branch weights are passed as one view of the flat parameter buffer (no torch.stack per block and step)."""

    @staticmethod
    def forward(ctx, x, cfg, fp, fb, fe, wp, gp, bp, ap, w0, w1, w2, w3, g2, b2, a2, we, ge, be, am):
        x, wp, we = _c(x), _c(wp), _c(we)
        stride, dil, groups, residual = cfg['stride'], cfg['dil'], cfg['groups'], cfg['residual']
        N, Cin, H, W = x.shape
        n = wp.shape[0]
        c1 = torch.empty((N, n, H, W), device=x.device, dtype=torch.float32)
        o1 = ops.conv1x1(x, wp, groups, Epi(fp['scale'], fp['shift'], ap, raw_out=c1))
        w4 = _stack4((w0, w1, w2, w3))
        # K2 writes the hierarchical sums (what br_after_cat's backward reads) and their BN + PReLU in one launch
        z2 = torch.empty((N, 4 * n, (H - 1) // stride + 1, (W - 1) // stride + 1), device=x.device, dtype=torch.float32)
        y2 = ops.eesp_dw_hff(o1, w4, dil, stride, Epi(fb['scale'], fb['shift'], a2, raw_out=z2))
        Cout = we.shape[0]
        c3 = torch.empty((N, Cout) + tuple(z2.shape[2:]), device=x.device, dtype=torch.float32)
        y = ops.conv1x1(y2, we, groups, Epi(fe['scale'], fe['shift'], am, residual=x if residual else None, raw_out=c3))
        ctx.save_for_backward(x, c1, o1, z2, y2, c3, w4, wp, we, ap, a2, am, fp['scale'], fp['shift'], fp['mean'], fp['inv'],
                              fb['scale'], fb['shift'], fb['mean'], fb['inv'], fe['scale'], fe['shift'], fe['mean'], fe['inv'])
        ctx.cfg = cfg
        ctx.wshape = tuple(w0.shape)
        ctx.sinks = {'wp': _sink(wp), 'p': (_sink(gp), _sink(bp), _sink(ap)), 'w4': [_sink(t) for t in (w0, w1, w2, w3)],
                     'b': (_sink(g2), _sink(b2), _sink(a2)), 'we': _sink(we), 'e': (_sink(ge), _sink(be), _sink(am))}
        return y

    @staticmethod
    def backward(ctx, gy):
        (x, c1, o1, z2, y2, c3, w4, wp, we, ap, a2, am, sp, hp, mp, ip, sb, hb, mb, ib, se, he, me, ie) = ctx.saved_tensors
        cfg, sk = ctx.cfg, ctx.sinks
        stride, dil, groups, residual = cfg['stride'], cfg['dil'], cfg['groups'], cfg['residual']
        gy = _c(gy)
        N, Cin, H, W = x.shape
        n = wp.shape[0]
        Ho, Wo = z2.shape[2:]
        dev = x.device
        # conv_1x1_exp's BatchNorm (+ residual, module_act) and the convolution itself
        gc3, gres, r_ge, r_be, r_am = _affine_backward(c3, se, he, am, None, x if residual else None, me, ie, True, sk['e'], gy)
        gy2, gwe = _conv_backward(y2, we, (1, groups, 1), sk['we'], gc3, True, True)
        # br_after_cat backward + HFF suffix sum
        C4 = 4 * n
        s_g2, s_b2, s_a2 = sk['b']
        acc = torch.zeros(3, C4, device=dev) if (s_g2 is None or s_b2 is None or s_a2 is None) else None
        d_g2 = s_g2 if s_g2 is not None else acc[0]
        d_b2 = s_b2 if s_b2 is not None else acc[1]
        d_a2 = s_a2 if s_a2 is not None else acc[2]
        go1 = torch.empty_like(o1)
        wsinks = sk['w4']
        tmp = torch.zeros((4,) + ctx.wshape, device=dev, dtype=torch.float32) if any(t is None for t in wsinks) else None
        dst = [wsinks[k] if wsinks[k] is not None else tmp[k] for k in range(4)]
        ptrs = (ctypes.c_void_p * 4)(*[d.data_ptr() for d in dst])
        dil_c = (ctypes.c_int32 * 4)(*dil)
        r_gp = r_bp = r_ap = None
        if stride == 1 and _FUSED_EESP_BWD and lib.mspl_eesp_bwd_fused_fits(N, n, H, W, dil_c):
            # one launch: BatchNorm/PReLU backward + suffix sum + both gradients of the four branches (no suffix-summed tensor in
            # memory) + proj_1x1's BatchNorm/PReLU backward on the way out: what is written is dL/d(projection's convolution result)
            s_gp, s_bp, s_ap = sk['p']
            pacc = torch.zeros(3, n, device=dev) if (s_gp is None or s_bp is None or s_ap is None) else None
            d_gp = s_gp if s_gp is not None else pacc[0]
            d_bp = s_bp if s_bp is not None else pacc[1]
            d_ap = s_ap if s_ap is not None else pacc[2]
            gc1 = go1
            check(lib.mspl_eesp_bwd_fused(_p(z2), _p(gy2), _p(o1), _p(w4), dil_c, _p(sb), _p(hb), _p(a2), _p(mb), _p(ib), N, n, H, W,
                                          _p(gc1), ptrs, _p(d_g2), _p(d_b2), _p(d_a2), _p(c1), _p(sp), _p(hp), _p(ap), _p(mp), _p(ip),
                                          _p(d_gp), _p(d_bp), _p(d_ap), _stream()))
            r_gp, r_bp, r_ap = (None if s_gp is not None else d_gp), (None if s_bp is not None else d_bp), (None if s_ap is not None else d_ap)
        else:
            gs = torch.empty((4, N, n, Ho, Wo), device=dev, dtype=torch.float32)
            check(lib.mspl_hff_bn_prelu_suffix_bwd(_p(z2), _p(gy2), _p(sb), _p(hb), _p(a2), _p(mb), _p(ib), N, n, Ho * Wo, _p(gs),
                                                   _p(d_g2), _p(d_b2), _p(d_a2), _stream()))
            # the four depthwise branches
            check(lib.mspl_eesp_dw_bwd(_p(gs), _p(o1), _p(w4), dil_c, stride, N, n, H, W, _p(go1), ptrs, _stream()))
            # proj_1x1's BatchNorm + PReLU
            gc1, _, r_gp, r_bp, r_ap = _affine_backward(c1, sp, hp, ap, None, None, mp, ip, True, sk['p'], go1)
        # proj_1x1's convolution; the residual link's gradient rides on the data gradient's epilogue
        gx = gwp = None
        if ctx.needs_input_grad[0]:
            wt = _transposed_weights(wp, groups, 1)
            gx = ops.conv1x1(gc1, wt, groups, Epi(residual=gres) if gres is not None else None)
        s_wp = sk['wp']
        gwp = torch.empty_like(wp) if s_wp is None else None
        if s_wp is not None:
            _wgrad_1x1_into_sink(gc1, x, N, Cin, n, groups, H, W, s_wp)
        else:
            check(lib.mspl_conv_bwd_weight(_p(gc1), _p(x), N, Cin, n, groups, H, W, 1, 1, 1, 0, _p(gwp), _stream()))
        ret = lambda sink, t: None if sink is not None else t          # noqa: E731
        gws = [ret(wsinks[k], tmp[k] if tmp is not None else None) for k in range(4)]
        return (gx, None, None, None, None, gwp, r_gp, r_bp, r_ap, *gws, ret(s_g2, d_g2), ret(s_b2, d_b2), ret(s_a2, d_a2),
                gwe, r_ge, r_be, r_am if am is not None else None)


_PYR_DOWN_FUSED = os.environ.get('MSPL_PYR_DOWN_FUSED', '1') != '0'   # low-resolution pyramid branches: one launch each way (A/B aid)


def _pyr_down_forward(x, sizes, stage_ws, down):
    """The scale < 1 branches of a pyramid in the training forward: (pooled maps, dw3x3(pooled) maps) per branch index.  One launch
    for all of them (the inference prologue kernel, which also leaves the pooled maps the weight gradient needs) instead of an
    adaptive pool + a depthwise 3x3 per branch."""
    N, P, h, w = x.shape
    pooled, es = {}, {}
    if not down:
        return pooled, es
    dsz = [sizes[i] for i in down]
    if _PYR_DOWN_FUSED and len(down) <= 4 and ops.pyr_down_prep_fits(x.shape, dsz):
        outs, pools = ops.pyr_down_prep(x, dsz, [stage_ws[i] for i in down], keep_pooled=True)
        for k, i in enumerate(down):
            pooled[i], es[i] = pools[k], outs[k]
        return pooled, es
    for i in down:
        pooled[i] = ops.adaptive_avgpool(x, sizes[i])
        es[i] = ops.conv3x3(pooled[i], stage_ws[i], P)
    return pooled, es


def _pyr_down_backward(gt, x_shape, sizes, stage_ws, pooled, down, g_stage):
    """Backward of those branches: gt[i] (N,P,h,w) = dL/d(branch value) -> the full-resolution gradients [g_x_i], the depthwise
    weight gradients added into g_stage[i].  Transposed bilinear interpolation per branch (existing kernels), then ONE launch for
    the depthwise 3x3's two gradients and the adaptive pool's transpose of every branch (csrc/pyr_down_bwd.hip; three launches per
    branch before)."""
    N, P, h, w = x_shape
    dev = gt.device
    g_es = []
    for i in down:
        hs_, ws_ = sizes[i]
        g_e = torch.empty((N, P, hs_, ws_), device=dev, dtype=torch.float32)
        check(lib.mspl_bilinear_bwd(_p(gt[i]), N, P, hs_, ws_, h, w, _p(g_e), _stream()))
        g_es.append(g_e)
    adds = []
    if _PYR_DOWN_FUSED and 1 <= len(down) <= 2 and all(((sizes[i][0] + 2) * (sizes[i][1] + 2) + sizes[i][0] * sizes[i][1]) * 4 + h * 16 <= 128 * 1024 for i in down):
        nbd = len(down)
        adds = [torch.empty((N, P, h, w), device=dev, dtype=torch.float32) for _ in down]
        arr = lambda ts: (ctypes.c_void_p * nbd)(*[t.data_ptr() for t in ts])         # noqa: E731
        sws = [_c(stage_ws[i]) for i in down]
        hsa = (ctypes.c_int32 * nbd)(*[sizes[i][0] for i in down])
        wsa = (ctypes.c_int32 * nbd)(*[sizes[i][1] for i in down])
        check(lib.mspl_pyr_down_mid_bwd(arr(g_es), arr([pooled[i] for i in down]), arr(sws), N, P, h, w, nbd, hsa, wsa,
                                        arr([g_stage[i] for i in down]), arr(adds), _stream()))
        return adds
    for k, i in enumerate(down):
        hs_, ws_ = sizes[i]
        g_pool = ops.conv3x3(g_es[k], _transposed_weights(stage_ws[i], P, 3), P, 1)
        check(lib.mspl_conv_bwd_weight(_p(g_es[k]), _p(pooled[i]), N, P, P, P, hs_, ws_, 3, 1, 1, 1, _p(g_stage[i]), _stream()))
        g_xi = torch.empty((N, P, h, w), device=dev, dtype=torch.float32)
        check(lib.mspl_adaptive_avgpool_bwd(_p(g_pool), N, P, h, w, hs_, ws_, _p(g_xi), _stream()))
        adds.append(g_xi)
    return adds


class PyrBodyFn(torch.autograd.Function):
    """The EfficientPyrPool body between projection_layer and the last 1x1 (nn_layers/efficient_pyramid_pool.py:39-58: the five
    branches, merge_layer.0 BatchNorm + PReLU over the concatenation, Shuffle, merge_layer.2 grouped 3x3 + BatchNorm + PReLU) as ONE
    autograd node with frozen BatchNorms.

    forward   one launch of the LDS-tiled fused kernel (+ the low-resolution branches' pooled / convolved maps), which also keeps the
              branch values before merge_layer.0 (zcat) and the bare merge convolution (mraw).
    backward  mspl_pyrpool_merge_bwd (everything from the output gradient down to dL/d(branch value), all parameter gradients of the
              two merge layers), the existing small-map kernels for the scale < 1 branches, and mspl_pyrpool_branch_bwd for the
              scale >= 1 branches: the concatenation is read once and written once, nothing at up-sampled resolution touches memory.
    The node-per-op form it replaces moved the 5x-wide concatenation through memory 17 times and the up-sampled intermediates ~44
    times per pyramid (4.9 of the 11.5 ms of kernel time of a uest step were pyramids: profiles/r03_train_trace_before.txt)."""

    @staticmethod
    def forward(ctx, x, sizes, bn0, bn2, br_gamma, br_beta, br_alpha, merge_w, m_gamma, m_beta, m_alpha, *stage_ws):
        x = _c(x)
        N, P, h, w = x.shape
        nb = len(sizes)
        br_scale, br_shift = bn0['scale'], bn0['shift']
        m_scale, m_shift = bn2['scale'], bn2['shift']
        down = [i for i, (hs_, ws_) in enumerate(sizes) if (hs_ < h or ws_ < w)]
        pooled, down_es = _pyr_down_forward(x, sizes, stage_ws, down)
        hs = (ctypes.c_int32 * nb)(*[int(s_[0]) for s_ in sizes])
        ws = (ctypes.c_int32 * nb)(*[int(s_[1]) for s_ in sizes])
        sw, de = (ctypes.c_void_p * nb)(), (ctypes.c_void_p * nb)()
        keep = []
        for i in range(nb):
            if i in down:
                de[i] = down_es[i].data_ptr()
            else:
                t = _c(stage_ws[i])
                keep.append(t)
                sw[i] = t.data_ptr()
        merge_w = _c(merge_w)
        y = torch.empty((N, P, h, w), device=x.device, dtype=torch.float32)
        mraw = torch.empty_like(y)
        zcat = torch.empty((N, nb * P, h, w), device=x.device, dtype=torch.float32)
        ep, keep2 = ops._build(Epi(m_scale, m_shift, m_alpha, raw_out=mraw), y, 0, N, P, h * w)
        check(lib.mspl_pyrpool_fused_train_fwd(_p(x), N, P, h, w, nb, hs, ws, sw, de, _p(br_scale), _p(br_shift), _p(_c(br_alpha)),
                                               _p(merge_w), ctypes.byref(ep), _p(y), _p(zcat), _stream()))
        ctx.save_for_backward(x, zcat, mraw, br_scale, br_shift, br_alpha, merge_w, m_scale, m_shift, m_alpha,
                              bn0['mean'], bn0['inv'], bn2['mean'], bn2['inv'], *stage_ws, *[pooled[i] for i in down])
        ctx.sizes, ctx.down = [tuple(int(v) for v in s_) for s_ in sizes], down
        ctx.sinks = ([_sink(t) for t in (br_gamma, br_beta, br_alpha, merge_w, m_gamma, m_beta, m_alpha)],
                     [_sink(t) for t in stage_ws])
        return y

    @staticmethod
    def backward(ctx, gy):
        sv = ctx.saved_tensors
        x, zcat, mraw, br_scale, br_shift, br_alpha, merge_w, m_scale, m_shift, m_alpha, mean0, inv0, mean2, inv2 = sv[:14]
        sizes, down = ctx.sizes, ctx.down
        nb = len(sizes)
        stage_ws = sv[14:14 + nb]
        pooled = dict(zip(down, sv[14 + nb:]))
        gy = _c(gy)
        N, P, h, w = x.shape
        dev = x.device
        psinks, wsinks = ctx.sinks

        def dst(sink, shape):
            return sink if sink is not None else torch.zeros(shape, device=dev, dtype=torch.float32)
        g_br_gamma, g_br_beta, g_br_alpha = dst(psinks[0], (nb * P,)), dst(psinks[1], (nb * P,)), dst(psinks[2], (nb * P,))
        g_merge_w = dst(psinks[3], tuple(merge_w.shape))
        g_m_gamma, g_m_beta = dst(psinks[4], (P,)), dst(psinks[5], (P,))
        g_m_alpha = dst(psinks[6], (P,)) if m_alpha is not None else None
        gt = torch.empty((nb, N, P, h, w), device=dev, dtype=torch.float32)
        check(lib.mspl_pyrpool_merge_bwd(_p(gy), _p(mraw), _p(zcat), N, P, h, w, nb, _p(br_scale), _p(br_shift), _p(br_alpha),
                                         _p(mean0), _p(inv0), _p(merge_w), _p(m_scale), _p(m_shift), _p(m_alpha), _p(mean2),
                                         _p(inv2), _p(gt), _p(g_br_gamma), _p(g_br_beta), _p(g_br_alpha), _p(g_merge_w),
                                         _p(g_m_gamma), _p(g_m_beta), _p(g_m_alpha), _stream()))
        g_stage = [dst(wsinks[i], tuple(stage_ws[i].shape)) for i in range(nb)]
        # scale < 1 branches (small maps): bilinear^T -> depthwise 3x3 backward -> adaptive pool^T, existing kernels
        adds = _pyr_down_backward(gt, (N, P, h, w), sizes, stage_ws, pooled, down, g_stage)
        # scale >= 1 branches (up to three, pyr_body_fits) in one launch, the low-resolution contributions (up to two) added in.
        # (always launched: the stage weights' gradients come from it)
        up = [i for i in range(nb) if i not in down]
        nbp = len(up)
        hsa = (ctypes.c_int32 * nbp)(*[sizes[i][0] for i in up])
        wsa = (ctypes.c_int32 * nbp)(*[sizes[i][1] for i in up])
        swp = (ctypes.c_void_p * nbp)(*[stage_ws[i].data_ptr() for i in up])
        gtp = (ctypes.c_void_p * nbp)(*[gt[i].data_ptr() for i in up])
        gwp = (ctypes.c_void_p * nbp)(*[g_stage[i].data_ptr() for i in up])
        gx = torch.empty_like(x)
        check(lib.mspl_pyrpool_branch_bwd(_p(x), N, P, h, w, nbp, hsa, wsa, swp, gtp, gwp, _p(adds[0] if adds else None),
                                          _p(adds[1] if len(adds) > 1 else None), _p(gx), _stream()))
        ret = lambda sink, t: None if sink is not None else t          # noqa: E731
        return (gx, None, None, None, ret(psinks[0], g_br_gamma), ret(psinks[1], g_br_beta), ret(psinks[2], g_br_alpha),
                ret(psinks[3], g_merge_w), ret(psinks[4], g_m_gamma), ret(psinks[5], g_m_beta),
                None if m_alpha is None else ret(psinks[6], g_m_alpha),
                *[ret(wsinks[i], g_stage[i]) for i in range(nb)])


_CONST_VECS = {}


def _const_vec(value, n, device):
    """A cached vector of `n` copies of `value` (identity BatchNorm folds of the batch-statistics pyramid node)."""
    key = (float(value), int(n), str(device))
    t = _CONST_VECS.get(key)
    if t is None:
        t = _CONST_VECS[key] = torch.full((int(n),), float(value), device=device, dtype=torch.float32)
    return t


class PyrBodyBNFn(torch.autograd.Function):
    """PyrBodyFn for BatchNorms in train() (the supervised loop): the same fused kernels, run around the two batch-statistics passes.

    forward   the fused training kernel for the branch values (zcat does not depend on the BatchNorms), the statistics of the
              concatenation -> merge_layer.0's fold, mspl_pyrpool_merge_fwd (fold + PReLU + Shuffle + grouped 3x3 from the kept
              branch values), the statistics of its bare result -> merge_layer.2's fold, one affine / PReLU launch.
    backward  merge_layer.2's BatchNorm + PReLU as in BNTrainPReLUFn (direct path + p * z + q); mspl_pyrpool_merge_bwd with that
              BatchNorm as the IDENTITY (its gradient is already applied) and RAW (d scale, d shift) sums for merge_layer.0, whose
              statistics path is then added to the branch-major gradient (mspl_bn_stats_path_add); the branch backward as in PyrBodyFn.
    bnp0 / bnp2: (running_mean, running_var, eps, momentum, workspace, num_batches_tracked) of the two BatchNorms."""

    @staticmethod
    def forward(ctx, x, sizes, bnp0, bnp2, br_gamma, br_beta, br_alpha, merge_w, m_gamma, m_beta, m_alpha, *stage_ws):
        x = _c(x)
        N, P, h, w = x.shape
        nb = len(sizes)
        dev = x.device
        down = [i for i, (hs_, ws_) in enumerate(sizes) if (hs_ < h or ws_ < w)]
        pooled, down_es = _pyr_down_forward(x, sizes, stage_ws, down)
        hs = (ctypes.c_int32 * nb)(*[int(s_[0]) for s_ in sizes])
        ws = (ctypes.c_int32 * nb)(*[int(s_[1]) for s_ in sizes])
        sw, de = (ctypes.c_void_p * nb)(), (ctypes.c_void_p * nb)()
        keep = []
        for i in range(nb):
            if i in down:
                de[i] = down_es[i].data_ptr()
            else:
                t = _c(stage_ws[i])
                keep.append(t)
                sw[i] = t.data_ptr()
        merge_w = _c(merge_w)
        br_alpha_c = _c(br_alpha)
        C0 = nb * P
        one0, zero0 = _const_vec(1.0, C0, dev), _const_vec(0.0, C0, dev)
        mraw = torch.empty((N, P, h, w), device=dev, dtype=torch.float32)
        zcat = torch.empty((N, C0, h, w), device=dev, dtype=torch.float32)
        y = torch.empty_like(mraw)
        # (the streaming form of the kernel wants the raw-output slot; y is scratch until step 4)
        ep, keep2 = ops._build(Epi(raw_out=mraw), y, 0, N, P, h * w)
        # (1) branch values
        check(lib.mspl_pyrpool_fused_train_fwd(_p(x), N, P, h, w, nb, hs, ws, sw, de, _p(one0), _p(zero0), _p(br_alpha_c),
                                               _p(merge_w), ctypes.byref(ep), _p(y), _p(zcat), _stream()))
        # (2) merge_layer.0: statistics of the concatenation
        rm0, rv0, eps0, mom0, ws0, nbt0 = bnp0
        st0 = torch.empty(4, C0, dtype=torch.float32, device=dev)             # mean, invstd, scale, shift
        g0, b0 = _c(br_gamma), _c(br_beta)
        check(lib.mspl_bn_batch_stats_fused_fwd(_p(zcat), N, C0, h * w, eps0, mom0, _p(rm0), _p(rv0), _p(g0), _p(b0), _p(ws0),
                                                _p(st0[0]), _p(st0[1]), _p(st0[2]), _p(st0[3]), _p(nbt0), _stream()))
        # (3) the merge convolution over PReLU(BN(zcat)), from the kept branch values
        check(lib.mspl_pyrpool_merge_fwd(_p(zcat), N, P, h, w, nb, _p(st0[2]), _p(st0[3]), _p(br_alpha_c), _p(merge_w), _p(mraw),
                                         _stream()))
        # (4) merge_layer.2's BatchNorm + PReLU
        rm2, rv2, eps2, mom2, ws2, nbt2 = bnp2
        st2 = torch.empty(4, P, dtype=torch.float32, device=dev)
        g2, b2 = _c(m_gamma), _c(m_beta)
        check(lib.mspl_bn_batch_stats_fused_fwd(_p(mraw), N, P, h * w, eps2, mom2, _p(rm2), _p(rv2), _p(g2), _p(b2), _p(ws2),
                                                _p(st2[0]), _p(st2[1]), _p(st2[2]), _p(st2[3]), _p(nbt2), _stream()))
        y = ops.pointwise(mraw, Epi(st2[2], st2[3], m_alpha), out=y)
        ctx.save_for_backward(x, zcat, mraw, st0, st2, br_alpha_c, merge_w, m_alpha, g0, g2, ws2, *stage_ws, *[pooled[i] for i in down])
        ctx.sizes, ctx.down = [tuple(int(v) for v in s_) for s_ in sizes], down
        ctx.sinks = ([_sink(t) for t in (br_gamma, br_beta, br_alpha, merge_w, m_gamma, m_beta, m_alpha)],
                     [_sink(t) for t in stage_ws])
        return y

    @staticmethod
    def backward(ctx, gy):
        sv = ctx.saved_tensors
        x, zcat, mraw, st0, st2, br_alpha, merge_w, m_alpha, g0, g2, ws2 = sv[:11]
        sizes, down = ctx.sizes, ctx.down
        nb = len(sizes)
        stage_ws = sv[11:11 + nb]
        pooled = dict(zip(down, sv[11 + nb:]))
        gy = _c(gy)
        N, P, h, w = x.shape
        dev = x.device
        C0 = nb * P
        psinks, wsinks = ctx.sinks

        def dst(sink, shape):
            return sink if sink is not None else torch.zeros(shape, device=dev, dtype=torch.float32)
        # merge_layer.2's BatchNorm + PReLU (batch statistics): direct path + statistics path
        direct2 = psinks[4] is not None and psinks[5] is not None
        out2 = torch.empty(4, P, dtype=torch.float32, device=dev)             # d gamma, d beta, p, q
        gal2 = dst(psinks[6], (P,)) if m_alpha is not None else None
        check(lib.mspl_bn_train_prelu_bwd(_p(mraw), None, _p(gy), _p(st2[2]), _p(st2[3]), _p(m_alpha), _p(g2), _p(st2[0]), _p(st2[1]),
                                          N, P, h * w, None, None, _p(ws2), 1 if direct2 else 0,
                                          _p(psinks[4] if direct2 else out2[0]), _p(psinks[5] if direct2 else out2[1]), _p(gal2),
                                          _p(out2[2]), _p(out2[3]), _stream()))
        g_mraw = torch.empty_like(mraw)                                       # p * z + q + gc, gc recomputed from (z, gy)
        check(lib.mspl_bn_train_prelu_bwd_apply(_p(mraw), _p(gy), _p(st2[2]), _p(st2[3]), _p(m_alpha), _p(out2[2]), _p(out2[3]), N, P,
                                                h * w, _p(g_mraw), _stream()))
        # the merge convolution and merge_layer.0's direct path; raw (d scale, d shift) sums of merge_layer.0
        zbuf = torch.zeros(2 * C0 + 2 * P, device=dev, dtype=torch.float32)   # one fill for both accumulators
        raw0 = zbuf[:2 * C0].view(2, C0)
        scratch2 = zbuf[2 * C0:].view(2, P)                                   # (merge_layer.2 is the identity here: sums not used)
        g_br_alpha = dst(psinks[2], (C0,))
        g_merge_w = dst(psinks[3], tuple(merge_w.shape))
        gt = torch.empty((nb, N, P, h, w), device=dev, dtype=torch.float32)
        check(lib.mspl_pyrpool_merge_bwd(_p(g_mraw), _p(mraw), _p(zcat), N, P, h, w, nb, _p(st0[2]), _p(st0[3]), _p(br_alpha), None, None,
                                         _p(merge_w), _p(_const_vec(1.0, P, dev)), _p(_const_vec(0.0, P, dev)), None, None, None, _p(gt),
                                         _p(raw0[0]), _p(raw0[1]), _p(g_br_alpha), _p(g_merge_w), _p(scratch2[0]), _p(scratch2[1]), None,
                                         _stream()))
        direct0 = psinks[0] is not None and psinks[1] is not None
        out0 = torch.empty(4, C0, dtype=torch.float32, device=dev)            # d gamma, d beta, p, q
        check(lib.mspl_bn_batch_stats_bwd_coeffs(_p(raw0[0]), _p(raw0[1]), _p(g0), _p(st0[0]), _p(st0[1]), _p(st0[2]), C0,
                                                 float(N * h * w), 1 if direct0 else 0, _p(psinks[0] if direct0 else out0[0]),
                                                 _p(psinks[1] if direct0 else out0[1]), _p(out0[2]), _p(out0[3]), _stream()))
        check(lib.mspl_bn_stats_path_add(_p(gt), _p(zcat), _p(out0[2]), _p(out0[3]), N, P, nb, h * w, _stream()))
        # the branches (as in PyrBodyFn)
        g_stage = [dst(wsinks[i], tuple(stage_ws[i].shape)) for i in range(nb)]
        adds = _pyr_down_backward(gt, (N, P, h, w), sizes, stage_ws, pooled, down, g_stage)
        up = [i for i in range(nb) if i not in down]
        nbp = len(up)
        hsa = (ctypes.c_int32 * nbp)(*[sizes[i][0] for i in up])
        wsa = (ctypes.c_int32 * nbp)(*[sizes[i][1] for i in up])
        swp = (ctypes.c_void_p * nbp)(*[stage_ws[i].data_ptr() for i in up])
        gtp = (ctypes.c_void_p * nbp)(*[gt[i].data_ptr() for i in up])
        gwp = (ctypes.c_void_p * nbp)(*[g_stage[i].data_ptr() for i in up])
        gx = torch.empty_like(x)
        check(lib.mspl_pyrpool_branch_bwd(_p(x), N, P, h, w, nbp, hsa, wsa, swp, gtp, gwp, _p(adds[0] if adds else None),
                                          _p(adds[1] if len(adds) > 1 else None), _p(gx), _stream()))
        ret = lambda sink, t: None if sink is not None else t          # noqa: E731
        return (gx, None, None, None,
                None if direct0 else out0[0], None if direct0 else out0[1], ret(psinks[2], g_br_alpha), ret(psinks[3], g_merge_w),
                None if direct2 else out2[0], None if direct2 else out2[1],
                None if m_alpha is None else ret(psinks[6], gal2),
                *[ret(wsinks[i], g_stage[i]) for i in range(nb)])


def bn_train_params(bn, device):
    """(running_mean, running_var, eps, momentum, workspace, num_batches_tracked) of a BatchNorm2d in train() for the fused nodes."""
    if bn.momentum is None or not bn.track_running_stats or not bn.affine:
        raise RuntimeError('mspl_amd: BatchNorm2d variants without momentum / running statistics / affine parameters are '
                           'not on the path (the reference uses the defaults everywhere)')
    return (bn.running_mean, bn.running_var, float(bn.eps), float(bn.momentum), _bn_workspace(bn, device), _bn_nbt(bn))


def pyr_body_fits(shape, sizes):
    """True when the fused training body (PyrBodyFn) covers an (N,P,h,w) projection with these branch sizes: every branch purely up
    (>= the map in both directions) or purely down, one to three of the former, at most two of the latter, tiles fit LDS."""
    N, P, h, w = [int(v) for v in shape]
    sizes = [(int(a), int(b)) for a, b in sizes]
    up = [(a, b) for a, b in sizes if a >= h and b >= w]
    down = [(a, b) for a, b in sizes if (a, b) not in up]
    if any(a > h or b > w for a, b in down) or not (1 <= len(up) <= 3) or len(down) > 2 or len(sizes) > 5:
        return False
    nb = len(sizes)
    hs = (ctypes.c_int32 * nb)(*[a for a, _ in sizes])
    ws = (ctypes.c_int32 * nb)(*[b for _, b in sizes])
    if not lib.mspl_pyrpool_fused_train_fits(N, P, h, w, nb, hs, ws):
        return False
    hsa = (ctypes.c_int32 * len(up))(*[a for a, _ in up])
    wsa = (ctypes.c_int32 * len(up))(*[b for _, b in up])
    return bool(lib.mspl_pyrpool_branch_bwd_fits(N, P, h, w, len(up), hsa, wsa))


def fan_out(x, n):
    """n aliases of x whose gradients are summed in one launch (use where one tensor feeds n >= 3 branches)."""
    if n < 2 or n > 8 or not x.requires_grad:
        return (x,) * n
    return FanOutFn.apply(x, n)


class AvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.shape = x.shape
        return ops.avgpool3x3s2(_c(x))

    @staticmethod
    def backward(ctx, gy):
        N, C, H, W = ctx.shape
        gx = torch.empty(ctx.shape, device=gy.device, dtype=torch.float32)
        check(lib.mspl_avgpool3x3s2_bwd(_p(_c(gy)), N, C, H, W, _p(gx), _stream()))
        return gx


class BilinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, size):
        ctx.shape = x.shape
        return ops.bilinear(_c(x), size)

    @staticmethod
    def backward(ctx, gy):
        N, C, H, W = ctx.shape
        gx = torch.empty(ctx.shape, device=gy.device, dtype=torch.float32)
        check(lib.mspl_bilinear_bwd(_p(_c(gy)), N, C, H, W, gy.shape[2], gy.shape[3], _p(gx), _stream()))
        return gx, None


class TwoHeadSumFn(torch.autograd.Function):
    """up(main) + 0.5 * up(aux) at `size` (`outputs[0] + 0.5 * outputs[1]` of train_seg_ue, utilities/train_eval_seg.py:187, with the two
    final bilinear up-samplings of espdnet_ue.py:301-302) as one node of two launches: the second up-sampling applies the 0.5 and adds
    the first in its epilogue (same roundings as the three ATen / bilinear launches: the 0.5 is exact).  Backward: the two transposed
    interpolations of the SAME gradient and the 0.5 on the small aux-sized result (exact, commutes with the sums)."""

    @staticmethod
    def forward(ctx, main, aux, size):
        main, aux = _c(main), _c(aux)
        ctx.shapes = (main.shape, aux.shape)
        out = ops.bilinear(main, size)
        half = _const_vec(0.5, aux.shape[1], aux.device)
        return ops.bilinear(aux, size, Epi(scale=half, residual=out), out=(out, 0))

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        outs = []
        for shp in ctx.shapes:
            gx = torch.empty(shp, device=g.device, dtype=torch.float32)
            check(lib.mspl_bilinear_bwd(_p(g), shp[0], shp[1], shp[2], shp[3], g.shape[2], g.shape[3], _p(gx), _stream()))
            outs.append(gx)
        outs[1].mul_(0.5)
        return outs[0], outs[1], None


def two_head_sum(main, aux, size):
    return TwoHeadSumFn.apply(main, aux, tuple(int(v) for v in size))


class AdaptivePoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, size):
        ctx.shape = x.shape
        return ops.adaptive_avgpool(_c(x), size)

    @staticmethod
    def backward(ctx, gy):
        N, C, H, W = ctx.shape
        gx = torch.empty(ctx.shape, device=gy.device, dtype=torch.float32)
        check(lib.mspl_adaptive_avgpool_bwd(_p(_c(gy)), N, C, H, W, gy.shape[2], gy.shape[3], _p(gx), _stream()))
        return gx, None


_GATE_EXPANDED = os.environ.get('MSPL_GATE_EXPANDED', '1') != '0'      # A/B aid


class GapGateFn(torch.autograd.Function):
    """gate = sigmoid(W . mean_hw(x)) -> (N, Cout)."""

    @staticmethod
    def forward(ctx, x, w):
        x, w = _c(x), _c(w)
        N, Cin, H, W = x.shape
        Cout = w.shape[0]
        mean = torch.empty((N, Cin), device=x.device, dtype=torch.float32)
        gate = torch.empty((N, Cout), device=x.device, dtype=torch.float32)
        check(lib.mspl_gap_gate_fwd(_p(x), _p(w), N, Cin, Cout, H * W, _p(mean), _p(gate), _stream()))
        ctx.save_for_backward(mean, gate, w)
        ctx.shape = x.shape
        ctx.wsink = _sink(w)
        return gate

    @staticmethod
    def backward(ctx, ggate):
        mean, gate, w = ctx.saved_tensors
        N, Cin, H, W = ctx.shape
        Cout = w.shape[0]
        gmean = torch.empty_like(mean)
        if ctx.wsink is not None:      # atomically into the parameter's gradient buffer (other micro-batch lanes add to it as well)
            gw = None
            check(lib.mspl_gap_gate_bwd_accum(_p(_c(ggate)), _p(gate), _p(mean), _p(w), N, Cin, Cout, _p(ctx.wsink), _p(gmean), _stream()))
        else:
            gw = torch.empty_like(w)
            check(lib.mspl_gap_gate_bwd(_p(_c(ggate)), _p(gate), _p(mean), _p(w), N, Cin, Cout, _p(gw), _p(gmean), _stream()))
        gx = None
        if ctx.needs_input_grad[0]:
            if _GATE_EXPANDED:
                # d mean / d x is 1 / HW everywhere in the plane: hand out the (N,Cin,1,1) values expanded over the plane (strides 0).  A
                # FanOutFn downstream folds them into its one summation launch; any other consumer materialises them (_c).
                gsc = torch.empty_like(gmean)
                check(lib.mspl_plane_broadcast(_p(gmean), N * Cin, 1, 1.0 / (H * W), 0, _p(gsc), _stream()))       # (N * Cin values)
                gx = gsc.view(N, Cin, 1, 1).expand(N, Cin, H, W)
            else:
                gx = torch.empty(ctx.shape, device=w.device, dtype=torch.float32)
                check(lib.mspl_plane_broadcast(_p(gmean), N * Cin, H * W, 1.0 / (H * W), 0, _p(gx), _stream()))
        return gx, gw


class FusionGateFn(torch.autograd.Function):
    """FusionGate blend (nn_layers/fusion_gate.py:36-44); z is None for the non-trainable gate (rgb + depth)."""

    @staticmethod
    def forward(ctx, z, rgb, depth):
        rgb, depth = _c(rgb), _c(depth)
        z = None if z is None else _c(z)
        out = torch.empty_like(rgb)
        check(lib.mspl_fusion_gate_fwd(None if z is None else _p(z), _p(rgb), _p(depth), rgb.numel(), _p(out), _stream()))
        ctx.gated = z is not None
        ctx.save_for_backward(*((z, rgb, depth) if ctx.gated else ()))
        return out

    @staticmethod
    def backward(ctx, gy):
        gy = _c(gy)
        grgb, gdepth = torch.empty_like(gy), torch.empty_like(gy)
        if ctx.gated:
            z, rgb, depth = ctx.saved_tensors
            gz = torch.empty_like(gy)
            check(lib.mspl_fusion_gate_bwd(_p(z), _p(rgb), _p(depth), _p(gy), gy.numel(), _p(gz), _p(grgb), _p(gdepth),
                                           _stream()))
            return gz, grgb, gdepth
        check(lib.mspl_fusion_gate_bwd(None, None, None, _p(gy), gy.numel(), None, _p(grgb), _p(gdepth), _stream()))
        return None, grgb, gdepth


class ChannelScaleFn(torch.autograd.Function):
    """y[n,c,:,:] * gate[n,c]."""

    @staticmethod
    def forward(ctx, y, gate):
        y, gate = _c(y), _c(gate)
        ctx.save_for_backward(y, gate)
        return ops.pointwise(y, Epi(gate=gate))

    @staticmethod
    def backward(ctx, g):
        y, gate = ctx.saved_tensors
        g = _c(g)
        N, C = y.shape[:2]
        hw = y[0, 0].numel()
        gy = ops.pointwise(g, Epi(gate=gate))
        ggate = torch.empty_like(gate)
        check(lib.mspl_plane_dot(_p(g), _p(y), N * C, hw, _p(ggate), _stream()))
        return gy, ggate


class UWLossFn(torch.autograd.Function):
    """K11: (criterion(pred + 0.5*aux, target, kld) * ce_scale + kld.mean()) * out_scale with kld = PixelwiseKLD(pred, aux).
    out_scale rides on the kernel's 1/npix (a micro-batch lane back-propagates loss / lanes).  root=True: the caller promises that
    this value is the ROOT of the backward (`loss.backward()` on it, upstream gradient 1): the backward then hands out the gradients
    the forward kernel wrote instead of multiplying both full-size tensors by a one."""

    @staticmethod
    def forward(ctx, pred, aux, target, class_weights, ce_scale, out_scale, root):
        pred, aux = _c(pred), _c(aux)
        N, C = pred.shape[:2]
        hw = pred[0, 0].numel()
        target = _c(target.to(torch.int64))
        loss = torch.zeros(1, device=pred.device, dtype=torch.float32)
        gpred, gaux = torch.empty_like(pred), torch.empty_like(aux)
        check(lib.mspl_uw_loss_scaled_fwd_bwd(_p(pred), _p(aux), _p(target), _p(_c(class_weights.float())), N, C, hw, float(ce_scale),
                                              float(out_scale), _p(loss), _p(gpred), _p(gaux), None, _stream()))
        ctx.save_for_backward(gpred, gaux)
        ctx.root = bool(root)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        gpred, gaux = ctx.saved_tensors
        if ctx.root:
            return gpred, gaux, None, None, None, None, None
        return gpred * g, gaux * g, None, None, None, None, None


class UWLossHeadsFn(torch.autograd.Function):
    """UWLossFn on the decoder's two outputs BEFORE their up-sampling to the label map (espdnet_ue.py:301-302): value and gradients of
    uw_loss(bilinear(main, size), bilinear(aux, size), ...) w.r.t. the low-resolution maps.  The up-sampling happens inside the loss
    kernel (the two full-size logit tensors never exist); the two transposed interpolations of its gradients run right behind it, so
    the node keeps two low-resolution tensors: three launches instead of five."""

    @staticmethod
    def forward(ctx, main, aux, target, class_weights, ce_scale, out_scale, root):
        main, aux = _c(main), _c(aux)
        N, C, Hm, Wm = main.shape
        Ha, Wa = aux.shape[2:]
        H, W = target.shape[-2:]
        target = _c(target.to(torch.int64))
        loss = torch.zeros(1, device=main.device, dtype=torch.float32)
        gfull = torch.empty((2, N, C, H, W), device=main.device, dtype=torch.float32)
        check(lib.mspl_uw_loss_heads_fwd_bwd(_p(main), _p(aux), _p(target), _p(_c(class_weights.float())), N, C, Hm, Wm, Ha, Wa, H, W,
                                             float(ce_scale), float(out_scale), _p(loss), _p(gfull[0]), _p(gfull[1]), _stream()))
        gmain, gaux = torch.empty_like(main), torch.empty_like(aux)
        check(lib.mspl_bilinear_bwd(_p(gfull[0]), N, C, Hm, Wm, H, W, _p(gmain), _stream()))
        check(lib.mspl_bilinear_bwd(_p(gfull[1]), N, C, Ha, Wa, H, W, _p(gaux), _stream()))
        ctx.save_for_backward(gmain, gaux)
        ctx.root = bool(root)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        gmain, gaux = ctx.saved_tensors
        if ctx.root:
            return gmain, gaux, None, None, None, None, None
        return gmain * g, gaux * g, None, None, None, None, None


# functional spellings
def conv(x, w, stride=1, groups=1):
    return ConvFn.apply(x, w, stride, groups)


def eesp_dw(x, ws, dil, stride):
    return EespDwFn.apply(x, ws[0], ws[1], ws[2], ws[3], dil, stride)


class DownTailFn(torch.autograd.Function):
    """y = PReLU(cat[a, b] + reinf): the tail of a DownSampler (nn_layers/eesp.py:131-144) as one forward and one backward launch.
    The node-per-op form ran torch.cat, then the affine/PReLU kernel; in the backward autograd's CatBackward handed out two channel
    SLICES of one gradient tensor, which had to be copied into contiguous tensors before the pool's and the EESP block's backward
    kernels could take them (3 cat + 6 copy launches per step over the three DownSamplers)."""

    @staticmethod
    def forward(ctx, a, b, alpha, reinf):
        a, b = _c(a), _c(b)
        reinf = None if reinf is None else _c(reinf)
        N, nin = a.shape[:2]
        C = nin + b.shape[1]
        hw = a[0, 0].numel()
        y = torch.empty((N, C) + tuple(a.shape[2:]), device=a.device, dtype=torch.float32)
        check(lib.mspl_down_tail_fwd(_p(a), _p(b), _p(reinf), _p(alpha), N, nin, C, hw, _p(y), _stream()))
        ctx.save_for_backward(a, b, alpha, reinf)
        ctx.sink = _sink(alpha)
        return y

    @staticmethod
    def backward(ctx, gy):
        a, b, alpha, reinf = ctx.saved_tensors
        gy = _c(gy)
        N, nin = a.shape[:2]
        C = nin + b.shape[1]
        hw = a[0, 0].numel()
        ga, gb = torch.empty_like(a), torch.empty_like(b)
        gr = torch.empty_like(reinf) if reinf is not None else None
        gal = ctx.sink if ctx.sink is not None else torch.zeros(C, device=a.device)
        check(lib.mspl_down_tail_bwd(_p(a), _p(b), _p(reinf), _p(gy), _p(alpha), N, nin, C, hw, _p(ga), _p(gb), _p(gr), _p(gal), _stream()))
        return ga, gb, (None if ctx.sink is not None else gal), gr


def down_tail(a, b, alpha, reinf=None):
    """PReLU(cat[a, b] + reinf); planes that are not a multiple of 4 pixels take the node-per-op form."""
    if a[0, 0].numel() % 4 or a.shape[0] != b.shape[0] or a.shape[2:] != b.shape[2:]:
        return affine_prelu(torch.cat([a, b], 1), None, None, alpha, residual=reinf)
    return DownTailFn.apply(a, b, alpha, reinf)


def affine_prelu(c, scale=None, shift=None, alpha=None, pre_add=None, residual=None):
    return AffinePReLUFn.apply(c, scale, shift, alpha, pre_add, residual, None, None, None, None)


def frozen_bn_inv(bn):
    """rsqrt(running_var + eps), cached on the module until running_var is written (frozen statistics)."""
    rv = bn.running_var
    key = (rv.data_ptr(), rv._version)
    c = bn.__dict__.get('_mspl_inv')
    if c is None or c[0] != key:
        with torch.no_grad():
            c = (key, torch.rsqrt(rv + bn.eps))
        bn.__dict__['_mspl_inv'] = c
    return c[1]


def conv_bn_prelu(x, w, stride, groups, bn, scale, shift, alpha=None, pre_add=None, residual=None):
    """PReLU(frozenBN(conv(x, w) + pre_add) + residual), one forward launch; (scale, shift) the current no-grad fold of bn."""
    return ConvAffinePReLUFn.apply(x, w, stride, groups, scale, shift, alpha, pre_add, residual, bn.weight, bn.bias, bn.running_mean,
                                   frozen_bn_inv(bn))


def conv_affine_prelu(x, w, stride, groups, scale=None, shift=None, alpha=None, pre_add=None, residual=None):
    return ConvAffinePReLUFn.apply(x, w, stride, groups, scale, shift, alpha, pre_add, residual, None, None, None, None)


def bn_prelu(c, bn, scale, shift, alpha=None, pre_add=None, residual=None):
    """PReLU(frozenBN(c + pre_add) + residual) with (scale, shift) the current no-grad fold of bn."""
    return AffinePReLUFn.apply(c, scale, shift, alpha, pre_add, residual, bn.weight, bn.bias, bn.running_mean, frozen_bn_inv(bn))


class BNBatchStatsFn(torch.autograd.Function):
    """Training-mode BatchNorm2d as a differentiable fold: (scale, shift) = (gamma*invstd, beta - mean*gamma*invstd) with
    mean / biased variance taken over (N,H,W) of z (nn.BatchNorm2d.forward in train(); the supervised loop of
    train_segmentation.py / utilities/train_eval_seg.py:174).  The statistics kernel also applies the running-statistics
    update (momentum, unbiased variance).  Backward: the statistics' dependence on z is an affine map per channel,
    gz = p_c*z + q_c, one pointwise launch; the direct path through (z*scale + shift) is AffinePReLUFn's."""

    @staticmethod
    def forward(ctx, z, gamma, beta, running_mean, running_var, eps, momentum):
        z = _c(z)
        N, C = z.shape[:2]
        hw = z[0, 0].numel()
        ws = torch.empty(2 * C, dtype=torch.float64, device=z.device)
        st = torch.empty(4, C, dtype=torch.float32, device=z.device)       # mean, invstd, scale, shift
        mean, invstd, scale, shift = st[0], st[1], st[2], st[3]
        gamma, beta = _c(gamma), _c(beta)
        check(lib.mspl_bn_batch_stats_fold_fwd(_p(z), N, C, hw, eps, momentum, _p(running_mean), _p(running_var), _p(gamma), _p(beta),
                                               _p(ws), _p(mean), _p(invstd), _p(scale), _p(shift), _stream()))
        ctx.save_for_backward(z, gamma, mean, invstd, scale)
        return scale, shift

    @staticmethod
    def backward(ctx, gsc, gsh):
        z, gamma, mean, invstd, scale = ctx.saved_tensors
        C = z.shape[1]
        M = z.numel() // C
        out = torch.empty(3, C, dtype=torch.float32, device=z.device)       # ggamma, p, q: one launch (was eight ATen launches)
        gsc, gsh = _c(gsc), _c(gsh)
        check(lib.mspl_bn_batch_stats_bwd_coeffs(_p(gsc), _p(gsh), _p(gamma), _p(mean), _p(invstd), _p(scale), C, float(M), 0,
                                                 _p(out[0]), None, _p(out[1]), _p(out[2]), _stream()))
        gz = ops.pointwise(z, Epi(out[1], out[2]))
        return gz, out[0], gsh, None, None, None, None


class BNTrainPReLUFn(torch.autograd.Function):
    """y = PReLU(BatchNorm_train(z) + residual) with batch statistics (nn.BatchNorm2d in train(), the supervised loop), as ONE
    autograd node of FOUR launches: statistics + fold (the workgroup that adds a channel's last partial finishes the channel), the
    affine / PReLU kernel; backward: the affine backward whose last workgroup per channel turns the sums into (d gamma, d beta, p, q),
    and ONE launch gz = p * z + q + gc (without a residual gc is recomputed there from (z, gy), not stored and read back).  `ws`: the BatchNorm's persistent workspace (zeroed once, handed back zeroed by both
    kernels: no memset, no finalize launch, no coefficient launch, no zeros() per call -- eight launches before).  As two nodes
    (BNBatchStatsFn + AffinePReLUFn) autograd added the two full-size gradients of z with an ATen kernel per BatchNorm and accumulated
    d gamma / d beta with two more (320 `add_` launches, 1.85 ms of a 17.5 ms iteration)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, alpha, residual, running_mean, running_var, eps, momentum, ws, nbt):
        z = _c(z)
        residual = None if residual is None else _c(residual)
        N, C = z.shape[:2]
        hw = z[0, 0].numel()
        st = torch.empty(4, C, dtype=torch.float32, device=z.device)       # mean, invstd, scale, shift
        gamma_c, beta_c = _c(gamma), _c(beta)
        ctx.small = _SMALL_BN and bool(lib.mspl_bn_train_small_fits(N, C, hw))
        if ctx.small:
            # small planes: the channel's whole node in one workgroup, one launch (statistics + fold + apply)
            y = torch.empty_like(z)
            check(lib.mspl_bn_train_small_fwd(_p(z), _p(residual), _p(gamma_c), _p(beta_c), _p(None if alpha is None else _c(alpha)), N, C, hw,
                                              eps, momentum, _p(running_mean), _p(running_var), _p(nbt), _p(st[0]), _p(st[1]), _p(st[2]),
                                              _p(st[3]), _p(y), _stream()))
        else:
            check(lib.mspl_bn_batch_stats_fused_fwd(_p(z), N, C, hw, eps, momentum, _p(running_mean), _p(running_var), _p(gamma_c),
                                                    _p(beta_c), _p(ws), _p(st[0]), _p(st[1]), _p(st[2]), _p(st[3]), _p(nbt), _stream()))
            y = ops.pointwise(z, Epi(st[2], st[3], alpha, residual=residual))
        ctx.save_for_backward(z, gamma_c, alpha, residual, st, ws)
        ctx.sinks = (_sink(gamma), _sink(beta), _sink(alpha))
        return y

    @staticmethod
    def backward(ctx, gy):
        z, gamma, alpha, residual, st, ws = ctx.saved_tensors
        mean, invstd, scale, shift = st[0], st[1], st[2], st[3]
        gy = _c(gy)
        N, C = z.shape[:2]
        hw = z[0, 0].numel()
        s_g, s_b, s_a = ctx.sinks
        gres = torch.empty_like(z) if residual is not None else None
        if ctx.small:
            gal = None
            if alpha is not None:
                gal = s_a if s_a is not None else torch.zeros(C, device=z.device)
            direct = s_g is not None and s_b is not None
            out = torch.empty(2, C, dtype=torch.float32, device=z.device)
            gz = torch.empty_like(z)
            check(lib.mspl_bn_train_small_bwd(_p(z), _p(residual), _p(gy), _p(scale), _p(shift), _p(alpha), _p(gamma), _p(mean), _p(invstd),
                                              N, C, hw, 1 if direct else 0, _p(gz), _p(gres), _p(s_g if direct else out[0]),
                                              _p(s_b if direct else out[1]), _p(gal), _stream()))
            return (gz, None if direct else out[0], None if direct else out[1],
                    None if (alpha is None or s_a is not None) else gal, gres, None, None, None, None, None, None)
        gc = torch.empty_like(z) if residual is not None else None       # no residual: the second pass recomputes it from (z, gy)
        gal = None
        if alpha is not None:
            gal = s_a if s_a is not None else torch.zeros(C, device=z.device)
        direct = s_g is not None and s_b is not None
        out = torch.empty(4, C, dtype=torch.float32, device=z.device)      # d gamma, d beta, p, q
        check(lib.mspl_bn_train_prelu_bwd(_p(z), _p(residual), _p(gy), _p(scale), _p(shift), _p(alpha), _p(gamma), _p(mean), _p(invstd),
                                          N, C, hw, _p(gres), _p(gc), _p(ws), 1 if direct else 0, _p(s_g if direct else out[0]),
                                          _p(s_b if direct else out[1]), _p(gal), _p(out[2]), _p(out[3]), _stream()))
        if gc is not None:
            gz = ops.pointwise(z, Epi(out[2], out[3], residual=gc))       # p * z + q + gc
        else:
            gz = torch.empty_like(z)
            check(lib.mspl_bn_train_prelu_bwd_apply(_p(z), _p(gy), _p(scale), _p(shift), _p(alpha), _p(out[2]), _p(out[3]), N, C, hw,
                                                    _p(gz), _stream()))
        return (gz, None if direct else out[0], None if direct else out[1],
                None if (alpha is None or s_a is not None) else gal, gres, None, None, None, None, None, None)


_SMALL_BN = os.environ.get('MSPL_BN_SMALL', '1') != '0'       # small planes: one launch per BatchNorm node and direction


def _bn_nbt(bn):
    """bn.num_batches_tracked as the statistics kernels take it (they add 1 through the raw pointer): an int64 tensor ON THE DEVICE --
    a CPU or mistyped buffer would become a device-side write through a host pointer."""
    nbt = bn.num_batches_tracked
    if nbt is None or not (nbt.is_cuda and nbt.dtype == torch.int64):
        raise RuntimeError('mspl_amd: num_batches_tracked must be an int64 tensor on the device')
    return nbt


def _bn_workspace(bn, device):
    """The BatchNorm's persistent, zeroed workspace of the two fused kernels (they hand it back zeroed)."""
    # one workspace per (module, stream): the kernels' "handed back zeroed" contract holds per stream of launches; the same module on
    # two streams at once (side streams of the layers, two lanes) must not share partial sums
    key = (str(device), int(torch.cuda.current_stream(device).cuda_stream))
    tab = bn.__dict__.setdefault('_mspl_bn_ws', {})
    ws = tab.get(key)
    if ws is None:
        ws = tab[key] = torch.zeros(int(lib.mspl_bn_fused_workspace_bytes(bn.num_features)) // 8 + 1, dtype=torch.float64, device=device)
    return ws


def bn_train_prelu(z, bn, alpha=None, residual=None):
    """PReLU(bn(z) + residual) for a BatchNorm2d in train(); updates its running statistics like nn.BatchNorm2d does."""
    if bn.momentum is None or not bn.track_running_stats or not bn.affine:
        raise RuntimeError('mspl_amd: BatchNorm2d variants without momentum / running statistics / affine parameters are '
                           'not on the path (the reference uses the defaults everywhere)')
    nbt = _bn_nbt(bn)                     # incremented by the statistics kernel (an int64 CUDA scalar; nn.BatchNorm2d adds 1 per forward)
    return BNTrainPReLUFn.apply(z, bn.weight, bn.bias, alpha, residual, bn.running_mean, bn.running_var, bn.eps, bn.momentum,
                                _bn_workspace(bn, z.device), nbt)


def bn_batch_stats(z, bn):
    """(scale, shift) of a BatchNorm2d in train() for the batch z; updates running_mean / running_var /
    num_batches_tracked like nn.BatchNorm2d does."""
    if bn.momentum is None or not bn.track_running_stats or not bn.affine:
        raise RuntimeError('mspl_amd: BatchNorm2d variants without momentum / running statistics / affine parameters are '
                           'not on the path (the reference uses the defaults everywhere)')
    out = BNBatchStatsFn.apply(z, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum)
    with torch.no_grad():
        bn.num_batches_tracked += 1
    return out


avgpool = AvgPoolFn.apply
bilinear = BilinearFn.apply
adaptive_avgpool = AdaptivePoolFn.apply
gap_gate = GapGateFn.apply
channel_scale = ChannelScaleFn.apply


def uw_loss(pred, aux, target, class_weights, ce_scale=20.0, out_scale=1.0, root=False):
    return UWLossFn.apply(pred, aux, target, class_weights, ce_scale, out_scale, root)


def uw_loss_heads_supported(classes):
    return bool(lib.mspl_uw_loss_heads_supported(int(classes)))


def uw_loss_heads(main, aux, target, class_weights, ce_scale=20.0, out_scale=1.0, root=False):
    """uw_loss(bilinear(main, target.shape[-2:]), bilinear(aux, ...), ...) from the low-resolution heads."""
    return UWLossHeadsFn.apply(main, aux, target, class_weights, ce_scale, out_scale, root)
