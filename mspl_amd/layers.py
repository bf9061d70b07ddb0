"""Layer modules with the reference's class names, constructor signatures and state-dict keys, whose
forward passes run on the HIP kernels of libmspl_hip.so.

Reference surface mirrored (paths relative to the reference root):
  nn_layers/espnet_utils.py  CBR:8 BR:39 CB:62 C:91 CDilated:118        (encoder dialect: .conv/.bn/.act)
  nn_layers/cnn_utils.py     CBR:26 CB:56 BR:85 Shuffle:108              (decoder dialect: .cbr.0/1/2, .br.0/1)
  nn_layers/eesp.py          EESP:15-93 DownSampler:96-144
  nn_layers/efficient_pyramid_pool.py  EfficientPyrPool:12-61
  nn_layers/efficient_pt.py  EfficientPWConv:10-29

torch.nn.Conv2d / BatchNorm2d / PReLU objects are used as PARAMETER CONTAINERS only (so that
state_dict() keys, shapes and init match the reference); their ATen forward is never called.
BatchNorm runs in eval mode (running statistics), which is what the label pass and the uest
self-training loop use (uest_seg_multi_os.py:605-608); it is folded into per-channel scale/shift
vectors that are cached until a parameter changes.
"""
import math
import os
import weakref

import torch
from torch import nn

from . import autograd as ag
from . import ops
from .ops import Epi

config_inp_reinf = 3          # model/classification/espnetv2_config.py:20
PYR_SCALES = (2.0, 1.5, 1.0, 0.5, 0.1)


# ------------------------------------------------------------------ caching helpers
_PARAM_EPOCH = [0]


def bump_param_epoch():
    """Invalidate every cached fold / packed weight: call after parameters were modified through raw pointers
    (the fused Adam kernel), which torch's version counters cannot see."""
    _PARAM_EPOCH[0] += 1


def _key(tensors):
    return (_PARAM_EPOCH[0],) + tuple((t.data_ptr(), t._version) for t in tensors)


def cached(module, name, deps, builder):
    """Memoise builder() on `module` until any tensor in deps is modified, moved or replaced."""
    store = module.__dict__.setdefault('_mspl_cache', {})
    k = _key(deps)
    hit = store.get(name)
    if hit is None or hit[0] != k:
        with torch.no_grad():
            val = builder()
        store[name] = hit = (k, val)
    return hit[1]


def _require_eval(bn):
    if bn.training:
        raise RuntimeError('mspl_amd: BatchNorm in training mode under torch.no_grad(): the fused inference kernels fold the '
                           'running statistics; call model.eval() (the uest label pass runs frozen BN, '
                           'uest_seg_multi_os.py:605-608), or enable gradients for the batch-statistics training path')


def bn_fold(bn):
    """(scale, shift) with y = x*scale + shift == eval-mode BatchNorm2d."""
    _require_eval(bn)

    def build():
        scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
        return scale.contiguous(), (bn.bias - bn.running_mean * scale).contiguous()
    return cached(bn, 'fold', [bn.weight, bn.bias, bn.running_mean, bn.running_var], build)


# ------------------------------------------------------------------ independent branches on side streams
# The forward has branches that do not depend on each other (the image-reinforcement chain of a DownSampler vs its
# EESP; the three EfficientPWConv skip connections and the auxiliary decoder vs the main decoder).  Most kernels of
# the path are single-round launches whose ramp, latency chains and tail leave CUs idle, so these branches are issued
# on side HIP streams (fork: side waits for the current stream; join: the current stream waits for the side).  Under
# hipGraph capture the fork/join become parallel graph branches.  Inference path only; MSPL_SIDE_STREAMS=0 disables.
_SIDE_ENABLED = os.environ.get('MSPL_SIDE_STREAMS', '1') != '0'
_SIDE_STREAMS = {}


class side_streams(object):
    """with side_streams(False): run (or capture) the forward on one stream only.  PipelinedLabelPass captures its lanes this
    way: a linear hipGraph executes on its launch stream alone, so two lanes on two hardware queues overlap for certain,
    whereas the internal streams a branched graph is given at instantiation may land on the other lane's queue."""

    def __init__(self, enabled):
        self.enabled = enabled

    def __enter__(self):
        global _SIDE_ENABLED
        self.prev = _SIDE_ENABLED
        _SIDE_ENABLED = self.enabled and os.environ.get('MSPL_SIDE_STREAMS', '1') != '0'

    def __exit__(self, *exc):
        global _SIDE_ENABLED
        _SIDE_ENABLED = self.prev


def _side_stream(idx, device, parent):
    """Side stream idx of the stream `parent` (keyed by parent so that concurrent source models do not share one)."""
    dev = device.index if device.index is not None else torch.cuda.current_device()
    key = (dev, idx, parent.cuda_stream)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
    return st


class fork(object):
    """with fork(idx, inputs): ...   -- the body is issued on side stream idx; `inputs` are tensors produced on the
    current stream that the body reads (their memory must not be recycled while the side stream still uses them)."""

    def __init__(self, idx, inputs=()):
        self.idx, self.inputs, self.ctx = idx, inputs, None

    def __enter__(self):
        if not _SIDE_ENABLED or _training_path() or not self.inputs:
            return self
        cur = torch.cuda.current_stream(self.inputs[0].device)
        side = _side_stream(self.idx, self.inputs[0].device, cur)
        side.wait_stream(cur)
        for t in self.inputs:
            t.record_stream(side)
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False

    def mark(self):
        """An event after the work issued so far in this branch (None when the branch runs inline): lets the main stream
        wait for an early result (`wait_mark`) without waiting for everything the side stream was given."""
        if self.ctx is None:
            return None
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        return ev


def wait_mark(ev, outputs=()):
    """The current stream waits for a fork.mark() event; `outputs` were produced before it on the side stream."""
    if ev is None:
        return
    cur = torch.cuda.current_stream()
    cur.wait_event(ev)
    for t in outputs:
        if t is not None:
            t.record_stream(cur)


def join(idx, outputs=()):
    """The current stream waits for side stream idx; `outputs` were produced there and are consumed here."""
    outs = [t for t in outputs if t is not None]
    if not _SIDE_ENABLED or _training_path() or not outs:
        return
    cur = torch.cuda.current_stream(outs[0].device)
    cur.wait_stream(_side_stream(idx, outs[0].device, cur))
    for t in outs:
        t.record_stream(cur)


def _training_path():
    """True when gradients are enabled: the modules then run the autograd form (HIP forward + HIP backward per op,
    mspl_amd/autograd.py) instead of the fused inference kernels."""
    return torch.is_grad_enabled()


def _conv_train(x, conv):
    if conv.kernel_size[0] == 3 and conv.dilation[0] != 1:
        raise RuntimeError('mspl_amd: dilated 3x3 convolutions only occur inside EESP (fused branch op)')
    return ag.conv(x, conv.weight, conv.stride[0], conv.groups)


def _bn_act(z, bn, alpha=None, pre_add=None, residual=None):
    """PReLU(BN(z + pre_add) + residual) on the training path.  bn.eval(): frozen statistics folded into (scale, shift)
    (the uest loop, uest_seg_multi_os.py:605-608).  bn.train(): batch statistics (the supervised loop, model.train() at
    utilities/train_eval_seg.py:174) -- (scale, shift) then depend on z, running statistics are updated in place."""
    if bn.training:
        if pre_add is not None:
            z, pre_add = z + pre_add, None
        if _FUSED_BN_TRAIN:
            return ag.bn_train_prelu(z, bn, alpha, residual)
        scale, shift = ag.bn_batch_stats(z, bn)
    else:
        scale, shift = train_fold(bn)
        return ag.bn_prelu(z, bn, scale, shift, alpha, pre_add=pre_add, residual=residual)
    return ag.affine_prelu(z, scale, shift, alpha, pre_add=pre_add, residual=residual)


_FUSED_TRAIN_FWD = os.environ.get('MSPL_TRAIN_FUSED_FWD', '1') != '0'
_EESP_DW_BN = os.environ.get('MSPL_EESP_DW_BN', '1') != '0'    # EESP in train(): K2 + br_after_cat as one autograd node
_CONV_SKIP = os.environ.get('MSPL_CONV_SKIP', '1') != '0'      # EESP in train(): projection + skip connection as one autograd node
_FUSED_BN_TRAIN = os.environ.get('MSPL_FUSED_BN_TRAIN', '1') != '0'      # batch-statistics BN + PReLU as one autograd node
_FUSED_DW_EXP = os.environ.get('MSPL_EESP_EXP', '1') != '0'   # inference: K2 + K3 of a stride-1 EESP block as one launch
_FUSED_NEXT_PROJ = os.environ.get('MSPL_EESP_NEXT', '1') != '0'   # ... and the following block's proj_1x1 inside that launch
_FUSED_EESP_TRAIN = os.environ.get('MSPL_FUSED_EESP_TRAIN', '1') != '0'  # EESP block as one autograd node (frozen BatchNorm)
_FUSED_PYR_TRAIN = os.environ.get('MSPL_FUSED_PYR_TRAIN', '1') != '0'    # pyramid body as one autograd node (frozen BatchNorm)


def _conv_bn_act(x, conv, bn, alpha=None, pre_add=None, residual=None):
    """PReLU(BN(conv(x) + pre_add) + residual) on the training path.  Frozen BatchNorm: one forward launch (the convolution applies
    the folded BatchNorm / PReLU and keeps its bare result for the backward, autograd.ConvAffinePReLUFn).  Batch statistics: the
    convolution result is needed first (two launches, as before)."""
    if bn.training or not _FUSED_TRAIN_FWD:
        return _bn_act(_conv_train(x, conv), bn, alpha, pre_add, residual)
    if conv.kernel_size[0] == 3 and conv.dilation[0] != 1:
        raise RuntimeError('mspl_amd: dilated 3x3 convolutions only occur inside EESP (fused branch op)')
    scale, shift = train_fold(bn)
    return ag.conv_bn_prelu(x, conv.weight, conv.stride[0], conv.groups, bn, scale, shift, alpha, pre_add, residual)


def train_fold(bn):
    """No-grad (scale, shift) of a frozen BatchNorm for the training forward, cached until a parameter changes (every
    optimizer step); prefold_frozen_bn() fills the caches of a whole model with two multi-tensor launches."""
    def build():
        scale = bn.weight * ag.frozen_bn_inv(bn)
        return scale, torch.addcmul(bn.bias, bn.running_mean, scale, value=-1.0)
    return cached(bn, 'tfold', [bn.weight, bn.bias, bn.running_mean, bn.running_var], build)


def prefold_frozen_bn(model):
    """Fold every eval-mode BatchNorm2d of `model` at once (torch._foreach: a handful of launches instead of two per
    module) and store the results where train_fold() looks."""
    bns = [m for m in model.modules() if isinstance(m, nn.BatchNorm2d) and not m.training]
    if not bns:
        return
    with torch.no_grad():
        invs = [ag.frozen_bn_inv(m) for m in bns]
        scales = torch._foreach_mul([m.weight for m in bns], invs)
        shifts = torch._foreach_addcmul([m.bias for m in bns], [m.running_mean for m in bns], scales, value=-1.0)
    for m, sc, sh in zip(bns, scales, shifts):
        store = m.__dict__.setdefault('_mspl_cache', {})
        store['tfold'] = (_key([m.weight, m.bias, m.running_mean, m.running_var]), (sc, sh))


def _conv_fwd(x, conv, ep, out=None, shuffle_groups=0):
    """Dispatch a bias-free Conv2d container to the matching kernel."""
    k = conv.kernel_size[0]
    if k == 1:
        return ops.conv1x1(x, conv.weight, conv.groups, ep, out)
    if k == 3 and conv.dilation[0] == 1:
        return ops.conv3x3(x, conv.weight, conv.groups, conv.stride[0], shuffle_groups, ep, out)
    raise RuntimeError('mspl_amd: no kernel for conv k=%d dilation=%d outside an EESP block' % (k, conv.dilation[0]))


# ------------------------------------------------------------------ encoder dialect (espnet_utils)
class CBR(nn.Module):
    def __init__(self, nIn, nOut, kSize, stride=1, groups=1):
        super().__init__()
        padding = int((kSize - 1) / 2)
        self.conv = nn.Conv2d(nIn, nOut, kSize, stride=stride, padding=padding, bias=False, groups=groups)
        self.bn = nn.BatchNorm2d(nOut)
        self.act = nn.PReLU(nOut)

    def forward(self, input):
        if _training_path():
            return _conv_bn_act(input, self.conv, self.bn, self.act.weight)
        scale, shift = bn_fold(self.bn)
        return _conv_fwd(input, self.conv, Epi(scale, shift, self.act.weight))


class BR(nn.Module):
    def __init__(self, nOut):
        super().__init__()
        self.bn = nn.BatchNorm2d(nOut)
        self.act = nn.PReLU(nOut)

    def forward(self, input):
        if _training_path():
            return _bn_act(input, self.bn, self.act.weight)
        scale, shift = bn_fold(self.bn)
        return ops.pointwise(input, Epi(scale, shift, self.act.weight))


class CB(nn.Module):
    def __init__(self, nIn, nOut, kSize, stride=1, groups=1):
        super().__init__()
        padding = int((kSize - 1) / 2)
        self.conv = nn.Conv2d(nIn, nOut, kSize, stride=stride, padding=padding, bias=False, groups=groups)
        self.bn = nn.BatchNorm2d(nOut)

    def forward(self, input):
        if _training_path():
            return _conv_bn_act(input, self.conv, self.bn)
        scale, shift = bn_fold(self.bn)
        return _conv_fwd(input, self.conv, Epi(scale, shift))


class C(nn.Module):
    def __init__(self, nIn, nOut, kSize, stride=1, groups=1):
        super().__init__()
        padding = int((kSize - 1) / 2)
        self.conv = nn.Conv2d(nIn, nOut, kSize, stride=stride, padding=padding, bias=False, groups=groups)

    def forward(self, input):
        if _training_path():
            return _conv_train(input, self.conv)
        return _conv_fwd(input, self.conv, None)


class CDilated(nn.Module):
    """Parameter container of one dilated depthwise branch; EESP runs the four branches fused (K2)."""

    def __init__(self, nIn, nOut, kSize, stride=1, d=1, groups=1):
        super().__init__()
        padding = int((kSize - 1) / 2) * d
        self.conv = nn.Conv2d(nIn, nOut, kSize, stride=stride, padding=padding, bias=False, dilation=d,
                              groups=groups)

    def forward(self, input):
        if _training_path():
            return _conv_train(input, self.conv)
        return _conv_fwd(input, self.conv, None)


# ------------------------------------------------------------------ decoder dialect (cnn_utils)
class DecCBR(nn.Module):
    def __init__(self, nIn, nOut, kSize, stride=1, dilation=1, groups=1, act_name='prelu'):
        super().__init__()
        if act_name != 'prelu':
            raise NotImplementedError('mspl_amd: only PReLU activations are on the path')
        padding = int((kSize - 1) / 2) * dilation
        self.cbr = nn.Sequential(
            nn.Conv2d(nIn, nOut, kSize, stride=stride, padding=padding, bias=False, groups=groups, dilation=dilation),
            nn.BatchNorm2d(nOut),
            nn.PReLU(nOut))

    def epi(self, **kw):
        scale, shift = bn_fold(self.cbr[1])
        return Epi(scale, shift, self.cbr[2].weight, **kw)

    def forward(self, x):
        if _training_path():
            return _conv_bn_act(x, self.cbr[0], self.cbr[1], self.cbr[2].weight)
        return _conv_fwd(x, self.cbr[0], self.epi())


class DecBR(nn.Module):
    def __init__(self, nOut, act_name='prelu'):
        super().__init__()
        if act_name != 'prelu':
            raise NotImplementedError('mspl_amd: only PReLU activations are on the path')
        self.br = nn.Sequential(nn.BatchNorm2d(nOut), nn.PReLU(nOut))

    def forward(self, x):
        if _training_path():
            return _bn_act(x, self.br[0], self.br[1].weight)
        scale, shift = bn_fold(self.br[0])
        return ops.pointwise(x, Epi(scale, shift, self.br[1].weight))


class Shuffle(nn.Module):
    """Channel shuffle (cnn_utils.py:108-125).  On the path it is folded into the next conv's input
    indexing (mspl_conv3x3_fwd shuffle_groups); this standalone form only exists for API parity."""

    def __init__(self, groups):
        super().__init__()
        self.groups = groups

    def forward(self, x):
        n, c, h, w = x.shape
        return x.view(n, self.groups, c // self.groups, h, w).transpose(1, 2).contiguous().view(n, c, h, w)


# ------------------------------------------------------------------ EESP / DownSampler
def eesp_dilations(r_lim, k=4):
    """nn_layers/eesp.py:38-53."""
    ks = sorted((3 + 2 * i) if (3 + 2 * i) <= r_lim else 3 for i in range(k))
    return [(s - 1) // 2 for s in ks]


class EESP(nn.Module):
    def __init__(self, nIn, nOut, stride=1, k=4, r_lim=7, down_method='esp'):
        super().__init__()
        self.stride = stride
        n = int(nOut / k)
        n1 = nOut - (k - 1) * n
        assert down_method in ['avg', 'esp'], 'One of these is suppported (avg or esp)'
        assert n == n1, "n(={}) and n1(={}) should be equal for Depth-wise Convolution ".format(n, n1)
        if k != 4:
            raise NotImplementedError('mspl_amd: the fused EESP kernel is built for k=4 branches')
        self.proj_1x1 = CBR(nIn, n, 1, stride=1, groups=k)
        self.dilations = eesp_dilations(r_lim, k)
        self.spp_dw = nn.ModuleList(CDilated(n, n, kSize=3, stride=stride, groups=n, d=d) for d in self.dilations)
        self.conv_1x1_exp = CB(nOut, nOut, 1, 1, groups=k)
        self.br_after_cat = BR(nOut)
        self.module_act = nn.PReLU(nOut)
        self.downAvg = True if down_method == 'avg' else False
        self.k = k

    def _dw_weights(self):
        ws = [m.conv.weight for m in self.spp_dw]
        return cached(self, 'w4', ws, lambda: torch.stack([w.reshape(-1, 3, 3) for w in ws]).contiguous())

    def _dw_exp_packed(self, H, W):
        """Parameter block of the fused K2 + K3 launch (depthwise weights + br_after_cat's fold per reduced channel, expansion
        weights in matrix-operand order): rebuilt when any of its sources changes."""
        br, exp = self.br_after_cat, self.conv_1x1_exp
        deps = [m.conv.weight for m in self.spp_dw] + [br.bn.weight, br.bn.bias, br.bn.running_mean, br.bn.running_var,
                                                      br.act.weight, exp.conv.weight]

        def build():
            bs, bb = bn_fold(br.bn)
            return ops.eesp_dw_exp_pack(self._dw_weights(), bs, bb, br.act.weight, exp.conv.weight, H, W, self.dilations)
        return cached(self, 'dwexp%dx%d' % (H, W), deps, build)       # (the chunking of the packed block follows the launch plan of the shape)

    def reduce_transform(self, input):
        """K1 + K2: returns the BN+PReLU'd concatenation that feeds conv_1x1_exp."""
        scale, shift = bn_fold(self.br_after_cat.bn)
        pj = self.proj_1x1
        if ops.eesp_proj_dw_hff_fits(input.shape, pj.conv.out_channels, pj.conv.groups, self.dilations, self.stride):
            # level 4 (planes that fit LDS): projection on the matrix cores straight into K2's LDS tile, one launch
            ps, pb = bn_fold(pj.bn)
            return ops.eesp_proj_dw_hff(input, pj.conv.weight, ps, pb, pj.act.weight, self._dw_weights(), self.dilations,
                                        pj.conv.groups, Epi(scale, shift, self.br_after_cat.act.weight))
        o1 = pj(input)
        return ops.eesp_dw_hff(o1, self._dw_weights(), self.dilations, self.stride,
                               Epi(scale, shift, self.br_after_cat.act.weight))

    def _forward_train(self, input):
        pj, br, exp = self.proj_1x1, self.br_after_cat, self.conv_1x1_exp
        if _FUSED_EESP_TRAIN and not (pj.bn.training or br.bn.training or exp.bn.training):
            # the whole block as one autograd node (frozen BatchNorms): autograd.EESPFn
            def fold(bn):
                scale, shift = train_fold(bn)
                return {'scale': scale, 'shift': shift, 'mean': bn.running_mean, 'inv': ag.frozen_bn_inv(bn)}
            strided_avg = self.stride == 2 and self.downAvg
            residual = (not strided_avg) and self.stride == 1 and exp.conv.out_channels == input.shape[1]
            cfg = {'stride': self.stride, 'dil': tuple(self.dilations), 'groups': pj.conv.groups, 'residual': residual}
            ws = [m.conv.weight for m in self.spp_dw]
            return ag.EESPFn.apply(input, cfg, fold(pj.bn), fold(br.bn), fold(exp.bn), pj.conv.weight, pj.bn.weight, pj.bn.bias,
                                   pj.act.weight, ws[0], ws[1], ws[2], ws[3], br.bn.weight, br.bn.bias, br.act.weight,
                                   exp.conv.weight, exp.bn.weight, exp.bn.bias, None if strided_avg else self.module_act.weight)
        has_res = self.stride == 1 and exp.conv.out_channels == input.shape[1] and not (self.stride == 2 and self.downAvg)
        skip = input
        if has_res and pj.bn.training and _CONV_SKIP and input.requires_grad:
            # projection and skip connection on one autograd node: the projection's data gradient adds the skip gradient in its epilogue
            z1, skip = ag.conv_skip(input, pj.conv.weight, pj.conv.groups)
            o1 = _bn_act(z1, pj.bn, pj.act.weight)
        else:
            o1 = self.proj_1x1(input)
        if br.bn.training and _EESP_DW_BN and o1.requires_grad:
            # K2 + br_after_cat (batch statistics) as one node: its backward is two launches for the stride-1 blocks, three otherwise
            # (autograd.EespDwBNFn)
            cat = ag.eesp_dw_bn(o1, [m.conv.weight for m in self.spp_dw], self.dilations, br.bn, br.act.weight, self.stride)
        else:
            cat = ag.eesp_dw(o1, [m.conv.weight for m in self.spp_dw], self.dilations, self.stride)
            cat = self.br_after_cat(cat)
        if self.stride == 2 and self.downAvg:
            return _conv_bn_act(cat, exp.conv, exp.bn)
        return _conv_bn_act(cat, exp.conv, exp.bn, self.module_act.weight, residual=skip if has_res else None)

    def _fused_dw_exp(self, shape):
        """True when K2 + K3 of this block run as one launch for an input of `shape` (inference path)."""
        exp = self.conv_1x1_exp
        return (self.stride == 1 and exp.conv.out_channels == shape[1] and _FUSED_DW_EXP
                and ops.eesp_dw_exp_fits((shape[0], self.proj_1x1.conv.out_channels) + tuple(shape[2:]), self.dilations))

    def _next_proj(self):
        """This block's proj_1x1 in the form the PREVIOUS block's fused launch computes it: (packed weight, scale, shift, alpha)."""
        pj = self.proj_1x1
        ps, pb = bn_fold(pj.bn)
        return cached(self, 'nextproj', [pj.conv.weight], lambda: ops.eesp_dw_exp_next_pack(pj.conv.weight)), ps, pb, pj.act.weight

    def forward(self, input, _reduced=None, _next=None):
        """_reduced: this block's proj_1x1 output when the previous block's launch has already produced it; _next: the following
        block of a chain (run_eesp_chain): the call then returns (output, that block's reduced tensor or None)."""
        if _training_path():
            return self._forward_train(input)
        exp = self.conv_1x1_exp
        if self._fused_dw_exp(input.shape):
            # K2 + K3 in one launch: the 4n-channel concatenation stays on the CU (csrc/eesp_exp.hip)
            o1 = self.proj_1x1(input) if _reduced is None else _reduced
            scale, shift = bn_fold(exp.bn)
            ep = Epi(scale, shift, self.module_act.weight, residual=input)
            packed = self._dw_exp_packed(input.shape[2], input.shape[3])
            nxt = _next is not None and _FUSED_NEXT_PROJ and _next._fused_dw_exp(input.shape) \
                and _next.proj_1x1.conv.out_channels == self.proj_1x1.conv.out_channels and _next.proj_1x1.conv.groups == 4
            if nxt:
                return ops.eesp_dw_exp(o1, packed, self.dilations, ep, next_proj=_next._next_proj())
            y = ops.eesp_dw_exp(o1, packed, self.dilations, ep)
            return (y, None) if _next is not None else y
        if _next is not None:
            return self.forward(input), None
        cat = self.reduce_transform(input)
        scale, shift = bn_fold(self.conv_1x1_exp.bn)
        if self.stride == 2 and self.downAvg:
            return ops.conv1x1(cat, self.conv_1x1_exp.conv.weight, self.k, Epi(scale, shift))
        residual = input if (self.stride == 1 and self.conv_1x1_exp.conv.out_channels == input.shape[1]) else None
        return ops.conv1x1(cat, self.conv_1x1_exp.conv.weight, self.k,
                           Epi(scale, shift, self.module_act.weight, residual=residual))


def run_eesp_chain(blocks, x):
    """x through consecutive stride-1 EESP blocks (a level's stack).  On the inference path each block's fused K2 + K3 launch also
    computes the NEXT block's proj_1x1 (csrc/eesp_exp.hip), so a chain of B blocks is B + 1 launches instead of 3 B."""
    blocks = list(blocks)
    if _training_path():
        for m in blocks:
            x = m(x)
        return x
    reduced = None
    for i, m in enumerate(blocks):
        nxt = blocks[i + 1] if i + 1 < len(blocks) else None
        if nxt is None:
            x = m(x, _reduced=reduced)
        else:
            x, reduced = m(x, _reduced=reduced, _next=nxt)
    return x


class ImagePyramid:
    """The input image and its repeated 3x3/s2 average pools, computed once per forward and shared by the
    three DownSamplers (the reference recomputes the chain inside each one, nn_layers/eesp.py:136-140)."""

    def __init__(self, image):
        self.levels = [image]

    def at_height(self, h):
        """Pool at least once, then until the height matches (the reference's `while True` loop)."""
        i = 1
        while True:
            if i == len(self.levels):
                prev = self.levels[-1]
                if prev.shape[2] == 1 and prev.shape[3] == 1:
                    raise RuntimeError('mspl_amd: image pyramid never reaches height %d' % h)
                self.levels.append(ops.avgpool3x3s2(prev))
            if self.levels[i].shape[2] == h:
                return self.levels[i]
            i += 1


class DownSampler(nn.Module):
    def __init__(self, nin, nout, k=4, r_lim=9, reinf=True):
        super().__init__()
        nout_new = nout - nin
        self.eesp = EESP(nin, nout_new, stride=2, k=k, r_lim=r_lim, down_method='avg')
        self.avg = nn.AvgPool2d(kernel_size=3, padding=1, stride=2)
        if reinf:
            self.inp_reinf = nn.Sequential(
                CBR(config_inp_reinf, config_inp_reinf, 3, 1),
                CB(config_inp_reinf, nout, 1, 1))
        self.act = nn.PReLU(nout)
        self.nin, self.nout = nin, nout

    def _epilogue_vectors(self, with_reinf):
        """Per-destination-channel constants for the two writers of the (N, nout, Ho, Wo) result:
        channels [0,nin) <- avg-pool branch (scale 1), [nin,nout) <- conv_1x1_exp (its folded BN);
        the reinforcement CB(3,nout,1) is folded to a (nout,3) matrix + shift added on both."""
        bn = self.eesp.conv_1x1_exp.bn
        deps = [bn.weight, bn.bias, bn.running_mean, bn.running_var]
        if with_reinf:
            cb = self.inp_reinf[1]
            deps += [cb.conv.weight, cb.bn.weight, cb.bn.bias, cb.bn.running_mean, cb.bn.running_var]

        def build():
            es, eb = bn_fold(bn)
            scale = torch.cat([torch.ones(self.nin, device=es.device), es])
            shift = torch.cat([torch.zeros(self.nin, device=es.device), eb])
            rw = None
            if with_reinf:
                rs, rb = bn_fold(cb.bn)
                rw = (cb.conv.weight.reshape(self.nout, config_inp_reinf) * rs[:, None]).contiguous()
                shift = shift + rb
            return scale.contiguous(), shift.contiguous(), rw
        return cached(self, 'epi%d' % int(with_reinf), deps, build)

    def _forward_train(self, input, input2, alias=None):
        # (alias: a second autograd alias of `input` from the caller's fan_out, so that the gradients of all consumers of the
        # tensor are summed by ONE launch instead of pairwise ATen adds)
        avg_out = ag.avgpool(input)
        eesp_out = self.eesp(input if alias is None else alias)
        reinf = None
        if input2 is not None:
            with torch.no_grad():                          # the image carries no gradient
                pyr = input2 if isinstance(input2, ImagePyramid) else ImagePyramid(input2)
                img = pyr.at_height(avg_out.shape[2])
            if img.shape[3] != avg_out.shape[3]:
                raise RuntimeError('The size of tensor a (%d) must match the size of tensor b (%d) at non-singleton '
                                   'dimension 3' % (avg_out.shape[3], img.shape[3]))
            reinf = self.inp_reinf(img)
        # PReLU(cat[avg_out, eesp_out] + reinf) without the concatenation (autograd.DownTailFn)
        return ag.down_tail(avg_out, eesp_out, self.act.weight, reinf)

    def forward(self, input, input2=None, _alias=None):
        if _training_path():
            return self._forward_train(input, input2, _alias)
        N, _, H, W = input.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        out = torch.empty((N, self.nout, Ho, Wo), device=input.device, dtype=torch.float32)
        r = None
        if input2 is not None:
            pyr = input2 if isinstance(input2, ImagePyramid) else ImagePyramid(input2)
            with fork(0, (pyr.levels[0],)):                      # image pyramid + 3x3: independent of the EESP branch
                r = self.inp_reinf[0](pyr.at_height(Ho))
            if r.shape[3] != Wo:
                raise RuntimeError('The size of tensor a (%d) must match the size of tensor b (%d) at non-singleton '
                                   'dimension 3' % (Wo, r.shape[3]))
        cat = self.eesp.reduce_transform(input)             # projection + K2 (do not need the reinforcement)
        join(0, (r,))
        scale, shift, rw = self._epilogue_vectors(r is not None)
        ep = Epi(scale, shift, self.act.weight, reinf_r=r, reinf_w=rw)
        # the pool reads `input` completely: it also leaves the plane sums that the decoder's EfficientPWConv gate over the same
        # tensor needs (remembered per tensor; consumed -- or dropped with the tensor -- by EfficientPWConv.forward)
        _, sums = ops.avgpool3x3s2(input, ep, out=(out, 0), plane_sums=True)
        _remember_plane_sums(input, sums)
        ops.conv1x1(cat, self.eesp.conv_1x1_exp.conv.weight, self.eesp.k, ep, out=(out, self.nin))
        return out


# Plane sums of a tensor, left behind by the kernel that pooled it (DownSampler) for the gate that needs its mean
# (EfficientPWConv).  Keyed by the tensor's identity through a weak reference, so an entry can never outlive or be confused
# with another tensor; one entry per tensor, overwritten per forward.
_PLANE_SUMS = weakref.WeakKeyDictionary()


class _TensorKey(object):
    __slots__ = ('__weakref__',)


def _remember_plane_sums(t, sums):
    key = getattr(t, '_mspl_key', None)
    if key is None:
        key = _TensorKey()
        t._mspl_key = key                     # lives exactly as long as the tensor object
    _PLANE_SUMS[key] = (t.data_ptr(), t._version, sums)


def _recall_plane_sums(t):
    key = getattr(t, '_mspl_key', None)
    hit = _PLANE_SUMS.get(key) if key is not None else None
    if hit is None or hit[0] != t.data_ptr() or hit[1] != t._version:
        return None
    return hit[2]


# ------------------------------------------------------------------ decoder units
class EfficientPyrPool(nn.Module):
    def __init__(self, in_planes, proj_planes, out_planes, scales=[2.0, 1.5, 1.0, 0.5, 0.1], last_layer_br=True):
        super().__init__()
        scales = sorted(scales, reverse=True)   # the reference sorts its (shared, mutable) default in place
        self.stages = nn.ModuleList()
        self.projection_layer = DecCBR(in_planes, proj_planes, 1, 1)
        for _ in scales:
            self.stages.append(nn.Conv2d(proj_planes, proj_planes, kernel_size=3, stride=1, padding=1, bias=False,
                                         groups=proj_planes))
        self.merge_layer = nn.Sequential(
            DecBR(proj_planes * len(scales)),
            Shuffle(groups=len(scales)),
            DecCBR(proj_planes * len(scales), proj_planes, 3, 1, groups=proj_planes),
            nn.Conv2d(proj_planes, out_planes, kernel_size=1, stride=1, bias=not last_layer_br))
        if last_layer_br:
            self.br = DecBR(out_planes)
        self.last_layer_br = last_layer_br
        self.scales = scales
        self.proj_planes = proj_planes

    def _final_epilogue(self):
        conv = self.merge_layer[3]
        if not self.last_layer_br:
            return Epi(shift=conv.bias)
        bn, act = self.br.br[0], self.br.br[1]
        scale, shift = bn_fold(bn)
        return Epi(scale, shift, act.weight)

    def branch_sizes(self, height, width):
        """efficient_pyramid_pool.py:41-44: ceil(size*scale), clamped to >= 5."""
        return [(max(int(math.ceil(height * s)), 5), max(int(math.ceil(width * s)), 5)) for s in self.scales]

    def _fusable(self, sizes, height, width):
        """The fused kernel handles pure up / same / down branches with up-scales <= 4x."""
        for hs, ws in sizes:
            up, down = (hs >= height and ws >= width), (hs <= height and ws <= width)
            if not (up or down) or hs > 4 * height or ws > 4 * width:
                return False
        return len(sizes) <= 5

    def forward_unfused(self, x):
        """Branch-by-branch form (one kernel per resample / depthwise step); kept for shapes the fused kernel
        does not cover and as its cross-check."""
        N, P, height, width = x.shape
        S = len(self.scales)
        cat = torch.empty((N, P * S, height, width), device=x.device, dtype=torch.float32)
        br = self.merge_layer[0].br
        bscale, bshift = bn_fold(br[0])
        ep = Epi(bscale, bshift, br[1].weight)       # merge_layer.0 (one big BN+PReLU) fused into each branch writer
        for i, (stage, (h_s, w_s)) in enumerate(zip(self.stages, self.branch_sizes(height, width))):
            dst = (cat, i * P)
            if self.scales[i] < 1.0:
                h = ops.adaptive_avgpool(x, (h_s, w_s))
                h = ops.conv3x3(h, stage.weight, P)
                ops.bilinear(h, (height, width), ep, out=dst)
            elif self.scales[i] > 1.0:
                h = ops.bilinear(x, (h_s, w_s))
                h = ops.conv3x3(h, stage.weight, P)
                ops.adaptive_avgpool(h, (height, width), ep, out=dst)
            else:
                ops.conv3x3(x, stage.weight, P, ep=ep, out=dst)
        mcbr = self.merge_layer[2]
        return ops.conv3x3(cat, mcbr.cbr[0].weight, P, 1, shuffle_groups=S, ep=mcbr.epi())

    def forward_fused(self, x, sizes):
        N, P, height, width = x.shape
        stage_ws, down_es = [], []
        down = [i for i, (h_s, w_s) in enumerate(sizes) if self.scales[i] < 1.0 and (h_s, w_s) != (height, width)]
        prepped = {}
        if down and ops.pyr_down_prep_fits(x.shape, [sizes[i] for i in down]):        # all low-res maps in one launch
            maps = ops.pyr_down_prep(x, [sizes[i] for i in down], [self.stages[i].weight for i in down])
            prepped = dict(zip(down, maps))
        for i, (stage, (h_s, w_s)) in enumerate(zip(self.stages, sizes)):
            if i in down:                                                 # (same-size pool/resize are identities)
                stage_ws.append(None)
                down_es.append(prepped[i] if i in prepped else
                               ops.conv3x3(ops.adaptive_avgpool(x, (h_s, w_s)), stage.weight, P))
            else:
                stage_ws.append(stage.weight)
                down_es.append(None)
        br = self.merge_layer[0].br
        bscale, bshift = bn_fold(br[0])
        mcbr = self.merge_layer[2]
        return ops.pyrpool_fused(x, sizes, stage_ws, down_es, bscale, bshift, br[1].weight, mcbr.cbr[0].weight,
                                 mcbr.epi())

    def _body_train_fused(self, x, sizes):
        """Branches + merge_layer.0/1/2 as one autograd node (autograd.PyrBodyFn); frozen BatchNorms only."""
        bn0, act0 = self.merge_layer[0].br[0], self.merge_layer[0].br[1]
        mcbr = self.merge_layer[2].cbr
        conv2, bn2, act2 = mcbr[0], mcbr[1], mcbr[2]

        def fold(bn):
            scale, shift = train_fold(bn)
            return {'scale': scale, 'shift': shift, 'mean': bn.running_mean, 'inv': ag.frozen_bn_inv(bn)}
        return ag.PyrBodyFn.apply(x, sizes, fold(bn0), fold(bn2), bn0.weight, bn0.bias, act0.weight, conv2.weight, bn2.weight,
                                  bn2.bias, act2.weight, *[st.weight for st in self.stages])

    def _body_train_bn(self, x, sizes):
        """The same body with BatchNorms in train() (batch statistics: the supervised loop), autograd.PyrBodyBNFn."""
        bn0, act0 = self.merge_layer[0].br[0], self.merge_layer[0].br[1]
        mcbr = self.merge_layer[2].cbr
        conv2, bn2, act2 = mcbr[0], mcbr[1], mcbr[2]
        return ag.PyrBodyBNFn.apply(x, sizes, ag.bn_train_params(bn0, x.device), ag.bn_train_params(bn2, x.device), bn0.weight, bn0.bias,
                                    act0.weight, conv2.weight, bn2.weight, bn2.bias, act2.weight, *[st.weight for st in self.stages])

    def _forward_train(self, x):
        x = self.projection_layer(x)
        height, width = x.shape[2:]
        P = self.proj_planes
        sizes = self.branch_sizes(height, width)
        bn0, bn2 = self.merge_layer[0].br[0], self.merge_layer[2].cbr[1]
        frozen = not (bn0.training or bn2.training)
        both_train = bn0.training and bn2.training
        if _FUSED_PYR_TRAIN and (frozen or both_train) and ag.pyr_body_fits(x.shape, sizes) and (frozen or (height * width) % 4 == 0):
            out = self._body_train_fused(x, sizes) if frozen else self._body_train_bn(x, sizes)
            conv = self.merge_layer[3]
            if self.last_layer_br:
                return _conv_bn_act(out, conv, self.br.br[0], self.br.br[1].weight)
            return ag.conv_affine_prelu(out, conv.weight, 1, 1, None, conv.bias, None)
        hs = []
        xs = ag.fan_out(x, len(self.stages))               # one alias per branch: their gradients are summed by one launch
        for xb, stage, sc, (h_s, w_s) in zip(xs, self.stages, self.scales, self.branch_sizes(height, width)):
            if sc < 1.0:
                h = ag.adaptive_avgpool(xb, (h_s, w_s))
                h = ag.conv(h, stage.weight, 1, P)
                h = ag.bilinear(h, (height, width))
            elif sc > 1.0:
                h = ag.bilinear(xb, (h_s, w_s))
                h = ag.conv(h, stage.weight, 1, P)
                h = ag.adaptive_avgpool(h, (height, width))
            else:
                h = ag.conv(xb, stage.weight, 1, P)
            hs.append(h)
        out = self.merge_layer[0](torch.cat(hs, 1))        # BN+PReLU over the concatenation
        out = self.merge_layer[1](out)                      # Shuffle: view/transpose copy
        out = self.merge_layer[2](out)
        conv = self.merge_layer[3]
        if self.last_layer_br:
            return _conv_bn_act(out, conv, self.br.br[0], self.br.br[1].weight)
        if _FUSED_TRAIN_FWD:
            return ag.conv_affine_prelu(out, conv.weight, 1, 1, None, conv.bias, None)
        return ag.affine_prelu(ag.conv(out, conv.weight, 1, 1), None, conv.bias, None)

    def forward(self, x, fused=True):
        if _training_path():
            return self._forward_train(x)
        x = self.projection_layer(x)
        height, width = x.shape[2:]
        sizes = self.branch_sizes(height, width)
        # a scale<1 branch whose clamped size exceeds the map (tiny maps) or a scale>1 branch are "up" for the kernel
        ok = fused and self._fusable(sizes, height, width) and all(
            (s >= 1.0) or (hs <= height and ws <= width) for s, (hs, ws) in zip(self.scales, sizes))
        m = self.forward_fused(x, sizes) if ok else self.forward_unfused(x)
        return ops.conv1x1(m, self.merge_layer[3].weight, 1, self._final_epilogue())


class EfficientPWConv(nn.Module):
    def __init__(self, nin, nout):
        super().__init__()
        self.wt_layer = nn.Sequential(
            nn.AdaptiveAvgPool2d(output_size=1),
            nn.Conv2d(nin, nout, kernel_size=1, stride=1, padding=0, groups=1, bias=False),
            nn.Sigmoid())
        self.groups = math.gcd(nin, nout)
        self.expansion_layer = DecCBR(nin, nout, kSize=3, stride=1, groups=self.groups)
        self.out_size = nout
        self.in_size = nin

    def forward(self, x, _alias=None):
        if _training_path():
            gate = ag.gap_gate(x, self.wt_layer[1].weight)
            return ag.channel_scale(self.expansion_layer(x if _alias is None else _alias), gate)
        sums = _recall_plane_sums(x)
        if sums is not None:
            gate = ops.gate_from_sums(sums, self.wt_layer[1].weight, x.shape[2] * x.shape[3])
        else:
            gate = ops.gap_gate(x, self.wt_layer[1].weight)
        return ops.conv3x3(x, self.expansion_layer.cbr[0].weight, self.groups, ep=self.expansion_layer.epi(gate=gate))

    def __repr__(self):
        return '%s(in_channels=%d, out_channels=%d)' % (self.__class__.__name__, self.in_size, self.out_size)


def decoder_merge(pw_out, bu_lowres, br_seq):
    """bu_br(pw_out + upsample2(bu)): model/segmentation/espdnet_ue.py:276-280 in one kernel
    (bilinear x2, align_corners=True, pre-add, folded BN, PReLU)."""
    size = (bu_lowres.shape[2] * 2, bu_lowres.shape[3] * 2)
    if tuple(pw_out.shape[2:]) != size:
        raise RuntimeError('The size of tensor a (%d) must match the size of tensor b (%d) at non-singleton dimension 2'
                           % (pw_out.shape[2], size[0]))
    if _training_path():
        return _bn_act(ag.bilinear(bu_lowres, size), br_seq[0], br_seq[1].weight, pre_add=pw_out)
    scale, shift = bn_fold(br_seq[0])
    return ops.bilinear(bu_lowres, size, Epi(scale, shift, br_seq[1].weight, pre_add=pw_out))
