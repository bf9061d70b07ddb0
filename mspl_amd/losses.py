"""Drop-in loss classes of loss_fns/segmentation_loss.py on the HIP path (forward and backward kernels in csrc/losses.hip).

    PixelwiseKLD()(d1, d2)                                            :177-189
    UncertaintyWeightedSegmentationLoss(num_classes, class_weights=None, ignore_idx=None, device='cuda')(pred, target, u_weight)   :146-175
    SegmentationLoss(n_classes, loss_type='ce', device, ignore_idx, class_weights)(inputs, target)   :11-52
    NIDLoss(image_bin=16, label_bin=4, bw_camera=0.005, bw_label=0.001)(camera, label)   :54-121

Both weight arguments also accept the alias `class_wts` the reference's callers use (uest_seg_multi_os.py:509).  The uest
training step composes the first two as criterion(pred + 0.5*aux, labels, kld) * 20 + kld.mean(); mspl_amd.training.uest_loss
is that composition as ONE fused kernel (K11) and is what train_step uses.  SelectiveBCE / SoftArgMax as a public class / the 'bce' loss type
are out of scope: 'bce' raises here, the two classes resolve to the reference's own through the overlay of
mspl_amd.dropin.
"""
import torch
from torch import nn

from ._native import check, lib
from .ops import _p, _stream


def _logits(t, name):
    if not t.is_cuda:
        raise RuntimeError('mspl_amd: %s must be a CUDA tensor (there is no CPU path)' % name)
    if t.dtype != torch.float32 or t.dim() != 4:
        raise RuntimeError('mspl_amd: %s must be float32 (N,C,H,W), got %s %s' % (name, t.dtype, tuple(t.shape)))
    return t.contiguous()


class _KLDFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, d1, d2):
        d1, d2 = _logits(d1, 'dist1'), _logits(d2, 'dist2')
        if d1.shape != d2.shape:
            raise RuntimeError('mspl_amd: PixelwiseKLD shapes differ: %s vs %s' % (tuple(d1.shape), tuple(d2.shape)))
        N, C, H, W = d1.shape
        out = torch.empty((N, H, W), device=d1.device, dtype=torch.float32)
        check(lib.mspl_pixelwise_kld_fwd(_p(d1), _p(d2), N, C, H * W, _p(out), _stream()))
        ctx.save_for_backward(d1, d2)
        return out

    @staticmethod
    def backward(ctx, g):
        d1, d2 = ctx.saved_tensors
        N, C, H, W = d1.shape
        g1 = torch.empty_like(d1) if ctx.needs_input_grad[0] else None
        g2 = torch.empty_like(d2) if ctx.needs_input_grad[1] else None
        check(lib.mspl_pixelwise_kld_bwd(_p(d1), _p(d2), _p(g.contiguous()), N, C, H * W, _p(g1), _p(g2), _stream()))
        return g1, g2


class _WeightedCEFn(torch.autograd.Function):
    """sums -> loss; mode 'all': / (N*H*W) (UW loss), mode 'weights': / sum of valid weights (nn.CrossEntropyLoss)."""

    @staticmethod
    def forward(ctx, pred, target, u, cw, ignore, mode):
        pred = _logits(pred, 'pred')
        N, C, H, W = pred.shape
        target = target.to(torch.int64).contiguous()
        if not target.is_cuda or target.numel() != N * H * W:
            raise RuntimeError('mspl_amd: target must be a CUDA tensor of %d labels, got %s' % (N * H * W, tuple(target.shape)))
        if u is not None:
            u = u.to(torch.float32).contiguous()
            if u.numel() != N * H * W:
                raise RuntimeError('mspl_amd: u_weight must hold %d values, got %s' % (N * H * W, tuple(u.shape)))
        sums = torch.zeros(2, device=pred.device, dtype=torch.float32)
        check(lib.mspl_weighted_ce_fwd(_p(pred), _p(target), _p(u), _p(cw), ignore, N, C, H * W, _p(sums), _stream()))
        ctx.save_for_backward(pred, target, u, cw, sums)
        ctx.ignore, ctx.mode = ignore, mode
        return sums[0] / (N * H * W) if mode == 'all' else sums[0] / sums[1]

    @staticmethod
    def backward(ctx, g):
        pred, target, u, cw, sums = ctx.saved_tensors
        N, C, H, W = pred.shape
        gp = torch.empty_like(pred) if ctx.needs_input_grad[0] else None
        gu = torch.empty((N, H, W), device=pred.device, dtype=torch.float32) if (u is not None and ctx.needs_input_grad[2]) else None
        if gp is None and gu is None:
            return None, None, None, None, None, None
        g = g.to(torch.float32).reshape(1).contiguous()
        den = sums[1:] if ctx.mode == 'weights' else None
        check(lib.mspl_weighted_ce_bwd(_p(pred), _p(target), _p(u), _p(cw), ctx.ignore, N, C, H * W, _p(g), _p(den), _p(gp), _p(gu),
                                       _stream()))
        return gp, None, gu, None, None, None


class PixelwiseKLD(nn.Module):
    def forward(self, dist1, dist2):
        return _KLDFn.apply(dist1, dist2)


class UncertaintyWeightedSegmentationLoss(nn.Module):
    def __init__(self, num_classes, class_weights=None, ignore_idx=None, device='cuda', class_wts=None):
        super().__init__()
        if class_weights is None:
            class_weights = class_wts
        self.num_classes = num_classes
        self.class_weights = (class_weights if class_weights is not None else torch.ones(num_classes)).to(device, torch.float32)
        self.ignore_idx = ignore_idx
        if ignore_idx is not None:
            self.class_weights[ignore_idx] = 0.0          # in place, like the reference (:152-153)

    def forward(self, pred, target, u_weight, epsilon=1e-12):
        cw = self.class_weights.to(pred.device).contiguous()
        # the ignore class has weight 0 and the mean runs over all pixels, so no pixel is dropped by index here
        return _WeightedCEFn.apply(pred, target, u_weight, cw, -1, 'all')


class SegmentationLoss(nn.Module):
    def __init__(self, n_classes=21, loss_type='ce', device='cuda', ignore_idx=255, class_weights=None, class_wts=None):
        super().__init__()
        if loss_type != 'ce':
            raise RuntimeError("mspl_amd: SegmentationLoss loss_type %r is not on the HIP path (only 'ce')" % (loss_type,))
        if class_weights is None:
            class_weights = class_wts
        self.loss_type, self.n_classes, self.device, self.ignore_idx = loss_type, n_classes, device, ignore_idx
        self.class_wts = None if class_weights is None else class_weights.to(device, torch.float32)

    def _one(self, inputs, target):
        cw = None if self.class_wts is None else self.class_wts.to(inputs.device).contiguous()
        return _WeightedCEFn.apply(inputs, target, None, cw, int(self.ignore_idx), 'weights')

    def forward(self, inputs, target):
        if isinstance(inputs, tuple):
            assert len(inputs) == 2
            return self._one(inputs[0], target) + self._one(inputs[1], target)
        return self._one(inputs, target)


class _NIDHistFn(torch.autograd.Function):
    """(joint (K,C), p_c (K), p_l (C)) of NIDLoss.get_probabilities applied to the soft-arg-max of the label logits."""

    @staticmethod
    def forward(ctx, camera, label, K, bw_c, bw_l):
        camera, label = camera.contiguous().float(), label.contiguous().float()
        B, C, H, W = label.shape
        if camera.shape != (B, 3, H, W):
            raise RuntimeError('mspl_amd: NIDLoss expects camera (B,3,H,W) matching label (B,C,H,W), got %s / %s'
                               % (tuple(camera.shape), tuple(label.shape)))
        Cl = min(C, K)
        n_ws = lib.mspl_nid_workspace_floats(C, H, W, K)
        if n_ws < 0:
            check(int(n_ws))
        ws = torch.empty(n_ws, dtype=torch.float32, device=label.device)
        out = torch.empty(K * Cl + K + Cl, dtype=torch.float32, device=label.device)
        check(lib.mspl_nid_hist_fwd(_p(camera), _p(label), B, C, H, W, K, bw_c, bw_l, _p(ws), _p(out), _stream()))
        ctx.save_for_backward(camera, label)
        ctx.cfg = (K, bw_c, bw_l, Cl)
        joint = torch.zeros(K, C, dtype=torch.float32, device=label.device)
        joint[:, :Cl] = out[:K * Cl].view(K, Cl)
        p_l = torch.zeros(C, dtype=torch.float32, device=label.device)
        p_l[:Cl] = out[K * Cl + K:]
        return joint, out[K * Cl:K * Cl + K].clone(), p_l

    @staticmethod
    def backward(ctx, gj, gpc, gpl):
        camera, label = ctx.saved_tensors
        K, bw_c, bw_l, Cl = ctx.cfg
        B, C, H, W = label.shape
        inv = 1.0 / (B * H * W)
        gj = (gj[:, :Cl] * inv).contiguous()
        gpl = (gpl[:Cl] * inv).contiguous()
        glabel = torch.zeros_like(label) if Cl < C else torch.empty_like(label)
        check(lib.mspl_nid_hist_bwd(_p(camera), _p(label), B, C, H, W, K, bw_c, bw_l, _p(gj), _p(gpl), _p(glabel), _stream()))
        return None, glabel, None, None, None


class NIDLoss(nn.Module):
    """loss_fns/segmentation_loss.py:54-121: (NID(camera grey levels, soft-arg-max labels) - 0.95) * 20.  The soft histograms
    run in two HIP kernels (nid.hip); the K x C entropy arithmetic below is the reference's, on tiny device tensors.  The
    reference's hard-coded `.to('cuda')` calls have no counterpart (tensors stay on the inputs' device)."""

    def __init__(self, image_bin=16, label_bin=4, bw_camera=0.005, bw_label=0.001):
        super().__init__()
        self.K, self.C = image_bin, label_bin
        self.bw_camera, self.bw_label = bw_camera, bw_label

    def nid(self, p_cl, p_c, p_l, eps=1e-7):
        I = torch.sum(p_cl * (torch.log(p_cl + eps) - torch.log(torch.mm(p_c, torch.t(p_l)) + eps)))
        H = -torch.sum(p_cl * torch.log(p_cl + eps))
        return 1 - I / H

    def forward(self, camera, label):
        if label.shape[1] < self.C:
            raise RuntimeError('mspl_amd: NIDLoss(label_bin=%d) got logits with %d classes' % (self.C, label.shape[1]))
        p_cl, p_c, p_l = _NIDHistFn.apply(camera, label, self.K, float(self.bw_camera), float(self.bw_label))
        # label_bin rows of the label histogram (:91-93 fills k < self.C); classes beyond label_bin never get a bin
        p_cl, p_l = p_cl[:, :self.C], p_l[:self.C]
        p_cl = p_cl / p_cl.sum()
        p_c = p_c / p_c.sum()
        p_l = p_l / p_l.sum()
        nid = self.nid(p_cl, p_c.reshape(-1, 1), p_l.reshape(-1, 1))
        return (nid - 0.95) * 20
