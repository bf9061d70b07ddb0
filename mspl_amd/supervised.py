"""The supervised source-model loop on the HIP path: what train_segmentation.py needs beyond the frozen-BN
uest step.

Reference surface mirrored (paths relative to the reference root):
  utilities/train_eval_seg.py:164-225   train_seg_ue: model.train() (batch-statistics BatchNorm), loss =
                                        criterion(out + 0.5*aux, target).mean() [+ add_criterion(inputs, out)*weight],
                                        flooding `(loss - b).abs() + b` with b = 0.015 (:221), zero_grad/backward/step
  train_segmentation.py:241-253         torch.optim.SGD over 2-3 learning-rate groups (base net lr, segmentation head and
                                        depth encoder lr*lr_mult), momentum, weight_decay
  train_segmentation.py:353-362         per-epoch learning rates written into optimizer.param_groups[i]['lr']
  utilities/lr_scheduler.py             -> mspl_amd/lr_scheduler.py

Batch-statistics BatchNorm runs through mspl_amd.autograd.BNBatchStatsFn (statistics kernel + the existing affine/PReLU
kernels); `FlatSGD` keeps parameters, gradients and momentum buffers of all groups in three flat fp32 buffers laid out group
after group, so a step is one kernel per group and the multi-GPU exchange one all-reduce.
"""
import os

import torch

from . import autograd as ag
from . import dist as mdist
from . import layers
from ._native import check, lib
from .ops import _p, _stream

FLOOD_LEVEL = 0.015          # utilities/train_eval_seg.py:178


class FlatSGD:
    """torch.optim.SGD semantics (momentum, L2 weight_decay, dampening 0, no Nesterov; parameters whose gradient is None
    are skipped) over torch-style parameter groups `[{'params': iterable, 'lr': float}, ...]`.  Build it AFTER the first
    backward, like FlatAdam.  A parameter listed in two groups raises, as torch.optim does."""

    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0):
        groups = list(params)
        if groups and not isinstance(groups[0], dict):
            groups = [{'params': groups}]
        self.param_groups, seen, flat = [], set(), []
        for g in groups:
            ps = [p for p in g['params'] if p.requires_grad and p.grad is not None]
            for p in ps:
                if id(p) in seen:
                    raise ValueError('some parameters appear in more than one parameter group')
                seen.add(id(p))
            self.param_groups.append({'params': ps, 'lr': g.get('lr', lr), 'momentum': g.get('momentum', momentum),
                                      'weight_decay': g.get('weight_decay', weight_decay)})
            flat += ps
        if not flat:
            raise RuntimeError('FlatSGD: run one backward before constructing the optimizer (no parameter has a gradient)')
        self.bucket = mdist.GradBucket(flat, with_params=True)            # the flat layout shared with FlatAdam and the all-reduce
        self.flat_p, self.flat_g = self.bucket.flat_p, self.bucket.flat
        self.buf = torch.zeros_like(self.flat_p)
        first = 0
        for g in self.param_groups:
            g['_lo'], g['_hi'] = self.bucket.span(first, len(g['params']))
            first += len(g['params'])
        self.params = flat
        self.step_count = 0

    def zero_grad(self):
        self.flat_g.zero_()

    def all_reduce_grads(self):
        self.bucket.all_reduce()

    def step(self):
        first = 1 if self.step_count == 0 else 0
        self.step_count += 1
        for g in self.param_groups:
            lo, hi = g['_lo'], g['_hi']
            if hi > lo:
                check(lib.mspl_sgd_step(_p(self.flat_p[lo:hi]), _p(self.flat_g[lo:hi]), _p(self.buf[lo:hi]), hi - lo, float(g['lr']),
                                        float(g['momentum']), float(g['weight_decay']), first, _stream()))
        layers.bump_param_epoch()


def segmentation_param_groups(model, lr, lr_mult, use_depth=False):
    """train_segmentation.py:244-250."""
    groups = [{'params': model.get_basenet_params(), 'lr': lr},
              {'params': model.get_segment_params(), 'lr': lr * lr_mult}]
    if use_depth:
        groups.append({'params': model.get_depth_encoder_params(), 'lr': lr * lr_mult})
    return groups


def set_epoch_learning_rates(optimizer, lr_base, lr_mult, use_depth=False):
    """train_segmentation.py:356-362 (including its depth-group rule: group 2 gets lr_base, not lr_base*lr_mult)."""
    optimizer.param_groups[0]['lr'] = lr_base
    if len(optimizer.param_groups) > 1:
        optimizer.param_groups[1]['lr'] = lr_base * lr_mult
    if use_depth:
        optimizer.param_groups[2]['lr'] = lr_base
    return lr_base, lr_base * lr_mult


def flood(loss, b=FLOOD_LEVEL):
    """utilities/train_eval_seg.py:221."""
    return (loss - b).abs() + b


_TWO_HEAD_SUM = os.environ.get('MSPL_TWO_HEAD_SUM', '1') != '0'


def two_head_outputs(model, inputs, depth=None):
    """`outputs[0] + 0.5 * outputs[1]` of the two-head model (train_eval_seg.py:185-187): through the low-resolution heads and
    autograd.TwoHeadSumFn when the model offers them (two launches instead of two up-samplings + a multiply + an add on full-size
    logits), the reference's three steps otherwise."""
    if _TWO_HEAD_SUM and hasattr(model, 'forward_lowres') and getattr(model, 'aux_layer', -1) >= 0:
        main, aux = model.forward_lowres(inputs, depth) if depth is not None else model.forward_lowres(inputs)
        if aux is not None and main.shape[1] == aux.shape[1]:
            return ag.two_head_sum(main, aux, inputs.shape[2:])
        out = (ag.bilinear(main, tuple(inputs.shape[2:])), ag.bilinear(aux, tuple(inputs.shape[2:])))
    else:
        out = model(inputs, depth) if depth is not None else model(inputs)
    return out[0] + 0.5 * out[1]


def train_seg_ue_step(model, inputs, target, criterion, optimizer=None, depth=None, add_criterion=None, weight=1.0,
                      lr=0.009, lr_mult=10.0, momentum=0.9, weight_decay=4e-5, b=FLOOD_LEVEL):
    """One iteration of train_seg_ue (utilities/train_eval_seg.py:179-225) for a two-head model in train() mode.
    Returns (flooded loss, (main + 0.5*aux) logits detached -- what the reference hands to MIOU --, optimizer).  Pass
    optimizer=None on the first call: it is built after the first backward from segmentation_param_groups."""
    if optimizer is not None:
        optimizer.zero_grad()
    tr = getattr(optimizer, 'transposer', None)
    with torch.enable_grad(), ag.grad_sinks(), (tr.active() if tr is not None else ag.collect_conv_weights()) as got:
        layers.prefold_frozen_bn(model)
        outputs = two_head_outputs(model, inputs, depth)
        loss = criterion(outputs, target).mean()
        if add_criterion is not None:
            loss = loss + add_criterion(inputs, outputs) * weight
        loss = flood(loss, b)
        loss.backward()
    if optimizer is None:
        optimizer = FlatSGD(segmentation_param_groups(model, lr, lr_mult, depth is not None), lr=lr * lr_mult,
                            momentum=momentum, weight_decay=weight_decay)
        optimizer.transposer = ag.WeightTransposer(got)      # (after FlatSGD: the parameters now live in its flat buffer)
    optimizer.all_reduce_grads()
    optimizer.step()
    return loss.detach(), outputs.detach(), optimizer


class GraphedSupervisedStep:
    """train_seg_ue_step with zero_grad + forward + loss + backward replayed as ONE hipGraph (the iteration is ~900 launches);
    the gradient all-reduce and the SGD kernels (their learning rates change per epoch) stay outside.  The first call runs one
    eager iteration (reveals the gradient-bearing parameters, builds FlatSGD, consumes SGD's first-step rule) and captures;
    shapes are fixed at construction.  BatchNorm running statistics and num_batches_tracked advance inside the graph."""

    def __init__(self, model, inputs, target, criterion, depth=None, lr=0.009, lr_mult=10.0, momentum=0.9, weight_decay=4e-5,
                 b=FLOOD_LEVEL):
        self.model, self.criterion, self.b = model, criterion, b
        self.inputs = inputs.detach().clone()
        self.target = target.detach().clone()
        self.depth = None if depth is None else depth.detach().clone()
        _, _, self.optimizer = train_seg_ue_step(model, self.inputs, self.target, criterion, None, self.depth, None, 1.0, lr, lr_mult,
                                                 momentum, weight_decay, b)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.optimizer.zero_grad()
            with torch.enable_grad(), ag.grad_sinks(), self.optimizer.transposer.active():
                layers.prefold_frozen_bn(model)
                self.outputs = two_head_outputs(model, self.inputs, self.depth)
                self.loss = flood(criterion(self.outputs, self.target).mean(), b)
                self.loss.backward()
        self._finish()                                  # the capture did not execute: run the iteration it recorded

    def _finish(self):
        self.graph.replay()
        self.optimizer.all_reduce_grads()
        self.optimizer.step()
        return self.loss.detach(), self.outputs.detach()

    def __call__(self, inputs, target, depth=None):
        self.inputs.copy_(inputs)
        self.target.copy_(target)
        if self.depth is not None:
            self.depth.copy_(depth)
        return self._finish()
