"""One uest self-training step on the HIP path (uest_seg_multi_os.py:958-1089, train()).

    (pred, aux) = model(images)                               # frozen BatchNorm (model.eval(), Appendix B-3)
    kld  = PixelwiseKLD()(pred, aux)                          # NOT detached
    loss = criterion(pred + 0.5*aux, labels, kld) * 20 + kld.mean()
    loss.backward(); optimizer.step()                         # Adam over model.parameters(), weight decay as L2

`uest_loss` is K11 (one fused forward+backward kernel); `FlatAdam` keeps every gradient-bearing parameter, its
gradient and the two Adam moments in four flat fp32 buffers, so the optimizer is ONE kernel and the multi-GPU
gradient exchange ONE all-reduce (mspl_amd.dist).  Parameters whose gradient stays None (depth encoder, fusion gates,
module_act of strided EESPs: 230 of 570 tensors) are left out, exactly like torch.optim.Adam skips them.
"""
import ctypes
import os

import torch

from . import autograd as ag
from . import dist as mdist
from . import layers
from ._native import check, lib
from .ops import _p, _stream


def _device_class_weights(class_weights, device, ignore_idx):
    cw = class_weights.detach().to(device, torch.float32).clone()
    if ignore_idx is not None:
        cw[ignore_idx] = 0.0
    return cw


def uest_loss(pred, aux, labels, class_weights, ignore_idx=None, ce_scale=20.0):
    """criterion(pred + 0.5*aux, labels, kld)*ce_scale + kld.mean() with kld = PixelwiseKLD(pred, aux).
    class_weights: tensor of num_classes weights; class_weights[ignore_idx] is treated as 0 (the reference zeroes it
    in place at construction, loss_fns/segmentation_loss.py:152-153)."""
    return ag.uw_loss(pred, aux, labels, _device_class_weights(class_weights, pred.device, ignore_idx), ce_scale)


_LOSS_AT_HEADS = os.environ.get('MSPL_LOSS_HEADS', '1') != '0'      # A/B aid: 0 = up-sample both heads, then the loss


def forward_loss(model, images, labels, cw, ce_scale=20.0, out_scale=1.0, root=False):
    """model forward + uest loss (uest_seg_multi_os.py:1010-1023) on device class weights `cw`.  A model that exposes its decoder
    outputs (`forward_lowres`: main at H/2, auxiliary at H/4) gets the loss taken at head resolution -- the up-sampling of
    espdnet_ue.py:301-302 happens inside the loss kernel (ag.uw_loss_heads), same value and gradients; any other model is called as
    the reference calls it and the loss takes its two full-size outputs."""
    lowres = getattr(model, 'forward_lowres', None) if _LOSS_AT_HEADS else None
    if lowres is not None:
        main, aux = lowres(images)
        if aux is not None and ag.uw_loss_heads_supported(main.shape[1]):
            return ag.uw_loss_heads(main, aux, labels, cw, ce_scale, out_scale, root)
        size = tuple(images.shape[2:])
        pred, aux = ag.bilinear(main, size), ag.bilinear(aux, size)
    else:
        pred, aux = model(images)
    return ag.uw_loss(pred, aux, labels, cw, ce_scale, out_scale, root)


class FlatAdam:
    """torch.optim.Adam semantics (lr, betas, eps, L2 weight_decay, bias correction) on flat buffers.

    Build it AFTER the first backward: the parameters that received a gradient define the flat layout.  Parameter
    .data and .grad are re-pointed to views of the flat buffers (values preserved)."""

    def __init__(self, params, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        try:
            self.bucket = mdist.GradBucket(params, with_params=True)      # the flat layout shared with FlatSGD and the all-reduce
        except RuntimeError:
            raise RuntimeError('FlatAdam: run one backward before constructing the optimizer (no parameter has a gradient)')
        self.params = self.bucket.params
        self.flat_p, self.flat_g = self.bucket.flat_p, self.bucket.flat
        self.m = torch.zeros_like(self.flat_p)
        self.v = torch.zeros_like(self.flat_p)
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        # torch.optim spelling: the reference's adjust_learning_rate writes optimizer.param_groups[0]['lr'] (:1325-1328)
        self.param_groups = [{'lr': lr, 'params': self.params}]
        self.step_count = 0

    @property
    def lr(self):
        return self.param_groups[0]['lr']

    @lr.setter
    def lr(self, value):
        self.param_groups[0]['lr'] = value

    def zero_grad(self):
        self.flat_g.zero_()

    def all_reduce_grads(self):
        """Average gradients over the ranks: one collective on the flat bucket (RCCL over xGMI on GPUs)."""
        self.bucket.all_reduce()

    def step(self):
        self.step_count += 1
        check(lib.mspl_adam_step(_p(self.flat_p), _p(self.flat_g), _p(self.m), _p(self.v), self.flat_p.numel(),
                                 float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps),
                                 float(self.weight_decay), self.step_count, _stream()))
        layers.bump_param_epoch()       # the kernel wrote through raw pointers: invalidate folded-BN / packed-weight caches


def lr_poly(base_lr, iter_n, max_iter, power):
    """uest_seg_multi_os.py:1317-1321."""
    return base_lr * ((1 - float(iter_n) / max_iter) ** power)


def adjust_learning_rate(optimizer, i_iter, tot_iter, base_lr, power=0.0):
    """uest_seg_multi_os.py:1325-1328 (base_lr / power are script globals there: args.learning_rate, args.power)."""
    lr = lr_poly(base_lr, i_iter, tot_iter, power)
    optimizer.param_groups[0]['lr'] = lr
    return lr


def train_step(model, images, labels, class_weights, optimizer=None, ignore_idx=None, lr=5e-4, weight_decay=5e-4,
               ce_scale=20.0):
    """One optimisation step; returns (loss tensor, optimizer).  Pass optimizer=None on the first call: it is built
    after the first backward (which reveals the gradient-bearing parameters)."""
    if optimizer is not None:
        optimizer.zero_grad()
    tr = getattr(optimizer, 'transposer', None)
    with torch.enable_grad(), ag.grad_sinks(), (tr.active() if tr is not None else ag.collect_conv_weights()) as got:
        layers.prefold_frozen_bn(model)
        loss = forward_loss(model, images, labels, _device_class_weights(class_weights, images.device, ignore_idx), ce_scale)
        loss.backward()
    if optimizer is None:
        optimizer = FlatAdam(model.parameters(), lr=lr, weight_decay=weight_decay)
        # (built after FlatAdam: the parameters now live in its flat buffer)
        optimizer.transposer = ag.WeightTransposer(got)
    optimizer.all_reduce_grads()
    optimizer.step()
    return loss.detach(), optimizer


SETTLE_THRESHOLD = 0.55      # four lanes overlapping take 0.44-0.45 of the same graphs back to back; two lanes sharing a queue ~0.7


def settle_seating(serial_ms, first_ms, measure, new_candidate, attempts=3, threshold=SETTLE_THRESHOLD):
    """Decision logic of `GraphedTrainStep._settle_streams`, free of GPU calls.  `first_ms` is the time of all lanes on seating 0,
    `serial_ms` of the same graphs back to back on one stream; `new_candidate()` makes another seating and returns its index,
    `measure(index)` times it.  Returns {'seating', 'lanes_ms', 'settled', 'attempts'}: the first seating under `threshold` x serial,
    else -- after `attempts` candidates -- the fastest one seen with settled = False (the caller warns and reports it)."""
    best_ms, best = first_ms, 0
    used = 0
    while best_ms > threshold * serial_ms and used < attempts:
        k = new_candidate()
        used += 1
        ms = measure(k)
        if ms < best_ms:
            best_ms, best = ms, k
    return {'seating': best, 'lanes_ms': best_ms, 'settled': bool(best_ms <= threshold * serial_ms), 'attempts': used}


class GraphedTrainStep:
    """train_step with zero_grad + forward + loss + backward replayed as hipGraphs (the step is ~1400 launches of 5-100 us; eager
    issue from Python is slower than the GPU executes them).  The gradient all-reduce and the Adam kernel stay outside the graphs,
    so the same object serves N = 1 and N > 1.

    lanes = 1: ONE graph.  lanes = L > 1: the batch is cut into L micro-batches that run as L graphs on L streams AT THE SAME TIME
    (the kernels of a batch-16 step are too small to fill the chip; graphs on separate streams overlap, branches inside one
    graph did not).  Same step: BatchNorm is frozen, the loss is a plain mean over the pixels of the batch (K11: 1 / (N*H*W)), so
    each lane back-propagates loss_lane / L and every parameter-gradient kernel adds into the shared flat gradient buffer with
    atomics (autograd.grad_sinks); only the floating-point summation order differs.  A parameter whose gradient would go through
    autograd's (non-atomic) AccumulateGrad in a lane is refused at construction.

    The first call runs one eager step (it reveals the gradient-bearing parameters and builds FlatAdam) and captures;
    later calls copy the batch into static buffers and replay.  Shapes are fixed at construction."""

    def __init__(self, model, images, labels, class_weights, ignore_idx=None, lr=5e-4, weight_decay=5e-4, ce_scale=20.0, lanes=1):
        self.model = model
        self.images = images.detach().clone()
        self.labels = labels.detach().to(torch.int64).clone()
        self.cw = _device_class_weights(class_weights, images.device, ignore_idx)
        self.ce_scale = ce_scale
        B = self.images.shape[0]
        self.lanes = lanes if (lanes > 1 and B % lanes == 0) else 1
        _, self.optimizer = train_step(model, self.images, self.labels, class_weights, None, ignore_idx, lr, weight_decay,
                                       ce_scale)
        torch.cuda.synchronize()
        tr = self.optimizer.transposer
        if self.lanes == 1:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.optimizer.zero_grad()
                with torch.enable_grad(), ag.grad_sinks(), tr.active():
                    layers.prefold_frozen_bn(model)
                    self.loss = forward_loss(model, self.images, self.labels, self.cw, ce_scale, root=True)
                    self.loss.backward()
        else:
            # what every lane needs first: zeroed gradients, this step's transposed weights and folded BatchNorms (their tensors live
            # in this graph's pool, the lanes' graphs read them)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.optimizer.zero_grad()
                tr.run()
                with torch.no_grad():
                    layers.prefold_frozen_bn(model)
            b = B // self.lanes
            # streams that really run side by side (HIP multiplexes streams onto a few hardware queues; two lanes on one queue run one
            # after the other, and four serialised micro-batches are SLOWER than the one-graph step: 12.2 vs 9.0 ms, both seen in one
            # process with plain new streams, depending on how many streams existed before)
            from .uest import _concurrent_streams
            self.streams = _concurrent_streams(self.lanes, self.images.device)
            self.lane_graphs, self.lane_losses = [], []
            stray = []
            # (a tensor hook sees None when the op accumulated into the sink itself, a tensor when autograd is about to add one)
            hooks = [p.register_hook(lambda g_: stray.append(1) if g_ is not None else None) for p in self.optimizer.params]
            try:
                for i, st in enumerate(self.streams):
                    st.wait_stream(torch.cuda.current_stream())
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=st):
                        with torch.enable_grad(), ag.grad_sinks(), tr.active(refresh=False):
                            loss = forward_loss(model, self.images[i * b:(i + 1) * b], self.labels[i * b:(i + 1) * b], self.cw, ce_scale,
                                                out_scale=1.0 / self.lanes, root=True)
                            loss.backward()
                    self.lane_graphs.append(g)
                    self.lane_losses.append(loss)
            finally:
                for h in hooks:
                    h.remove()
            if stray:
                raise RuntimeError('GraphedTrainStep(lanes=%d): %d parameter gradients went through AccumulateGrad instead of an '
                                   'atomic gradient sink; concurrent lanes would race on them' % (self.lanes, len(stray)))
        if self.lanes > 1:
            # the capture did not execute: the lanes read this step's transposed weights and BatchNorm folds from self.graph's pool,
            # so it runs once before the lanes are timed (they would otherwise run on uninitialised memory)
            self.graph.replay()
            self._settle_streams()
        self._finish()      # run the step the capture recorded

    def _lanes_ms(self, streams, reps=2):
        """Wall time of one replay of every lane graph, lane i on streams[i % len(streams)] (the gradients they add up are thrown away by
        the next step's zero fill)."""
        import time
        dev = self.images.device
        best = float('inf')
        for _ in range(reps):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            cur = torch.cuda.current_stream()
            ev = cur.record_event()
            for i, g in enumerate(self.lane_graphs):
                st = streams[i % len(streams)]
                st.wait_event(ev)
                with torch.cuda.stream(st):
                    g.replay()
            torch.cuda.synchronize(dev)
            best = min(best, (time.perf_counter() - t0) * 1e3)
        return best

    def _settle_streams(self, attempts=3):
        """The lanes must really overlap: measured on the captured graphs themselves.  `_concurrent_streams` picks streams with a spin
        kernel, and that has been seen to let two lanes share a hardware queue now and then (a 4-lane step of 10.6-11 ms instead of
        6.7: twice in ~25 processes) -- a replay may run on any stream, so the lanes are re-seated until all lanes together take
        clearly less than the same graphs back to back on one stream; the best seating seen is kept (`settle_seating` is the decision,
        a pure function of the timings: tests/test_host.py drives it with injected ones).  The lanes are timed on REAL data: the
        caller has replayed `self.graph` (zeroed gradients, this step's transposed weights and BatchNorm folds) before."""
        from .uest import _concurrent_streams
        self._lanes_ms(self.streams, 1)                                  # warm-up
        serial = self._lanes_ms(self.streams[:1])
        pool = [self.streams]           # candidate stream sets: bounded, and the losers are dropped with this list

        def candidate():
            pool.append(_concurrent_streams(self.lanes, self.images.device))
            return len(pool) - 1

        # (four lanes that overlap take 0.44-0.45 of the serial time, two sharing a queue ~0.7; two lanes that overlap ~0.57)
        d = settle_seating(serial, self._lanes_ms(self.streams), lambda k: self._lanes_ms(pool[k]), candidate, attempts,
                           threshold=min(0.9, 1.0 / self.lanes + 0.3))
        self.streams = pool[d['seating']]
        self.lane_overlap = {'lanes_ms': round(d['lanes_ms'], 3), 'serial_ms': round(serial, 3), 'settled': d['settled'],
                             'attempts': d['attempts'], 'lanes': self.lanes}
        if not d['settled']:
            import warnings
            warnings.warn('GraphedTrainStep(lanes=%d): the lane graphs did not overlap after %d re-seatings (%.2f ms against %.2f ms '
                          'back to back); the step runs on the best seating seen' % (self.lanes, d['attempts'], d['lanes_ms'], serial))

    def _finish(self):
        self.graph.replay()
        if self.lanes > 1:
            cur = torch.cuda.current_stream()
            ev = cur.record_event()
            for st, g in zip(self.streams, self.lane_graphs):
                st.wait_event(ev)
                with torch.cuda.stream(st):
                    g.replay()
            for st in self.streams:
                cur.wait_stream(st)
            self.loss = torch.stack([l.detach() for l in self.lane_losses]).sum()
        self.optimizer.all_reduce_grads()
        self.optimizer.step()
        return self.loss.detach()

    def __call__(self, images, labels):
        self.images.copy_(images)
        self.labels.copy_(labels)
        return self._finish()
