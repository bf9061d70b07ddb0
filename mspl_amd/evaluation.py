"""The evaluation step of the loop on the HIP path.

Reference functions mirrored (paths relative to the reference root):
  val_seg_ue          utilities/train_eval_seg.py:249-324      model(x) -> out + 0.5*aux -> criterion(...).mean() -> MIOU.get_iou
  test() (loop body)  uest_seg_multi_os.py:1150-1200           model(x) -> criterion(pred) -> MIOU.get_iou(pred)   (main head alone)

The reference writes both heads at full resolution (2*C*H*W floats per image), adds them with an ATen kernel, runs the loss, then
copies prediction and labels to the host for three torch.histc calls per batch.  Here `EvalPass` feeds the low-resolution decoder
outputs (`model.forward_lowres`) to ONE epilogue kernel (mspl_eval_epilogue_fwd: up-sampling, head sum, weighted cross entropy sums,
argmax, the three MIOU area histograms); nothing at full resolution is written, nothing is copied to the host until the end of the
loop, and the whole batch step replays as one hipGraph.  One process per GPU: rank r evaluates the loader's batches b == r (mod
world) and the sums (3*K areas, loss sum, image count, batch count) are all-reduced once at the end.
"""
import numpy as np
import torch

from . import dist as mdist
from . import ops
from ._native import check, lib
from .ops import _f32, _p, _stream
from .uest import _GraphedPassMixin, _lowres


def eval_epilogue(main, aux, target, class_weights, size, aux_weight, ignore_index, miou_classes, loss_sums, areas, labels=None):
    """One launch of mspl_eval_epilogue_fwd; accumulates into loss_sums (2 float64) and areas ((3, K) int64)."""
    main = _f32(main, 'main')
    N, C, Hm, Wm = main.shape
    Ha = Wa = 0
    if aux is not None:
        aux = _f32(aux, 'aux')
        if aux.shape[0] != N or aux.shape[1] != C:
            raise RuntimeError('mspl_amd: aux logits %s do not match main %s' % (tuple(aux.shape), tuple(main.shape)))
        Ha, Wa = aux.shape[2:]
    H, W = int(size[0]), int(size[1])
    if not target.is_cuda or target.dtype != torch.int64 or not target.is_contiguous() or target.numel() != N * H * W:
        raise RuntimeError('mspl_amd: eval target must be a contiguous CUDA int64 tensor of %d x %d x %d labels, got %s %s'
                           % (N, H, W, tuple(target.shape), target.dtype))
    if loss_sums.dtype != torch.float64 or loss_sums.numel() < 2 or areas.dtype != torch.int64 or areas.numel() < 3 * miou_classes:
        raise RuntimeError('mspl_amd: eval accumulators must be float64[2] and int64[3*K]')
    cw = None
    if class_weights is not None:
        cw = _f32(class_weights, 'class_weights')
        if cw.numel() != C:
            raise RuntimeError('mspl_amd: %d class weights for %d classes' % (cw.numel(), C))
    if labels is not None and (labels.dtype != torch.uint8 or labels.numel() != N * H * W or not labels.is_contiguous()):
        raise RuntimeError('mspl_amd: eval label output must be a contiguous uint8 (N,H,W) tensor')
    check(lib.mspl_eval_epilogue_fwd(_p(main), _p(aux), _p(target), _p(cw), N, C, Hm, Wm, Ha, Wa, H, W, float(aux_weight),
                                     int(ignore_index), int(miou_classes), _p(loss_sums), _p(areas), _p(labels), _stream()))


class EvalSums(object):
    """What an evaluation loop accumulates and how the final numbers follow from it (host logic, shared by EvalPass and by the
    CPU stand-ins of the multi-rank tests).  Subclasses provide `K` and `sums()` = one float64 tensor
    [3*K areas (inter | pred | mask) | sum loss_b * n_b | sum n_b | batches]."""

    def result(self, reduce=True):
        """(iou float64[K], average loss) of the batches seen so far -- `inter_meter.sum / (union_meter.sum + 1e-10)` and `losses.avg`
        of the reference loop (union per batch = pred + mask - inter + 1e-6, so the 1e-6 counts once per batch).  With
        torch.distributed initialised and reduce=True the sums of all ranks are added first (one all-reduce)."""
        s = self.sums()
        if reduce and mdist.collective_needed():
            torch.distributed.all_reduce(s, op=torch.distributed.ReduceOp.SUM)
        s = s.cpu().numpy()
        K = self.K
        inter, pred, mask = s[:K], s[K:2 * K], s[2 * K:3 * K]
        loss_sum, n_img, n_batches = s[3 * K], s[3 * K + 1], s[3 * K + 2]
        union = pred + mask - inter + 1e-6 * n_batches
        iou = inter / (union + 1e-10)
        return iou, (float(loss_sum / n_img) if n_img > 0 else 0.0)


class EvalPass(EvalSums, _GraphedPassMixin):
    """One evaluation batch step on the device, accumulating over calls.

        ep = EvalPass(model, num_classes=5, class_weights=cw, ignore_idx=4)         # val_seg_ue: out + 0.5 * aux
        for images, labels in loader: ep(images, labels)
        iou, loss = ep.result()

    aux_weight=0.5 is val_seg_ue's `outputs + 0.5 * out_aux`; aux_weight=0 is the uest script's test() (criterion and MIOU on the main
    head alone); single-head models always use the main head.  MIOU runs over num_classes - 1 classes like the reference
    (`MIOU(num_classes=num_classes-1)`).  use_graph=True captures forward + epilogue per input shape and replays it."""

    def __init__(self, model, num_classes, class_weights=None, ignore_idx=255, aux_weight=0.5, device='cuda', use_graph=False):
        self.model = model.to(device).eval()
        self.device = torch.device(device)
        self.num_classes = int(num_classes)
        self.K = self.num_classes - 1
        self.cw = None if class_weights is None else class_weights.detach().to(self.device, torch.float32).contiguous()
        self.ignore_idx = int(ignore_idx)
        self.aux_weight = float(aux_weight)
        self.use_graph = use_graph
        self.areas = torch.zeros((3, self.K), dtype=torch.int64, device=self.device)
        self.loss_sums = torch.zeros(2, dtype=torch.float64, device=self.device)       # this batch's (sum w*nll, sum w)
        self.acc = torch.zeros(3, dtype=torch.float64, device=self.device)             # sum loss_b * n_b, sum n_b, last loss_b
        self.batches = 0
        self._graphs = {}
        self._splits = None          # batch sizes of the reference's batches inside one launch (PipelinedEvalPass(group > 1)); None: one

    def reset(self):
        self.areas.zero_()
        self.loss_sums.zero_()
        self.acc.zero_()
        self.batches = 0

    # ---- _GraphedPassMixin hooks: the input is (images[, depth], labels)
    def _graph_models(self):
        return [self.model]

    def _graph_state(self):
        return [self.areas, self.loss_sums, self.acc]

    @staticmethod
    def _graph_outputs(run_result):
        return run_result

    def _shape_key(self, inputs):
        return tuple(tuple(t.shape) for t in inputs) + (self._splits,)

    def static_inputs(self, inputs_like, splits):
        """The input buffers of the graph captured for a launch of `splits` batches shaped like `inputs_like` (one batch), or None."""
        n = sum(splits)
        key = tuple((n,) + tuple(t.shape[1:]) for t in inputs_like) + (tuple(splits),)
        g = self._graphs.get(key)
        return None if g is None else g.static_in

    @staticmethod
    def _clone_input(inputs):
        return tuple(t.clone() for t in inputs)

    @staticmethod
    def _copy_input(static_in, inputs, copy_always):
        for s, t in zip(static_in, inputs):
            if copy_always or s.data_ptr() != t.data_ptr():
                s.copy_(t)

    def _run(self, inputs):
        images, labels = inputs[0], inputs[-1]
        if len(inputs) == 3:
            out = self.model.forward_lowres(images, inputs[1])
            main, aux = out if isinstance(out, tuple) else (out, None)
        else:
            main, aux = _lowres(self.model, images)
        aw = self.aux_weight if aux is not None else 0.0
        # one forward over the whole launch; loss sums and the meter update per BATCH of the reference's loop (the loss is a mean over a
        # batch's valid pixels, train_eval_seg.py:288-297), areas add up over everything
        off = 0
        for nb in (self._splits or (int(images.shape[0]),)):
            sl = slice(off, off + nb)
            eval_epilogue(main[sl], aux[sl] if aw != 0.0 else None, labels[sl], self.cw, images.shape[2:], aw, self.ignore_idx, self.K,
                          self.loss_sums, self.areas)
            check(lib.mspl_eval_batch_finalize(_p(self.loss_sums), _p(self.acc), int(nb), _stream()))
            off += nb
        return self.acc

    def __call__(self, images, labels, depth=None, splits=None):
        """splits: sizes of the consecutive batches of the reference's loop that `images` holds (default: one batch)."""
        with torch.no_grad():
            images = images.to(self.device)
            labels = labels.to(self.device, torch.int64).contiguous()
            if tuple(labels.shape) != (images.shape[0],) + tuple(images.shape[2:]):
                raise RuntimeError('mspl_amd: eval labels %s do not match images %s' % (tuple(labels.shape), tuple(images.shape)))
            splits = (int(images.shape[0]),) if splits is None else tuple(int(v) for v in splits)
            if sum(splits) != images.shape[0] or min(splits) < 1:
                raise RuntimeError('mspl_amd: eval splits %s do not add up to %d images' % (splits, images.shape[0]))
            self._splits = splits
            inputs = (images, labels) if depth is None else (images, depth.to(self.device), labels)
            self.batches += len(splits)
            if self.use_graph:
                return self._replay(inputs)
            return self._run(inputs)

    def sums(self):
        """Device tensor of everything the final numbers need: [3*K areas | sum loss_b*n_b | sum n_b | batches] as float64 (exact: the
        counts stay far below 2^53) -- ONE tensor, so a multi-rank evaluation needs one all-reduce."""
        return torch.cat([self.areas.reshape(-1).to(torch.float64), self.acc[:2],
                          torch.tensor([float(self.batches)], dtype=torch.float64, device=self.device)])


class PipelinedEvalPass(EvalSums):
    """`depth` EvalPass lanes taken in turn, each on a stream of its own with its own hipGraph, static inputs and accumulators: the
    evaluation step of consecutive batches overlaps on the GPU the way the label pass's lanes do (one graph replay alone leaves the
    chip half idle: its kernels are small and run one after the other).  The lanes never share an accumulator, so nothing races;
    `sums()` adds them up (integer areas: exact; the loss sums are float64 sums of per-batch means either way).

    group: consecutive batches that ONE launch of a lane evaluates (the label pass's batches-per-launch): the batches are copied into
    the lane's input buffer as they arrive (the caller may reuse its tensors at once) and the lane is launched when `group` of them are
    staged, or earlier on a batch of another shape / `flush()` / `sums()`.  Per-batch semantics are kept: the loss is a mean per batch
    and the meters are updated batch by batch (EvalPass._run), so the results do not depend on `group`."""

    def __init__(self, model, num_classes, depth=3, group=1, **kw):
        from .uest import _concurrent_streams
        kw.setdefault('use_graph', True)
        self.lanes = [EvalPass(model, num_classes, **kw) for _ in range(max(1, int(depth)))]
        self.device = self.lanes[0].device
        self.K = self.lanes[0].K
        self.group = max(1, int(group))
        self.streams = _concurrent_streams(len(self.lanes), self.device) if len(self.lanes) > 1 else [None]
        self._next = 0
        self._staged = []            # batches waiting in the current lane: (images, depth, labels, in_static_slot)

    @property
    def batches(self):
        return sum(l.batches for l in self.lanes) + len(self._staged)

    def reset(self):
        self._join()
        for l in self.lanes:
            l.reset()

    def flush(self):
        """Launch what is staged in the current lane (fewer than `group` batches)."""
        if self._staged:
            self._launch()

    def _join(self):
        self.flush()
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams:
            if st is not None:
                cur.wait_stream(st)

    def _on_lane(self):
        st = self.streams[self._next]
        return torch.cuda.stream(st) if st is not None else torch.no_grad()

    def __call__(self, images, labels, depth=None):
        lane, st = self.lanes[self._next], self.streams[self._next]
        images = images.to(self.device)
        labels = labels.to(self.device)
        depth = None if depth is None else depth.to(self.device)
        if self._staged and (self._staged[0][0].shape != images.shape or (self._staged[0][1] is None) != (depth is None)):
            self._launch()                                              # a batch of another shape: evaluate what is staged first
            lane, st = self.lanes[self._next], self.streams[self._next]
        k = len(self._staged)
        if st is not None:
            st.wait_stream(torch.cuda.current_stream(self.device))      # the caller's batch is complete before the lane copies it
        with self._on_lane(), torch.no_grad():
            # the lane takes its copy now: straight into slot k of the graph's input buffer once that graph exists
            like = (images, labels) if depth is None else (images, depth, labels)
            n = images.shape[0]
            static = lane.static_inputs(like, (n,) * self.group) if self.group > 1 and lane.use_graph else None
            if static is not None and all(b[3] for b in self._staged):
                sl = slice(k * n, (k + 1) * n)
                static[0][sl].copy_(images)
                static[-1][sl].copy_(labels)
                if depth is not None:
                    static[1][sl].copy_(depth)
                self._staged.append((static[0][sl], None if depth is None else static[1][sl], static[-1][sl], True))
            elif self.group == 1:
                self._staged.append((images, depth, labels, False))      # launched below, before this call returns
            else:
                self._staged.append((images.clone(), None if depth is None else depth.clone(), labels.to(torch.int64).clone(), False))
        for t in (images, labels, depth):
            if t is not None and st is not None:
                t.record_stream(st)
        if len(self._staged) == self.group:
            self._launch()
        return lane.acc

    def _launch(self):
        lane, st = self.lanes[self._next], self.streams[self._next]
        staged, self._staged = self._staged, []
        self._next = (self._next + 1) % len(self.lanes)
        splits = tuple(int(b[0].shape[0]) for b in staged)
        with self._on_lane(), ops.launch_flags(throughput=True):          # launch shapes for kernels that share the GPU (see uest.py)
            if len(staged) == 1:
                images, depth, labels = staged[0][:3]
            elif all(b[3] for b in staged) and len(staged) == self.group:
                like = staged[0][:3] if staged[0][1] is not None else (staged[0][0], staged[0][2])
                static = lane.static_inputs(like, splits)                 # the slots the batches were copied into: no further copy
                images, labels = static[0], static[-1]
                depth = static[1] if staged[0][1] is not None else None
            else:
                images = torch.cat([b[0] for b in staged])
                labels = torch.cat([b[2] for b in staged])
                depth = torch.cat([b[1] for b in staged]) if staged[0][1] is not None else None
            lane(images, labels, depth, splits=splits)

    def sums(self):
        self._join()
        total = self.lanes[0].sums()
        for l in self.lanes[1:]:
            total = total + l.sums()
        return total


def val_seg_ue(model, dataset_loader, criterion=None, num_classes=21, device='cuda', use_depth=False, add_criterion=None,
               greenhouse_use_trav=False, use_graph=True, pre_sharded=False, _eval_pass=None, lanes=3, group=2):
    """Drop-in for utilities/train_eval_seg.py:249-324: returns (iou, average loss) ((iou, 0) without a criterion).

    criterion: a SegmentationLoss-like object -- its `class_wts` / `class_weights` and `ignore_idx` are read (loss_type 'ce'); None
    evaluates the MIOU only.  add_criterion (the NID term of the supervised loop) has no fused form and is refused.
    One process per GPU: rank r takes batches b == r (mod world) unless the loader is pre_sharded; every rank returns the same result.
    lanes: evaluation steps in flight on the GPU (PipelinedEvalPass; 1 = one graph replay after the other); group: consecutive
    batches per launch of a lane."""
    if add_criterion is not None:
        raise NotImplementedError('mspl_amd: val_seg_ue(add_criterion=...) is not on the path (the shipped scripts pass None)')
    cw = ign = None
    if criterion is not None:
        if getattr(criterion, 'loss_type', 'ce') != 'ce':
            raise NotImplementedError("mspl_amd: only loss_type='ce' is on the path")
        cw = getattr(criterion, 'class_wts', None)
        if cw is None:
            cw = getattr(criterion, 'class_weights', None)
        ign = getattr(criterion, 'ignore_idx', 255)
    # (_eval_pass: an EvalSums stand-in with EvalPass's call signature, for host-logic tests without a GPU)
    if _eval_pass is not None:
        ep = _eval_pass
    elif use_graph and lanes > 1:
        ep = PipelinedEvalPass(model, num_classes, depth=lanes, group=group, class_weights=cw, ignore_idx=255 if ign is None else ign, aux_weight=0.5,
                               device=device)
    else:
        ep = EvalPass(model, num_classes, class_weights=cw, ignore_idx=255 if ign is None else ign, aux_weight=0.5, device=device,
                      use_graph=use_graph)
    rank, world = mdist.world()
    for b, batch in enumerate(dataset_loader):
        if world > 1 and not pre_sharded and b % world != rank:
            continue
        ep(batch[0], batch[1], batch[2] if use_depth else None)
    iou, loss = ep.result()
    return iou, (loss if criterion is not None else 0)


def miou_percent(iou, use_traversable=False):
    """The summary line of both loops: mean over all classes with --use-traversable, else over classes 1..3 (train_eval_seg.py:314-319,
    uest_seg_multi_os.py:1204-1207)."""
    iou = np.asarray(iou)
    return float(iou.mean() * 100) if use_traversable else float(iou[[1, 2, 3]].mean() * 100)
