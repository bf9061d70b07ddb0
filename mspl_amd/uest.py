"""The pseudo-label pass of uest_seg_multi_os.py on the HIP path.

Reference functions mirrored (same names / argument meaning; paths relative to the reference root):
  get_output                          uest_seg_multi_os.py:669-693
  merge_outputs                       uest_seg_multi_os.py:695-718
  generate_pseudo_label_multi_model   uest_seg_multi_os.py:832-956  -> PseudoLabelPass (batched, on device) + the function itself
  id_*_to_greenhouse                  data_loader/segmentation/greenhouse.py:15-58

The reference runs batch size 1, one source model after another, and post-processes on the host with
numpy (two D2H copies of C*H*W floats per model per image).  Here a batch of images goes through all
source models on the GPU; the only tensors that ever exist at full resolution are uint8 class maps
(1 byte/pixel/source), merged by an integer kernel that also accumulates the class histogram.
"""
import numpy as np
import torch

from . import layers, ops

# data_loader/segmentation/greenhouse.py:15-58 (literal tables: source class id -> greenhouse class id)
id_camvid_to_greenhouse = np.array([4, 2, 2, 3, 3, 1, 2, 2, 2, 4, 4, 2, 4])
id_cityscapes_to_greenhouse = np.array([3, 3, 2, 2, 2, 2, 2, 2, 1, 3, 4, 4, 4, 2, 2, 2, 2, 2, 2, 4])
id_forest_to_greenhouse = np.array([3, 1, 1, 2, 2])
LUTS = {'camvid': id_camvid_to_greenhouse, 'cityscapes': id_cityscapes_to_greenhouse,
        'forest': id_forest_to_greenhouse}
GREENHOUSE_CLASSES = 5
NO_AGREEMENT_CLASS = 4      # the literal written at uest_seg_multi_os.py:716


def resolve_thresh(num_data, thresh):
    """Vote threshold rule of merge_outputs (uest_seg_multi_os.py:697-705)."""
    if thresh is None or thresh == 'half':
        return num_data // 2 + 1
    if thresh == 'all':
        return num_data
    if isinstance(thresh, int) and not isinstance(thresh, bool) and thresh <= num_data:
        return thresh
    return num_data // 2 + 1


def _lowres(model, image, pyr=None):
    out = model.forward_lowres(image, pyr=pyr) if pyr is not None else model.forward_lowres(image)
    return out if isinstance(out, tuple) else (out, None)


def _lowres_batch_stats(model, images, pyr=None):
    """`--eval-training` label generation (uest_seg_multi_os.py:749-752, 871-876): the model is in train() mode under no_grad, so
    every BatchNorm normalises with the statistics of ITS OWN batch -- and the reference's loader has batch size 1 (:746), so the
    statistics are per image.  Reproduced image by image through the batch-statistics forward kernels (the autograd form of the
    layers: the inference kernels fold RUNNING statistics and cannot serve this mode); the running statistics receive the same
    momentum updates as in the reference.  Returns the heads of the whole batch."""
    mains, auxs = [], []
    for k in range(images.shape[0]):
        with torch.enable_grad():
            main, aux = _lowres(model, images[k:k + 1], pyr)
        mains.append(main.detach())
        auxs.append(None if aux is None else aux.detach())
    return torch.cat(mains), (None if auxs[0] is None else torch.cat(auxs))


def get_output(model, image, model_name='espdnetue', device='cuda'):
    """Drop-in for uest_seg_multi_os.get_output: (softmax(pred + 0.5*aux) of batch element 0 as a numpy
    (C,H,W) array, KL(pred||aux) map as numpy (H,W)).  The upsample, the softmax and the KLD run in one
    kernel; only the two requested maps are copied to the host."""
    with torch.no_grad():
        image = image.to(device)
        main, aux = _lowres(model, image)
        r = ops.label_epilogue(main, aux, image.shape[2:], want_labels=False, want_prob=True, want_kld=True)
    return r['prob'][0].cpu().numpy(), r['kld'][0].cpu().numpy()


def merge_outputs(amax_outputs, seg_classes=GREENHOUSE_CLASSES, thresh=None):
    """Drop-in for uest_seg_multi_os.merge_outputs.  amax_outputs: (S, ...) class maps, numpy or torch.
    numpy in -> numpy int64 out (like counts_np.argmax); CUDA uint8 in -> CUDA uint8 out."""
    is_np = isinstance(amax_outputs, np.ndarray)
    t = torch.from_numpy(np.ascontiguousarray(amax_outputs).astype(np.uint8)).cuda() if is_np else amax_outputs
    if t.dtype != torch.uint8:
        t = t.to(torch.uint8)
    S = t.shape[0]
    if t[0].numel() == 0:
        out = torch.empty_like(t[0])
    else:
        out = ops.merge_labels([t[s] for s in range(S)], seg_classes, resolve_thresh(S, thresh), NO_AGREEMENT_CLASS)
    return out.cpu().numpy().astype(np.int64) if is_np else out


def class_weights_from_histogram(class_array, policy='normal'):
    """uest_seg_multi_os.py:942-947 (host-side, 5 float64 values)."""
    class_array = np.asarray(class_array, dtype=np.float64)
    if policy == 'normal':
        freq = class_array / class_array.sum()
        w = 1.0 / (freq + 1e-10)
        w[0] = 0.0
        return w
    return np.ones(len(class_array))


import os as _os
_NO_SIGNATURE = _os.environ.get('MSPL_GRAPH_SIGNATURE', '1') == '0'      # measurement aid: epoch check only (unsafe with torch-side edits)


class _GraphedPassMixin:
    """hipGraph capture / replay shared by PseudoLabelPass and SelfLabelPass.

    One captured pass = `_Captured(graph, static_in, static_out, signature)`.  Two hazards are closed here (both seen in
    round 1, DESIGN.md section 4 "Failures on record"):
      * lifetime -- a hipGraph bakes in raw pointers: its input buffer, the pass's histogram, the folded-BN / packed-weight
        caches of every module.  Everything the capture touched that is NOT allocated from the graph's own memory pool is
        referenced from the graph object itself (`graph._mspl_keep`), so whoever holds the graph (or a `static_input()` view,
        which carries the graph along) keeps those buffers alive: dropping the pass object can no longer free memory that a
        live graph replays on (the round-1 fault: a freed `static_in` under a separately held graph, unmapped by the next
        capture's empty_cache()).
      * staleness -- the graph is valid for the parameter values it was captured with.  Each graph stores the parameter
        signature (layers._PARAM_EPOCH, which the raw-pointer optimizer kernels bump, plus (data_ptr, _version) of every
        parameter and buffer); a replay with a different signature re-captures first (FlatAdam re-points .data, every
        optimizer step rebuilds the folded caches elsewhere)."""

    def _signature(self):
        """(epoch, (id, data_ptr, _version) of every floating-point parameter / buffer SLOT of the models).  The slots (module, dict,
        name) are cached, the tensors are looked up on every call: a replaced nn.Parameter / buffer object (model surgery,
        `module.weight = nn.Parameter(...)`) changes the id and the pointer.  An in-place edit through `.data` bumps neither: callers
        that do that must call layers.bump_param_epoch() or invalidate_graphs()."""
        if _NO_SIGNATURE:
            return (layers._PARAM_EPOCH[0],)
        slots = self.__dict__.get('_sig_slots')
        if slots is None:
            slots = []
            for m in self._graph_models():
                for mod in m.modules():
                    slots += [(mod._parameters, k) for k, t in mod._parameters.items() if t is not None and t.is_floating_point()]
                    slots += [(mod._buffers, k) for k, t in mod._buffers.items() if t is not None and t.is_floating_point()]
            self._sig_slots = slots
        sig = [layers._PARAM_EPOCH[0]]
        for d, k in slots:
            t = d.get(k)
            sig.append(None if t is None else (id(t), t.data_ptr(), t._version))
        return tuple(sig)

    def invalidate_graphs(self):
        """Drop every captured graph (after editing parameters in a way the signature cannot see, e.g. through `.data`)."""
        self._graphs.clear()
        self.__dict__.pop('_sig_slots', None)

    def _cache_tensors(self):
        keep = []
        for m in self._graph_models():
            for mod in m.modules():
                for _, val in mod.__dict__.get('_mspl_cache', {}).values():
                    keep.append(val)
        return keep

    def _replay(self, images, copy_always=False):
        key = self._shape_key(images)
        sig = self._signature()
        g = self._graphs.get(key)
        if g is not None and g.signature != sig:
            del self._graphs[key]                     # parameters changed since the capture: this graph holds stale values
            g = None
        if g is None:
            static_in = self._clone_input(images)
            state = self._graph_state()               # accumulators the pass adds to (histogram, ...): the warm-up must not count
            before = [t.clone() for t in state]
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):             # warm-up: fills the folded-BN caches outside the capture
                self._run(static_in)
            torch.cuda.current_stream().wait_stream(side)
            for t, b in zip(state, before):
                t.copy_(b)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode='thread_local'):      # other threads (the PNG writer) keep using HIP
                static_out = self._graph_outputs(self._run(static_in))
            for t, b in zip(state, before):           # capture does not execute, but keep the invariant explicit
                t.copy_(b)
            graph._mspl_keep = (static_in, state, self._cache_tensors(), list(getattr(self, 'luts', ())))
            g = self._graphs[key] = _Captured(graph, static_in, static_out, sig)
        self._copy_input(g.static_in, images, copy_always)
        g.graph.replay()
        return g.static_out

    # hooks (single-tensor input, one histogram by default)
    @staticmethod
    def _shape_key(images):
        return tuple(images.shape)

    def _graph_state(self):
        return [self.hist]

    @staticmethod
    def _clone_input(images):
        return images.clone()

    @staticmethod
    def _copy_input(static_in, images, copy_always):
        if copy_always or static_in.data_ptr() != images.data_ptr():
            static_in.copy_(images)

    def static_input(self, shape):
        """The graph's own input buffer for `shape` (write batches straight into it to skip the copy); None before the first
        call with that shape.  The returned tensor keeps the graph's buffers alive (`_mspl_graph`)."""
        g = self._graphs.get(tuple(shape))
        if g is None:
            return None
        view = g.static_in.view(g.static_in.shape)
        view._mspl_graph = g.graph
        return view


class _Captured(object):
    __slots__ = ('graph', 'static_in', 'static_out', 'signature')

    def __init__(self, graph, static_in, static_out, signature):
        self.graph, self.static_in, self.static_out, self.signature = graph, static_in, static_out, signature

    def __iter__(self):                                # (graph, static_in, static_out), the round-1 tuple layout
        return iter((self.graph, self.static_in, self.static_out))

    def __getitem__(self, i):
        return (self.graph, self.static_in, self.static_out)[i]


class PseudoLabelPass(_GraphedPassMixin):
    """Batched multi-source pseudo-label generation (the loop body of generate_pseudo_label_multi_model).

    model_list / os_data_list as in the reference (os_data in {'camvid','cityscapes','forest', other=identity}).
    __call__(images) -> merged uint8 label maps (N,H,W) on the device; the per-class pixel histogram
    accumulates in self.hist (int64[classes], device) until reset().  With use_graph=True the whole pass
    for one batch shape is captured once into a hipGraph and replayed.
    """

    def __init__(self, model_list, os_data_list, classes=GREENHOUSE_CLASSES, merge_label_policy='all',
                 device='cuda', use_graph=False, eval_training=False):
        if len(model_list) != len(os_data_list) or not model_list:
            raise ValueError('model_list and os_data_list must be non-empty and of equal length')
        # eval_training: args.eval_training of the reference (:871-876): models in train() mode, per-image batch statistics
        self.eval_training = bool(eval_training)
        self.models = [(m.to(device).train() if self.eval_training else m.to(device).eval()) for m in model_list]
        self.os_data = list(os_data_list)
        self.classes = classes
        self.thresh = resolve_thresh(len(model_list), merge_label_policy)
        self.device = torch.device(device)
        self.luts = []
        for m, d in zip(self.models, self.os_data):
            lut = LUTS.get(d)
            self.luts.append(None if lut is None else torch.from_numpy(lut.astype(np.uint8)).to(self.device))
        self.hist = torch.zeros(classes, dtype=torch.int64, device=self.device)
        self.use_graph = use_graph and not self.eval_training      # (the running statistics change with every image: nothing to replay)
        self._graphs = {}

    def reset(self):
        self.hist.zero_()

    def _run(self, images):
        maps = []
        if self.eval_training:
            for m, lut in zip(self.models, self.luts):
                main, aux = _lowres_batch_stats(m, images)
                maps.append(ops.label_epilogue(main, aux, images.shape[2:], lut=lut)['labels'])
            return ops.merge_labels(maps, self.classes, self.thresh, NO_AGREEMENT_CLASS, self.hist), maps
        # the average-pool pyramid of the batch (input reinforcement of every DownSampler) does not depend on the model: once per batch
        shared = all(hasattr(m, 'depth_base_net') and getattr(m.base_net, 'input_reinforcement', False) for m in self.models)
        pyr = layers.ImagePyramid(images) if shared and len(self.models) > 1 else None
        for m, lut in zip(self.models, self.luts):
            # (sources run one after the other: forking one HIP stream per source model on top of the models' own side
            # streams crashed hipStreamEndCapture on ROCm 7.2 -- nested fork/join graphs are avoided)
            main, aux = _lowres(m, images, pyr)
            maps.append(ops.label_epilogue(main, aux, images.shape[2:], lut=lut)['labels'])
        return ops.merge_labels(maps, self.classes, self.thresh, NO_AGREEMENT_CLASS, self.hist), maps

    def source_maps(self, images):
        """Per-source class maps (after the id LUT) -- what the reference appends to output_list."""
        with torch.no_grad():
            return self._run(images.to(self.device))[1]

    def __call__(self, images):
        with torch.no_grad():
            images = images.to(self.device)
            if not self.use_graph:
                return self._run(images)[0]
            return self._replay(images)

    def _graph_models(self):
        return self.models

    @staticmethod
    def _graph_outputs(run_result):
        return run_result[0]

    def class_weights(self, policy='normal'):
        return torch.from_numpy(class_weights_from_histogram(self.hist.cpu().numpy(), policy)).float().to(self.device)


def _concurrent_streams(n, device, candidates=12, spin_cycles=400000):
    """n streams whose kernels really run side by side.  HIP multiplexes streams onto a few hardware queues (4 by default,
    GPU_MAX_HW_QUEUES); two streams on the same queue execute in order, and which queue a torch stream lands on depends on how
    many streams the process created before.  Measured, not assumed: a spin kernel on the candidate and on every stream
    chosen so far must take about as long as one spin, not the sum.  Falls back to plain new streams when no n-subset overlaps
    (e.g. GPU_MAX_HW_QUEUES=1)."""
    import time
    if n == 1 or not hasattr(torch.cuda, '_sleep'):
        return [torch.cuda.Stream(device=device) for _ in range(n)]

    def spin_ms(streams):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for st in streams:
            with torch.cuda.stream(st):
                torch.cuda._sleep(spin_cycles)
        torch.cuda.synchronize(device)
        return (time.perf_counter() - t0) * 1e3

    pool = [torch.cuda.Stream(device=device) for _ in range(candidates)]
    # The spin kernel counts CYCLES: on an idle chip the first spins run at a low clock and take longer, and a baseline taken then makes
    # two serialised spins at the boosted clock look like one (seen as a four-lane train step of 11.8 ms instead of 7.5 ms: all lanes
    # on one queue).  So: clocks warmed up first, and the final choice VALIDATED against a baseline taken after it; a failed
    # validation repeats the selection.
    for _ in range(30):
        spin_ms(pool[:1])
    chosen = [pool[0]]
    for attempt in range(3):
        one = min(spin_ms(pool[:1]) for _ in range(3))
        chosen = [pool[0]]
        for st in pool[1:]:
            if len(chosen) == n:
                break
            if min(spin_ms(chosen + [st]) for _ in range(2)) < one * (1.0 + 0.5 * len(chosen)):
                chosen.append(st)
        if len(chosen) < n:
            break                                       # fewer hardware queues than lanes: nothing to validate
        together = min(spin_ms(chosen) for _ in range(3))
        alone = min(spin_ms(chosen[:1]) for _ in range(3))
        if together < 1.6 * alone:
            break
    while len(chosen) < n:
        chosen.append(torch.cuda.Stream(device=device))
    return chosen


class PipelinedLabelPass:
    """`depth` label passes in flight: lane i owns a pass object (its hipGraph, static buffers and histogram) and a stream; batch k
    goes to lane k % depth.  Images are independent in the label loop (the reference runs them one at a time), so consecutive
    batches may overlap -- and they should: most kernels of a pass are single-round launches whose ramp, tail and
    load/compute/store phases leave the chip partly idle; a second pass in flight fills those holes (+22 % images/s at
    depth 2, +33 % at depth 3 on MI355X; at depth 4 the lanes outnumber the free hardware queues and it drops again).  Two things make the overlap
    dependable rather than lucky: the lanes' streams are CHOSEN by measurement (`_concurrent_streams`: HIP multiplexes
    streams onto 4 hardware queues and two streams on one queue run in order), and the lanes are captured WITHOUT the
    models' internal side streams -- a branched hipGraph is handed internal streams at instantiation that may land on
    the other lane's queue, which silently serialises the two passes (seen as x1.00 on two of four pass types).

        plp = PipelinedLabelPass(lambda: SelfLabelPass(model, use_graph=True), depth=3)
        for images in loader:
            out = plp(images)            # outputs of the batch submitted depth-1 calls earlier, None while the pipe fills
            if out is not None: consume(out)
        for out in plp.flush(): consume(out)

    What `plp(...)` returns is valid on the current stream until the next call (its lane's static buffers are reused depth calls
    later, and every submit first waits for the work already queued on the current stream).  `hist` sums the lanes' histograms.

    group = g > 1: g consecutive batches are staged in a lane's input buffer and labelled by ONE launch of the lane's pass; outputs
    still come back batch by batch in submission order, depth * g - 1 calls after their batch went in; a batch of another shape or
    flush() labels a partly filled lane with a shorter launch."""

    def __init__(self, make_pass, depth=3, device='cuda', group=1):
        if depth < 1 or group < 1:
            raise ValueError('PipelinedLabelPass: depth and group must be >= 1')
        self.device = torch.device(device)
        self.depth, self.group = depth, group
        self.lanes = [make_pass() for _ in range(depth)]
        self.streams = _concurrent_streams(depth, self.device)
        self._pending = []
        self._n = 0
        # group > 1: `group` consecutive batches are staged in a lane's input buffer and labelled by ONE launch of the lane's pass
        # (images are independent, so this changes nothing but the launch shapes: at 16 x 3 x 288 x 480 most kernels of a pass are one
        # round of workgroups, and 32 images per launch run at 16 100 images/s where 16 run at 15 000 -- tools: bench.py
        # --profile-batch).  Outputs are handed back per batch, in submission order.
        self._staged = None              # [lane, count, batch shape, buffer] of the lane being filled
        self._lane_no = 0
        self._bufs = {}

    def join(self):
        """The current stream waits for everything queued on the lanes' streams (results consumed with on_lane=True are otherwise
        ordered only against their own lane)."""
        cur = torch.cuda.current_stream(self.device)
        for st in self.streams:
            cur.wait_stream(st)

    @property
    def hist(self):
        self.join()          # the lanes' launches add to their histograms on the lanes' streams
        total = self.lanes[0].hist.clone()
        for lane in self.lanes[1:]:
            total += lane.hist
        return total

    def reset(self):
        for lane in self.lanes:
            lane.reset()

    def class_weights(self, policy='normal'):
        return torch.from_numpy(class_weights_from_histogram(self.hist.cpu().numpy(), policy)).float().to(self.device)

    @property
    def next_lane(self):
        """Index into static_inputs() of the buffer the next submitted batch (of the shape being staged) will use.  Derived from
        the staging state, not from the number of calls: after a partly filled launch (flush() or a batch of another shape) the
        next batch starts a fresh lane at slot 0 whatever the call count is."""
        if self._staged is not None:
            return self._staged[0] * self.group + self._staged[1]
        return (self._lane_no % self.depth) * self.group

    def _group_buffer(self, i, bshape):
        """Input buffer of lane i for `group` batches of shape bshape: the captured graph's own buffer once there is one (no copy
        between staging and launch), a plain tensor before."""
        full = (self.group * bshape[0],) + tuple(bshape[1:])
        key = (i, full)
        buf, owned = self._bufs.get(key, (None, False))
        if not owned:
            own = self.lanes[i].static_input(full) if hasattr(self.lanes[i], 'static_input') else None
            if own is not None:
                buf, owned = own, True
            elif buf is None:
                buf = torch.empty(full, device=self.device, dtype=torch.float32)
            self._bufs[key] = (buf, owned)
        return buf

    def static_inputs(self, shape):
        """Static input buffers of the captured graphs, one per batch slot in submission order (depth * group of them; None for
        eager lanes or before a lane's first launch): filling slot k and passing it to the k-th call skips the input copy."""
        if self.group == 1:
            return [lane.static_input(shape) for lane in self.lanes]
        B = shape[0]
        full = (self.group * B,) + tuple(shape[1:])
        out = []
        for lane in self.lanes:
            buf = lane.static_input(full)
            out += [None if buf is None else buf[j * B:(j + 1) * B] for j in range(self.group)]
        return out

    def _launch(self, i, images):
        """One launch of lane i's pass on its stream; returns (outputs, event)."""
        st = self.streams[i]
        # launch shapes for a shared chip (per call, read at capture); lanes are captured as linear graphs
        with torch.cuda.stream(st), layers.side_streams(self.depth == 1), ops.launch_flags(throughput=self.depth > 1):
            out = self.lanes[i](images)
        with torch.cuda.stream(st):
            ev = torch.cuda.Event()
            ev.record(st)
        return out, ev

    def _launch_staged(self):
        i, count, bshape, buf = self._staged
        self._staged = None
        B = bshape[0]
        out, ev = self._launch(i, buf if count == self.group else buf[:count * B])

        def part(t, j):
            return t[j * B:(j + 1) * B] if (torch.is_tensor(t) and t.dim() > 0 and t.shape[0] == count * B) else t
        for j in range(count):
            self._pending.append((tuple(part(t, j) for t in out) if isinstance(out, (tuple, list)) else part(out, j), ev, i))

    def submit(self, images):
        self._n += 1
        cur = torch.cuda.current_stream(self.device)
        if self.group == 1:
            i = self._lane_no % self.depth
            self._lane_no += 1
            self.streams[i].wait_stream(cur)    # the inputs, and whatever still reads this lane's previous outputs, are on `cur`
            out, ev = self._launch(i, images)
            if torch.is_tensor(images) and images.is_cuda:
                images.record_stream(self.streams[i])
            self._pending.append((out, ev, i))
            return
        bshape = tuple(images.shape)
        if self._staged is not None and self._staged[2] != bshape:
            self._launch_staged()               # a batch of another shape (the loader's last one): label what is staged first
        if self._staged is None:
            i = self._lane_no % self.depth
            self._lane_no += 1
            self._staged = [i, 0, bshape, self._group_buffer(i, bshape)]
        i, count, _, buf = self._staged
        st = self.streams[i]
        st.wait_stream(cur)
        slot = buf[count * bshape[0]:(count + 1) * bshape[0]]
        if not (images.is_cuda and images.data_ptr() == slot.data_ptr()):
            with torch.cuda.stream(st):
                slot.copy_(images, non_blocking=True)
            if images.is_cuda:
                images.record_stream(st)
        self._staged[1] = count + 1
        if self._staged[1] == self.group:
            self._launch_staged()

    @property
    def next_stream(self):
        """Stream of the lane the next submitted batch goes to.  Work issued on it BEFORE the submit (the upload and the loader
        transform of that batch, written into static_inputs()[next_lane]) is ordered behind the lane's previous launch and ahead of
        the next one without an event, and uses no hardware queue beyond the lanes' own."""
        if self._staged is not None:
            return self.streams[self._staged[0]]
        return self.streams[self._lane_no % self.depth]

    def pop(self, on_lane=False):
        """Oldest batch's outputs.  on_lane=False: the current stream waits for them.  on_lane=True: returns (outputs, lane stream)
        and waits for nothing -- the consumer queues its work on that stream (LabelWriter.submit(..., stream=)): it then runs right
        behind the launch that produced the outputs and ahead of the lane's next launch, which is what overwrites them."""
        if not self._pending and self._staged is not None:
            self._launch_staged()
        out, ev, i = self._pending.pop(0)
        if on_lane:
            return out, self.streams[i]
        torch.cuda.current_stream(self.device).wait_event(ev)
        return out

    def __call__(self, images, on_lane=False):
        self.submit(images)
        waiting = len(self._pending) + (self._staged[1] if self._staged is not None else 0)
        return self.pop(on_lane) if waiting >= self.depth * self.group else None

    def flush(self, on_lane=False):
        if self._staged is not None:
            self._launch_staged()
        while self._pending:
            yield self.pop(on_lane)


def _label_loop(p, testloader, save_path, labels_of, class_weighting, use_depth, device, writer_workers, pre_sharded, transform,
                eager_input):
    """Loop + files + list + class weights shared by the two label functions (uest_seg_multi_os.py:783-828 and :889-954 are the same
    code around different loop bodies).  `p`: PipelinedLabelPass interface; labels_of(result) -> the (N,H,W) uint8 maps of one batch.
    transform(image, out=slot) -> network input: the loader may then yield decoded uint8 frames (N,Hs,Ws,3), host or device, and the
    transform (mspl_amd.io.Preprocessor) writes the network input straight into the static input slot of the lane that labels it."""
    import os.path as osp
    from . import dist as mdist
    from .io import LabelWriter, default_writer_workers, update_image_list
    rank, world = mdist.world()
    p.reset()
    tgt_train_lst = osp.join(save_path, 'tgt_train.lst')
    if writer_workers is None:
        writer_workers = default_writer_workers(world)
    writer = LabelWriter(osp.join(save_path, 'pred'), workers=writer_workers, use_depth=use_depth)
    names, batch_sizes = [], []
    slots = {}
    # A real PipelinedLabelPass: everything around a batch runs on the stream of the lane that labels it -- upload + transform ahead of
    # the launch, the device -> host copy of the maps behind it.  No extra stream means no extra hardware queue: HIP has four, the three
    # lanes use three, and a loader stream or a writer stream that lands on a lane's queue waits behind that lane's whole launch.
    on_lane = isinstance(p, PipelinedLabelPass)

    def consume(r):
        if on_lane:
            writer.submit(names.pop(0), labels_of(r[0]), stream=r[1])
        else:
            writer.submit(names.pop(0), labels_of(r))
    for b, batch in enumerate(testloader):
        if world > 1 and not pre_sharded and b % world != rank:
            continue
        image = batch[0]
        names.append(list(batch[-2]))
        batch_sizes.append(len(names[-1]))
        if transform is not None:
            slot = None
            if hasattr(p, 'static_inputs'):
                W, H = transform.size
                shape = (image.shape[0], 3, H, W)
                xs = slots.get(shape)
                if not xs or any(x is None for x in xs):            # a lane has no graph (hence no slot) before its first launch
                    xs = slots[shape] = p.static_inputs(shape)
                slot = xs[p.next_lane] if xs else None
            if on_lane:
                with torch.cuda.stream(p.next_stream):
                    image = transform(image, out=slot)
            else:
                image = transform(image, out=slot)
            image = image[0] if isinstance(image, tuple) else image
        elif not (image.is_cuda or eager_input):
            image = image.to(device, non_blocking=True)
        out = p(image, on_lane=True) if on_lane else p(image)
        if out is not None:
            consume(out)
    for out in (p.flush(on_lane=True) if on_lane else p.flush()):
        consume(out)
    lists = writer.close()
    hist = p.hist
    if mdist.collective_needed():
        # per-batch records in this rank's order -> loader order (batch b came from rank b % world), then flattened
        recs, at = [], 0
        for n in batch_sizes:
            recs.append(tuple(l[at:at + n] for l in lists))
            at += n
        merged_recs = mdist.gather_lists(recs)
        lists = tuple([x for r in merged_recs for x in r[k]] for k in range(len(lists)))
        hist = mdist.reduce_histogram(hist.clone())
    if rank == 0:
        update_image_list(tgt_train_lst, *lists)
    mdist.barrier()                                     # the list file exists before any rank builds its train loader from it
    weights = torch.from_numpy(class_weights_from_histogram(hist.cpu().numpy(), class_weighting)).float().to(hist.device)
    return tgt_train_lst, weights


def generate_pseudo_label_multi_model(model_list, os_data_list, testloader, save_path, classes=GREENHOUSE_CLASSES,
                                      merge_label_policy='all', class_weighting='normal', use_depth=False, device='cuda',
                                      use_graph=True, writer_workers=None, in_flight=3, pre_sharded=False, _label_pass=None,
                                      batches_per_launch=1, transform=None, eval_training=False):
    """uest_seg_multi_os.py:832-956 end to end: label every batch of `testloader` with all source models, merge, write
    `<save_path>/pred/<image_name>.png`, write `<save_path>/tgt_train.lst` and return (tgt_train_lst, class_weights).

    testloader yields the reference's tuples `(image, label, name, _)` (or `(image, label, depth, name, _)` with
    use_depth; depth is only used for the list file, like the reference's :936) with any batch size.  The label maps never
    visit the host on the critical path: PseudoLabelPass keeps them on the device, mspl_amd.io.LabelWriter copies and
    encodes them asynchronously while the next batch runs; `in_flight` batches overlap on the GPU (PipelinedLabelPass).

    One process per GPU (torch.distributed initialised): the images are independent (:849 runs them one at a time), so rank r
    labels the batches b == r (mod world) of the loader and nothing crosses ranks on the data path.  The two pieces of
    cross-image state are exchanged once at the end: the class histogram (`class_array`, :887,920-921) by one all-reduce,
    the path lists (:933-940) gathered in loader order; rank 0 writes the list file, every rank returns the same class
    weights.  pre_sharded=True: the caller's loader already yields this rank's batches only (e.g. a sampler over
    dist.shard_indices) -- use it when skipping a foreign batch is not free (the loader decodes it first).
    `_label_pass`: an object with PipelinedLabelPass's interface, for host-logic tests.  batches_per_launch: PipelinedLabelPass's
    `group` (consecutive loader batches labelled by one launch; worth ~5 % at batch 16).  transform: see _label_loop."""
    p = _label_pass if _label_pass is not None else PipelinedLabelPass(
        lambda: PseudoLabelPass(model_list, os_data_list, classes=classes, merge_label_policy=merge_label_policy,
                                device=device, use_graph=use_graph, eval_training=eval_training),
        # eval_training: the images update the models' running statistics in loader order -- one lane, one batch per launch
        depth=1 if eval_training else in_flight, device=device, group=1 if eval_training else batches_per_launch)
    return _label_loop(p, testloader, save_path, lambda out: out, class_weighting, use_depth, device, writer_workers, pre_sharded,
                       transform, _label_pass is not None)


def generate_pseudo_label(model, testloader, save_path, classes=GREENHOUSE_CLASSES, class_weighting='normal', use_depth=False,
                          device='cuda', use_graph=True, writer_workers=None, in_flight=3, batches_per_launch=2, pre_sharded=False,
                          _label_pass=None, transform=None, eval_training=False):
    """uest_seg_multi_os.py:730-830 end to end, the single-model relabelling of every self-training round (called at :527-528):
    loader -> get_output (`softmax(pred + 0.5 aux)`, :795) -> argmax (:798) -> `class_array` (:800-801) -> `<save_path>/pred/<image_name>.png`
    (:803-811) -> path lists (:813-816) -> update_image_list (:820) -> class weights (:822-828); returns (tgt_train_lst, class_weights).

    The loop body is SelfLabelPass (forward + fused epilogue: up-sampling of both heads, argmax of pred + 0.5 aux -- softmax is
    monotone, the probabilities are never written -- and the class histogram), `in_flight` launches of `batches_per_launch` loader
    batches each overlap on the GPU, the PNG files are written by LabelWriter's native threads.  Loader tuples, rank sharding,
    `pre_sharded`, `_label_pass` and `transform` as in generate_pseudo_label_multi_model.  The reference's loader is batch size 1
    (:746); BatchNorm is in eval mode, so any batch size gives the same maps.  eval_training=True is the reference's
    `--eval-training` (:749-752): train() mode under no_grad, i.e. BatchNorm with the statistics of each single image (and the
    momentum updates of the running statistics that come with it) -- `_lowres_batch_stats`; one lane, no graph."""
    p = _label_pass if _label_pass is not None else PipelinedLabelPass(
        lambda: SelfLabelPass(model, classes=classes, device=device, use_graph=use_graph, with_kld=False, eval_training=eval_training),
        depth=1 if eval_training else in_flight, device=device, group=1 if eval_training else batches_per_launch)
    return _label_loop(p, testloader, save_path, lambda out: out[0] if isinstance(out, (tuple, list)) else out, class_weighting,
                       use_depth, device, writer_workers, pre_sharded, transform, _label_pass is not None)


class SelfLabelPass(_GraphedPassMixin):
    """Batched single-model relabelling (the loop body of generate_pseudo_label, uest_seg_multi_os.py:730-830):
    forward -> pred + 0.5*aux -> argmax -> class histogram, plus the KL(pred||aux) uncertainty map that
    get_output computes (:691) -- kept on the device for the uncertainty-weighted loss instead of being
    copied to the host and dropped.  __call__(images) -> (labels uint8 (N,H,W), kld fp32 (N,H,W))."""

    def __init__(self, model, classes=GREENHOUSE_CLASSES, device='cuda', use_graph=False, with_kld=True, eval_training=False):
        self.eval_training = bool(eval_training)        # args.eval_training (:749-752): train() mode, per-image batch statistics
        self.model = model.to(device).train() if self.eval_training else model.to(device).eval()
        self.classes = classes
        self.device = torch.device(device)
        self.hist = torch.zeros(classes, dtype=torch.int64, device=self.device)
        self.use_graph = use_graph and not self.eval_training
        self.with_kld = with_kld
        self._graphs = {}

    def reset(self):
        self.hist.zero_()

    def _run(self, images):
        main, aux = _lowres_batch_stats(self.model, images) if self.eval_training else _lowres(self.model, images)
        if main.shape[1] <= min(24, self.classes, 32) and ops.label_epilogue_hist_fits(main, aux, images.shape[2:]):
            # every argmax is a counted class: the epilogue kernel accumulates the histogram itself (one launch; the S = 1 merge
            # would be the identity on these labels)
            r = ops.label_epilogue_hist(main, aux, images.shape[2:], self.hist, self.classes, want_kld=self.with_kld)
            return r['labels'], r.get('kld')
        r = ops.label_epilogue(main, aux, images.shape[2:], want_kld=self.with_kld)
        # S = 1, threshold 1: out-of-range classes become NO_AGREEMENT_CLASS, the merge kernel accumulates the histogram
        labels = ops.merge_labels([r['labels']], self.classes, 1, NO_AGREEMENT_CLASS, self.hist)
        return labels, r.get('kld')

    def __call__(self, images):
        with torch.no_grad():
            images = images.to(self.device)
            if not self.use_graph:
                return self._run(images)
            return self._replay(images)

    def _graph_models(self):
        return [self.model]

    @staticmethod
    def _graph_outputs(run_result):
        return run_result
