"""One-process-per-GPU helpers (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" on CPU for tests).

The reference's only multi-GPU mechanism is single-process nn.DataParallel (utilities/parallel_wrapper.py:17-106:
per-step parameter broadcast, input scatter, gradient reduce to GPU 0).  Here weights stay resident on every rank:
  * label pass  -- embarrassingly parallel over images (uest_seg_multi_os.py:849: batch_size=1, no cross-image state);
                   the only exchange is the final sum of the class histogram (:887,920-921);
  * train step  -- one all-reduce of a flat fp32 bucket holding the gradients of the parameters that actually receive
                   gradients (340 of 570 tensors); loss = mean of per-rank means, which equals
                   DataParallelCriteria + .mean() (utilities/train_eval_seg.py:202) for equal shards.
"""
import torch
import torch.distributed as dist


_LOCAL_ONLY = [False]


class local_only(object):
    """with local_only(): every helper of this module behaves as in a single process (no collective is issued) although a process
    group exists.  For set-up work that may fail on ONE rank (graph capture, allocation): a rank that raises inside a collective
    section leaves the others blocked in theirs; do the risky part locally, agree on success with one all-reduce, then enter the
    collective section (bench.py --gpus N)."""

    def __enter__(self):
        self.prev = _LOCAL_ONLY[0]
        _LOCAL_ONLY[0] = True

    def __exit__(self, *exc):
        _LOCAL_ONLY[0] = self.prev


_FORCE = [False]


class force_collectives(object):
    """with force_collectives(): the helpers issue their collective even when the group has ONE rank (where it is the identity).
    Lets a one-GPU box execute the RCCL code path of the product (backend "nccl", device buffers) -- tests/test_gpu_rccl.py."""

    def __enter__(self):
        self.prev = _FORCE[0]
        _FORCE[0] = True

    def __exit__(self, *exc):
        _FORCE[0] = self.prev


def collective_needed():
    """True when a helper must issue its collective: more than one rank, or a forced one-rank group."""
    if not (dist.is_available() and dist.is_initialized()) or _LOCAL_ONLY[0]:
        return False
    return dist.get_world_size() > 1 or _FORCE[0]


def world():
    if dist.is_available() and dist.is_initialized() and not _LOCAL_ONLY[0]:
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_indices(n_items, rank=None, world_size=None):
    """Indices of the image list handled by this rank: i == rank (mod world), like a DistributedSampler without padding."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    return list(range(rank, n_items, world_size))


def reduce_histogram(hist):
    """Sum the per-rank int64 class histograms in place (the label pass's only collective)."""
    if collective_needed():
        dist.all_reduce(hist, op=dist.ReduceOp.SUM)
    return hist


def gather_lists(local_items):
    """Rank-ordered concatenation of per-rank python lists (image / label path lists of update_image_list)."""
    _, w = world()
    if not collective_needed():
        return list(local_items)
    out = [None] * w
    dist.all_gather_object(out, list(local_items))
    n = max(len(o) for o in out)
    merged = []
    for i in range(n):          # undo the i == rank (mod world) sharding
        for o in out:
            if i < len(o):
                merged.append(o[i])
    return merged


def all_reduce_mean(flat):
    """Average a flat tensor over the ranks in place (sum / world): the train step's only collective, RCCL over xGMI on
    GPUs.  Equals the reference's mean of per-replica mean losses (utilities/train_eval_seg.py:202) for equal shards."""
    _, w = world()
    if collective_needed():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(w)
    return flat


def barrier():
    if collective_needed():
        dist.barrier()


class GradBucket:
    """Flat fp32 bucket over the parameters that receive gradients -- the ONE implementation of the flat layout: FlatAdam and
    FlatSGD (mspl_amd/training.py, supervised.py) build their buffers through it.

    Call after the first backward (which reveals the unused parameters, exactly the set torch.optim skips):
    `GradBucket(model.parameters())`.  The parameters' .grad tensors become views into one contiguous buffer `flat`, so
    `all_reduce()` is a single collective and the optimizer kernel runs on the flat views.  With `with_params=True` the
    parameters' .data are re-pointed into a second flat buffer `flat_p` in the same layout (values preserved).  `params`
    keeps the order given (FlatSGD lays its learning-rate groups out one after the other)."""

    def __init__(self, params, with_params=False):
        self.params = [p for p in params if p.requires_grad and p.grad is not None]
        if not self.params:
            raise RuntimeError('GradBucket: no parameter has a gradient yet (run one backward first)')
        dev = self.params[0].device
        # Every parameter starts at a multiple of ALIGN floats (16 bytes): the HIP kernels read weights as 16-byte vectors and
        # fall back to slower forms for unaligned tensors (a 13-element PReLU slope in front of a convolution weight would
        # misalign everything behind it).  The padding elements are zero in both buffers and stay zero under SGD / Adam.
        n = sum(self._padded(p.numel()) for p in self.params)
        self.numel_params = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_p = torch.zeros(n, dtype=torch.float32, device=dev) if with_params else None
        self.offsets = []
        off = 0
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                view = self.flat[off:off + k].view_as(p)
                view.copy_(p.grad)
                p.grad = view
                if with_params:
                    pv = self.flat_p[off:off + k].view_as(p)
                    pv.copy_(p.data)
                    p.data = pv
                self.offsets.append(off)
                off += self._padded(k)

    ALIGN = 4

    @classmethod
    def _padded(cls, k):
        return (k + cls.ALIGN - 1) // cls.ALIGN * cls.ALIGN

    def span(self, first, count):
        """[lo, hi) of the flat buffers covering parameters first .. first+count-1 (with their padding)."""
        if count <= 0:
            lo = self.offsets[first] if first < len(self.offsets) else self.flat.numel()
            return lo, lo
        last = first + count - 1
        return self.offsets[first], self.offsets[last] + self._padded(self.params[last].numel())

    def numel(self):
        return self.flat.numel()

    def zero(self):
        self.flat.zero_()

    def all_reduce(self):
        """Average the gradients over the ranks (sum / world)."""
        return all_reduce_mean(self.flat)
