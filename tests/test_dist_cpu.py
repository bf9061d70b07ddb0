"""World-size-2 gloo tests of the sharding / reduction helpers (CPU; the N>1 path of the label pass and of the
gradient bucket).  Spawns two processes on 127.0.0.1."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _toy(seed):
    torch.manual_seed(seed)
    net = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.Tanh(), torch.nn.Linear(4, 2))
    unused = torch.nn.Parameter(torch.randn(5))           # never receives a gradient: stays out of the bucket (Appendix B-5)
    return net, unused


def _check_flat_optimizers(rank, world):
    """The all-reduce the PRODUCT uses: FlatAdam.all_reduce_grads (training.py) and FlatSGD.all_reduce_grads (supervised.py)
    both go through dist.GradBucket.  Per-rank mean gradients of equal shards must average to the full-batch gradient, the
    flat layout must follow the parameter (group) order, and gradient-less parameters must stay out."""
    from mspl_amd import supervised, training
    x = torch.randn(8, 3, generator=torch.Generator().manual_seed(3))
    for kind in ('adam', 'sgd'):
        net, unused = _toy(1)
        ref, _ = _toy(1)
        net(x[rank::world]).pow(2).mean().backward()
        ref(x).pow(2).mean().backward()                 # equal shards: mean of the shard means == full-batch mean
        before = [p.detach().clone() for p in net.parameters()]
        if kind == 'adam':
            opt = training.FlatAdam(list(net.parameters()) + [unused], lr=1e-3)
            order = list(net.parameters())
        else:
            groups = [{'params': list(net[2].parameters()), 'lr': 0.1}, {'params': list(net[0].parameters()) + [unused], 'lr': 0.01}]
            opt = supervised.FlatSGD(groups, lr=0.1, momentum=0.9)
            order = list(net[2].parameters()) + list(net[0].parameters())
            pad = lambda k: (k + 3) // 4 * 4                      # every parameter starts at a multiple of 16 bytes
            n2, n0 = (sum(pad(p.numel()) for p in net[i].parameters()) for i in (2, 0))
            assert [(g['_lo'], g['_hi']) for g in opt.param_groups] == [(0, n2), (n2, n2 + n0)]
            assert all(o % 4 == 0 for o in opt.bucket.offsets)
        assert opt.bucket.params == order and unused.grad is None and opt.flat_g is opt.bucket.flat
        assert all(torch.equal(a, p.detach()) for a, p in zip(before, net.parameters()))       # values preserved by the re-pointing
        assert all(p.data_ptr() == opt.flat_p[o:].data_ptr() and p.grad.data_ptr() == opt.flat_g[o:].data_ptr()
                   for p, o in zip(order, opt.bucket.offsets))
        opt.all_reduce_grads()
        for p, r in zip(net.parameters(), ref.parameters()):
            assert torch.allclose(p.grad, r.grad, atol=1e-6), kind
        opt.zero_grad()
        assert float(opt.flat_g.abs().sum()) == 0.0 and all(float(p.grad.abs().sum()) == 0.0 for p in net.parameters())


class _StubPipelinedPass:
    """PipelinedLabelPass's interface with one batch in flight and no GPU: labels = a deterministic function of the batch."""

    def __init__(self):
        self.hist = torch.zeros(5, dtype=torch.int64)
        self._held = None

    def reset(self):
        self.hist.zero_()

    def _label(self, images):
        lab = (images.sum(1).round().to(torch.int64) % 5).to(torch.uint8)
        self.hist += torch.bincount(lab.flatten().to(torch.int64), minlength=5)
        return lab

    def __call__(self, images):
        out, self._held = self._held, self._label(images)
        return out

    def flush(self):
        if self._held is not None:
            out, self._held = self._held, None
            yield out


def _check_sharded_label_function(rank, world, tmp):
    """generate_pseudo_label_multi_model under world 2: every rank labels its own batches, the histogram is all-reduced, the
    path lists come back in loader order, rank 0 alone writes tgt_train.lst, both ranks return the same class weights."""
    import numpy as np
    from mspl_amd import io as mio, uest
    from oracle import imageio as oio
    g = torch.Generator().manual_seed(11)
    sizes = [2, 1, 3, 2, 1]                              # ragged batches, odd count: rank 0 gets 3 batches, rank 1 gets 2
    batches, k = [], 0
    for n in sizes:
        names = ['/data/color/frame_%03d.jpg' % (k + i) for i in range(n)]
        batches.append((torch.rand(n, 3, 8, 12, generator=g) * 4, None, names, None))
        k += n

    def loader():
        for b in batches:
            yield b
    save = tmp + '/shared'                               # both ranks write into the same directory tree, like a shared run dir
    lst, w = uest.generate_pseudo_label_multi_model(None, None, loader(), save, writer_workers=2, _label_pass=_StubPipelinedPass())
    stub = _StubPipelinedPass()
    all_labels = [stub._label(b[0]) for b in batches]
    expect_w = uest.class_weights_from_histogram(stub.hist.numpy(), 'normal')
    assert np.allclose(w.numpy(), expect_w.astype(np.float32)), (w, expect_w)
    import os.path as osp
    assert osp.isfile(lst)                               # visible to every rank after the function's barrier
    images, masks = mio.read_image_list(lst, check_files=False)[:2]
    flat_names = [n for b in batches for n in b[2]]
    assert images == flat_names, images                  # loader order restored across the ranks
    assert masks == ['%s/pred/frame_%03d.png' % (save, i) for i in range(len(flat_names))]
    # this rank wrote exactly its own batches' files, with the right pixels
    mine = [i for i in range(len(batches)) if i % world == rank]
    at = np.cumsum([0] + sizes)
    for bi in mine:
        for j in range(sizes[bi]):
            arr = oio.png_decode_gray8(open(masks[at[bi] + j], 'rb').read())
            assert np.array_equal(arr, all_labels[bi][j].numpy())


def _check_sharded_self_label_function(rank, world, tmp):
    """generate_pseudo_label (uest_seg_multi_os.py:730-830) under world 2 with a stub pass that returns SelfLabelPass's (labels, kld)
    pairs and depth paths in the list: rank r labels batches b == r (mod world); list lines in loader order with the third column
    (:815-816), one histogram all-reduce, both ranks return the oracle loop's class weights."""
    import numpy as np
    from mspl_amd import io as mio, uest
    from oracle import imageio as oio, labels as olab
    g = torch.Generator().manual_seed(12)
    sizes = [1, 3, 2, 2, 1, 2, 1]
    batches, k = [], 0
    for n in sizes:
        names = ['/data/color/im_%03d.png' % (k + i) for i in range(n)]
        batches.append((torch.randn(n, 5, 8, 12, generator=g), None, None, names, None))
        k += n

    class Stub(_StubPipelinedPass):
        def _label(self, images):
            lab = images.argmax(1).to(torch.uint8)        # the batch IS the logits: forward = identity for both heads
            self.hist += torch.bincount(lab.flatten().to(torch.int64), minlength=5)
            return lab, torch.zeros(lab.shape)
    save = tmp + '/self'
    lst, w = uest.generate_pseudo_label(None, iter(batches), save, use_depth=True, writer_workers=2, _label_pass=Stub())
    ri, rl, rd, rmaps, rw = olab.generate_pseudo_label(lambda x: (x, torch.zeros_like(x)), batches, 5, save + '/pred', 'normal', True)
    assert np.allclose(w.numpy(), rw.astype(np.float32), rtol=1e-6), (w, rw)
    assert mio.read_image_list(lst, use_depth=True, check_files=False) == (ri, rl, rd)
    assert rd[0] == '/data/depth/im_000.png'
    at = np.cumsum([0] + sizes)
    for bi in [i for i in range(len(batches)) if i % world == rank]:
        for j in range(sizes[bi]):
            assert np.array_equal(oio.png_decode_gray8(open(rl[at[bi] + j], 'rb').read()), rmaps[at[bi] + j])


def _check_sharded_eval(rank, world):
    """evaluation.val_seg_ue under world 2 with a CPU stand-in for the device pass: rank r evaluates batches b == r (mod world), ONE
    all-reduce of the sums, every rank returns what the reference loop gives on the whole loader (oracle.labels.val_seg_ue)."""
    import numpy as np
    import torch.nn.functional as F
    from mspl_amd import evaluation as ev
    from oracle import labels as olab
    C, K = 6, 5
    g = torch.Generator().manual_seed(23)
    sizes = [2, 3, 1, 2, 2]                               # ragged batches, odd count
    batches = []
    for n in sizes:
        y = torch.randint(0, C, (n, 10, 14), generator=g)
        y[torch.rand(n, 10, 14, generator=g) < 0.1] = 255
        batches.append((torch.randn(n, 3, 10, 14, generator=g), y))
    proj = torch.randn(C, 3, generator=g)
    cw = torch.rand(C, generator=g) + 0.5

    def fwd(x):
        main = torch.einsum('cj,njhw->nchw', proj, x)
        return main, main.flip(1) * 0.3

    class Stub(ev.EvalSums):
        def __init__(self):
            self.K = K
            self.areas = torch.zeros(3, K, dtype=torch.float64)
            self.acc = torch.zeros(2, dtype=torch.float64)
            self.batches = 0

        def __call__(self, images, labels, depth=None):
            main, aux = fwd(images)
            out = main + 0.5 * aux
            pred = (out.argmax(1).to(torch.uint8) + 1) * ((labels.to(torch.uint8) + 1) > 0)
            tgt = labels.to(torch.uint8) + 1
            inter = pred * (pred == tgt)
            for i, t in enumerate((inter, pred, tgt)):
                self.areas[i] += torch.histc(t.float(), bins=K, min=1, max=K).double()
            loss = F.cross_entropy(out, labels, weight=cw, ignore_index=255)
            self.acc[0] += float(loss) * images.shape[0]
            self.acc[1] += images.shape[0]
            self.batches += 1

        def sums(self):
            return torch.cat([self.areas.reshape(-1), self.acc, torch.tensor([float(self.batches)], dtype=torch.float64)])

    class Crit:
        loss_type, class_wts, ignore_idx = 'ce', cw, 255
    stub = Stub()
    iou, loss = ev.val_seg_ue(None, batches, criterion=Crit(), num_classes=C, device='cpu', _eval_pass=stub)
    assert stub.batches == len([b for b in range(len(batches)) if b % world == rank])      # this rank saw only its own batches
    ref_iou, ref_loss = olab.val_seg_ue(fwd, batches, cw, 255, C, aux_weight=0.5)
    assert np.allclose(iou, ref_iou, rtol=0, atol=1e-6), (iou, ref_iou)      # (the reference adds its 1e-6 in float32)
    assert abs(loss - ref_loss) < 1e-6, (loss, ref_loss)
    iou2, zero = ev.val_seg_ue(None, batches, criterion=None, num_classes=C, device='cpu', _eval_pass=Stub())
    assert zero == 0 and np.allclose(iou2, ref_iou, atol=1e-6)


def _worker(rank, world, port, q, tmp):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from mspl_amd import dist as md
    try:
        n = 11
        mine = md.shard_indices(n)
        # every image exactly once across ranks
        allidx = md.gather_lists(mine)
        assert allidx == list(range(n)), allidx
        # histogram: each rank counts its own shard; the sum equals the single-process count
        labels = torch.arange(n) % 5
        hist = torch.bincount(labels[mine], minlength=5).to(torch.int64)
        md.reduce_histogram(hist)
        assert torch.equal(hist, torch.bincount(labels, minlength=5))
        # gradient bucket: per-rank mean gradients of equal shards average to the full-batch gradient;
        # parameters without gradients stay out of the bucket and untouched
        torch.manual_seed(0)
        w1 = torch.nn.Parameter(torch.randn(4, 3))
        w2 = torch.nn.Parameter(torch.randn(3))
        unused = torch.nn.Parameter(torch.randn(2))
        x = torch.randn(8, 3)
        shard = x[rank::world]
        loss = ((shard @ w1.t()).pow(2).mean() + (shard * w2).sum(1).mean())
        loss.backward()
        b = md.GradBucket([w1, w2, unused])
        assert len(b.params) == 2 and unused.grad is None
        b.all_reduce()
        w1r, w2r = w1.detach().clone().requires_grad_(True), w2.detach().clone().requires_grad_(True)
        full = sum(((x[r::world] @ w1r.t()).pow(2).mean() + (x[r::world] * w2r).sum(1).mean()) for r in range(world)) / world
        full.backward()
        assert torch.allclose(w1.grad, w1r.grad, atol=1e-6) and torch.allclose(w2.grad, w2r.grad, atol=1e-6)
        assert w1.grad.data_ptr() == b.flat.data_ptr()      # grads are views of the flat bucket
        _check_flat_optimizers(rank, world)
        _check_sharded_label_function(rank, world, tmp)
        _check_sharded_self_label_function(rank, world, tmp)
        _check_sharded_eval(rank, world)
        q.put((rank, 'ok'))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_histogram_and_grad_bucket(tmp_path):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, 'ok'), (1, 'ok')], res


def test_single_process_defaults():
    from mspl_amd import dist as md
    assert md.world() == (0, 1)
    assert md.shard_indices(5) == [0, 1, 2, 3, 4]
    assert md.shard_indices(7, rank=1, world_size=3) == [1, 4]
    h = torch.tensor([1, 2, 3])
    assert md.reduce_histogram(h) is h
    assert md.gather_lists(['a', 'b']) == ['a', 'b']
