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


def _worker(rank, world, port, q):
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
        q.put((rank, 'ok'))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_histogram_and_grad_bucket():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
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
