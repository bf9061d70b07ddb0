"""The product's collectives through backend "nccl" (= RCCL on ROCm) on DEVICE buffers, on the one GPU a test box has.

A group of one rank makes every collective the identity, so the numbers prove nothing about the reduction itself (the world-2 gloo
tests in test_dist_cpu.py do that); what this covers is that the RCCL code path of every helper the N > 1 run depends on executes on
hardware: communicator creation, all_reduce on the flat gradient bucket / the int64 histogram / the float64 evaluation sums,
all_gather_object of the path records, the barrier, and a GraphedTrainStep whose all-reduce sits between the graph replay and the Adam
kernel.  `dist.force_collectives()` makes the helpers issue their collective although world == 1.
"""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def nccl_world1():
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip('a process group already exists in this process')
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
    try:
        yield dist
    finally:
        dist.destroy_process_group()


def test_helpers_issue_rccl_collectives_on_device_buffers(nccl_world1):
    from mspl_amd import dist as md
    dist = nccl_world1
    assert dist.get_backend() == 'nccl' and md.world() == (0, 1)
    assert not md.collective_needed()
    with md.force_collectives():
        assert md.collective_needed()
        flat = torch.arange(1024, device='cuda', dtype=torch.float32)
        assert md.all_reduce_mean(flat) is flat and torch.equal(flat.cpu(), torch.arange(1024, dtype=torch.float32))
        hist = torch.tensor([5, 4, 3, 2, 1], device='cuda', dtype=torch.int64)
        assert md.reduce_histogram(hist).tolist() == [5, 4, 3, 2, 1]
        assert md.gather_lists([('a', 1), ('b', 2)]) == [('a', 1), ('b', 2)]
        md.barrier()
        # the gradient bucket of the optimizers: .grad views of ONE flat device buffer, one collective
        torch.manual_seed(0)
        w1 = torch.nn.Parameter(torch.randn(8, 3, device='cuda'))
        w2 = torch.nn.Parameter(torch.randn(13, device='cuda'))
        ((torch.randn(4, 3, device='cuda') @ w1.t()).pow(2).mean() + w2.sum()).backward()
        g1, g2 = w1.grad.clone(), w2.grad.clone()
        b = md.GradBucket([w1, w2])
        b.all_reduce()
        assert torch.equal(w1.grad, g1) and torch.equal(w2.grad, g2) and w1.grad.data_ptr() == b.flat.data_ptr()
        with md.local_only():
            assert not md.collective_needed()
    torch.cuda.synchronize()


def test_eval_sums_all_reduce_float64_on_device(nccl_world1):
    from mspl_amd import dist as md, evaluation as ev

    class Sums(ev.EvalSums):
        K = 4

        def sums(self):
            return torch.tensor([3, 2, 1, 0, 4, 4, 2, 1, 5, 2, 3, 0, 1.5, 3, 1], dtype=torch.float64, device='cuda')
    with md.force_collectives():
        iou, loss = Sums().result()
    iou0, loss0 = Sums().result(reduce=False)
    assert np.array_equal(iou, iou0) and loss == loss0 == 0.5


def test_train_step_with_rccl_all_reduce_between_graph_and_adam(nccl_world1):
    """GraphedTrainStep under an initialised nccl group: the flat-bucket all-reduce runs (forced, world 1) after the graph replay and
    before the Adam kernel, and the step equals the same step without a process group collective."""
    import argparse
    from mspl_amd import dist as md, models, training
    from tests.synth import synth_input, synth_state_dict

    def build():
        a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
        m = models.ESPDNetwithUncertaintyEstimation(a, classes=5, dataset='greenhouse', fix_pyr_plane_proj=True)
        m.load_state_dict(synth_state_dict(m.state_dict(), 5))
        return m.cuda().eval()
    x = synth_input((2, 3, 64, 96), 3).cuda()
    y = (torch.arange(2 * 64 * 96, device='cuda').reshape(2, 64, 96) % 5).to(torch.int64)
    cw = torch.ones(5, device='cuda')
    import contextlib
    outs = []
    for forced in (False, True):
        m = build()
        with (md.force_collectives() if forced else contextlib.nullcontext()):
            step = training.GraphedTrainStep(m, x, y, cw, ignore_idx=4, lr=5e-4, weight_decay=5e-4)   # (its first, eager step too)
            losses = [float(step(x, y)) for _ in range(2)]
        outs.append((losses, step.optimizer.flat_p.detach().cpu().numpy()))
    # float atomics order the gradient sums differently from run to run: same tolerances as test_micro_batch_lanes_equal_one_graph
    np.testing.assert_allclose(outs[1][0], outs[0][0], rtol=2e-4, atol=1e-6)
    from tests.synth import assert_weights_close_after_adam
    assert_weights_close_after_adam(outs[1][1], outs[0][1], lr=5e-4, steps=3)
