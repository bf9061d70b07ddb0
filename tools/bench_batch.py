"""Label-pass throughput vs images per launch: one hipGraph, one pass in flight, N = 16..64 (is a merged batch faster than
several batches of 16 in flight?)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspl_amd import models, uest
from tests.synth import synth_state_dict
a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
m = models.ESPDNetwithUncertaintyEstimation(a, classes=13, dataset='camvid', fix_pyr_plane_proj=True)
m.load_state_dict(synth_state_dict(m.state_dict(), 0))
for N in [int(v) for v in (sys.argv[1:] or ['16', '32', '48', '64'])]:
    x = torch.randn(N, 3, 288, 480, device='cuda')
    for depth in (1, 2, 3):
        plp = uest.PipelinedLabelPass(lambda: uest.SelfLabelPass(m, classes=13, device='cuda', use_graph=True), depth=depth)
        for _ in range(2 * depth + 2):
            plp(x)
        list(plp.flush())
        xs = [xi if xi is not None else x for xi in plp.static_inputs(x.shape)]
        for xi in xs:
            xi.copy_(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        K = 60
        for _ in range(K):
            plp(xs[plp.next_lane])
        list(plp.flush())
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
        print('N=%d in flight=%d: %.3f ms/launch -> %.0f images/s' % (N, depth, dt * 1e3, N / dt), flush=True)
        del plp
