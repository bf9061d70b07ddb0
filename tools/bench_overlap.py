#!/usr/bin/env python3
"""Throughput of the single-source label pass with ONE pass in flight (one hipGraph replayed back to back on one stream) vs TWO
independent batches in flight (two graph instances with their own static buffers, replayed alternately on two streams)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspl_amd import models, uest
from tests.synth import synth_state_dict

a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
m = models.ESPDNetwithUncertaintyEstimation(a, classes=13, dataset='camvid', fix_pyr_plane_proj=True)
m.load_state_dict(synth_state_dict(m.state_dict(), 0))
m = m.cuda().eval()
x = torch.randn(16, 3, 288, 480, device='cuda')
K = 60


def capture(stream):
    p = uest.SelfLabelPass(m, classes=13, use_graph=True)
    with torch.cuda.stream(stream):
        p(x)
        p(x)
    stream.synchronize()
    graph, static_in, static_out = p._graphs[tuple(x.shape)]
    return p, graph


NL = int(sys.argv[1]) if len(sys.argv) > 1 else 2
SEQ = len(sys.argv) > 2 and sys.argv[2] == 'seq'
lanes, keep = [], []                                   # `keep`: the pass objects own the graphs' static buffers
for _ in range(NL):
    s_ = torch.cuda.Stream()
    p_, g_ = capture(s_)
    keep.append(p_)
    lanes.append((s_, g_))
torch.cuda.synchronize()
print('captured', NL, 'lanes', flush=True)
for label, plan in [('%d in flight%s' % (n, ' (sequential: sync after every replay)' if SEQ else ''), lanes[:n]) for n in range(1, NL + 1)]:
    for _ in range(5):
        for s, g in plan:
            with torch.cuda.stream(s):
                g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        s, g = plan[i % len(plan)]
        with torch.cuda.stream(s):
            g.replay()
        if SEQ:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print('%s: %.3f ms per batch of 16 -> %.0f images/s' % (label, dt * 1e3, 16 / dt), flush=True)
