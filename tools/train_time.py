#!/usr/bin/env python3
"""uest train step timing (bench.py's train_step workload): hipGraph replay with 1 / 2 / 4 micro-batch lanes.
usage: python tools/train_time.py [lanes ...]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspl_amd import models, training
from tests.synth import synth_state_dict
a = argparse.Namespace(s=2.0, channels=3, num_classes=1000)
g = torch.Generator().manual_seed(7)
x = torch.randn((16, 3, 256, 480), generator=g).cuda()
y = torch.randint(0, 5, (16, 256, 480), generator=g).cuda()
for lanes in [int(v) for v in sys.argv[1:]] or [1, 2, 4]:
    m = models.ESPDNetwithUncertaintyEstimation(a, classes=5, dataset='greenhouse', fix_pyr_plane_proj=True)
    m.load_state_dict(synth_state_dict(m.state_dict(), 9))
    m = m.cuda().eval()
    step = training.GraphedTrainStep(m, x, y, torch.ones(5), ignore_idx=4, lanes=lanes)
    for _ in range(3):
        step(x, y)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(10):
            loss = step(x, y)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 10)
    print('lanes %d: %.3f ms/step (min of 3 x 10), loss %.5f' % (lanes, min(ts) * 1e3, float(loss)), flush=True)
    del step, m
