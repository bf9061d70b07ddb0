#!/usr/bin/env python3
"""BASELINE configs[4]: ASPP_Bottleneck(num_classes=20) on (16, 2048, 32, 64) = 1024x512 at OS16 -- time per batch, images/s,
TFLOP/s against the fp32 matrix-core peak (157.3 TF), and the three dilated convs alone."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspl_amd import aspp, ops
from tests.synth import synth_state_dict

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = 'cuda'
m = aspp.ASPP_Bottleneck(num_classes=20)
m.load_state_dict(synth_state_dict(m.state_dict(), 0))
m = m.to(dev).eval()
x = torch.randn(N, 2048, 32, 64, device=dev)
with torch.no_grad():
    for _ in range(2):
        m(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = m(x)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
macs = N * 32 * 64 * (2048 * 256 * (1 + 27) + 1280 * 256 + 256 * 20) + N * 2048 * 256
print('ASPP_Bottleneck bs%d 2048x32x64: %.2f ms/batch -> %.0f img/s, %.1f TFLOP/s (%.1f%% of 157.3 TF fp32 MFMA)'
      % (N, dt * 1e3, N / dt, 2 * macs / dt / 1e12, 100 * 2 * macs / dt / 157.3e12))
with torch.no_grad():
    wp = ops.pack_dense_weight(m.conv_3x3_2.weight)
    out = torch.empty(N, 256, 32, 64, device=dev)
    for _ in range(2):
        ops.dense_conv(x, wp, 3, 12, out=(out, 0))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ops.dense_conv(x, wp, 3, 12, out=(out, 0))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
fl = 2 * N * 32 * 64 * 2048 * 256 * 9
print('dense 3x3 d=12 alone: %.2f ms -> %.1f TFLOP/s (%.1f%%)' % (dt * 1e3, fl / dt / 1e12, 100 * fl / dt / 157.3e12))
