#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspl_amd import ops
from mspl_amd.ops import Epi
from tools.bench_ops import timeit
N = 16
for name, ci, co, g, h, w in [('l3 PWConv 128->48 g16 72x120', 128, 48, 16, 72, 120), ('l2 PWConv 256->64 g64 36x60', 256, 64, 64, 36, 60),
                              ('inp_reinf 3->3 72x120', 3, 3, 1, 72, 120), ('inp_reinf 3->3 36x60', 3, 3, 1, 36, 60)]:
    x = torch.randn(N, ci, h, w, device='cuda')
    wt = torch.randn(co, ci // g, 3, 3, device='cuda') * 0.1
    sc, sh, al = torch.rand(co, device='cuda') + 0.5, torch.randn(co, device='cuda'), torch.rand(co, device='cuda') * 0.3
    out = torch.empty(N, co, h, w, device='cuda')
    ep = Epi(sc, sh, al)
    t = timeit(lambda: ops.conv3x3(x, wt, g, 1, 0, ep, out=(out, 0)))
    by = 4 * N * h * w * (ci + co)
    print('%-32s %8.1f us  %7.1f GB/s' % (name, t, by / t / 1e3))
