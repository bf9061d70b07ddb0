#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspl_amd import ops
from mspl_amd.ops import Epi
from tools.bench_ops import timeit
for N in (16, 32):
    Cin, n, G, H, W, dil = 512, 128, 4, 18, 30, [1, 1, 2, 3]
    x = torch.randn(N, Cin, H, W, device='cuda')
    wp = torch.randn(n, Cin // G, 1, 1, device='cuda') * 0.1
    ps, pb, pa = torch.rand(n, device='cuda') + 0.5, torch.randn(n, device='cuda'), torch.rand(n, device='cuda') * 0.3
    w = torch.randn(4, n, 3, 3, device='cuda') * 0.3
    sc, sh, al = torch.rand(4 * n, device='cuda') + 0.5, torch.randn(4 * n, device='cuda'), torch.rand(4 * n, device='cuda') * 0.3
    ep, pe = Epi(sc, sh, al), Epi(ps, pb, pa)
    out = torch.empty(N, 4 * n, H, W, device='cuda')
    o1 = torch.empty(N, n, H, W, device='cuda')
    tf = timeit(lambda: ops.eesp_proj_dw_hff(x, wp, ps, pb, pa, w, dil, G, ep, out=(out, 0)))
    t1 = timeit(lambda: ops.conv1x1(x, wp, G, pe, out=(o1, 0)))
    t2 = timeit(lambda: ops.eesp_dw_hff(o1, w, dil, 1, ep, out=(out, 0)))
    print('N=%d  fused %.1f us   proj %.1f + K2 %.1f = %.1f us' % (N, tf, t1, t2, t1 + t2))
