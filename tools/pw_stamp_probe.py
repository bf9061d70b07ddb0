#!/usr/bin/env python3
"""STAMPS=1 build + MSPL_PW_STAMP=1: phase timeline of the 1x1 kernels at the level-3 / level-4 shapes (eager launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspl_amd import ops
from mspl_amd.ops import Epi
N = 16
for name, ci, co, g, h, w, res in [('L4 proj 512->128', 512, 128, 4, 18, 30, False), ('L4 exp 512->512 +res', 512, 512, 4, 18, 30, True),
                                   ('L3 proj 256->64', 256, 64, 4, 36, 60, False), ('L3 exp 256->256 +res', 256, 256, 4, 36, 60, True)]:
    x = torch.randn(N, ci, h, w, device='cuda')
    wt = torch.randn(co, ci // g, 1, 1, device='cuda') * 0.1
    sc, sh, al = torch.rand(co, device='cuda') + 0.5, torch.randn(co, device='cuda'), torch.rand(co, device='cuda') * 0.3
    r = torch.randn(N, co, h, w, device='cuda') if res else None
    out = torch.empty(N, co, h, w, device='cuda')
    ep = Epi(sc, sh, al, residual=r)
    print('==', name, file=sys.stderr, flush=True)
    for _ in range(4):
        ops.conv1x1(x, wt, g, ep, out=(out, 0))
    torch.cuda.synchronize()
