"""The uest loss at head resolution against the three-step form at the bench shape (16 and 4 x 5 x 256x480), for rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspl_amd import autograd as ag
g = torch.Generator().manual_seed(1)
for n in (16, 4):
    m = torch.randn(n, 5, 128, 240, generator=g).cuda().requires_grad_(True)
    a = torch.randn(n, 5, 64, 120, generator=g).cuda().requires_grad_(True)
    t = torch.randint(0, 5, (n, 256, 480), generator=g).cuda()
    cw = torch.ones(5).cuda()
    for _ in range(5):
        ag.uw_loss_heads(m, a, t, cw).backward()
    for _ in range(5):
        ag.uw_loss(ag.bilinear(m, (256, 480)), ag.bilinear(a, (256, 480)), t, cw).backward()
torch.cuda.synchronize()
