"""The two loss kernels at the bench shapes (uest train step 16 x 5 x 256x480 two heads; supervised 16 x 13 x 288x480), for rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspl_amd import autograd as ag, losses
g = torch.Generator().manual_seed(1)
p = torch.randn(16, 5, 256, 480, generator=g).cuda().requires_grad_(True)
a = torch.randn(16, 5, 256, 480, generator=g).cuda().requires_grad_(True)
t = torch.randint(0, 5, (16, 256, 480), generator=g).cuda()
cw = torch.ones(5).cuda()
for _ in range(5):
    l = ag.uw_loss(p, a, t, cw)
    l.backward()
p4, a4, t4 = p[:4].detach().clone().requires_grad_(True), a[:4].detach().clone().requires_grad_(True), t[:4].clone()
for _ in range(5):
    ag.uw_loss(p4, a4, t4, cw).backward()
crit = losses.SegmentationLoss(n_classes=13, device='cuda', ignore_idx=255)
x = torch.randn(16, 13, 288, 480, generator=g).cuda().requires_grad_(True)
y = torch.randint(0, 13, (16, 288, 480), generator=g).cuda()
for _ in range(5):
    crit(x, y).mean().backward()
torch.cuda.synchronize()
