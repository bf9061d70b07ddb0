"""BatchNorm-train node (statistics + apply, backward sums + apply) on the supervised loop's tensor shapes, for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspl_amd import autograd as ag
shapes = [(16, 32, 144, 240), (16, 64, 72, 120), (16, 128, 36, 60), (16, 256, 18, 30), (16, 16, 144, 240), (16, 512, 9, 15)]
for (N, C, H, W) in shapes:
    bn = torch.nn.BatchNorm2d(C).cuda().train()
    al = torch.full((C,), 0.25, device='cuda', requires_grad=True)
    z = torch.randn(N, C, H, W, device='cuda', requires_grad=True)
    go = torch.randn(N, C, H, W, device='cuda')
    for _ in range(5):
        y = ag.bn_train_prelu(z, bn, al)
        y.backward(go)
torch.cuda.synchronize()
