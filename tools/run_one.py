"""Run one op a few times (for rocprofv3 --pmc).  Usage: python tools/run_one.py pyr4|k2_l3|pw_l3exp"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspl_amd import layers as L, ops
from mspl_amd.ops import Epi
what = sys.argv[1]
N = 16
with torch.no_grad():
    if what == 'pyr4':
        m = L.EfficientPyrPool(32, 16, 13, last_layer_br=False).cuda().eval()
        x = torch.randn(N, 16, 144, 240, device='cuda')
        sizes = m.branch_sizes(144, 240)
        for _ in range(3):
            m.forward_fused(x, sizes)
    elif what == 'k2_l3':
        x = torch.randn(N, 64, 36, 60, device='cuda'); w4 = torch.randn(4, 64, 3, 3, device='cuda')
        for _ in range(3):
            ops.eesp_dw_hff(x, w4, [1, 2, 3, 4], 1)
    elif what == 'k2_l4':
        x = torch.randn(N, 128, 18, 30, device='cuda'); w4 = torch.randn(4, 128, 3, 3, device='cuda')
        for _ in range(3):
            ops.eesp_dw_hff(x, w4, [1, 1, 2, 3], 1)
    elif what == 'k2_l2':
        x = torch.randn(N, 24, 144, 240, device='cuda'); w4 = torch.randn(4, 24, 3, 3, device='cuda')
        for _ in range(3):
            ops.eesp_dw_hff(x, w4, [1, 2, 3, 4], 2)
    elif what == 'prep4':
        x = torch.randn(N, 16, 144, 240, device='cuda')
        ws = [torch.randn(16, 1, 3, 3, device='cuda') for _ in range(2)]
        for _ in range(3):
            ops.pyr_down_prep(x, [(72, 120), (15, 24)], ws)
    elif what == 'pw_l4exp':
        x = torch.randn(N, 512, 18, 30, device='cuda'); w = torch.randn(512, 128, 1, 1, device='cuda')
        r = torch.randn(N, 512, 18, 30, device='cuda')
        sc = torch.ones(512, device='cuda')
        for _ in range(3):
            ops.conv1x1(x, w, 4, Epi(sc, sc, sc, residual=r))
    elif what == 'pw_l3exp':
        x = torch.randn(N, 256, 36, 60, device='cuda'); w = torch.randn(256, 64, 1, 1, device='cuda')
        r = torch.randn(N, 256, 36, 60, device='cuda')
        sc = torch.ones(256, device='cuda')
        for _ in range(3):
            ops.conv1x1(x, w, 4, Epi(sc, sc, sc, residual=r))
torch.cuda.synchronize()
