import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspl_amd._native import lib, check
from mspl_amd.ops import _p, _stream
shapes = [(2, 13, 32, 48, 64, 96), (2, 13, 16, 24, 64, 96), (2, 32, 16, 24, 32, 48), (2, 16, 4, 6, 8, 12), (2, 16, 8, 12, 16, 24), (2, 16, 5, 6, 8, 12),
          (2, 16, 5, 6, 4, 6), (2, 16, 8, 12, 32, 48), (2, 8, 16, 24, 32, 48), (2, 16, 5, 5, 8, 12), (2, 16, 5, 12, 16, 24), (2, 16, 16, 24, 32, 48)]
g = torch.Generator().manual_seed(3)
for (N, C, Hi, Wi, Ho, Wo) in shapes:
    gy = torch.randn(N, C, Ho, Wo, generator=g).cuda()
    gx = torch.full((N, C, Hi, Wi), float('nan'), device='cuda')
    check(lib.mspl_bilinear_bwd(_p(gy), N, C, Hi, Wi, Ho, Wo, _p(gx), _stream()))
    ref = gx.clone()
    gx2 = torch.full((N, C, Hi, Wi), float('nan'), device='cuda')
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        check(lib.mspl_bilinear_bwd(_p(gy), N, C, Hi, Wi, Ho, Wo, _p(gx2), _stream()))
    gr.replay(); torch.cuda.synchronize()
    d = (gx2 - ref).abs()
    print((N, C, Hi, Wi, Ho, Wo), 'equal' if torch.equal(gx2, ref) else 'DIFF max %g nan %d' % (float(d[~d.isnan()].max()) if (~d.isnan()).any() else -1, int(gx2.isnan().sum())))
