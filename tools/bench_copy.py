import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.bench_ops import timeit
for mb in (2, 8, 22, 44, 106, 400, 1600):
    n = mb * 1024 * 1024 // 8
    x = torch.randn(n, device='cuda'); y = torch.empty_like(x)
    t = timeit(lambda: y.copy_(x))
    print('copy %5d MB total traffic: %8.1f us  %7.1f GB/s' % (mb, t, 2 * n * 4 / t / 1e3))
x = torch.randn(16, device='cuda'); y = torch.empty_like(x)
print('tiny copy (launch floor in graph): %.2f us' % timeit(lambda: y.copy_(x), iters=50))
