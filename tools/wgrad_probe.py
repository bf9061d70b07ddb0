"""mspl_conv1x1_wgrad_batch on the problem shapes of a uest train step (tools/wgrad_census.py), one problem per launch and the level-4
run of 16 as one launch: us and operand GB/s.  With the tuning library, MSPL_WGRAD_QUAD / MSPL_WGRAD_* select the seating."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspl_amd._native import check, lib
shapes = [(16, 512, 512, 4, 480), (16, 512, 128, 4, 480), (16, 256, 256, 4, 1920), (16, 256, 64, 4, 1920), (16, 128, 128, 4, 1920),
          (16, 128, 32, 4, 7680), (16, 96, 96, 4, 7680), (16, 32, 24, 4, 30720), (16, 32, 16, 1, 30720), (16, 16, 5, 1, 30720),
          (16, 48, 16, 1, 7680), (16, 512, 16, 1, 480), (4, 512, 512, 4, 480), (4, 256, 256, 4, 1920)]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(probs, reps=20):
    n = len(probs)
    gy = [torch.randn(N, Co, HW, device='cuda') for N, Ci, Co, G, HW in probs]
    x = [torch.randn(N, Ci, HW, device='cuda') for N, Ci, Co, G, HW in probs]
    gw = [torch.zeros(Co, Ci // G, device='cuda') for N, Ci, Co, G, HW in probs]
    arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
    col = lambda k: (ctypes.c_int32 * n)(*[p[k] for p in probs])
    rs = (ctypes.c_void_p * n)(*[None] * n)
    call = lambda: check(lib.mspl_conv1x1_wgrad_batch(arr(gy), arr(x), arr(gw), rs, col(0), col(1), col(2), col(3), col(4), n, st))
    flush = torch.empty(512 << 20, dtype=torch.uint8, device='cuda')
    ts = []
    for _ in range(reps):
        flush.zero_()                       # operands out of L2 / the Infinity Cache, as in a step
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]
for p in shapes:
    us = run([p])
    N, Ci, Co, G, HW = p
    mb = 4.0 * N * HW * (Ci + Co) / 1e6
    print('N %2d %4d -> %4d g%d HW %6d: %7.1f us  %6.1f MB  %5.2f TB/s  %5.1f TFLOP/s' % (N, Ci, Co, G, HW, us, mb, mb / us, 2.0 * N * HW * Ci * Co / G / us / 1e6))
lvl4 = [(16, 512, 512, 4, 480), (16, 512, 128, 4, 480)] * 7 + [(16, 256, 256, 4, 480), (16, 256, 64, 4, 1920)]
us = run(lvl4)
mb = sum(4.0 * N * HW * (Ci + Co) for N, Ci, Co, G, HW in lvl4) / 1e6
print('run of 16 (level 4): %.1f us, %.1f MB, %.2f TB/s' % (us, mb, mb / us))
