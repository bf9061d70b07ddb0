#!/usr/bin/env python3
"""Micro-benchmark single ops at the shapes of ESPDNet-UE s=2.0 (bs=16, 288x480): time per launch (HIP events,
median of batches), algorithmic GB/s and GFLOP/s.  Usage: python tools/bench_ops.py [conv1x1|k2|all]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mspl_amd import ops
from mspl_amd.ops import Epi

DEV = 'cuda'


def timeit(fn, iters=20, reps=5):
    """Device time per launch: `iters` launches captured into one hipGraph (no host launch overhead in between)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / iters * 1e3)
    best.sort()
    return best[len(best) // 2]


def conv1x1_cases(N=16):
    # (name, Cin, Cout, groups, H, W, residual)
    return [('L2_0 exp 96->96 g4', 96, 96, 4, 72, 120, False), ('L3_0 proj 128->32', 128, 32, 4, 72, 120, False),
            ('L3_0 exp 128->128', 128, 128, 4, 36, 60, False), ('L3 proj 256->64', 256, 64, 4, 36, 60, False),
            ('L3 exp 256->256 +res', 256, 256, 4, 36, 60, True), ('L4_0 exp 256->256', 256, 256, 4, 18, 30, False),
            ('L4 proj 512->128', 512, 128, 4, 18, 30, False), ('L4 exp 512->512 +res', 512, 512, 4, 18, 30, True),
            ('dec1 proj 512->16', 512, 16, 1, 18, 30, False), ('dec1 out 16->64', 16, 64, 1, 18, 30, False),
            ('dec2 proj 64->16', 64, 16, 1, 36, 60, False), ('dec3 proj 48->16', 48, 16, 1, 72, 120, False),
            ('dec4 proj 32->16', 32, 16, 1, 144, 240, False), ('dec4 out 16->13', 16, 13, 1, 144, 240, False)]


def bench_conv1x1():
    N = 16
    for name, ci, co, g, h, w, res in conv1x1_cases():
        x = torch.randn(N, ci, h, w, device=DEV)
        wt = torch.randn(co, ci // g, 1, 1, device=DEV) * 0.1
        sc, sh, al = torch.rand(co, device=DEV) + 0.5, torch.randn(co, device=DEV), torch.rand(co, device=DEV) * 0.3
        r = torch.randn(N, co, h, w, device=DEV) if res else None
        out = torch.empty(N, co, h, w, device=DEV)
        ep = Epi(sc, sh, al, residual=r)
        t = timeit(lambda: ops.conv1x1(x, wt, g, ep, out=(out, 0)))
        by = 4 * N * h * w * (ci + co * (2 if res else 1))
        fl = 2 * N * h * w * co * ci // g
        print('%-24s %8.1f us  %7.1f GB/s  %7.1f GFLOP/s' % (name, t, by / t / 1e3, fl / t / 1e3))


def bench_k2():
    N = int(os.environ.get('K2_N', '16'))
    for name, n, h, w, stride, dil in [('L2_0 s2 n=24', 24, 144, 240, 2, [1, 2, 3, 4]), ('L3_0 s2 n=32', 32, 72, 120, 2, [1, 2, 3, 4]),
                                       ('L3 s1 n=64', 64, 36, 60, 1, [1, 2, 3, 4]), ('L4_0 s2 n=64', 64, 36, 60, 2, [1, 2, 3, 4]),
                                       ('L4 s1 n=128', 128, 18, 30, 1, [1, 1, 2, 3])]:
        x = torch.randn(N, n, h, w, device=DEV)
        w4 = torch.randn(4, n, 3, 3, device=DEV) * 0.2
        sc, sh, al = torch.rand(4 * n, device=DEV) + 0.5, torch.randn(4 * n, device=DEV), torch.rand(4 * n, device=DEV) * 0.3
        ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
        out = torch.empty(N, 4 * n, ho, wo, device=DEV)
        ep = Epi(sc, sh, al)
        t = timeit(lambda: ops.eesp_dw_hff(x, w4, dil, stride, ep, out=(out, 0)))
        by = 4 * N * n * (h * w + 4 * ho * wo)
        print('%-24s %8.1f us  %7.1f GB/s (%.1f%% of 8 TB/s)' % (name, t, by / t / 1e3, by / t / 1e3 / 80))


def bench_pyr():
    from mspl_amd import layers as L
    N = 16
    for name, cin, cout, h, w in [('dec1 512->64 18x30', 512, 64, 18, 30), ('dec2 64->48 36x60', 64, 48, 36, 60),
                                  ('dec3 48->32 72x120', 48, 32, 72, 120), ('dec4 32->13 144x240', 32, 13, 144, 240)]:
        m = L.EfficientPyrPool(cin, 16, cout, last_layer_br=(cout != 13)).to(DEV).eval()
        x = torch.randn(N, cin, h, w, device=DEV)
        with torch.no_grad():
            xp = m.projection_layer(x)
            sizes = m.branch_sizes(h, w)
            t_all = timeit(lambda: m(x))
            t_fused = timeit(lambda: m.forward_fused(xp, sizes))
            import mspl_amd.ops as O
            down = [O.conv3x3(O.adaptive_avgpool(xp, sz), st.weight, 16) if sc < 1 else None for sc, sz, st in zip(m.scales, sizes, m.stages)]
            sw = [None if sc < 1 else st.weight for sc, st in zip(m.scales, m.stages)]
            br = m.merge_layer[0].br
            bs, bh = L.bn_fold(br[0])
            mc = m.merge_layer[2]
            t_k = timeit(lambda: O.pyrpool_fused(xp, sizes, sw, down, bs, bh, br[1].weight, mc.cbr[0].weight, mc.epi()))
            print('   kernel only %.1f us' % t_k)
        by = 4 * N * 16 * h * w * 2
        print('%-24s whole %8.1f us | fused body %8.1f us (%.0f GB/s of x+y traffic)' % (name, t_all, t_fused, by / t_fused / 1e3))


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if what in ('conv1x1', 'all'):
        bench_conv1x1()
    if what in ('k2', 'all'):
        bench_k2()
    if what in ('pyr', 'all'):
        bench_pyr()


def bench_prep():
    from mspl_amd import ops as O
    N, P = 16, 16
    for h, w in [(18, 30), (36, 60), (72, 120), (144, 240)]:
        import math
        sizes = [(max(math.ceil(h * s), 5), max(math.ceil(w * s), 5)) for s in (0.5, 0.1)]
        x = torch.randn(N, P, h, w, device=DEV)
        ws = [torch.randn(P, 1, 3, 3, device=DEV) for _ in sizes]
        t_f = timeit(lambda: O.pyr_down_prep(x, sizes, ws))
        t_s = timeit(lambda: [O.conv3x3(O.adaptive_avgpool(x, sz), wt, P) for sz, wt in zip(sizes, ws)])
        print('prep %3dx%3d  fused %7.1f us   separate %7.1f us' % (h, w, t_f, t_s))


if len(sys.argv) > 1 and sys.argv[1] == 'prep':
    bench_prep()


def bench_stream():
    from mspl_amd import ops as O
    N = 16
    x = torch.randn(N, 32, 144, 240, device=DEV)
    t = timeit(lambda: O.avgpool3x3s2(x))
    print('avgpool 32x144x240 -> 72x120: %.1f us (%.0f GB/s of in+out)' % (t, (x.numel() * 4 * 1.25) / t / 1e3))
    y = torch.randn(N, 32, 72, 120, device=DEV)
    pre = torch.randn(N, 32, 144, 240, device=DEV)
    sc = torch.ones(32, device=DEV)
    t = timeit(lambda: O.bilinear(y, (144, 240), Epi(sc, sc, sc, pre_add=pre)))
    print('bilinear 32x72x120 -> 144x240 (+pre_add, BN, PReLU): %.1f us (%.0f GB/s)' % (t, (y.numel() * 4 + 2 * pre.numel() * 4) / t / 1e3))
    t = timeit(lambda: O.pointwise(pre, Epi(sc, sc, sc)))
    print('pointwise 32x144x240: %.1f us (%.0f GB/s)' % (t, 2 * pre.numel() * 4 / t / 1e3))
    img = torch.randn(N, 3, 288, 480, device=DEV)
    w = torch.randn(32, 3, 3, 3, device=DEV)
    sc32 = torch.ones(32, device=DEV)
    t = timeit(lambda: O.conv3x3(img, w, 1, 2, ep=Epi(sc32, sc32, sc32)))
    print('stem conv3x3 s2 3->32: %.1f us (%.0f GB/s)' % (t, (img.numel() * 4 + N * 32 * 144 * 240 * 4) / t / 1e3))
    xe = torch.randn(N, 32, 144, 240, device=DEV)
    we = torch.randn(16, 2, 3, 3, device=DEV)
    sc16 = torch.ones(16, device=DEV)
    t = timeit(lambda: O.conv3x3(xe, we, 16, 1, ep=Epi(sc16, sc16, sc16)))
    print('pwconv expand 3x3 g16 32->16 @144x240: %.1f us (%.0f GB/s)' % (t, (xe.numel() * 4 * 1.5) / t / 1e3))


if len(sys.argv) > 1 and sys.argv[1] == 'stream':
    bench_stream()


def bench_k2x():
    """Eager launches of the K2 shapes (for a STAMPS=1 build: the launcher prints the phase timeline of every launch)."""
    N = int(os.environ.get('K2_N', '16'))
    for name, n, h, w, stride, dil in [('L2_0 s2 n=24', 24, 144, 240, 2, [1, 2, 3, 4]), ('L3_0 s2 n=32', 32, 72, 120, 2, [1, 2, 3, 4]),
                                       ('L3 s1 n=64', 64, 36, 60, 1, [1, 2, 3, 4]), ('L4_0 s2 n=64', 64, 36, 60, 2, [1, 2, 3, 4]),
                                       ('L4 s1 n=128', 128, 18, 30, 1, [1, 1, 2, 3])]:
        x = torch.randn(N, n, h, w, device=DEV)
        w4 = torch.randn(4, n, 3, 3, device=DEV) * 0.2
        sc, sh, al = torch.rand(4 * n, device=DEV) + 0.5, torch.randn(4 * n, device=DEV), torch.rand(4 * n, device=DEV) * 0.3
        ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
        out = torch.empty(N, 4 * n, ho, wo, device=DEV)
        ep = Epi(sc, sh, al)
        sys.stderr.write('--- %s\n' % name)
        sys.stderr.flush()
        for _ in range(3):
            ops.eesp_dw_hff(x, w4, dil, stride, ep, out=(out, 0))
            torch.cuda.synchronize()


if len(sys.argv) > 1 and sys.argv[1] == 'k2x':
    bench_k2x()


def bench_exp():
    """K2 + K3 of the stride-1 EESP blocks: the fused launch (csrc/eesp_exp.hip) against the two launches it replaces."""
    N = int(os.environ.get('K2_N', '16'))
    for name, n, h, w, dil in [('L3 s1 n=64 36x60', 64, 36, 60, [1, 2, 3, 4]), ('L4 s1 n=128 18x30', 128, 18, 30, [1, 1, 2, 3])]:
        r = torch.randn(N, n, h, w, device=DEV)
        xin = torch.randn(N, 4 * n, h, w, device=DEV)
        w4 = torch.randn(4, n, 3, 3, device=DEV) * 0.2
        bs, bb, ba = torch.rand(4 * n, device=DEV) + 0.5, torch.randn(4 * n, device=DEV), torch.rand(4 * n, device=DEV) * 0.3
        wexp = torch.randn(4 * n, n, 1, 1, device=DEV) * 0.1
        es, eb, ea = torch.rand(4 * n, device=DEV) + 0.5, torch.randn(4 * n, device=DEV), torch.rand(4 * n, device=DEV) * 0.3
        cat = torch.empty(N, 4 * n, h, w, device=DEV)
        out = torch.empty(N, 4 * n, h, w, device=DEV)
        epb, epe = Epi(bs, bb, ba), Epi(es, eb, ea, residual=xin)
        t_k2 = timeit(lambda: ops.eesp_dw_hff(r, w4, dil, 1, epb, out=(cat, 0)))
        t_k3 = timeit(lambda: ops.conv1x1(cat, wexp, 4, epe, out=(out, 0)))
        t_two = timeit(lambda: (ops.eesp_dw_hff(r, w4, dil, 1, epb, out=(cat, 0)), ops.conv1x1(cat, wexp, 4, epe, out=(out, 0))))
        packed = ops.eesp_dw_exp_pack(w4, bs, bb, ba, wexp, h, w, dil)
        t_f = timeit(lambda: ops.eesp_dw_exp(r, packed, dil, epe))
        by = 4 * N * h * w * 9 * n
        fl = 2 * N * h * w * 4 * n * n
        print('%-20s N=%d  K2 %6.1f + K3 %6.1f (back to back %6.1f) us | fused %6.1f us  %6.0f GB/s (%.2f of 8 TB/s)  %5.1f TFLOP/s (%.2f of 157)'
              % (name, N, t_k2, t_k3, t_two, t_f, by / t_f / 1e3, by / t_f / 8e6, fl / t_f / 1e6, fl / t_f / 157.3e6))


if len(sys.argv) > 1 and sys.argv[1] == 'exp':
    bench_exp()


def bench_expx():
    """Eager launches of the fused K2+K3 kernel (for a STAMPS=1 build with MSPL_XE_STAMP=1: the launcher prints the step timeline)."""
    N = int(os.environ.get('K2_N', '16'))
    for name, n, h, w, dil in [('L3 s1 n=64 36x60', 64, 36, 60, [1, 2, 3, 4]), ('L4 s1 n=128 18x30', 128, 18, 30, [1, 1, 2, 3])]:
        r = torch.randn(N, n, h, w, device=DEV)
        xin = torch.randn(N, 4 * n, h, w, device=DEV)
        w4 = torch.randn(4, n, 3, 3, device=DEV) * 0.2
        bs, bb, ba = torch.rand(4 * n, device=DEV) + 0.5, torch.randn(4 * n, device=DEV), torch.rand(4 * n, device=DEV) * 0.3
        wexp = torch.randn(4 * n, n, 1, 1, device=DEV) * 0.1
        es, eb, ea = torch.rand(4 * n, device=DEV) + 0.5, torch.randn(4 * n, device=DEV), torch.rand(4 * n, device=DEV) * 0.3
        packed = ops.eesp_dw_exp_pack(w4, bs, bb, ba, wexp, h, w, dil)
        epe = Epi(es, eb, ea, residual=xin)
        sys.stderr.write('--- %s\n' % name)
        sys.stderr.flush()
        for _ in range(4):
            ops.eesp_dw_exp(r, packed, dil, epe)
            torch.cuda.synchronize()


if len(sys.argv) > 1 and sys.argv[1] == 'expx':
    bench_expx()


def bench_k2_cold():
    """The large strided K2 launch with WARM caches (re-issued back to back: its input and output fit the 256 MB Infinity Cache) and
    with COLD ones (a 1 GB buffer is rewritten before every launch): explains the in-pass / isolated gap of that launch."""
    for N in (16, 32):
        n, h, w, stride, dil = 24, 144, 240, 2, [1, 2, 3, 4]
        x = torch.randn(N, n, h, w, device=DEV)
        w4 = torch.randn(4, n, 3, 3, device=DEV) * 0.2
        sc, sh, al = torch.rand(4 * n, device=DEV) + 0.5, torch.randn(4 * n, device=DEV), torch.rand(4 * n, device=DEV) * 0.3
        ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
        out = torch.empty(N, 4 * n, ho, wo, device=DEV)
        ep = Epi(sc, sh, al)
        big = torch.empty(256 * 1024 * 1024, device=DEV)          # 1 GB
        res = {}
        for mode in ('warm', 'cold', 'input-fresh'):
            ts = []
            for _ in range(12):
                if mode == 'cold':
                    big.fill_(1.0)
                elif mode == 'input-fresh':
                    big.fill_(1.0)
                    x.mul_(1.0)                                   # the producer has just written the input (as in the pass)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda._sleep(2_000_000)
                e0.record()
                ops.eesp_dw_hff(x, w4, dil, stride, ep, out=(out, 0))
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            res[mode] = ts[len(ts) // 2]
        by = 4 * N * n * (h * w + 4 * ho * wo)
        print('L2_0 K2 N=%d (%.0f MB in + out): event-pair time warm %.1f us, cold %.1f us, cold with a freshly written input %.1f us '
              '(each includes the ~2.7 us of the event pair)' % (N, by / 1e6, res['warm'], res['cold'], res['input-fresh']))


if len(sys.argv) > 1 and sys.argv[1] == 'k2cold':
    bench_k2_cold()


def bench_label():
    """The label epilogue (up-sampling of both heads + argmax + histogram) at the path's shapes: C = 13 / 20 / 5 at 288x480 / 256x480,
    C = 20 at 512x1024; alone, back to back inside a hipGraph."""
    from mspl_amd import ops as O
    N = 16
    for C, H, W in [(13, 288, 480), (20, 256, 480), (5, 256, 480), (20, 512, 1024)]:
        main = torch.randn(N, C, H // 2, W // 2, device=DEV)
        aux = torch.randn(N, C, H // 4, W // 4, device=DEV)
        hist = torch.zeros(C, dtype=torch.int64, device=DEV)
        fits = O.label_epilogue_hist_fits(main, aux, (H, W))
        t_h = timeit(lambda: O.label_epilogue_hist(main, aux, (H, W), hist, C, want_kld=False)) if fits else float('nan')
        t_l = timeit(lambda: O.label_epilogue(main, aux, (H, W)))
        t_k = timeit(lambda: O.label_epilogue(main, aux, (H, W), want_kld=True))
        by = 4 * N * C * (H * W // 4 + H * W // 16) + N * H * W
        print('label C=%2d %4dx%4d  with histogram %7.1f us (%.0f GB/s)   labels only %7.1f us   labels + kld %7.1f us'
              % (C, H, W, t_h, by / t_h / 1e3, t_l, t_k))


if len(sys.argv) > 1 and sys.argv[1] == 'label':
    bench_label()
