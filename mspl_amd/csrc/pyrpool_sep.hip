// K6, second form -- fused EfficientPyrPool body with the up-sampled branches collapsed into position-dependent
// separable stencils on x itself.
//
// Reference arithmetic (nn_layers/efficient_pyramid_pool.py:36-61), per projected channel c, for a branch with
// scale > 1:   b = adaptive_avg_pool2d( dw3x3_c( bilinear_up(x_c) ) ).
// Up-sampling R and pooling M are separable linear maps, the 3x3 kernel is a sum of three row kernels, hence
//   b[py, px] = sum_ky  sum_r sum_s  A_ky[py, r] * C_ky[px, s] * x[py - R + r, px - R + s]
//   A_ky = M_y . shift_ky . R_y                 (channel independent; TAPS = 2R+1 coefficients per output row)
//   C_ky = sum_kx w_c[ky, kx] * G_kx,   G_kx = M_x . shift_kx . R_x     (G channel independent)
// with shift_k the zero-padded one-step shift of the 3x3 convolution on the up-sampled grid.  For the scales the
// reference uses (2.0 and 1.5, align_corners=True) the sources of output pixel p stay inside [p-1, p+1] resp.
// [p-2, p+2]; the launcher verifies this for the actual sizes and otherwise leaves the call to the table-driven
// kernel in pyrpool.hip.  Compared with that kernel there is no up-sampled tile in LDS at all: a thread produces a
// 1x4 strip of all branches from one 5x8 register patch of x (~150 FMA per pixel and plane, 4-5x fewer instructions).
//
// Branches with scale < 1 interpolate the small pre-convolved map E (staged per tile in LDS); scale = 1 is a plain
// 3x3.  Then merge_layer.0 (BN+PReLU), Shuffle, merge_layer.2 (grouped 3x3 + BN + PReLU) exactly as before.
#include <stdlib.h>

#include "common.hpp"

namespace mspl {

constexpr int P2_MAXB = 5;
constexpr int P2_NDESC = 8;         // ints per branch descriptor in LDS

struct Pyr2Geom {
    int N, P, h, w, nb;
    int kind[P2_MAXB];              // 0: up (stencil form), 1: same, 2: down
    int taps[P2_MAXB];              // up branches: 3 or 5
    int hs[P2_MAXB], ws[P2_MAXB];
    float sh[P2_MAXB], sw[P2_MAXB]; // bilinear scales (up: x -> U grid, down: E grid -> output grid)
    const float* stage_w[P2_MAXB];
    const float* down_e[P2_MAXB];
    const float* br_scale; const float* br_shift; const float* br_alpha;
    const float* merge_w;
    int TH, TW, tiles_y, tiles_x;
    int NS, BW, XW, NCH;            // strips per branch row, branch-tile row stride (4*NS), x-tile stride (BW+4), float4 chunks per x row
    int aoff[P2_MAXB], goff[P2_MAXB], coff[P2_MAXB];   // A / G / C tables of the up branches
    int doff[P2_MAXB], eoff[P2_MAXB], EH[P2_MAXB], EW[P2_MAXB], epre[P2_MAXB];   // down branches: tables, E tile, prefix of E elements
    int nE;                         // total E-tile elements (<= 512)
    int boff, woff, dscoff;
    int CPB, cblocks;
    int stop_after;                 // tuning aid (MSPL_PYR_STOP): skip the phases after k; 0 = run everything
    unsigned xcd_per, total;        // XCD-contiguous tile order (common.hpp): tiles of a plane share halos and the low-res maps
};

// Table geometry of an up branch with T taps: a table row holds [ky][KS] floats, KS = 4 (T = 3) or 8 (T = 5).
__host__ __device__ constexpr int p2_ks(int T) { return T <= 3 ? 4 : 8; }
__host__ __device__ constexpr int p2_ts(int T) { return 3 * p2_ks(T); }
// C tables: floats per strip = 4 rows + 4 pad, so that the strips of a wave land on distinct LDS banks
__host__ __device__ constexpr int p2_cs(int T) { return 4 * p2_ts(T) + 4; }

__device__ __forceinline__ int p2_ada_s(int o, int I, int O) { return (int)(((unsigned)o * (unsigned)I) / (unsigned)O); }
__device__ __forceinline__ int p2_ada_e(int o, int I, int O) { return (int)((((unsigned)(o + 1)) * (unsigned)I + O - 1) / (unsigned)O); }

template <int T>
__device__ __forceinline__ void p2_load_row(const float* __restrict__ p, float (&v)[5]) {
    if (T <= 3) {
        const float4 t4 = *reinterpret_cast<const float4*>(p);
        v[0] = t4.x; v[1] = t4.y; v[2] = t4.z; v[3] = 0.f; v[4] = 0.f;
    } else {
        const float4 t4 = *reinterpret_cast<const float4*>(p);
        v[0] = t4.x; v[1] = t4.y; v[2] = t4.z; v[3] = t4.w; v[4] = p[4];
    }
}

// One up branch for a strip: xp = 5x8 patch (patch row rr <-> image row py-2+rr, col s <-> image col px0-2+s).
// Evaluated one kernel row ky at a time to keep few values live.
template <int T>
__device__ __forceinline__ void up_branch_strip(const float (&xp)[5][8], const float* __restrict__ Arow,
                                                const float* __restrict__ Cq, float (&b)[4]) {
    constexpr int R0 = (5 - T) / 2;
    constexpr int NCOL = T + 3;
    constexpr int KS = p2_ks(T), TS = p2_ts(T);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        float a[5];
        p2_load_row<T>(Arow + ky * KS, a);
        float cs[NCOL];
#pragma unroll
        for (int s = 0; s < NCOL; ++s) {
            float v = a[0] * xp[R0][R0 + s];
#pragma unroll
            for (int rr = 1; rr < T; ++rr) v = fmaf(a[rr], xp[R0 + rr][R0 + s], v);
            cs[s] = v;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float c[5];
            p2_load_row<T>(Cq + j * TS + ky * KS, c);
            float v = b[j];
#pragma unroll
            for (int s = 0; s < T; ++s) v = fmaf(c[s], cs[j + s], v);
            b[j] = v;
        }
        asm volatile("" ::: "memory");      // keep the table reads of the next kernel row behind this one's arithmetic
    }
}

// Channel-independent stencil table of one up branch: rows [pos][ky][KS]; thread t handles (pos, k).
template <int T>
__device__ __forceinline__ void p2_fill_up_tables(float* A, float* G, int BH, int BW, int y0, int x0, int h, int w,
                                                  int hs, int ws, float sh, float sw, int tid) {
    constexpr int R = (T - 1) / 2, KS = p2_ks(T), TS = p2_ts(T);
    const int nrow = BH * 3, ncol = BW * 3;
    for (int t = tid; t < nrow + ncol; t += 256) {
        const bool isrow = t < nrow;
        const int u = isrow ? t : t - nrow;
        const int pos = u / 3, k = u - pos * 3;
        const int p = (isrow ? y0 : x0) - 1 + pos;
        const int I = isrow ? h : w, S = isrow ? hs : ws;
        const float sc = isrow ? sh : sw;
        float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        if (p >= 0 && p < I) {
            const int us = p2_ada_s(p, S, I), ue = p2_ada_e(p, S, I);
            const float inv = 1.0f / (float)(ue - us);
            for (int uu = us; uu < ue; ++uu) {
                const int v = uu + k - 1;
                if (v < 0 || v >= S) continue;          // zero padding of the 3x3 on the up-sampled grid
                int ia, ib;  float w0, w1;
                bilinear_src(sc, v, I, ia, ib, w0, w1);
                const int ta = ia - (p - R), tb = ib - (p - R);
#pragma unroll
                for (int q = 0; q < T; ++q) {
                    if (q == ta) acc[q] += w0 * inv;
                    if (q == tb) acc[q] += w1 * inv;
                }
            }
        }
        float* dst = (isrow ? A : G) + pos * TS + k * KS;
#pragma unroll
        for (int q = 0; q < KS; ++q) if (q < T || KS == 4) dst[q] = q < T ? acc[q] : 0.f;
    }
}

// Per-plane column table C[q][ky][s] = sum_kx w[ky][kx] * G[q][kx][s]; thread t handles (q, ky).
template <int T>
__device__ __forceinline__ void p2_fill_c(const float* __restrict__ G, float* __restrict__ C, const float* __restrict__ w9,
                                          int BW, int tid) {
    constexpr int KS = p2_ks(T), TS = p2_ts(T), CS = p2_cs(T);
    for (int t = tid; t < BW * 3; t += 256) {
        const int q = t / 3, ky = t - q * 3;
        const float w0 = w9[ky * 3], w1 = w9[ky * 3 + 1], w2 = w9[ky * 3 + 2];
        float g0[5], g1[5], g2[5];
        p2_load_row<T>(G + q * TS, g0);
        p2_load_row<T>(G + q * TS + KS, g1);
        p2_load_row<T>(G + q * TS + 2 * KS, g2);
        float* dst = C + (q >> 2) * CS + (q & 3) * TS + ky * KS;
#pragma unroll
        for (int s = 0; s < T; ++s) dst[s] = fmaf(w2, g2[s], fmaf(w1, g1[s], w0 * g0[s]));
    }
}

// T0 / T1: stencil taps (3 or 5) of the first / second up branch (in branch order).
// Epilogue of the merge convolution: y = PReLU(acc * ep_scale[c] + ep_shift[c]) written to channel ep_coff + c of ep_ctot.
template <int T0, int T1>
__global__ __launch_bounds__(256, 3) void pyrpool_sep_kernel(const float* __restrict__ x, const float* __restrict__ ep_scale,
                                                             const float* __restrict__ ep_shift,
                                                             const float* __restrict__ ep_alpha, int ep_ctot, int ep_coff,
                                                             Pyr2Geom g, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                               // (TH+6) x XW, col j <-> image col x0-3+j, row r <-> image row y0-3+r
    float* wl = smem + g.woff;                      // [nb][9] stage weights, [nb][9] merge weights, [nb][3] BR consts
    float* B = smem + g.boff;                       // nb x (TH+2) x BW, col q <-> image col x0-1+q
    int* bdesc = reinterpret_cast<int*>(smem + g.dscoff);   // per branch: kind, slot, aoff, coff, doff, eoff
    int bid = (int)xcd_contiguous(blockIdx.x, g.xcd_per);
    if ((unsigned)bid >= g.total) return;
    const int txi = bid % g.tiles_x;  bid /= g.tiles_x;
    const int tyi = bid % g.tiles_y;  bid /= g.tiles_y;
    const int cb = bid % g.cblocks;
    const int n = bid / g.cblocks;
    const int y0 = tyi * g.TH, x0 = txi * g.TW;
    const int tid = threadIdx.x;
    const int XH = g.TH + 6, BH = g.TH + 2;
    const int hl = g.h, wl_ = g.w, XWl = g.XW, BWl = g.BW, NS = g.NS, nbl = g.nb;
    const int c_first = cb * g.CPB;

    // ---- phase 0 (once per workgroup): channel-independent tables and the branch descriptors
    {
        int upslot = 0;
#pragma unroll
        for (int i = 0; i < P2_MAXB; ++i) {
            if (i >= g.nb) break;
            if (tid == 0) {
                int* d = bdesc + i * P2_NDESC;
                d[0] = g.kind[i]; d[1] = upslot; d[2] = g.aoff[i]; d[3] = g.coff[i]; d[4] = g.doff[i]; d[5] = g.eoff[i];
            }
            if (g.stop_after & 256) continue;
            if (g.kind[i] == 0) {
                if (upslot == 0) p2_fill_up_tables<T0>(smem + g.aoff[i], smem + g.goff[i], BH, BWl, y0, x0, hl, wl_, g.hs[i], g.ws[i], g.sh[i], g.sw[i], tid);
                else p2_fill_up_tables<T1>(smem + g.aoff[i], smem + g.goff[i], BH, BWl, y0, x0, hl, wl_, g.hs[i], g.ws[i], g.sh[i], g.sw[i], tid);
                ++upslot;
            } else if (g.kind[i] == 2) {
                float* DR = smem + g.doff[i];  float* DC = DR + 4 * BH;
                int ya0, yb0, xa0, xb0;  float f0, f1;
                bilinear_src(g.sh[i], min(max(y0 - 1, 0), hl - 1), g.hs[i], ya0, yb0, f0, f1);
                bilinear_src(g.sw[i], min(max(x0 - 1, 0), wl_ - 1), g.ws[i], xa0, xb0, f0, f1);
                for (int t = tid; t < BH + BWl; t += 256) {
                    if (t < BH) {
                        const int py = min(max(y0 - 1 + t, 0), hl - 1);
                        int ya, yb;  float w0, w1;
                        bilinear_src(g.sh[i], py, g.hs[i], ya, yb, w0, w1);
                        ya = min(ya - ya0, g.EH[i] - 1);  yb = min(yb - ya0, g.EH[i] - 1);
                        DR[4 * t] = __int_as_float(ya * g.EW[i]); DR[4 * t + 1] = __int_as_float(yb * g.EW[i]); DR[4 * t + 2] = w0; DR[4 * t + 3] = w1;
                    } else {
                        // column entries are grouped per strip (4 entries + 1 pad float4): strips on distinct banks
                        const int q = t - BH, px = min(max(x0 - 1 + q, 0), wl_ - 1);
                        int xa, xb;  float w0, w1;
                        bilinear_src(g.sw[i], px, g.ws[i], xa, xb, w0, w1);
                        xa = min(xa - xa0, g.EW[i] - 1);  xb = min(xb - xa0, g.EW[i] - 1);
                        float* d = DC + (q >> 2) * 20 + (q & 3) * 4;
                        d[0] = __int_as_float(xa); d[1] = __int_as_float(xb); d[2] = w0; d[3] = w1;
                    }
                }
            }
        }
    }

    // ---- staging descriptors, computed once per thread (per plane only a base address changes)
    //  x tile: chunk i = (row r, chunk m): image cols x0-4+4m .. +3 of row y0-3+r (w % 4 == 0: all in or all out)
    const int NCH = g.NCH;
    const int nx = XH * NCH;                        // <= 512
    int xg0 = -1, xg1 = -1, xd0 = -1, xd1 = -1;     // offset inside the plane (floats) or -1; LDS chunk position | flags
    {
        auto mk = [&](int i, int& goff, int& dst) {
            if (i >= nx) return;
            const int r = i / NCH, m = i - r * NCH;
            const int iy = y0 - 3 + r, ix = x0 - 4 + 4 * m;
            if (iy >= 0 && iy < hl && ix >= 0 && ix < wl_ && !(g.stop_after & 1024)) goff = iy * wl_ + ix;
            const int nvalid = min(4, wl_ - ix);            // (only the row's last chunk is partial, when w % 4 != 0)
            dst = (r * XWl + 4 * m) | (m > 0 ? (1 << 30) : 0) | (4 * m < XWl ? (1 << 29) : 0) | ((nvalid & 7) << 25);     // +1: see store_x
        };
        mk(tid, xg0, xd0);
        mk(tid + 256, xg1, xd1);
    }
    //  E tiles: element t of the concatenated low-resolution tiles
    const float* ep0 = nullptr;  const float* ep1 = nullptr;
    int es0 = 0, es1 = 0, ed0 = -1, ed1 = -1;
    {
        auto mk = [&](int t, const float*& ptr, int& stride, int& dst) {
#pragma unroll
            for (int i = 0; i < P2_MAXB; ++i) {
                if (i >= g.nb || g.kind[i] != 2 || (g.stop_after & 512)) continue;
                const int u = t - g.epre[i];
                if (u >= 0 && u < g.EH[i] * g.EW[i]) {
                    int ya0, yb0, xa0, xb0;  float f0, f1;
                    bilinear_src(g.sh[i], min(max(y0 - 1, 0), hl - 1), g.hs[i], ya0, yb0, f0, f1);
                    bilinear_src(g.sw[i], min(max(x0 - 1, 0), wl_ - 1), g.ws[i], xa0, xb0, f0, f1);
                    const int r = u / g.EW[i], q = u - r * g.EW[i];
                    const int ey = min(ya0 + r, g.hs[i] - 1), ex = min(xa0 + q, g.ws[i] - 1);
                    stride = g.hs[i] * g.ws[i];
                    ptr = g.down_e[i] + ((size_t)n * g.P + c_first) * stride + (size_t)ey * g.ws[i] + ex;
                    dst = g.eoff[i] + u;
                }
            }
        };
        mk(tid, ep0, es0, ed0);
        mk(tid + 256, ep1, es1, ed1);
    }
    //  per-plane constants: element t < 2*nb*9 + 3*nb
    const int nconst = 2 * g.nb * 9 + 3 * g.nb;
    const float* cp = nullptr;
    int cstride = 0;
    if (tid < nconst) {
        const int n9 = g.nb * 9, t = tid;
        if (t < n9) {
            const int i = t / 9, k = t - i * 9;
#pragma unroll
            for (int j = 0; j < P2_MAXB; ++j)
                if (j == i && g.kind[j] != 2) { cp = g.stage_w[j] + (size_t)c_first * 9 + k; cstride = 9; }
        } else if (t < 2 * n9) {
            cp = g.merge_w + (size_t)c_first * n9 + (t - n9);  cstride = n9;
        } else {
            const int u = t - 2 * n9, i = u / 3, k = u - i * 3;
            cp = (k == 0 ? g.br_scale : (k == 1 ? g.br_shift : g.br_alpha)) + (size_t)i * g.P + c_first;  cstride = 1;
        }
    }
    // pin the descriptors: without this the compiler re-derives them (and their ~50 temporaries) inside the plane loop
    asm volatile("" : "+v"(xg0), "+v"(xg1), "+v"(xd0), "+v"(xd1), "+v"(es0), "+v"(es1), "+v"(ed0), "+v"(ed1), "+v"(cstride));
    asm volatile("" : "+v"(ep0), "+v"(ep1), "+v"(cp));

    const size_t plane = (size_t)hl * wl_;
    const float* xpl = x + ((size_t)n * g.P + c_first) * plane;
    const bool w4 = (wl_ & 3) == 0;                 // rows 16-byte aligned: whole chunks, vector loads / stores
    auto load_x = [&](int goff, int dst) -> float4 {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (goff < 0) return v;
        if (w4) return *reinterpret_cast<const float4*>(xpl + goff);
        const int nvalid = (dst >> 25) & 7;
        const float* q = xpl + goff;
        v.x = q[0];
        if (nvalid > 1) v.y = q[1];
        if (nvalid > 2) v.z = q[2];
        if (nvalid > 3) v.w = q[3];
        return v;
    };
    auto store_x = [&](int dst, const float4& v) {
        float* d = xs + (dst & 0x1ffffff) - 1;         // chunk m covers tile cols 4m-1 .. 4m+2
        if (dst & (1 << 30)) d[0] = v.x;
        if (dst & (1 << 29)) { d[1] = v.y; d[2] = v.z; d[3] = v.w; }
    };
    float4 xr0 = load_x(xg0, xd0), xr1 = load_x(xg1, xd1);
    float er0 = ed0 >= 0 ? *ep0 : 0.f, er1 = ed1 >= 0 ? *ep1 : 0.f;
    float cr = cp ? *cp : 0.f;

    for (int ci = 0; ci < g.CPB; ++ci) {
        const int c = c_first + ci;
        // ---- phase 1: registers -> LDS, then prefetch the next plane
        if (xd0 >= 0) store_x(xd0, xr0);
        if (xd1 >= 0) store_x(xd1, xr1);
        if (tid < nconst) wl[tid] = cr;
        if (ed0 >= 0) smem[ed0] = er0;
        if (ed1 >= 0) smem[ed1] = er1;
        if (ci + 1 < g.CPB) {
            xpl += plane;
            xr0 = load_x(xg0, xd0);  xr1 = load_x(xg1, xd1);
            if (ed0 >= 0) { ep0 += es0; er0 = *ep0; }
            if (ed1 >= 0) { ep1 += es1; er1 = *ep1; }
            if (cp) { cp += cstride; cr = *cp; }
        }
        __syncthreads();
        if ((g.stop_after & 255) == 1) continue;

        // ---- phase 2: per-plane column tables C_ky = sum_kx w[ky][kx] * G_kx
        {
            int upslot = 0;
#pragma unroll
            for (int i = 0; i < P2_MAXB; ++i) {
                if (i < g.nb && g.kind[i] == 0) {
                    if (upslot == 0) p2_fill_c<T0>(smem + g.goff[i], smem + g.coff[i], wl + i * 9, BWl, tid);
                    else p2_fill_c<T1>(smem + g.goff[i], smem + g.coff[i], wl + i * 9, BWl, tid);
                    ++upslot;
                }
            }
        }
        __syncthreads();
        if ((g.stop_after & 255) == 2) continue;

        // ---- phase 3: all branches of a 1x4 strip from one 5x8 register patch of x; BN+PReLU'd; zero outside the image
        if (tid < BH * NS) {
            const int r = tid / NS, k = tid - r * NS;
            const int py = y0 - 1 + r, px0 = x0 - 1 + 4 * k;
            const bool rowin = py >= 0 && py < hl;
            float xp[5][8];
#pragma unroll
            for (int rr = 0; rr < 5; ++rr) {
                const float* row = xs + (r + rr) * XWl + 4 * k;
                const float4 a4 = *reinterpret_cast<const float4*>(row);
                const float4 b4 = *reinterpret_cast<const float4*>(row + 4);
                xp[rr][0] = a4.x; xp[rr][1] = a4.y; xp[rr][2] = a4.z; xp[rr][3] = a4.w;
                xp[rr][4] = b4.x; xp[rr][5] = b4.y; xp[rr][6] = b4.z; xp[rr][7] = b4.w;
            }
#pragma unroll 1
            for (int i = 0; i < nbl; ++i) {        // not unrolled: one branch's tables and temporaries live at a time
                const int4 d0 = *reinterpret_cast<const int4*>(bdesc + i * P2_NDESC);       // kind, slot, aoff, coff
                const int2 d1 = *reinterpret_cast<const int2*>(bdesc + i * P2_NDESC + 4);   // doff, eoff
                float b[4] = {0.f, 0.f, 0.f, 0.f};
                if (d0.x == 0) {
                    if (d0.y == 0) up_branch_strip<T0>(xp, smem + d0.z + r * p2_ts(T0), smem + d0.w + k * p2_cs(T0), b);
                    else up_branch_strip<T1>(xp, smem + d0.z + r * p2_ts(T1), smem + d0.w + k * p2_cs(T1), b);
                } else if (d0.x == 1) {
                    const float* w9 = wl + i * 9;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const float w0 = w9[ky * 3], w1 = w9[ky * 3 + 1], w2 = w9[ky * 3 + 2];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            b[j] = fmaf(w0, xp[1 + ky][j + 1], b[j]);
                            b[j] = fmaf(w1, xp[1 + ky][j + 2], b[j]);
                            b[j] = fmaf(w2, xp[1 + ky][j + 3], b[j]);
                        }
                    }
                } else {
                    const float* DR = smem + d1.x;
                    const float* E = smem + d1.y;
                    const float4 rr4 = *reinterpret_cast<const float4*>(DR + 4 * r);
                    const float* ra = E + __float_as_int(rr4.x);
                    const float* rb = E + __float_as_int(rr4.y);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float4 cc = *reinterpret_cast<const float4*>(DR + 4 * BH + k * 20 + 4 * j);
                        const int xa = __float_as_int(cc.x), xb = __float_as_int(cc.y);
                        const float top = cc.z * ra[xa] + cc.w * ra[xb];
                        const float bot = cc.z * rb[xa] + cc.w * rb[xb];
                        b[j] = rr4.z * top + rr4.w * bot;
                    }
                }
                const float* kbr = wl + 2 * nbl * 9 + i * 3;
                const float bsc = kbr[0], bsh = kbr[1], bal = kbr[2];
                float4 o;
                {
                    float v = fmaf(b[0], bsc, bsh);  v = v > 0.f ? v : bal * v;  o.x = (rowin && px0 >= 0 && px0 < wl_) ? v : 0.f;
                    v = fmaf(b[1], bsc, bsh);  v = v > 0.f ? v : bal * v;  o.y = (rowin && px0 + 1 < wl_) ? v : 0.f;
                    v = fmaf(b[2], bsc, bsh);  v = v > 0.f ? v : bal * v;  o.z = (rowin && px0 + 2 < wl_) ? v : 0.f;
                    v = fmaf(b[3], bsc, bsh);  v = v > 0.f ? v : bal * v;  o.w = (rowin && px0 + 3 < wl_) ? v : 0.f;
                }
                *reinterpret_cast<float4*>(B + (i * BH + r) * BWl + 4 * k) = o;
            }
        }
        __syncthreads();
        if ((g.stop_after & 255) == 3) continue;

        // ---- phase 4: merge convolution (sum over branches of a 3x3) + BN + PReLU, 1x4 strips
        {
            const int cabs = ep_coff + c;
            const float esc = ep_scale ? ep_scale[cabs] : 1.f, esh = ep_shift ? ep_shift[cabs] : 0.f;
            const float eal = ep_alpha ? ep_alpha[cabs] : 1.f;
            const int XS = g.TW >> 2;
            const int nstrip = g.TH * XS;
            for (int t = tid; t < nstrip; t += 256) {
                const int ty = t / XS, xsi = t - ty * XS;
                const int y = y0 + ty, xb = x0 + xsi * 4;
                if (y >= hl || xb >= wl_) continue;
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
                for (int i = 0; i < nbl; ++i) {
                    const float* wm = wl + nbl * 9 + i * 9;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const float* row = B + (i * BH + ty + ky) * BWl + xsi * 4;
                        const float4 a = *reinterpret_cast<const float4*>(row);
                        const float2 b2 = *reinterpret_cast<const float2*>(row + 4);
                        const float rv[6] = {a.x, a.y, a.z, a.w, b2.x, b2.y};
                        const float w0 = wm[ky * 3], w1 = wm[ky * 3 + 1], w2 = wm[ky * 3 + 2];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            acc[j] = fmaf(w0, rv[j], acc[j]);
                            acc[j] = fmaf(w1, rv[j + 1], acc[j]);
                            acc[j] = fmaf(w2, rv[j + 2], acc[j]);
                        }
                    }
                }
                float* dst = out + ((size_t)n * ep_ctot + cabs) * plane + (size_t)y * wl_ + xb;
                float4 v;
                v.x = fmaf(acc[0], esc, esh);  v.x = (ep_alpha && v.x <= 0.f) ? eal * v.x : v.x;
                v.y = fmaf(acc[1], esc, esh);  v.y = (ep_alpha && v.y <= 0.f) ? eal * v.y : v.y;
                v.z = fmaf(acc[2], esc, esh);  v.z = (ep_alpha && v.z <= 0.f) ? eal * v.z : v.z;
                v.w = fmaf(acc[3], esc, esh);  v.w = (ep_alpha && v.w <= 0.f) ? eal * v.w : v.w;
                if (w4) {
                    store_out4(dst, v);
                } else {
                    dst[0] = v.x;
                    if (xb + 1 < wl_) dst[1] = v.y;
                    if (xb + 2 < wl_) dst[2] = v.z;
                    if (xb + 3 < wl_) dst[3] = v.w;
                }
            }
        }
        __syncthreads();   // the next plane overwrites xs / wl / E / C / B
    }
}

// Host twin of bilinear_src's index part (same fp32 operations; -ffp-contract=off).
static void host_bilinear_idx(float scale, int dst, int in_size, int& i0, int& i1) {
    const float real = scale * (float)dst;
    int idx = (int)floorf(real);
    if (idx > in_size - 1) idx = in_size - 1;
    i0 = idx;
    i1 = idx + ((idx < in_size - 1) ? 1 : 0);
}

// Smallest R such that every source of output p lies in [p-R, p+R] (one dimension); large when unsupported.
static int stencil_radius(int I, int S) {
    const float sc = bilinear_scale(I, S);
    int R = 0;
    for (int p = 0; p < I; ++p) {
        const int us = (int)(((int64_t)p * S) / I), ue = (int)((((int64_t)p + 1) * S + I - 1) / I);
        for (int v = std::max(us - 1, 0); v <= std::min(ue, S - 1); ++v) {
            int a, b;
            host_bilinear_idx(sc, v, I, a, b);
            R = std::max(R, std::max(p - a, b - p));
        }
    }
    return R;
}

// Returns MSPL_OK when launched, 1 when the shape is left to the table-driven kernel, < 0 on error.
int pyrpool_sep_try(const float* x, int N, int P, int h, int w, int nb, const int32_t* hs, const int32_t* ws,
                    const float* const* stage_w, const float* const* down_e, const float* br_scale,
                    const float* br_shift, const float* br_alpha, const float* merge_w, const Epi& e, float* out,
                    hipStream_t stream, unsigned launch_flags) {
    if (e.pre_add || e.residual || e.reinf_r || e.gate) return 1;    // only the scale/shift/PReLU epilogue
    Pyr2Geom g;
    memset(&g, 0, sizeof(g));
    g.N = N; g.P = P; g.h = h; g.w = w; g.nb = nb;
    g.br_scale = br_scale; g.br_shift = br_shift; g.br_alpha = br_alpha; g.merge_w = merge_w;
    g.TW = w >= 32 ? 32 : ((w + 3) & ~3);
    g.NS = (g.TW + 2 + 3) / 4;
    g.BW = 4 * g.NS;
    g.XW = g.BW + 4;
    g.NCH = g.XW / 4 + 1;
    // rows: (TH+2)*NS strips should fill 256 threads once
    int th_max = 256 / g.NS - 2;
    if (th_max > 26) th_max = 26;
    if (th_max < 1) th_max = 1;
    g.tiles_y = ceil_div(h, th_max);
    g.TH = ceil_div(h, g.tiles_y);
    g.tiles_x = ceil_div(w, g.TW);
    if ((g.TH + 6) * g.NCH > 512) return 1;
    const int BH = g.TH + 2;
    int off = (g.TH + 6) * g.XW;
    int taps[2] = {3, 3}, nup = 0, nE = 0;
    for (int i = 0; i < nb; ++i) {
        if (hs[i] <= 0 || ws[i] <= 0) return 1;
        g.hs[i] = hs[i]; g.ws[i] = ws[i];
        g.stage_w[i] = stage_w[i]; g.down_e[i] = down_e[i];
        if (hs[i] == h && ws[i] == w) {
            g.kind[i] = 1;
            if (!stage_w[i]) return 1;
        } else if (hs[i] >= h && ws[i] >= w) {
            g.kind[i] = 0;
            if (!stage_w[i] || nup >= 2) return 1;
            const int R = std::max(stencil_radius(h, hs[i]), stencil_radius(w, ws[i]));
            if (R > 2) return 1;
            const int T = R <= 1 ? 3 : 5;
            taps[nup++] = T;
            g.taps[i] = T;
            g.sh[i] = bilinear_scale(h, hs[i]); g.sw[i] = bilinear_scale(w, ws[i]);
            g.aoff[i] = off;  off += BH * p2_ts(T);
            g.goff[i] = off;  off += g.BW * p2_ts(T);
            g.coff[i] = off;  off += g.NS * p2_cs(T);
        } else if (hs[i] <= h && ws[i] <= w) {
            g.kind[i] = 2;
            if (!down_e[i]) return 1;
            g.sh[i] = bilinear_scale(hs[i], h); g.sw[i] = bilinear_scale(ws[i], w);
            g.EH[i] = std::min(hs[i], (int)floorf((float)(g.TH + 1) * g.sh[i]) + 3);
            g.EW[i] = std::min(ws[i], (int)floorf((float)(g.BW - 1) * g.sw[i]) + 3);
            g.doff[i] = off;  off += 4 * BH + 20 * g.NS;
            g.eoff[i] = off;  off += (g.EH[i] * g.EW[i] + 3) & ~3;
            g.epre[i] = nE;   nE += g.EH[i] * g.EW[i];
        } else {
            return 1;
        }
    }
    if (nE > 512) return 1;
    g.nE = nE;
    g.boff = off;  off += nb * BH * g.BW;
    g.woff = off;  off += (nb * 9 * 2 + nb * 3 + 3) & ~3;
    g.dscoff = off;  off += P2_MAXB * P2_NDESC;
    const size_t lds = (size_t)off * sizeof(float);
    if (lds > 64 * 1024) return 1;
    // planes per workgroup (each workgroup walks them with the next plane's loads in flight): as many as keep >= 2048 workgroups
    // for a lone pass; with MSPL_LAUNCH_THROUGHPUT 512 are enough -- fewer, longer workgroups (+2.5 % images/s with three passes in flight)
    const int64_t min_blocks = (launch_flags & MSPL_LAUNCH_THROUGHPUT) ? 512 : 2048;
    int cpb = 1;
    while (cpb * 2 <= P && P % (cpb * 2) == 0 && (int64_t)N * (P / (cpb * 2)) * g.tiles_y * g.tiles_x >= min_blocks) cpb *= 2;
    static const int dbg_cpb = MSPL_TUNE_INT("MSPL_PYR_CPB", 0);
    if (dbg_cpb > 0 && P % dbg_cpb == 0) cpb = dbg_cpb;
    g.CPB = cpb; g.cblocks = P / cpb;
    static const int dbg_stop = MSPL_TUNE_INT("MSPL_PYR_STOP", 0);
    g.stop_after = dbg_stop;
    const int64_t blocks = (int64_t)N * g.cblocks * g.tiles_y * g.tiles_x;
    if (blocks >= (1ll << 31)) return 1;
    g.total = (unsigned)blocks; g.xcd_per = xcd_per(blocks);
    const dim3 grid(8u * g.xcd_per), blk(256);
#define MSPL_P2_LAUNCH(A, B) hipLaunchKernelGGL((pyrpool_sep_kernel<A, B>), grid, blk, lds, stream, x, e.scale, e.shift, e.alpha, e.ctot, e.coff, g, out)
    if (taps[0] == 3 && taps[1] == 3) MSPL_P2_LAUNCH(3, 3);
    else if (taps[0] == 3) MSPL_P2_LAUNCH(3, 5);
    else if (taps[1] == 3) MSPL_P2_LAUNCH(5, 3);
    else MSPL_P2_LAUNCH(5, 5);
#undef MSPL_P2_LAUNCH
    MSPL_CHECK_LAUNCH("pyrpool_fused(stencil form)");
    return MSPL_OK;
}

}  // namespace mspl
