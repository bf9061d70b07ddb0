// K6 -- fused EfficientPyrPool body: everything between the 1x1 projection and the final 1x1 convolution.
//
// Reference arithmetic (nn_layers/efficient_pyramid_pool.py:36-61, cnn_utils.py:108-125), per projected
// channel c (the unit is depthwise end to end, because Shuffle(groups=S) hands group c of the merge conv
// exactly channel c of each of the S branches):
//   scale > 1 : adaptive_avg_pool2d( dw3x3( bilinear_up(x_c) ) )            (align_corners=True)
//   scale = 1 : dw3x3(x_c)
//   scale < 1 : bilinear_up( dw3x3( adaptive_avg_pool2d(x_c) ) )  -- the low-res map E = dw3x3(pool(x)) is
//               tiny and produced beforehand by mspl_adaptive_avgpool_fwd + mspl_conv3x3_fwd
//   cat -> BatchNorm+PReLU (merge_layer.0) -> Shuffle -> grouped 3x3 (merge_layer.2) -> BatchNorm+PReLU.
//
// The unfused form moves the 2x and 1.5x resolution intermediates through HBM three times each
// (37% of all activation traffic of a forward).  Here a workgroup owns a
// TH x TW output tile of one (image, channel) plane: it stages the x tile (+4 halo) in LDS, builds the
// up-sampled tiles in LDS, evaluates all S branches (+1 halo) with the folded BN+PReLU into LDS, runs the
// merge convolution from LDS and writes one plane tile.  HBM traffic: read x once, write y once.
#include <stdlib.h>

#include "common.hpp"

namespace mspl {

constexpr int PYR_MAXB = 5;
constexpr int PYR_HALO = 4;      // x-tile halo that covers every bilinear source of the up-sampled tiles

struct PyrGeom {
    int N, P, h, w, nb;
    int kind[PYR_MAXB];          // 0: up (hs > h), 1: same, 2: down
    int hs[PYR_MAXB], ws[PYR_MAXB];
    float sh[PYR_MAXB], sw[PYR_MAXB];   // bilinear scales (up: x -> U grid, down: E grid -> output grid)
    const float* stage_w[PYR_MAXB];     // (P,1,3,3) depthwise weights (up / same branches)
    const float* down_e[PYR_MAXB];      // (N,P,hs,ws) conv'd low-res maps (down branches)
    const float* br_scale; const float* br_shift; const float* br_alpha;   // nb*P each (merge_layer.0)
    const float* merge_w;               // (P, nb, 3, 3)
    int TH, TW, tiles_y, tiles_x;
    int XW;                             // x tile row stride (TW + 2*HALO)
    int UH[PYR_MAXB], UW[PYR_MAXB], uoff[PYR_MAXB];   // up-sampled tile dims / LDS offsets (floats)
    int toff[PYR_MAXB];                 // per-branch index/weight tables (see fill_tables)
    int boff, BW;                       // branch tiles: nb x (TH+2) x BW
    int woff;                           // per-plane constants
    unsigned mag_uw[PYR_MAXB], mag_bw;  // exact small-range division by UW[i] / (TW+2): (t * magic) >> 24
    int CPB, cblocks;                   // planes (channels) per workgroup, P / CPB
    int stop_after;                     // tuning aid (MSPL_PYR_STOP): return after phase k; 0 = run everything
    float* zcat;                        // training forward: (N, nb*P, h, w), the branch values BEFORE merge_layer.0 (torch.cat order), or null
};

__device__ __forceinline__ int ada_s(int o, int I, int O) { return (int)(((unsigned)o * (unsigned)I) / (unsigned)O); }
__device__ __forceinline__ int ada_e(int o, int I, int O) { return (int)((((unsigned)(o + 1)) * (unsigned)I + O - 1) / (unsigned)O); }
__device__ __forceinline__ int fdiv(int t, unsigned magic) { return (int)(((unsigned)t * magic) >> 24); }

// Table layout of branch i inside LDS (float words; ints stored bit-cast):
//   up   : UR[UH][4] = {ya_l, yb_l, wy0, wy1}, UC[UW][4] = {xa_l, xb_l, wx0, wx1}   (bilinear sources in the x tile;
//          entries outside the hs x ws grid carry zero weights -> U = 0 = the dw conv's zero padding)
//          WR[BH][2] = {first U-tile row of the 3x3-extended window, rows}, WC[BW2][2] likewise for columns
//   down : DR[BH][4] = {ya*ws, yb*ws, wy0, wy1}, DC[BW2][4] = {xa, xb, wx0, wx1}    (bilinear sources in E)
__global__ __launch_bounds__(256) void pyrpool_fused_kernel(const float* __restrict__ x, PyrGeom g, Epi e,
                                                            float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                               // (TH+8) x XW
    float* wl = smem + g.woff;                      // [nb][9] stage weights, [nb][9] merge weights, [nb][3] BR consts
    int bid = blockIdx.x;
    const int txi = bid % g.tiles_x;  bid /= g.tiles_x;
    const int tyi = bid % g.tiles_y;  bid /= g.tiles_y;
    const int cb = bid % g.cblocks;
    const int n = bid / g.cblocks;
    const int y0 = tyi * g.TH, x0 = txi * g.TW;
    const int tid = threadIdx.x;
    const int XH = g.TH + 2 * PYR_HALO;
    const int BH = g.TH + 2, BW2 = g.TW + 2;
    const int py_lo = max(y0 - 1, 0), px_lo = max(x0 - 1, 0);

    // ---- phase 0 (once per workgroup): index / weight tables; they depend on the tile only, not on the plane
#pragma unroll
    for (int i = 0; i < PYR_MAXB; ++i) {
        if (i >= g.nb) break;
        float* T = smem + g.toff[i];
        if (g.kind[i] == 0) {
            const int UH = g.UH[i], UW = g.UW[i];
            const int u0 = ada_s(py_lo, g.hs[i], g.h) - 1, v0 = ada_s(px_lo, g.ws[i], g.w) - 1;
            float* UR = T; float* UC = UR + 4 * UH; float* WR = UC + 4 * UW; float* WC = WR + 2 * BH;
            for (int t = tid; t < UH + UW + BH + BW2; t += 256) {
                if (t < UH) {
                    const int u = u0 + t;
                    int ya = 0, yb = 0;  float w0 = 0.f, w1 = 0.f;
                    if (u >= 0 && u < g.hs[i]) bilinear_src(g.sh[i], u, g.h, ya, yb, w0, w1); else { ya = y0; yb = y0; }
                    UR[4 * t] = __int_as_float((ya - y0 + PYR_HALO) * g.XW); UR[4 * t + 1] = __int_as_float((yb - y0 + PYR_HALO) * g.XW);
                    UR[4 * t + 2] = w0; UR[4 * t + 3] = w1;
                } else if (t < UH + UW) {
                    const int q = t - UH, v = v0 + q;
                    int xa = 0, xb = 0;  float w0 = 0.f, w1 = 0.f;
                    if (v >= 0 && v < g.ws[i]) bilinear_src(g.sw[i], v, g.w, xa, xb, w0, w1); else { xa = x0; xb = x0; }
                    UC[4 * q] = __int_as_float(xa - x0 + PYR_HALO); UC[4 * q + 1] = __int_as_float(xb - x0 + PYR_HALO);
                    UC[4 * q + 2] = w0; UC[4 * q + 3] = w1;
                } else if (t < UH + UW + BH) {
                    const int r = t - UH - UW, py = y0 - 1 + r;
                    int first = 0, cnt = 0;
                    if (py >= 0 && py < g.h) { const int us = ada_s(py, g.hs[i], g.h); first = us - 1 - u0; cnt = ada_e(py, g.hs[i], g.h) - us; }
                    WR[2 * r] = __int_as_float(first * UW); WR[2 * r + 1] = __int_as_float(cnt);
                } else {
                    const int q = t - UH - UW - BH, px = x0 - 1 + q;
                    int first = 0, cnt = 0;
                    if (px >= 0 && px < g.w) { const int vs = ada_s(px, g.ws[i], g.w); first = vs - 1 - v0; cnt = ada_e(px, g.ws[i], g.w) - vs; }
                    WC[2 * q] = __int_as_float(first); WC[2 * q + 1] = __int_as_float(cnt);
                }
            }
        } else if (g.kind[i] == 2) {
            float* DR = T; float* DC = DR + 4 * BH;
            for (int t = tid; t < BH + BW2; t += 256) {
                if (t < BH) {
                    const int py = min(max(y0 - 1 + t, 0), g.h - 1);
                    int ya, yb;  float w0, w1;
                    bilinear_src(g.sh[i], py, g.hs[i], ya, yb, w0, w1);
                    DR[4 * t] = __int_as_float(ya * g.ws[i]); DR[4 * t + 1] = __int_as_float(yb * g.ws[i]); DR[4 * t + 2] = w0; DR[4 * t + 3] = w1;
                } else {
                    const int q = t - BH, px = min(max(x0 - 1 + q, 0), g.w - 1);
                    int xa, xb;  float w0, w1;
                    bilinear_src(g.sw[i], px, g.ws[i], xa, xb, w0, w1);
                    DC[4 * q] = __int_as_float(xa); DC[4 * q + 1] = __int_as_float(xb); DC[4 * q + 2] = w0; DC[4 * q + 3] = w1;
                }
            }
        }
    }

    // The workgroup walks CPB planes of its tile; the tables above are reused, and the next plane's x tile and
    // constants are fetched into registers while the current plane is being computed.
    const int xv = g.XW >> 2;                       // float4 per x-tile row
    const bool w4 = (g.w & 3) == 0;
    auto load_x = [&](int c, int i) -> float4 {     // element i of the x tile of plane c (zero outside the image)
        const float* xp = x + ((size_t)n * g.P + c) * (size_t)g.h * g.w;
        const int r = i / xv, v = i - r * xv;
        const int iy = y0 - PYR_HALO + r, ix = x0 - PYR_HALO + 4 * v;
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy >= 0 && iy < g.h) {
            const float* row = xp + (size_t)iy * g.w;
            if (w4 && ix >= 0 && ix + 3 < g.w) {
                val = *reinterpret_cast<const float4*>(row + ix);
            } else {
                if (ix >= 0 && ix < g.w) val.x = row[ix];
                if (ix + 1 >= 0 && ix + 1 < g.w) val.y = row[ix + 1];
                if (ix + 2 >= 0 && ix + 2 < g.w) val.z = row[ix + 2];
                if (ix + 3 >= 0 && ix + 3 < g.w) val.w = row[ix + 3];
            }
        }
        return val;
    };
    auto load_const = [&](int c, int t) -> float {   // t < 2*nb*9 + 3*nb
        const int n9 = g.nb * 9;
        if (t < n9) { const int i = t / 9, k = t - i * 9; return (g.kind[i] != 2) ? g.stage_w[i][c * 9 + k] : 0.f; }
        if (t < 2 * n9) { const int u = t - n9, i = u / 9, k = u - i * 9; return g.merge_w[((size_t)c * g.nb + i) * 9 + k]; }
        const int u = t - 2 * n9, i = u / 3, k = u - i * 3;
        return (k == 0 ? g.br_scale : (k == 1 ? g.br_shift : g.br_alpha))[i * g.P + c];
    };
    const int nx = XH * xv;                         // <= 2 float4 per thread for the default 16x32 tile
    const int nconst = 2 * g.nb * 9 + 3 * g.nb;
    const int c_first = cb * g.CPB;
    float4 xr0 = make_float4(0.f, 0.f, 0.f, 0.f), xr1 = xr0;
    float cr = 0.f;
    if (tid < nx) xr0 = load_x(c_first, tid);
    if (tid + 256 < nx) xr1 = load_x(c_first, tid + 256);
    if (tid < nconst) cr = load_const(c_first, tid);

    for (int ci = 0; ci < g.CPB; ++ci) {
    const int c = c_first + ci;
    // ---- phase 1: registers -> LDS, then prefetch the next plane
    if (tid < nx) *reinterpret_cast<float4*>(xs + (tid / xv) * g.XW + 4 * (tid % xv)) = xr0;
    if (tid + 256 < nx) *reinterpret_cast<float4*>(xs + ((tid + 256) / xv) * g.XW + 4 * ((tid + 256) % xv)) = xr1;
    for (int i = tid + 512; i < nx; i += 256) *reinterpret_cast<float4*>(xs + (i / xv) * g.XW + 4 * (i % xv)) = load_x(c, i);
    if (tid < nconst) wl[tid] = cr;
    if (ci + 1 < g.CPB) {
        if (tid < nx) xr0 = load_x(c + 1, tid);
        if (tid + 256 < nx) xr1 = load_x(c + 1, tid + 256);
        if (tid < nconst) cr = load_const(c + 1, tid);
    }
    __syncthreads();
    if (g.stop_after == 1) return;

    // ---- phase 2: up-sampled tiles U_i from the tables (no divisions, no branches)
#pragma unroll
    for (int i = 0; i < PYR_MAXB; ++i) {
        if (i < g.nb && g.kind[i] == 0) {
            float* U = smem + g.uoff[i];
            const int UH = g.UH[i], UW = g.UW[i];
            const float* UR = smem + g.toff[i];
            const float* UC = UR + 4 * UH;
            const unsigned mag = g.mag_uw[i];
            const int nU = UH * UW;
            for (int t = tid; t < nU; t += 256) {
                const int r = fdiv(t, mag), q = t - r * UW;
                const float4 rr = *reinterpret_cast<const float4*>(UR + 4 * r);
                const float4 cc = *reinterpret_cast<const float4*>(UC + 4 * q);
                const float* ra = xs + __float_as_int(rr.x);
                const float* rb = xs + __float_as_int(rr.y);
                const int xa = __float_as_int(cc.x), xb = __float_as_int(cc.y);
                const float top = cc.z * ra[xa] + cc.w * ra[xb];
                const float bot = cc.z * rb[xa] + cc.w * rb[xb];
                U[t] = rr.z * top + rr.w * bot;
            }
        }
    }
    __syncthreads();
    if (g.stop_after == 2) return;

    // ---- phase 3: branch values at every position of the (TH+2) x (TW+2) halo tile, BN+PReLU'd; zero outside
    // the image (the merge convolution's zero padding applies AFTER merge_layer.0).  One specialised loop per
    // branch (the branch kind is uniform), so the inner code has no kind dispatch.
    float* B = smem + g.boff;
    const int npos = BH * BW2;
    const int XWl = g.XW, BWl = g.BW, hl = g.h, wl_ = g.w;
    const unsigned magb = g.mag_bw;
    // training forward: the branch value before BN+PReLU at the tile's own (non-halo) positions goes to zcat (the backward needs it
    // for the PReLU sign and d gamma); position (r, q) of the halo tile is pixel (y0 - 1 + r, x0 - 1 + q)
    auto keep_raw = [&](int i, int r, int q, int py, int px, float b) {
        if (g.zcat && r >= 1 && r <= g.TH && q >= 1 && q <= g.TW)
            g.zcat[(((size_t)n * g.nb + i) * g.P + c) * (size_t)hl * wl_ + (size_t)py * wl_ + px] = b;
    };
#pragma unroll
    for (int i = 0; i < PYR_MAXB; ++i) {
        if (i >= g.nb) break;
        const float* ws9 = wl + i * 9;
        const float* kbr = wl + 2 * g.nb * 9 + i * 3;
        const float bsc = kbr[0], bsh = kbr[1], bal = kbr[2];
        float* Bi = B + i * BH * BWl;
        const float* T = smem + g.toff[i];
        if (g.kind[i] == 1) {
            const float w00 = ws9[0], w01 = ws9[1], w02 = ws9[2], w10 = ws9[3], w11 = ws9[4], w12 = ws9[5],
                        w20 = ws9[6], w21 = ws9[7], w22 = ws9[8];
            for (int t = tid; t < npos; t += 256) {
                const int r = fdiv(t, magb), q = t - r * BW2;
                const int py = y0 - 1 + r, px = x0 - 1 + q;
                float b = 0.f;
                if (py >= 0 && py < hl && px >= 0 && px < wl_) {
                    const float* p = xs + (r + PYR_HALO - 2) * XWl + (q + PYR_HALO - 2);
                    b = w00 * p[0];
                    b = fmaf(w01, p[1], b); b = fmaf(w02, p[2], b);
                    b = fmaf(w10, p[XWl], b); b = fmaf(w11, p[XWl + 1], b); b = fmaf(w12, p[XWl + 2], b);
                    b = fmaf(w20, p[2 * XWl], b); b = fmaf(w21, p[2 * XWl + 1], b); b = fmaf(w22, p[2 * XWl + 2], b);
                    keep_raw(i, r, q, py, px, b);
                    b = fmaf(b, bsc, bsh);
                    b = b > 0.f ? b : bal * b;
                }
                Bi[r * BWl + q] = b;
            }
        } else if (g.kind[i] == 0) {
            const float* U = smem + g.uoff[i];
            const int UW = g.UW[i];
            const float* WR = T + 4 * g.UH[i] + 4 * UW;
            const float* WC = WR + 2 * BH;
            const float w00 = ws9[0], w01 = ws9[1], w02 = ws9[2], w10 = ws9[3], w11 = ws9[4], w12 = ws9[5],
                        w20 = ws9[6], w21 = ws9[7], w22 = ws9[8];
            for (int t = tid; t < npos; t += 256) {
                const int r = fdiv(t, magb), q = t - r * BW2;
                const int py = y0 - 1 + r, px = x0 - 1 + q;
                float b = 0.f;
                if (py >= 0 && py < hl && px >= 0 && px < wl_) {
                    const float2 wr = *reinterpret_cast<const float2*>(WR + 2 * r);
                    const float2 wc = *reinterpret_cast<const float2*>(WC + 2 * q);
                    const int rcnt = __float_as_int(wr.y), ccnt = __float_as_int(wc.y);
                    const float* p0 = U + __float_as_int(wr.x) + __float_as_int(wc.x);
                    float s = 0.f;
                    if (rcnt == 2 && ccnt == 2) {
                        // 2x2 pooling window (scale 2.0 and 1.5 on even maps): a 4x4 block of U, 4 conv results
                        float u[4][4];
#pragma unroll
                        for (int a2 = 0; a2 < 4; ++a2)
#pragma unroll
                            for (int b2 = 0; b2 < 4; ++b2) u[a2][b2] = p0[a2 * UW + b2];
#pragma unroll
                        for (int du = 0; du < 2; ++du)
#pragma unroll
                            for (int dv = 0; dv < 2; ++dv) {
                                float cv = w00 * u[du][dv];
                                cv = fmaf(w01, u[du][dv + 1], cv); cv = fmaf(w02, u[du][dv + 2], cv);
                                cv = fmaf(w10, u[du + 1][dv], cv); cv = fmaf(w11, u[du + 1][dv + 1], cv); cv = fmaf(w12, u[du + 1][dv + 2], cv);
                                cv = fmaf(w20, u[du + 2][dv], cv); cv = fmaf(w21, u[du + 2][dv + 1], cv); cv = fmaf(w22, u[du + 2][dv + 2], cv);
                                s += cv;
                            }
                        b = s * 0.25f;
                    } else {
                        for (int du = 0; du < rcnt; ++du)
                            for (int dv = 0; dv < ccnt; ++dv) {
                                const float* p = p0 + du * UW + dv;
                                float cv = w00 * p[0];
                                cv = fmaf(w01, p[1], cv); cv = fmaf(w02, p[2], cv);
                                cv = fmaf(w10, p[UW], cv); cv = fmaf(w11, p[UW + 1], cv); cv = fmaf(w12, p[UW + 2], cv);
                                cv = fmaf(w20, p[2 * UW], cv); cv = fmaf(w21, p[2 * UW + 1], cv); cv = fmaf(w22, p[2 * UW + 2], cv);
                                s += cv;
                            }
                        b = s / (float)(rcnt * ccnt);
                    }
                    keep_raw(i, r, q, py, px, b);
                    b = fmaf(b, bsc, bsh);
                    b = b > 0.f ? b : bal * b;
                }
                Bi[r * BWl + q] = b;
            }
        } else {
            const float* E = g.down_e[i] + ((size_t)n * g.P + c) * (size_t)g.hs[i] * g.ws[i];
            for (int t = tid; t < npos; t += 256) {
                const int r = fdiv(t, magb), q = t - r * BW2;
                const int py = y0 - 1 + r, px = x0 - 1 + q;
                float b = 0.f;
                if (py >= 0 && py < hl && px >= 0 && px < wl_) {
                    const float4 rr = *reinterpret_cast<const float4*>(T + 4 * r);
                    const float4 cc = *reinterpret_cast<const float4*>(T + 4 * BH + 4 * q);
                    const float* ra = E + __float_as_int(rr.x);
                    const float* rb = E + __float_as_int(rr.y);
                    const int xa = __float_as_int(cc.x), xb = __float_as_int(cc.y);
                    const float top = cc.z * ra[xa] + cc.w * ra[xb];
                    const float bot = cc.z * rb[xa] + cc.w * rb[xb];
                    b = rr.z * top + rr.w * bot;
                    keep_raw(i, r, q, py, px, b);
                    b = fmaf(b, bsc, bsh);
                    b = b > 0.f ? b : bal * b;
                }
                Bi[r * BWl + q] = b;
            }
        }
    }
    __syncthreads();
    if (g.stop_after == 3) return;

    // ---- phase 4: merge convolution (sum over branches of a 3x3) + BN + PReLU, 1x4 strips
    const int cabs = e.coff + c;
    const EpiCh ec = epi_channel(e, cabs);
    const int XS = g.TW >> 2;
    const int nstrip = g.TH * XS, nbl = g.nb;
    for (int t = tid; t < nstrip; t += 256) {
        const int ty = t / XS, xsi = t - ty * XS;
        const int y = y0 + ty, xb = x0 + xsi * 4;
        if (y >= hl || xb >= wl_) continue;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < PYR_MAXB; ++i) {
            if (i >= nbl) break;
            const float* wm = wl + nbl * 9 + i * 9;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float* row = B + (i * BH + ty + ky) * BWl + xsi * 4;
                const float4 a = *reinterpret_cast<const float4*>(row);
                const float2 b2 = *reinterpret_cast<const float2*>(row + 4);
                const float rv[6] = {a.x, a.y, a.z, a.w, b2.x, b2.y};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j] = fmaf(wm[ky * 3 + 0], rv[j], acc[j]);
                    acc[j] = fmaf(wm[ky * 3 + 1], rv[j + 1], acc[j]);
                    acc[j] = fmaf(wm[ky * 3 + 2], rv[j + 2], acc[j]);
                }
            }
        }
        const int pix = y * wl_ + xb;
        float* dst = out + epi_offset(e, n, cabs, pix);
        if (e.raw) {                       // training forward: the bare merge convolution result (un-sliced destination)
            float* rdst = e.raw + ((size_t)n * g.P + c) * (size_t)hl * wl_ + pix;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (xb + j < wl_) rdst[j] = acc[j];
        }
        if ((wl_ & 3) == 0) {
            store_out4(dst, epi_apply4(e, ec, acc, n, cabs, pix));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (xb + j < wl_) dst[j] = epi_apply(e, ec, acc[j], n, cabs, pix + j);
        }
    }
    __syncthreads();   // the next plane overwrites xs / wl / B
    }
}

int pyrpool_sep_try(const float* x, int N, int P, int h, int w, int nb, const int32_t* hs, const int32_t* ws,
                    const float* const* stage_w, const float* const* down_e, const float* br_scale,
                    const float* br_shift, const float* br_alpha, const float* merge_w, const Epi& e, float* out,
                    hipStream_t stream, unsigned launch_flags);       // pyrpool_sep.hip
int pyrpool_stream_try(const float* x, int N, int P, int h, int w, int nb, const int32_t* hs, const int32_t* ws,
                       const float* const* stage_w, const float* const* down_e, const float* br_scale,
                       const float* br_shift, const float* br_alpha, const float* merge_w, const Epi& e, float* out,
                       hipStream_t stream, float* zcat = nullptr);    // pyrpool_stream.hip

}  // namespace mspl

using namespace mspl;

static int pyrpool_fused_launch(const float* x, int32_t N, int32_t P, int32_t h, int32_t w, int32_t nb,
                               const int32_t* hs, const int32_t* ws, const float* const* stage_w,
                               const float* const* down_e, const float* br_scale, const float* br_shift,
                               const float* br_alpha, const float* merge_w, const mspl_epilogue_t* ep,
                               float* out, float* zcat, void* stream, bool dry = false) {
    // dry: shape / LDS planning only (mspl_pyrpool_fused_train_fits): no pointer is looked at, nothing is launched
    MSPL_REQUIRE(dry || (x && hs && ws && stage_w && down_e && br_scale && br_shift && br_alpha && merge_w && out),
                 MSPL_ERR_NULL_POINTER, "pyrpool_fused: null pointer");
    MSPL_REQUIRE(hs && ws, MSPL_ERR_NULL_POINTER, "pyrpool_fused: null pointer");
    MSPL_REQUIRE(N > 0 && P > 0 && h > 0 && w > 0, MSPL_ERR_BAD_SHAPE, "pyrpool_fused: bad shape N=%d P=%d %dx%d", N, P, h, w);
    MSPL_REQUIRE(nb >= 1 && nb <= PYR_MAXB, MSPL_ERR_UNSUPPORTED, "pyrpool_fused: %d branches (1..%d)", nb, PYR_MAXB);
    if (int rc = check_epi(ep, P, "pyrpool_fused", zcat != nullptr)) return rc;
    if (!dry) {   // register-streaming form first, then the LDS-tiled stencil form; shapes they do not cover fall through to the table-driven kernel
        static const int force_tables = MSPL_TUNE_INT("MSPL_PYR_TABLES", 0);
        if (!force_tables) {
            // (the training forward -- zcat wanted -- exists in the streaming and in the table form)
            const int rc3 = pyrpool_stream_try(x, N, P, h, w, nb, hs, ws, stage_w, down_e, br_scale, br_shift, br_alpha, merge_w,
                                               make_epi(ep, P, h * w), out, (hipStream_t)stream, zcat);
            if (rc3 <= 0) return rc3;
            if (!zcat) {
                const int rc = pyrpool_sep_try(x, N, P, h, w, nb, hs, ws, stage_w, down_e, br_scale, br_shift, br_alpha, merge_w,
                                               make_epi(ep, P, h * w), out, (hipStream_t)stream, epi_flags(ep));
                if (rc <= 0) return rc;
            }
        }
    }
    PyrGeom g;
    memset(&g, 0, sizeof(g));
    g.N = N; g.P = P; g.h = h; g.w = w; g.nb = nb;
    g.br_scale = br_scale; g.br_shift = br_shift; g.br_alpha = br_alpha; g.merge_w = merge_w;
    g.zcat = zcat;
    g.TW = w >= 32 ? 32 : ((w + 3) & ~3);
    g.TH = h >= 16 ? 16 : h;
    g.tiles_x = ceil_div(w, g.TW);
    g.tiles_y = ceil_div(h, g.TH);
    g.XW = g.TW + 2 * PYR_HALO;
    int off = (g.TH + 2 * PYR_HALO) * g.XW;
    off = (off + 3) & ~3;
    for (int i = 0; i < nb; ++i) {
        MSPL_REQUIRE(hs[i] > 0 && ws[i] > 0, MSPL_ERR_BAD_SHAPE, "pyrpool_fused: branch %d size %dx%d", i, hs[i], ws[i]);
        g.hs[i] = hs[i]; g.ws[i] = ws[i];
        g.stage_w[i] = dry ? nullptr : stage_w[i]; g.down_e[i] = dry ? nullptr : down_e[i];
        if (hs[i] == h && ws[i] == w) {
            g.kind[i] = 1;
            MSPL_REQUIRE(dry || stage_w[i], MSPL_ERR_NULL_POINTER, "pyrpool_fused: branch %d needs stage weights", i);
        } else if (hs[i] >= h && ws[i] >= w) {
            g.kind[i] = 0;
            MSPL_REQUIRE(dry || stage_w[i], MSPL_ERR_NULL_POINTER, "pyrpool_fused: branch %d needs stage weights", i);
            // windows of an up-sampled branch must stay small (x-tile halo of 4 covers scales in [1, 4])
            MSPL_REQUIRE(hs[i] <= 4 * h && ws[i] <= 4 * w, MSPL_ERR_UNSUPPORTED, "pyrpool_fused: up-scale beyond 4x");
            g.sh[i] = bilinear_scale(h, hs[i]); g.sw[i] = bilinear_scale(w, ws[i]);
            g.UH[i] = (int)(((int64_t)(g.TH + 2) * hs[i] + h - 1) / h) + 4;
            g.UW[i] = (int)(((int64_t)(g.TW + 2) * ws[i] + w - 1) / w) + 4;
            g.uoff[i] = off;
            off += g.UH[i] * g.UW[i];
            off = (off + 3) & ~3;
        } else if (hs[i] <= h && ws[i] <= w) {
            g.kind[i] = 2;
            MSPL_REQUIRE(dry || down_e[i], MSPL_ERR_NULL_POINTER, "pyrpool_fused: branch %d needs its low-resolution map", i);
            g.sh[i] = bilinear_scale(hs[i], h); g.sw[i] = bilinear_scale(ws[i], w);
        } else {
            set_error("pyrpool_fused: branch %d mixes up- and down-sampling (%dx%d vs %dx%d)", i, hs[i], ws[i], h, w);
            return MSPL_ERR_UNSUPPORTED;
        }
    }
    auto magic = [](int d) { return (unsigned)(((1u << 24) + (unsigned)d - 1) / (unsigned)d); };
    for (int i = 0; i < nb; ++i) {       // per-branch tables (16-byte aligned)
        g.toff[i] = off;
        if (g.kind[i] == 0) { off += 4 * g.UH[i] + 4 * g.UW[i] + 2 * (g.TH + 2) + 2 * (g.TW + 2); g.mag_uw[i] = magic(g.UW[i]); }
        else if (g.kind[i] == 2) off += 4 * (g.TH + 2) + 4 * (g.TW + 2);
        off = (off + 3) & ~3;
    }
    g.mag_bw = magic(g.TW + 2);
    static const int dbg_stop = MSPL_TUNE_INT("MSPL_PYR_STOP", 0);
    g.stop_after = dbg_stop;
    g.BW = (g.TW + 2 + 3 + 4) & ~3;      // halo tile row stride, 16-byte aligned rows, +4 for the strip over-read
    g.boff = off;
    off += nb * (g.TH + 2) * g.BW;
    off = (off + 3) & ~3;
    g.woff = off;
    off += nb * 9 * 2 + nb * 3;
    const size_t lds = (size_t)off * sizeof(float);
    MSPL_REQUIRE(lds <= 64 * 1024, MSPL_ERR_UNSUPPORTED, "pyrpool_fused: tile needs %zu B of LDS", lds);
    if (dry) return MSPL_OK;
    const Epi e = make_epi(ep, P, h * w);
    int cpb = 1;
    while (cpb * 2 <= P && P % (cpb * 2) == 0 && (int64_t)N * (P / (cpb * 2)) * g.tiles_y * g.tiles_x >= 2048) cpb *= 2;
    static const int dbg_cpb = MSPL_TUNE_INT("MSPL_PYR_CPB", 0);
    if (dbg_cpb > 0 && P % dbg_cpb == 0) cpb = dbg_cpb;
    g.CPB = cpb; g.cblocks = P / cpb;
    const int64_t blocks = (int64_t)N * g.cblocks * g.tiles_y * g.tiles_x;
    MSPL_REQUIRE(blocks < (1ll << 31), MSPL_ERR_BAD_SHAPE, "pyrpool_fused: grid too large");
    hipLaunchKernelGGL(pyrpool_fused_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, x, g, e, out);
    MSPL_CHECK_LAUNCH("pyrpool_fused");
    return MSPL_OK;
}

extern "C" int mspl_pyrpool_fused_fwd(const float* x, int32_t N, int32_t P, int32_t h, int32_t w, int32_t nb,
                                      const int32_t* hs, const int32_t* ws, const float* const* stage_w,
                                      const float* const* down_e, const float* br_scale, const float* br_shift,
                                      const float* br_alpha, const float* merge_w, const mspl_epilogue_t* ep,
                                      float* out, void* stream) {
    return pyrpool_fused_launch(x, N, P, h, w, nb, hs, ws, stage_w, down_e, br_scale, br_shift, br_alpha, merge_w, ep, out, nullptr,
                                stream);
}

extern "C" int mspl_pyrpool_fused_train_fwd(const float* x, int32_t N, int32_t P, int32_t h, int32_t w, int32_t nb,
                                            const int32_t* hs, const int32_t* ws, const float* const* stage_w,
                                            const float* const* down_e, const float* br_scale, const float* br_shift,
                                            const float* br_alpha, const float* merge_w, const mspl_epilogue_t* ep,
                                            float* out, float* zcat, void* stream) {
    MSPL_REQUIRE(zcat, MSPL_ERR_NULL_POINTER, "pyrpool_fused_train: zcat is required");
    return pyrpool_fused_launch(x, N, P, h, w, nb, hs, ws, stage_w, down_e, br_scale, br_shift, br_alpha, merge_w, ep, out, zcat,
                                stream);
}

extern "C" int mspl_pyrpool_fused_train_fits(int32_t N, int32_t P, int32_t h, int32_t w, int32_t nb, const int32_t* hs,
                                             const int32_t* ws) {
    float dummy = 0.f;
    return pyrpool_fused_launch(nullptr, N, P, h, w, nb, hs, ws, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                &dummy, nullptr, true) == MSPL_OK;
}
