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
    int boff, BW;                       // branch tiles: nb x (TH+2) x BW
    int woff;                           // per-plane constants
};

__device__ __forceinline__ int ada_s(int o, int I, int O) { return (int)(((int64_t)o * I) / O); }
__device__ __forceinline__ int ada_e(int o, int I, int O) { return (int)((((int64_t)(o + 1)) * I + O - 1) / O); }

__global__ __launch_bounds__(256) void pyrpool_fused_kernel(const float* __restrict__ x, PyrGeom g, Epi e,
                                                            float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                               // (TH+8) x XW
    float* wl = smem + g.woff;                      // [nb][9] stage weights, [nb][9] merge weights, [nb][3] BR consts
    int bid = blockIdx.x;
    const int txi = bid % g.tiles_x;  bid /= g.tiles_x;
    const int tyi = bid % g.tiles_y;  bid /= g.tiles_y;
    const int c = bid % g.P;
    const int n = bid / g.P;
    const int y0 = tyi * g.TH, x0 = txi * g.TW;
    const int tid = threadIdx.x;
    const int XH = g.TH + 2 * PYR_HALO;
    const float* xp = x + ((size_t)n * g.P + c) * (size_t)g.h * g.w;

    // ---- phase 1: constants + x tile (zero outside the image)
    if (tid < g.nb * 9) {
        const int i = tid / 9, t = tid - i * 9;
        wl[tid] = (g.kind[i] != 2) ? g.stage_w[i][c * 9 + t] : 0.f;
        wl[g.nb * 9 + tid] = g.merge_w[((size_t)c * g.nb + i) * 9 + t];
    }
    if (tid < g.nb) {
        float* k = wl + 2 * g.nb * 9 + tid * 3;
        k[0] = g.br_scale[tid * g.P + c]; k[1] = g.br_shift[tid * g.P + c]; k[2] = g.br_alpha[tid * g.P + c];
    }
    for (int i = tid; i < XH * g.XW; i += 256) {
        const int r = i / g.XW, q = i - r * g.XW;
        const int iy = y0 - PYR_HALO + r, ix = x0 - PYR_HALO + q;
        xs[i] = (iy >= 0 && iy < g.h && ix >= 0 && ix < g.w) ? xp[(size_t)iy * g.w + ix] : 0.f;
    }
    __syncthreads();

    // output positions evaluated by the branch phase: the tile + 1 halo, clipped to the image
    const int py_lo = max(y0 - 1, 0), px_lo = max(x0 - 1, 0);

    // ---- phase 2: up-sampled tiles U_i (zero outside the hs x ws grid = the dw conv's zero padding)
    int u0[PYR_MAXB], v0[PYR_MAXB];
#pragma unroll
    for (int i = 0; i < PYR_MAXB; ++i) {
        u0[i] = 0; v0[i] = 0;
        if (i < g.nb && g.kind[i] == 0) {
            u0[i] = ada_s(py_lo, g.hs[i], g.h) - 1;
            v0[i] = ada_s(px_lo, g.ws[i], g.w) - 1;
            float* U = smem + g.uoff[i];
            const int UH = g.UH[i], UW = g.UW[i];
            for (int t = tid; t < UH * UW; t += 256) {
                const int r = t / UW, q = t - r * UW;
                const int u = u0[i] + r, v = v0[i] + q;
                float val = 0.f;
                if (u >= 0 && u < g.hs[i] && v >= 0 && v < g.ws[i]) {
                    int ya, yb, xa, xb;  float wy0, wy1, wx0, wx1;
                    bilinear_src(g.sh[i], u, g.h, ya, yb, wy0, wy1);
                    bilinear_src(g.sw[i], v, g.w, xa, xb, wx0, wx1);
                    const float* ra = xs + (ya - y0 + PYR_HALO) * g.XW + (PYR_HALO - x0);
                    const float* rb = xs + (yb - y0 + PYR_HALO) * g.XW + (PYR_HALO - x0);
                    const float top = wx0 * ra[xa] + wx1 * ra[xb];
                    const float bot = wx0 * rb[xa] + wx1 * rb[xb];
                    val = wy0 * top + wy1 * bot;
                }
                U[t] = val;
            }
        }
    }
    __syncthreads();

    // ---- phase 3: branch values at every position of the (TH+2) x (TW+2) halo tile, BN+PReLU'd; zero outside
    // the image (the merge convolution's zero padding applies AFTER merge_layer.0)
    float* B = smem + g.boff;
    const int BH = g.TH + 2;
    for (int t = tid; t < BH * (g.TW + 2); t += 256) {
        const int r = t / (g.TW + 2), q = t - r * (g.TW + 2);
        const int py = y0 - 1 + r, px = x0 - 1 + q;
        const bool inside = py >= 0 && py < g.h && px >= 0 && px < g.w;
#pragma unroll
        for (int i = 0; i < PYR_MAXB; ++i) {
            if (i >= g.nb) break;
            float b = 0.f;
            if (inside) {
                const float* ws9 = wl + i * 9;
                if (g.kind[i] == 1) {
                    const float* p0 = xs + (py - y0 + PYR_HALO - 1) * g.XW + (px - x0 + PYR_HALO - 1);
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) b = fmaf(ws9[ky * 3 + kx], p0[ky * g.XW + kx], b);
                } else if (g.kind[i] == 0) {
                    const float* U = smem + g.uoff[i];
                    const int UW = g.UW[i];
                    const int us = ada_s(py, g.hs[i], g.h), ue = ada_e(py, g.hs[i], g.h);
                    const int vs = ada_s(px, g.ws[i], g.w), ve = ada_e(px, g.ws[i], g.w);
                    float s = 0.f;
                    for (int u = us; u < ue; ++u)
                        for (int v = vs; v < ve; ++v) {
                            const float* p0 = U + (u - 1 - u0[i]) * UW + (v - 1 - v0[i]);
                            float cv = 0.f;
#pragma unroll
                            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                                for (int kx = 0; kx < 3; ++kx) cv = fmaf(ws9[ky * 3 + kx], p0[ky * UW + kx], cv);
                            s += cv;
                        }
                    b = s / (float)((ue - us) * (ve - vs));
                } else {
                    const float* E = g.down_e[i] + ((size_t)n * g.P + c) * (size_t)g.hs[i] * g.ws[i];
                    int ya, yb, xa, xb;  float wy0, wy1, wx0, wx1;
                    bilinear_src(g.sh[i], py, g.hs[i], ya, yb, wy0, wy1);
                    bilinear_src(g.sw[i], px, g.ws[i], xa, xb, wx0, wx1);
                    const float top = wx0 * E[ya * g.ws[i] + xa] + wx1 * E[ya * g.ws[i] + xb];
                    const float bot = wx0 * E[yb * g.ws[i] + xa] + wx1 * E[yb * g.ws[i] + xb];
                    b = wy0 * top + wy1 * bot;
                }
                const float* k = wl + 2 * g.nb * 9 + i * 3;
                b = fmaf(b, k[0], k[1]);
                b = b > 0.f ? b : k[2] * b;
            }
            B[(i * BH + r) * g.BW + q] = b;
        }
    }
    __syncthreads();

    // ---- phase 4: merge convolution (sum over branches of a 3x3) + BN + PReLU, 1x4 strips
    const int cabs = e.coff + c;
    const EpiCh ec = epi_channel(e, cabs);
    const int XS = g.TW >> 2;
    for (int t = tid; t < g.TH * XS; t += 256) {
        const int ty = t / XS, xsi = t - ty * XS;
        const int y = y0 + ty, xb = x0 + xsi * 4;
        if (y >= g.h || xb >= g.w) continue;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < PYR_MAXB; ++i) {
            if (i >= g.nb) break;
            const float* wm = wl + g.nb * 9 + i * 9;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float* row = B + (i * BH + ty + ky) * g.BW + xsi * 4;
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
        const int pix = y * g.w + xb;
        float* dst = out + epi_offset(e, n, cabs, pix);
        if ((g.w & 3) == 0) {
            float4 v;
            v.x = epi_apply(e, ec, acc[0], n, cabs, pix);
            v.y = epi_apply(e, ec, acc[1], n, cabs, pix + 1);
            v.z = epi_apply(e, ec, acc[2], n, cabs, pix + 2);
            v.w = epi_apply(e, ec, acc[3], n, cabs, pix + 3);
            *reinterpret_cast<float4*>(dst) = v;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (xb + j < g.w) dst[j] = epi_apply(e, ec, acc[j], n, cabs, pix + j);
        }
    }
}

}  // namespace mspl

using namespace mspl;

extern "C" int mspl_pyrpool_fused_fwd(const float* x, int32_t N, int32_t P, int32_t h, int32_t w, int32_t nb,
                                      const int32_t* hs, const int32_t* ws, const float* const* stage_w,
                                      const float* const* down_e, const float* br_scale, const float* br_shift,
                                      const float* br_alpha, const float* merge_w, const mspl_epilogue_t* ep,
                                      float* out, void* stream) {
    MSPL_REQUIRE(x && hs && ws && stage_w && down_e && br_scale && br_shift && br_alpha && merge_w && out,
                 MSPL_ERR_NULL_POINTER, "pyrpool_fused: null pointer");
    MSPL_REQUIRE(N > 0 && P > 0 && h > 0 && w > 0, MSPL_ERR_BAD_SHAPE, "pyrpool_fused: bad shape N=%d P=%d %dx%d", N, P, h, w);
    MSPL_REQUIRE(nb >= 1 && nb <= PYR_MAXB, MSPL_ERR_UNSUPPORTED, "pyrpool_fused: %d branches (1..%d)", nb, PYR_MAXB);
    if (int rc = check_epi(ep, P, "pyrpool_fused")) return rc;
    PyrGeom g;
    memset(&g, 0, sizeof(g));
    g.N = N; g.P = P; g.h = h; g.w = w; g.nb = nb;
    g.br_scale = br_scale; g.br_shift = br_shift; g.br_alpha = br_alpha; g.merge_w = merge_w;
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
        g.stage_w[i] = stage_w[i]; g.down_e[i] = down_e[i];
        if (hs[i] == h && ws[i] == w) {
            g.kind[i] = 1;
            MSPL_REQUIRE(stage_w[i], MSPL_ERR_NULL_POINTER, "pyrpool_fused: branch %d needs stage weights", i);
        } else if (hs[i] >= h && ws[i] >= w) {
            g.kind[i] = 0;
            MSPL_REQUIRE(stage_w[i], MSPL_ERR_NULL_POINTER, "pyrpool_fused: branch %d needs stage weights", i);
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
            MSPL_REQUIRE(down_e[i], MSPL_ERR_NULL_POINTER, "pyrpool_fused: branch %d needs its low-resolution map", i);
            g.sh[i] = bilinear_scale(hs[i], h); g.sw[i] = bilinear_scale(ws[i], w);
        } else {
            set_error("pyrpool_fused: branch %d mixes up- and down-sampling (%dx%d vs %dx%d)", i, hs[i], ws[i], h, w);
            return MSPL_ERR_UNSUPPORTED;
        }
    }
    g.BW = (g.TW + 2 + 3 + 4) & ~3;      // halo tile row stride, 16-byte aligned rows, +4 for the strip over-read
    g.boff = off;
    off += nb * (g.TH + 2) * g.BW;
    off = (off + 3) & ~3;
    g.woff = off;
    off += nb * 9 * 2 + nb * 3;
    const size_t lds = (size_t)off * sizeof(float);
    MSPL_REQUIRE(lds <= 64 * 1024, MSPL_ERR_UNSUPPORTED, "pyrpool_fused: tile needs %zu B of LDS", lds);
    const Epi e = make_epi(ep, P, h * w);
    const int64_t blocks = (int64_t)N * P * g.tiles_y * g.tiles_x;
    MSPL_REQUIRE(blocks < (1ll << 31), MSPL_ERR_BAD_SHAPE, "pyrpool_fused: grid too large");
    hipLaunchKernelGGL(pyrpool_fused_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, x, g, e, out);
    MSPL_CHECK_LAUNCH("pyrpool_fused");
    return MSPL_OK;
}
