// Low-resolution pyramid branches of EfficientPyrPool, prepared in one launch.
//
// Reference arithmetic (nn_layers/efficient_pyramid_pool.py:44-50), per branch with scale < 1 and per projected channel c:
//   E = dw3x3_c( adaptive_avg_pool2d(x_c, (hs, ws)) )       -- the fused K6 kernel then interpolates E back up.
// Branch by branch this is two launches per branch (pool, conv), each of which re-reads the full-resolution x from HBM
// (the pools) or runs on a few hundred pixels (the convs): ~90 us of mostly latency at the 144x240 decoder stage.
// Here one workgroup owns one (image, channel) plane: it pools every branch from the plane (which stays L2/L1 hot after
// the first branch), keeps the pooled maps in LDS, runs the depthwise 3x3 from LDS and writes the tiny E maps.
// Pooling windows follow ATen exactly (start = floor(o*I/O), end = ceil((o+1)*I/O)); windows of <= 16 pixels are summed
// by one thread, larger ones by a whole wave (shuffle reduction).
#include <stdlib.h>

#include "common.hpp"

namespace mspl {

constexpr int PP_MAXB = 4;

struct PrepGeom {
    int N, P, h, w, nb, S;                        // S: row bands per plane (one workgroup per (plane, band))
    int hs[PP_MAXB], ws[PP_MAXB], off[PP_MAXB];   // pooled map sizes and LDS offsets (floats) of a band (+2 halo rows)
    const float* wts[PP_MAXB];                    // (P,1,3,3)
    float* out[PP_MAXB];                          // (N,P,hs,ws)
    int xoff, WS;                                 // staged input rows: LDS offset, row stride (floats, multiple of 4)
    int boff;                                     // window-bound tables
    int csoff;                                    // column sums of the large-window branches: [band rows + 2][WS]
    int stop;                                     // tuning aid (MSPL_PREP_STOP): return after phase k
    unsigned xcd_per, total;                      // XCD-contiguous order (common.hpp): the bands of a plane overlap by their halo rows
};

__device__ __host__ __forceinline__ int pp_s(int o, int I, int O) { return (int)(((unsigned)o * (unsigned)I) / (unsigned)O); }
__device__ __host__ __forceinline__ int pp_e(int o, int I, int O) { return (int)((((unsigned)(o + 1)) * (unsigned)I + O - 1) / (unsigned)O); }

// Input rows [ylo, yhi) that band `band` of the S bands needs for all its branches (pooled rows ra-1 .. rb of each).
__device__ __host__ __forceinline__ void pp_band_rows(const PrepGeom& g, int band, int& ylo, int& yhi) {
    ylo = g.h;  yhi = 0;
    for (int i = 0; i < g.nb; ++i) {
        const int hs = g.hs[i];
        const int ra = band * hs / g.S, rb = (band + 1) * hs / g.S;
        if (rb <= ra) continue;
        const int a = ra - 1 < 0 ? 0 : ra - 1, b = rb + 1 > hs ? hs : rb + 1;      // pooled rows [a, b)
        const int lo = pp_s(a, g.h, hs), hi = pp_e(b - 1, g.h, hs);
        ylo = lo < ylo ? lo : ylo;  yhi = hi > yhi ? hi : yhi;
    }
}

__global__ __launch_bounds__(256) void pyr_down_prep_kernel(const float* __restrict__ x, PrepGeom g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const unsigned bid = xcd_contiguous(blockIdx.x, g.xcd_per);
    if (bid >= g.total) return;
    const int band = bid % g.S;
    const int plane = bid / g.S;                  // n * P + c
    const int c = plane % g.P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xp = x + (size_t)plane * g.h * g.w;
    float* X = smem + g.xoff;
    int ylo, yhi;
    pp_band_rows(g, band, ylo, yhi);
    if (yhi <= ylo) return;                       // empty band (uniform)

    // Everything below is division free: this kernel was instruction bound on integer divides (window bounds, index
    // splits) before the bounds moved into small LDS tables and the loops became 2-D.
    // phase 0: the band's input rows -> LDS; a wave takes whole rows, up to 8 rows' loads in flight per lane
    {
        const int WS = g.WS, nrows = yhi - ylo;
        if ((g.w & 3) == 0) {
            const int nv = g.w >> 2;
            constexpr int UL = 8;
            for (int q0 = 0; q0 < nv; q0 += 64) {
                const int q = q0 + lane;
                for (int rb0 = wave; rb0 < nrows; rb0 += 4 * UL) {
                    float4 v[UL];
#pragma unroll
                    for (int u = 0; u < UL; ++u) {
                        const int r = rb0 + 4 * u;
                        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (r < nrows && q < nv) v[u] = *reinterpret_cast<const float4*>(xp + (size_t)(ylo + r) * g.w + 4 * q);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < UL; ++u) {
                        const int r = rb0 + 4 * u;
                        if (r < nrows && q < nv) *reinterpret_cast<float4*>(X + r * WS + 4 * q) = v[u];
                    }
                }
            }
        } else {
            for (int r = wave; r < nrows; r += 4)
                for (int q = lane; q < g.w; q += 64) X[r * WS + q] = xp[(size_t)(ylo + r) * g.w + q];
        }
    }
    if (g.stop == 1) return;
    // window bounds: per branch [ws] x-bounds and [band rows + 2] y-bounds (relative to ylo), packed lo | hi << 16
    int* XB = reinterpret_cast<int*>(smem + g.boff);
    {
        int base = 0;
#pragma unroll
        for (int i = 0; i < PP_MAXB; ++i) {
            if (i >= g.nb) break;
            const int hs = g.hs[i], ws = g.ws[i];
            const int ra = band * hs / g.S, rb = (band + 1) * hs / g.S;
            const int nrow = rb > ra ? rb - ra + 2 : 0;
            for (int t = tid; t < ws + nrow; t += 256) {
                int lo, hi;
                if (t < ws) { lo = pp_s(t, g.w, ws); hi = pp_e(t, g.w, ws); }
                else {
                    const int oy = ra - 1 + (t - ws);
                    if (oy >= 0 && oy < hs) { lo = pp_s(oy, g.h, hs) - ylo; hi = pp_e(oy, g.h, hs) - ylo; } else { lo = 0; hi = 0; }
                }
                XB[base + t] = lo | (hi << 16);
            }
            base += ws + nrow;
        }
    }
    __syncthreads();

    if (g.stop == 2) return;
    // phase 1: pooled rows [ra-1, rb+1) of every branch from LDS (rows outside the map are zero = the conv's padding)
    {
        int base = 0;
#pragma unroll
        for (int i = 0; i < PP_MAXB; ++i) {
            if (i >= g.nb) break;
            float* Pm = smem + g.off[i];
            const int hs = g.hs[i], ws = g.ws[i];
            const int ra = band * hs / g.S, rb = (band + 1) * hs / g.S;
            const int nrow = rb > ra ? rb - ra + 2 : 0;
            const int* xb = XB + base;
            const int* yb = xb + ws;
            base += ws + nrow;
            if (nrow == 0) continue;
            const int wy = (g.h + hs - 1) / hs + 1, wx = (g.w + ws - 1) / ws + 1;      // uniform
            if (wy * wx <= 16) {                  // small windows: one thread per output, a wave per pooled row
                for (int ry = wave; ry < nrow; ry += 4) {
                    const int yv = yb[ry], y0 = yv & 0xffff, y1 = yv >> 16;
                    for (int ox = lane; ox < ws; ox += 64) {
                        const int xv = xb[ox], x0 = xv & 0xffff, x1 = xv >> 16;
                        float s2 = 0.f;
                        for (int yy = y0; yy < y1; ++yy)
                            for (int xx = x0; xx < x1; ++xx) s2 += X[yy * g.WS + xx];
                        Pm[ry * ws + ox] = y1 > y0 ? s2 / (float)((y1 - y0) * (x1 - x0)) : 0.f;
                    }
                }
            } else {
                // large windows, separable through LDS (no wave reductions, whose ds_bpermute chains made this phase
                // latency bound): column sums over each pooled row's input rows, then sums over the x window
                float* CS = smem + g.csoff;                       // [nrow][WS]
                __syncthreads();                                  // CS may still be read by a previous branch
                for (int ry = wave; ry < nrow; ry += 4) {
                    const int yv = yb[ry], y0 = yv & 0xffff, y1 = yv >> 16;
                    for (int xx = lane; xx < g.w; xx += 64) {
                        float a0 = 0.f, a1 = 0.f;
                        int yy = y0;
                        for (; yy + 1 < y1; yy += 2) { a0 += X[yy * g.WS + xx]; a1 += X[(yy + 1) * g.WS + xx]; }
                        if (yy < y1) a0 += X[yy * g.WS + xx];
                        CS[ry * g.WS + xx] = a0 + a1;
                    }
                }
                __syncthreads();
                for (int ry = wave; ry < nrow; ry += 4) {
                    const int yv = yb[ry], y0 = yv & 0xffff, y1 = yv >> 16;
                    for (int ox = lane; ox < ws; ox += 64) {
                        const int xv = xb[ox], x0 = xv & 0xffff, x1 = xv >> 16;
                        float a0 = 0.f, a1 = 0.f;
                        int xx = x0;
                        for (; xx + 1 < x1; xx += 2) { a0 += CS[ry * g.WS + xx]; a1 += CS[ry * g.WS + xx + 1]; }
                        if (xx < x1) a0 += CS[ry * g.WS + xx];
                        Pm[ry * ws + ox] = y1 > y0 ? (a0 + a1) / (float)((y1 - y0) * (x1 - x0)) : 0.f;
                    }
                }
            }
        }
    }
    __syncthreads();
    if (g.stop == 3) return;
    // phase 2: depthwise 3x3 on the band from LDS; a wave per output row
#pragma unroll
    for (int i = 0; i < PP_MAXB; ++i) {
        if (i >= g.nb) break;
        const float* Pm = smem + g.off[i];
        const int hs = g.hs[i], ws = g.ws[i];
        const int ra = band * hs / g.S, rb = (band + 1) * hs / g.S;
        if (rb <= ra) continue;
        const float* w9 = g.wts[i] + (size_t)c * 9;
        const float w00 = w9[0], w01 = w9[1], w02 = w9[2], w10 = w9[3], w11 = w9[4], w12 = w9[5], w20 = w9[6], w21 = w9[7], w22 = w9[8];
        float* dst = g.out[i] + ((size_t)plane * hs + ra) * ws;
        for (int ry = wave; ry < rb - ra; ry += 4) {              // LDS row ry+1 is map row ra+ry
            const float* r0p = Pm + ry * ws;
            for (int ox = lane; ox < ws; ox += 64) {
                auto at = [&](int dy, int xx) { return (xx >= 0 && xx < ws) ? r0p[dy * ws + xx] : 0.f; };
                float v = w00 * at(0, ox - 1);
                v = fmaf(w01, at(0, ox), v);      v = fmaf(w02, at(0, ox + 1), v);
                v = fmaf(w10, at(1, ox - 1), v);  v = fmaf(w11, at(1, ox), v);  v = fmaf(w12, at(1, ox + 1), v);
                v = fmaf(w20, at(2, ox - 1), v);  v = fmaf(w21, at(2, ox), v);  v = fmaf(w22, at(2, ox + 1), v);
                dst[ry * ws + ox] = v;
            }
        }
    }
}

}  // namespace mspl

using namespace mspl;

// Chooses the band count and the LDS layout; returns the dynamic LDS bytes, or 0 when no band count fits.
static size_t prep_plan(PrepGeom& g, int64_t planes) {
    g.WS = (g.w + 3) & ~3;
    // two bands per plane measured best for the 18x30 .. 72x120 stages (more bands = more halo rows and more tiny
    // workgroups; the chain of phases, not the grid size, sets the time) ...
    int S = planes >= 128 ? 2 : (int)ceil_div64(256, planes);
    if (S < 1) S = 1;
    if (S > 16) S = 16;
    static const int dbg_s = getenv("MSPL_PREP_S") ? atoi(getenv("MSPL_PREP_S")) : 0;
    if (dbg_s > 0) S = dbg_s;
    for (;; S *= 2) {                             // ... and few enough input rows per band to fit LDS
        g.S = S;
        int off = 0;
        for (int i = 0; i < g.nb; ++i) {
            g.off[i] = off;
            off += ((ceil_div(g.hs[i], S) + 3) * g.ws[i] + 3) & ~3;     // widest band + 2 halo rows
        }
        int rmax = 0;
        for (int b = 0; b < S; ++b) { int lo, hi; pp_band_rows(g, b, lo, hi); if (hi - lo > rmax) rmax = hi - lo; }
        g.xoff = off;
        off += rmax * g.WS;
        g.boff = off;
        for (int i = 0; i < g.nb; ++i) off += g.ws[i] + ceil_div(g.hs[i], S) + 3;
        off = (off + 3) & ~3;
        g.csoff = off;
        int csrows = 0;
        for (int i = 0; i < g.nb; ++i)
            if ((ceil_div(g.h, g.hs[i]) + 1) * (ceil_div(g.w, g.ws[i]) + 1) > 16 && ceil_div(g.hs[i], S) + 3 > csrows) csrows = ceil_div(g.hs[i], S) + 3;
        off += csrows * g.WS;
        const size_t lds = (size_t)off * sizeof(float);
        if (lds <= 56 * 1024) return lds;
        if (S >= 64) return lds <= 64 * 1024 ? lds : 0;
    }
}

extern "C" int64_t mspl_pyr_down_prep_lds_bytes(int32_t N, int32_t P, int32_t h, int32_t w, int32_t nb, const int32_t* hs,
                                                const int32_t* ws) {
    if (!hs || !ws || nb < 1 || nb > PP_MAXB || N < 1 || P < 1 || h < 1 || w < 1) return 0;
    PrepGeom g;
    memset(&g, 0, sizeof(g));
    g.N = N; g.P = P; g.h = h; g.w = w; g.nb = nb;
    for (int i = 0; i < nb; ++i) {
        if (hs[i] < 1 || ws[i] < 1 || hs[i] > h || ws[i] > w) return 0;
        g.hs[i] = hs[i]; g.ws[i] = ws[i];
    }
    return (int64_t)prep_plan(g, (int64_t)N * P);
}

extern "C" int mspl_pyr_down_prep_fwd(const float* x, int32_t N, int32_t P, int32_t h, int32_t w, int32_t nb, const int32_t* hs,
                                      const int32_t* ws, const float* const* stage_w, float* const* out, void* stream) {
    MSPL_REQUIRE(x && hs && ws && stage_w && out, MSPL_ERR_NULL_POINTER, "pyr_down_prep: null pointer");
    MSPL_REQUIRE(N > 0 && P > 0 && h > 0 && w > 0, MSPL_ERR_BAD_SHAPE, "pyr_down_prep: bad shape N=%d P=%d %dx%d", N, P, h, w);
    MSPL_REQUIRE(nb >= 1 && nb <= PP_MAXB, MSPL_ERR_UNSUPPORTED, "pyr_down_prep: %d branches (1..%d)", nb, PP_MAXB);
    PrepGeom g;
    memset(&g, 0, sizeof(g));
    g.N = N; g.P = P; g.h = h; g.w = w; g.nb = nb;
    const int64_t planes = (int64_t)N * P;
    for (int i = 0; i < nb; ++i) {
        MSPL_REQUIRE(hs[i] > 0 && ws[i] > 0 && hs[i] <= h && ws[i] <= w, MSPL_ERR_BAD_SHAPE,
                     "pyr_down_prep: branch %d size %dx%d for a %dx%d map", i, hs[i], ws[i], h, w);
        MSPL_REQUIRE(stage_w[i] && out[i], MSPL_ERR_NULL_POINTER, "pyr_down_prep: branch %d has a null pointer", i);
        g.hs[i] = hs[i]; g.ws[i] = ws[i]; g.wts[i] = stage_w[i]; g.out[i] = out[i];
    }
    const size_t lds = prep_plan(g, planes);
    static const int dbg_stop = getenv("MSPL_PREP_STOP") ? atoi(getenv("MSPL_PREP_STOP")) : 0;
    g.stop = dbg_stop;
    MSPL_REQUIRE(lds > 0, MSPL_ERR_UNSUPPORTED, "pyr_down_prep: no row band of a %dx%d map fits LDS (see mspl_pyr_down_prep_lds_bytes)", h, w);
    const int64_t blocks = planes * g.S;
    MSPL_REQUIRE(blocks < (1ll << 31), MSPL_ERR_BAD_SHAPE, "pyr_down_prep: grid too large");
    g.total = (unsigned)blocks; g.xcd_per = xcd_per(blocks);
    hipLaunchKernelGGL(pyr_down_prep_kernel, dim3(8u * g.xcd_per), dim3(256), lds, (hipStream_t)stream, x, g);
    MSPL_CHECK_LAUNCH("pyr_down_prep");
    return MSPL_OK;
}
