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

#include <algorithm>

#include "common.hpp"

namespace mspl {

constexpr int PP_MAXB = 4;

struct PrepGeom {
    int N, P, h, w, nb, S;                        // S: row bands per plane (one workgroup per (plane, band))
    int hs[PP_MAXB], ws[PP_MAXB], off[PP_MAXB];   // pooled map sizes and LDS offsets (floats) of a band (+2 halo rows)
    const float* wts[PP_MAXB];                    // (P,1,3,3)
    float* out[PP_MAXB];                          // (N,P,hs,ws)
    float* pool[PP_MAXB];                         // (N,P,hs,ws) or null: the pooled maps themselves (the training forward keeps them)
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
        float* pdst = g.pool[i] ? g.pool[i] + ((size_t)plane * hs + ra) * ws : nullptr;
        for (int ry = wave; ry < rb - ra; ry += 4) {              // LDS row ry+1 is map row ra+ry
            const float* r0p = Pm + ry * ws;
            for (int ox = lane; ox < ws; ox += 64) {
                if (pdst) pdst[ry * ws + ox] = r0p[ws + ox];
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


// ------------------------------------------------------------------------------------------------------------------
// Streaming form (round 2).  The band form above moves the full-resolution map global -> registers -> LDS -> registers and runs
// load / pool / convolve as phases separated by barriers in workgroups that all start together: 37 us for the 35 MB of the
// 144x240 stage (floor of moving those bytes: ~8 us).  Here one workgroup owns one (image, channel) plane and the map never
// goes through LDS: a group of lanes owns a pooled row, reads the input rows of its window straight from global memory (all of
// them in flight at once, 8- or 16-byte coalesced loads) and reduces them in registers.
//   * exact 2x2 branches (h = 2 hs, w = 2 ws: the 0.5 scale): a lane's 2 or 4 columns of two rows give 1 or 2 pooled values.
//   * large windows (the 0.1 scale, ~10 x 10): column sums of the window's rows in registers, written to a row buffer of the
//     wave in LDS (w floats), then lanes ox < ws add their x window.  Same two-accumulator order as the band form.
// Only the pooled maps (a quarter of the plane + a few hundred floats) are kept in LDS for the depthwise 3x3 that follows the
// one barrier.  Rows of the second branch are re-read from L2.
constexpr int PS_MAXWIN = 12;

struct PsGeom {
    int N, P, h, w, nb;
    int hs[PP_MAXB], ws[PP_MAXB], off[PP_MAXB], exact2[PP_MAXB];
    const float* wts[PP_MAXB];
    float* out[PP_MAXB];
    float* pool[PP_MAXB];      // pooled maps (training forward) or null
    int LPR, lprp_shift;       // lanes per input row (w / VW) and log2 of its power-of-two padding
    int rboff, RBS;            // row buffers: LDS offset, floats per buffer
};

template <int VW>
__global__ __launch_bounds__(1024) void pyr_prep_stream_kernel(const float* __restrict__ x, PsGeom g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int plane = blockIdx.x, c = plane % g.P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lprp = 1 << g.lprp_shift, rpw = 64 >> g.lprp_shift;        // sub-row groups of a wave
    const int sr = lane >> g.lprp_shift, cl = lane & (lprp - 1);
    const bool colok = cl < g.LPR;
    const int nthr = blockDim.x, nwaves = nthr >> 6;                    // 4 .. 16 waves: enough row loads in flight per CU
    const int gid = wave * rpw + sr, ngroups = nwaves * rpw;
    const float* xp = x + (size_t)plane * g.h * g.w;
    const int w = g.w;
    typedef float vecw __attribute__((ext_vector_type(VW)));
    auto load_row = [&](int y) {
        return *reinterpret_cast<const vecw*>(xp + (size_t)y * w + cl * VW);
    };
    // The pooled maps sit ZERO-HALOED in LDS (row stride ws + 2, a zero row above and below): the depthwise 3x3 below then reads its
    // nine taps unconditionally (four compares + selects per tap before: 55 of the ~80 vector instructions per output).
    for (int i = 0; i < g.nb; ++i) {
        float* Pz = smem + g.off[i];
        const int hs = g.hs[i], WP = g.ws[i] + 2;
        for (int t = tid; t < 2 * WP + 2 * hs; t += nthr) {
            int at;
            if (t < WP) at = t;                                             // top row
            else if (t < 2 * WP) at = (hs + 1) * WP + (t - WP);             // bottom row
            else { const int r = (t - 2 * WP) >> 1; at = (r + 1) * WP + ((t & 1) ? WP - 1 : 0); }      // left / right column
            Pz[at] = 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < PP_MAXB; ++i) {
        if (i >= g.nb) break;
        const int hs = g.hs[i], ws = g.ws[i];
        const int WP = ws + 2;
        float* Pm = smem + g.off[i] + WP + 1;              // (0, 0) of the map inside its halo
        if (g.exact2[i]) {
            constexpr int U = 4;                           // pooled rows per group and trip: 8 row loads in flight per lane
            for (int o0 = 0; o0 < hs; o0 += ngroups * U) {
                vecw r0[U], r1[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int o = o0 + u * ngroups + gid;
                    if (o < hs && colok) { r0[u] = load_row(2 * o); r1[u] = load_row(2 * o + 1); }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int o = o0 + u * ngroups + gid;
                    if (o < hs && colok) {
#pragma unroll
                        for (int j = 0; j < VW / 2; ++j) {
                            float s2 = 0.f;                  // (ATen's order inside the window: row-major)
                            s2 += r0[u][2 * j]; s2 += r0[u][2 * j + 1]; s2 += r1[u][2 * j]; s2 += r1[u][2 * j + 1];
                            Pm[o * WP + cl * (VW / 2) + j] = s2 / 4.0f;
                        }
                    }
                }
            }
        } else {
            float* RB = smem + g.rboff + (wave * rpw + sr) * g.RBS;
            for (int ob = 0; ob < hs; ob += ngroups) {     // uniform trip count
                const int o = ob + gid;
                const bool ok = o < hs;
                const int oc = ok ? o : hs - 1;
                const int y0 = pp_s(oc, g.h, hs), y1 = pp_e(oc, g.h, hs);
                vecw r[PS_MAXWIN];
#pragma unroll
                for (int t = 0; t < PS_MAXWIN; ++t) {
#pragma unroll
                    for (int j = 0; j < VW; ++j) r[t][j] = 0.f;
                    if (colok && y0 + t < y1) r[t] = load_row(y0 + t);
                }
                float a0[VW], a1[VW];
#pragma unroll
                for (int j = 0; j < VW; ++j) { a0[j] = 0.f; a1[j] = 0.f; }
#pragma unroll
                for (int t = 0; t < PS_MAXWIN; t += 2) {
                    if (y0 + t + 1 < y1) {
#pragma unroll
                        for (int j = 0; j < VW; ++j) { a0[j] += r[t][j]; a1[j] += r[t + 1][j]; }
                    } else if (y0 + t < y1) {
#pragma unroll
                        for (int j = 0; j < VW; ++j) a0[j] += r[t][j];
                    }
                }
                if (colok) {
#pragma unroll
                    for (int j = 0; j < VW; ++j) RB[cl * VW + j] = a0[j] + a1[j];
                }
                // (same wave: its LDS operations execute in order, the reads below see the writes above)
                for (int ox = cl; ox < ws; ox += lprp) {
                    const int x0 = pp_s(ox, w, ws), x1 = pp_e(ox, w, ws);
                    float b0 = 0.f, b1 = 0.f;
                    int xx = x0;
                    for (; xx + 1 < x1; xx += 2) { b0 += RB[xx]; b1 += RB[xx + 1]; }
                    if (xx < x1) b0 += RB[xx];
                    if (ok) Pm[o * WP + ox] = (b0 + b1) / (float)((y1 - y0) * (x1 - x0));
                }
            }
        }
    }
    __syncthreads();
    // depthwise 3x3 (zero padding = the halo) on the pooled maps
#pragma unroll
    for (int i = 0; i < PP_MAXB; ++i) {
        if (i >= g.nb) break;
        const int hs = g.hs[i], ws = g.ws[i];
        const int WP = ws + 2;
        const float* Pm = smem + g.off[i];                 // haloed: map element (y, x) at (y + 1) * WP + x + 1
        const float* w9 = g.wts[i] + (size_t)c * 9;
        const float w00 = w9[0], w01 = w9[1], w02 = w9[2], w10 = w9[3], w11 = w9[4], w12 = w9[5], w20 = w9[6], w21 = w9[7], w22 = w9[8];
        float* dst = g.out[i] + (size_t)plane * hs * ws;
        const int total = hs * ws;
        int oy = tid / ws, ox = tid - oy * ws;             // one division; then the index advances by the block size per trip
        const int dy256 = nthr / ws, dx256 = nthr - dy256 * ws;
        for (int idx = tid; idx < total; idx += nthr) {
            const float* q = Pm + oy * WP + ox;            // q[dy * WP + dx] = map(oy - 1 + dy, ox - 1 + dx)
            float v = w00 * q[0];
            v = fmaf(w01, q[1], v);  v = fmaf(w02, q[2], v);
            v = fmaf(w10, q[WP], v);  v = fmaf(w11, q[WP + 1], v);  v = fmaf(w12, q[WP + 2], v);
            v = fmaf(w20, q[2 * WP], v);  v = fmaf(w21, q[2 * WP + 1], v);  v = fmaf(w22, q[2 * WP + 2], v);
            dst[idx] = v;
            if (g.pool[i]) g.pool[i][(size_t)plane * total + idx] = q[WP + 1];
            oy += dy256;  ox += dx256;
            if (ox >= ws) { ox -= ws; ++oy; }
        }
    }
}

// Returns MSPL_OK when launched, 1 when the shape is left to the band form.
static int prep_stream_try(const float* x, int N, int P, int h, int w, int nb, const int32_t* hs, const int32_t* ws,
                           const float* const* stage_w, float* const* out, float* const* pooled, hipStream_t stream) {
    static const int off = (MSPL_TUNE_INT("MSPL_PREP_STREAM", 1) == 0);
    if (off || (w & 1) || (((uintptr_t)x) & 15)) return 1;
    const int VW = (w & 3) == 0 ? 4 : 2;
    PsGeom g;
    memset(&g, 0, sizeof(g));
    g.N = N; g.P = P; g.h = h; g.w = w; g.nb = nb;
    g.LPR = w / VW;
    if (g.LPR > 64) return 1;
    g.lprp_shift = 0;
    while ((1 << g.lprp_shift) < g.LPR) ++g.lprp_shift;
    int off_f = 0;
    bool any_large = false;
    for (int i = 0; i < nb; ++i) {
        g.hs[i] = hs[i]; g.ws[i] = ws[i]; g.wts[i] = stage_w[i]; g.out[i] = out[i]; g.pool[i] = pooled ? pooled[i] : nullptr;
        g.exact2[i] = (h == 2 * hs[i] && w == 2 * ws[i]) ? 1 : 0;
        if (!g.exact2[i]) {
            const int wy = ceil_div(h, hs[i]) + 1, wx = ceil_div(w, ws[i]) + 1;
            if (wy * wx <= 16 || wy > PS_MAXWIN) return 1;         // small ragged windows / very tall windows: band form
            any_large = true;
        }
        g.off[i] = off_f;
        off_f += ((hs[i] + 2) * (ws[i] + 2) + 3) & ~3;            // zero-haloed
    }
    g.rboff = off_f;
    g.RBS = (w + 3) & ~3;
    // waves per workgroup: one trip of the 2x2 branch should cover the map's rows (all of a plane's row loads in flight at once:
    // one workgroup per CU is all a launch of N * P = 256 planes gives, so the loads of a CU come from this workgroup alone)
    static const int dbg_waves = MSPL_TUNE_INT("MSPL_PREP_WAVES", 0);
    int hmax = 1;
    for (int i = 0; i < nb; ++i) hmax = std::max(hmax, (int)hs[i]);
    int waves = 4;
    while (waves < 16 && waves * (64 >> g.lprp_shift) * 4 < hmax) waves *= 2;
    if (dbg_waves >= 1 && dbg_waves <= 16) waves = dbg_waves;
    if (any_large) off_f += waves * (64 >> g.lprp_shift) * g.RBS;
    const size_t lds = (size_t)off_f * sizeof(float);
    if (lds > 64 * 1024) return 1;
    const int64_t planes = (int64_t)N * P;
    if (planes >= (1ll << 31)) return 1;
    if (VW == 4) hipLaunchKernelGGL(pyr_prep_stream_kernel<4>, dim3((unsigned)planes), dim3(64 * waves), lds, stream, x, g);
    else hipLaunchKernelGGL(pyr_prep_stream_kernel<2>, dim3((unsigned)planes), dim3(64 * waves), lds, stream, x, g);
    MSPL_CHECK_LAUNCH("pyr_down_prep(streaming form)");
    return MSPL_OK;
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
    static const int dbg_s = MSPL_TUNE_INT("MSPL_PREP_S", 0);
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

extern "C" int mspl_pyr_down_prep_train_fwd(const float* x, int32_t N, int32_t P, int32_t h, int32_t w, int32_t nb, const int32_t* hs,
                                            const int32_t* ws, const float* const* stage_w, float* const* out, float* const* pooled,
                                            void* stream);

extern "C" int mspl_pyr_down_prep_fwd(const float* x, int32_t N, int32_t P, int32_t h, int32_t w, int32_t nb, const int32_t* hs,
                                      const int32_t* ws, const float* const* stage_w, float* const* out, void* stream) {
    return mspl_pyr_down_prep_train_fwd(x, N, P, h, w, nb, hs, ws, stage_w, out, nullptr, stream);
}

// The training forward: the same launch also writes the pooled maps p_i = adaptive_avg_pool2d(x, (hs_i, ws_i)) (pooled[i], or
// pooled == NULL), which the depthwise convolutions' weight gradient needs.
extern "C" int mspl_pyr_down_prep_train_fwd(const float* x, int32_t N, int32_t P, int32_t h, int32_t w, int32_t nb, const int32_t* hs,
                                            const int32_t* ws, const float* const* stage_w, float* const* out, float* const* pooled,
                                            void* stream) {
    MSPL_REQUIRE(x && hs && ws && stage_w && out, MSPL_ERR_NULL_POINTER, "pyr_down_prep: null pointer");
    MSPL_REQUIRE(N > 0 && P > 0 && h > 0 && w > 0, MSPL_ERR_BAD_SHAPE, "pyr_down_prep: bad shape N=%d P=%d %dx%d", N, P, h, w);
    MSPL_REQUIRE(nb >= 1 && nb <= PP_MAXB, MSPL_ERR_UNSUPPORTED, "pyr_down_prep: %d branches (1..%d)", nb, PP_MAXB);
    for (int i = 0; i < nb; ++i) {
        MSPL_REQUIRE(hs[i] > 0 && ws[i] > 0 && hs[i] <= h && ws[i] <= w, MSPL_ERR_BAD_SHAPE,
                     "pyr_down_prep: branch %d size %dx%d for a %dx%d map", i, hs[i], ws[i], h, w);
        MSPL_REQUIRE(stage_w[i] && out[i], MSPL_ERR_NULL_POINTER, "pyr_down_prep: branch %d has a null pointer", i);
    }
    {
        const int rc = mspl::prep_stream_try(x, N, P, h, w, nb, hs, ws, stage_w, out, pooled, (hipStream_t)stream);
        if (rc <= 0) return rc;
    }
    PrepGeom g;
    memset(&g, 0, sizeof(g));
    g.N = N; g.P = P; g.h = h; g.w = w; g.nb = nb;
    const int64_t planes = (int64_t)N * P;
    for (int i = 0; i < nb; ++i) {
        MSPL_REQUIRE(hs[i] > 0 && ws[i] > 0 && hs[i] <= h && ws[i] <= w, MSPL_ERR_BAD_SHAPE,
                     "pyr_down_prep: branch %d size %dx%d for a %dx%d map", i, hs[i], ws[i], h, w);
        MSPL_REQUIRE(stage_w[i] && out[i], MSPL_ERR_NULL_POINTER, "pyr_down_prep: branch %d has a null pointer", i);
        g.hs[i] = hs[i]; g.ws[i] = ws[i]; g.wts[i] = stage_w[i]; g.out[i] = out[i]; g.pool[i] = pooled ? pooled[i] : nullptr;
    }
    const size_t lds = prep_plan(g, planes);
    static const int dbg_stop = MSPL_TUNE_INT("MSPL_PREP_STOP", 0);
    g.stop = dbg_stop;
    MSPL_REQUIRE(lds > 0, MSPL_ERR_UNSUPPORTED, "pyr_down_prep: no row band of a %dx%d map fits LDS (see mspl_pyr_down_prep_lds_bytes)", h, w);
    const int64_t blocks = planes * g.S;
    MSPL_REQUIRE(blocks < (1ll << 31), MSPL_ERR_BAD_SHAPE, "pyr_down_prep: grid too large");
    g.total = (unsigned)blocks; g.xcd_per = xcd_per(blocks);
    hipLaunchKernelGGL(pyr_down_prep_kernel, dim3(8u * g.xcd_per), dim3(256), lds, (hipStream_t)stream, x, g);
    MSPL_CHECK_LAUNCH("pyr_down_prep");
    return MSPL_OK;
}
