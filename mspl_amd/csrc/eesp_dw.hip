// K2 -- EESP split / transform / hierarchical-feature-fusion kernel (the headline HBM-bound kernel).
//
// Reference arithmetic: nn_layers/eesp.py:68-80 -- four CDilated depthwise 3x3 convolutions
// (espnet_utils.py:118-142, padding = dilation) of the SAME reduced tensor, out_k += out_{k-1},
// torch.cat over branches, then br_after_cat (BatchNorm + PReLU, espnet_utils.py:39-60).
//
// MI355X design (round 2): one pass over HBM by PERSISTENT workgroups.  A tile is a band of output rows of CP (image,
// channel) planes; a workgroup walks tiles t = b, b + G, ... of an XCD-contiguous tile order.  Per tile: the input rows
// (+ MAXD halo rows) arrive in registers (16-byte coalesced loads issued one tile ahead, so they fly during the previous
// tile's arithmetic and stores), are written to LDS (zero halo columns are written once per workgroup), and every thread
// produces 1x4 output strips for all four branches from register-resident row windows:
//   * stride 1: a row is staged as 4 zero columns + W values; a strip's window is 12 floats = 3 aligned ds_read_b128.
//   * stride 2: a row is staged DE-INTERLEAVED, E[j] = x[2j] and O[j] = x[2j+1]; output xo reads input column 2xo + dx, i.e.
//     E[xo + dx/2] or O[xo + (dx-1)/2]: four consecutive outputs again read a contiguous 12-float window, lanes stay 16 bytes
//     apart (conflict-free ds_read_b128) and write float4 (the round-1 form produced 2 outputs per lane from a 12-float
//     window: 2.3x the LDS reads per output and 8-byte stores).
//   * the centre row is read once for all four dilations; a row +-d is read once per distinct dilation; only the float4s
//     of a window that a dilation touches are read.
//   * weights and epilogue constants sit in LDS as one float4 per (branch, kernel row) / per branch.
// The hierarchical sum out_k = conv_k + out_{k-1} is carried in registers, the folded BN + PReLU applied, and the four
// concatenated planes written with 16-byte stores (8-byte when the row length is only even: the 18x30 level).
// Algorithmic bytes: 4*n*(H*W + 4*Ho*Wo) per image.
//
// What the round-1 timeline (s_memrealtime stamps, profiles/r02_k2_stamps_before.txt) showed and this form answers: at the
// 18x30 level the 2048 one-shot workgroups of a launch started over 4-7 us, loaded for 2.7 us, then all computed at once for
// 3 us with 56 % of the lanes idle (144 items on 256 threads) and ~450 vector instructions per item -- the vector pipe, not
// HBM, was the bound of that phase; at 144x240 (stride 2) a band was 5 output rows (70 % halo rows re-read) in 2.8 rounds.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include <mutex>

#include "common.hpp"

namespace mspl {

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // Give each XCD (ids b, b+8, ... share one) a contiguous chunk of the logical order so that neighbouring row bands
    // (which share halo rows) hit the same L2.  Bijective for any nwg.
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

template <int D0, int D1, int D2, int D3>
struct DilSet {
    static constexpr int d(int k) { return k == 0 ? D0 : k == 1 ? D1 : k == 2 ? D2 : D3; }
    static constexpr int maxd() { return D3 > D2 ? (D3 > D1 ? (D3 > D0 ? D3 : D0) : (D1 > D0 ? D1 : D0))
                                                 : (D2 > D1 ? (D2 > D0 ? D2 : D0) : (D1 > D0 ? D1 : D0)); }
    static constexpr bool any_odd() { return ((D0 | D1 | D2 | D3) & 1) != 0; }
};

struct DwGeom {
    int N, n, H, W, Ho, Wo;
    int TH;       // output rows per band
    int CP;       // planes per tile (same image, consecutive channels)
    int bands;    // ceil(Ho / TH)
    int cgroups;  // n / CP
    int ntiles;   // N * cgroups * bands
    int RS;       // LDS floats per staged input row (stride 2: E part then O part, RS / 2 each)
    int RIN;      // staged input rows per plane
    int XS;       // output strips (of 4) per output row
    int NCH;      // 16-byte chunks per input row: ceil(W / 4)
    int chunks;   // CP * RIN * NCH  (<= 256 * DW_PF)
    unsigned mag_xs, mag_nch, mag_rin;   // exact divisions by XS / NCH / RIN: (t * mag) >> 20 (ranges verified on the host)
    int wt;       // 1: write-through (sc1) stores -- the outputs leave L2 while the kernel runs instead of at its end
    unsigned long long* stamps;  // tuning aid (MSPL_DW_STAMP, STAMPS=1 builds): 4 s_memrealtime stamps per workgroup, or null
};

typedef float f4a8 __attribute__((ext_vector_type(4), aligned(8)));     // a float4 that is only known to be 8-byte aligned
typedef float f4a4 __attribute__((ext_vector_type(4), aligned(4)));

constexpr int DW_PF = 8;        // float4 prefetch registers per thread: a tile is at most 256 * 8 chunks = 32 KB of input

// 12-float window starting at p (16-byte aligned); only the float4s that hold indices LO..HI are read.
template <int LO, int HI>
__device__ __forceinline__ void dw_window(const float* __restrict__ p, float (&v)[12]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (4 * i + 3 >= LO && 4 * i <= HI) {
            const float4 q = *reinterpret_cast<const float4*>(p + 4 * i);
            // all four elements count as used: keeps this one ds_read_b128 (hipcc otherwise narrows the read to the
            // elements this dilation touches and emits more, smaller reads)
            asm volatile("" :: "v"(q.x), "v"(q.y), "v"(q.z), "v"(q.w));
            v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
        } else {
            v[4 * i] = v[4 * i + 1] = v[4 * i + 2] = v[4 * i + 3] = 0.f;
        }
    }
}

struct DwItemCtx {
    int CP, rows_here, XS, PS, RS, Wo, hw, y0, wt;
    unsigned mag_xs;
    bool o16, o8, has_act;
    size_t kstride;
    char* ob;                       // first output plane of the tile (branch 0)
    float* out;                     // destination tensor (base of the write-through descriptor)
    __amdgpu_buffer_rsrc_t orsrc;
    char* rawb;                     // training forward: the sums BEFORE the folded BN + PReLU go here too (same layout as ob), or null
};

// The arithmetic of a tile that sits in LDS (shared by the tiled kernel and the fused projection + K2 kernel, eesp_front.hip).
template <int STRIDE, class DS>
__device__ __forceinline__ void dw_compute_items(const float* __restrict__ tile, const float* __restrict__ wl, const float* __restrict__ el,
                                                 const DwItemCtx& c, int tid, int nthr) {
    constexpr int MAXD = DS::maxd();
    constexpr bool ODD = STRIDE == 2 && DS::any_odd();
        const int rows_here = c.rows_here;
        const int items = c.CP * rows_here * c.XS;
        char* ob = c.ob;
        float* out = c.out;
        const int hw = c.hw, PS = c.PS, y0 = c.y0;
        const __amdgpu_buffer_rsrc_t orsrc = c.orsrc;
        const bool o16 = c.o16, o8 = c.o8, has_act = c.has_act;
        const size_t kstride = c.kstride;
        const int OOFF = c.RS >> 1;
        const unsigned mag_rows = ((1u << 20) + (unsigned)rows_here - 1) / (unsigned)rows_here;   // uniform
        for (int it = tid; it < items; it += nthr) {
            const int t2 = (int)(((unsigned)it * c.mag_xs) >> 20);
            const int xs = it - t2 * c.XS;
            const int p = (int)(((unsigned)t2 * mag_rows) >> 20);
            const int ty = t2 - p * rows_here;
            const float* lp = tile + (size_t)p * PS + (ty * STRIDE + MAXD) * c.RS + 4 * xs;   // centre row window (A / E array)
            const float4* wp = reinterpret_cast<const float4*>(wl) + p * 12;
            const float4* ep = reinterpret_cast<const float4*>(el) + p * 4;
            const int xb = xs * 4;
            // ONE 32-bit lane offset; the branch part of the address is uniform and goes in the scalar base
            const unsigned voff = (unsigned)((((size_t)p * hw) + (size_t)(y0 + ty) * c.Wo + xb) * sizeof(float));
            float cA[12], cO[12];
            dw_window<0, 11>(lp, cA);
            if (ODD) dw_window<2, 8>(lp + OOFF, cO);      // odd taps use indices 4 + j + {-2, -1, 0, 1}, j = 0..3
            float prev[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int d = DS::d(k);
                if (k > 0 && DS::d(k - 1) == d) continue;       // evaluated together with the first branch of its run
                int R = 1;
#pragma unroll
                for (int q = k + 1; q < 4; ++q) if (DS::d(q) == d && q == k + R) ++R;
                float a[4][4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[r][j] = 0.f;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    float rA[12], rO[12];
                    if (ky == 1) {
#pragma unroll
                        for (int i = 0; i < 12; ++i) { rA[i] = cA[i]; rO[i] = ODD ? cO[i] : 0.f; }
                    } else {
                        const float* row = lp + (ky - 1) * d * c.RS;
#pragma unroll
                        for (int i = 0; i < 12; ++i) rO[i] = 0.f;
                        if (STRIDE == 1) {
                            if (d == 1) dw_window<3, 8>(row, rA); else if (d == 2) dw_window<2, 9>(row, rA);
                            else if (d == 3) dw_window<1, 10>(row, rA); else dw_window<0, 11>(row, rA);
                        } else if (d & 1) {
                            dw_window<4, 7>(row, rA);                                   // centre tap only
                            if (d == 1) dw_window<3, 7>(row + OOFF, rO); else dw_window<2, 8>(row + OOFF, rO);
                        } else {
                            if (d == 2) dw_window<3, 8>(row, rA); else dw_window<2, 9>(row, rA);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (r >= R) break;
                        const float4 w4 = wp[(k + r) * 3 + ky];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float tm, tc, tp;                   // taps at input column offsets -d, 0, +d
                            if (STRIDE == 1) { tm = rA[4 + j - d]; tc = rA[4 + j]; tp = rA[4 + j + d]; }
                            else if (d & 1) { tm = rO[4 + j - (d + 1) / 2]; tc = rA[4 + j]; tp = rO[4 + j + (d - 1) / 2]; }
                            else { tm = rA[4 + j - d / 2]; tc = rA[4 + j]; tp = rA[4 + j + d / 2]; }
                            a[r][j] = fmaf(w4.x, tm, a[r][j]);
                            a[r][j] = fmaf(w4.y, tc, a[r][j]);
                            a[r][j] = fmaf(w4.z, tp, a[r][j]);
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (r >= R) break;
                    const int kk = k + r;
                    // hierarchical feature fusion: out_k = conv_k + out_{k-1}   (nn_layers/eesp.py:72-76)
                    const float4 ec = ep[kk];
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        a[r][j] += prev[j];
                        prev[j] = a[r][j];
                        float q = fmaf(a[r][j], ec.x, ec.y);
                        if (has_act) q = q > 0.f ? q : ec.z * q;
                        v[j] = q;
                    }
                    if (c.rawb) {                                 // uniform; what br_after_cat's backward reads (training forward)
                        float* rd = reinterpret_cast<float*>(c.rawb + kk * kstride + voff);
                        if (o16) *reinterpret_cast<float4*>(rd) = make_float4(a[r][0], a[r][1], a[r][2], a[r][3]);
                        else {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (xb + j < c.Wo) rd[j] = a[r][j];
                        }
                    }
                    float* dst = reinterpret_cast<float*>(ob + kk * kstride + voff);
                    if (o16 && c.wt) {
                        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                        const u32x4 dv = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                        const int boff = (int)(reinterpret_cast<const char*>(dst) - reinterpret_cast<const char*>(out));
                        __builtin_amdgcn_raw_buffer_store_b128(dv, orsrc, boff, 0, 16);          // aux 16 = sc1
                    } else if (o8 && c.wt) {
                        // rows of an even number of floats: the strip starts 8-byte aligned; a 16-byte buffer store needs dword
                        // alignment only.  The row's last strip may hold two pixels.
                        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                        const int boff = (int)(reinterpret_cast<const char*>(dst) - reinterpret_cast<const char*>(out));
                        if (xb + 2 < c.Wo) {
                            const u32x4 dv = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                            __builtin_amdgcn_raw_buffer_store_b128(dv, orsrc, boff, 0, 16);
                        } else {
                            const u32x2 d0 = {__float_as_uint(v[0]), __float_as_uint(v[1])};
                            __builtin_amdgcn_raw_buffer_store_b64(d0, orsrc, boff, 0, 16);
                        }
                    } else if (o16) {
                        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                    } else if (o8) {
                        *reinterpret_cast<float2*>(dst) = make_float2(v[0], v[1]);
                        if (xb + 2 < c.Wo) *reinterpret_cast<float2*>(dst + 2) = make_float2(v[2], v[3]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (xb + j < c.Wo) dst[j] = v[j];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);   // one dilation at a time (keeps the register footprint small)
            }
        }
}

// PERSIST: a workgroup walks tiles b, b + G, ... with the next tile's rows prefetched into registers (many tiles per
// workgroup, 3 workgroups per CU); otherwise one tile per workgroup (the staging registers die before the arithmetic starts:
// 4-5 workgroups per CU, which is what hides latency when a launch is a single round of small tiles).
template <int STRIDE, class DS, bool PERSIST>
__global__ __launch_bounds__(PERSIST ? 256 : 1024, PERSIST ? 3 : 4) void eesp_dw_hff_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ w,
                                                             DwGeom g, Epi e, float* __restrict__ out) {
    constexpr int MAXD = DS::maxd();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int PS = g.RIN * g.RS;                          // floats per staged plane
    float* tile = smem;                                   // CP * PS (+16 floats of tail pad)
    float* wl = smem + (size_t)g.CP * PS + 16;            // [CP][branch*3 + ky][4]: (w0, w1, w2, 0)
    float* el = wl + g.CP * 48;                           // [CP][branch][4]: (scale, shift, alpha, 0)
    const int tid = threadIdx.x, nthr = blockDim.x;        // 64..256 threads: chosen so that the tile's items fill whole rounds
    const int OOFF = g.RS >> 1;                           // stride 2: offset of the odd-column array inside a staged row
    unsigned long long st0 = 0, st1 = 0, st2 = 0;
    if (g.stamps) st0 = __builtin_amdgcn_s_memrealtime();

    // ---- once per workgroup: zero the tile (halo columns / fill stay zero for every tile; data slots are overwritten)
    {
        const int tot4 = (g.CP * PS + 16) >> 2;
        for (int i = tid; i < tot4; i += nthr) *reinterpret_cast<float4*>(tile + 4 * i) = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    // ---- per-thread staging slots: chunk i = tid + nthr*u -> (plane p, staged row r, 16-byte chunk m of the input row)
    // (fixed for the whole kernel: only the tile's base pointer and first input row change)
    // packed per slot: bit 31 = no chunk, bit 30 = partial last chunk of its row (W % 4 != 0), p << 24 | r << 12 | m
    unsigned slot[DW_PF];
#pragma unroll
    for (int u = 0; u < DW_PF; ++u) {
        const int i = tid + nthr * u;
        slot[u] = 0x80000000u;
        if (i < g.chunks) {
            const int rr = (int)(((unsigned)i * g.mag_nch) >> 20), m = i - rr * g.NCH;
            const int p = (int)(((unsigned)rr * g.mag_rin) >> 20), r = rr - p * g.RIN;
            slot[u] = ((unsigned)p << 24) | ((unsigned)r << 12) | (unsigned)m | ((4 * m + 4 > g.W) ? 0x40000000u : 0u);
        }
    }
    const int wrem = g.W & 3;                              // elements of a row's last chunk (0: all four)
    const bool al16 = wrem == 0, al8 = (g.W & 1) == 0;

    float4 pre[DW_PF];
    auto tile_coords = [&](int t, int& img, int& c0, int& y0) {
        int L = xcd_remap(t, g.ntiles);
        const int band = L % g.bands;  L /= g.bands;
        const int cg = L % g.cgroups;
        img = L / g.cgroups;  c0 = cg * g.CP;  y0 = band * g.TH;
    };
    auto issue_loads = [&](int t) {
        int img, c0, y0;
        tile_coords(t, img, c0, y0);
        const int iy0 = y0 * STRIDE - MAXD;                // input row of staged row 0 (may be negative: offsets stay in range)
        const float* xb = x + ((size_t)img * g.n + c0) * g.H * (size_t)g.W + (ptrdiff_t)iy0 * g.W;
#pragma unroll
        for (int u = 0; u < DW_PF; ++u) {
            pre[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int sp = (slot[u] >> 24) & 3, sr = (slot[u] >> 12) & 0xfff, sm = slot[u] & 0xfff;
            const int iy = iy0 + sr;
            if (!(slot[u] & 0x80000000u) && iy >= 0 && iy < g.H) {
                const float* src = xb + ((sp * g.H + sr) * g.W + 4 * sm);
                if (al16) {
                    pre[u] = *reinterpret_cast<const float4*>(src);
                } else if (!(slot[u] & 0x40000000u)) {
                    if (al8) { const f4a8 q = *reinterpret_cast<const f4a8*>(src); pre[u] = make_float4(q.x, q.y, q.z, q.w); }
                    else { const f4a4 q = *reinterpret_cast<const f4a4*>(src); pre[u] = make_float4(q.x, q.y, q.z, q.w); }
                } else {
                    pre[u].x = src[0];
                    if (wrem > 1) pre[u].y = src[1];
                    if (wrem > 2) pre[u].z = src[2];
                }
            }
        }
    };
    auto write_tile = [&]() {
#pragma unroll
        for (int u = 0; u < DW_PF; ++u) {
            if (!(slot[u] & 0x80000000u)) {
                const int sp = (slot[u] >> 24) & 3, sr = (slot[u] >> 12) & 0xfff, sm = slot[u] & 0xfff;
                float* d = tile + (sp * PS + sr * g.RS + (STRIDE == 1 ? 4 + 4 * sm : 4 + 2 * sm));
                if (STRIDE == 1) {
                    *reinterpret_cast<float4*>(d) = pre[u];
                } else {
                    *reinterpret_cast<float2*>(d) = make_float2(pre[u].x, pre[u].z);
                    *reinterpret_cast<float2*>(d + OOFF) = make_float2(pre[u].y, pre[u].w);
                }
            }
        }
    };

    int t = blockIdx.x;
    if (t < g.ntiles) issue_loads(t);
    __syncthreads();                                       // zero fill complete before the first data writes
    const int hw = g.Ho * g.Wo;
    // write-through stores go through a buffer descriptor over the whole destination tensor (uniform base, 32-bit offsets)
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((size_t)g.N * e.ctot * hw * sizeof(float)), 0x00020000);
    const bool o16 = (g.Wo & 3) == 0, o8 = (g.Wo & 1) == 0;
    const bool has_act = e.alpha != nullptr;
    const size_t kstride = (size_t)g.n * hw * sizeof(float);

    for (; t < g.ntiles; t += PERSIST ? (int)gridDim.x : g.ntiles) {
        int img, c0, y0;
        tile_coords(t, img, c0, y0);
        // ---- weights / epilogue constants of this tile's planes (tiny, L2 resident), padded to float4 groups
        float wreg = 0.f, ereg = 0.f;
        const int nwts = g.CP * 48, neps = g.CP * 16;
        if (tid < nwts) {
            const int p = tid / 48, r = tid - p * 48, kq = r >> 2, kx = r & 3, k = kq / 3, ky = kq - 3 * k;
            if (kx < 3) wreg = w[((size_t)k * g.n + (c0 + p)) * 9 + ky * 3 + kx];
        }
        if (tid < neps) {
            const int p = tid >> 4, r = tid & 15, k = r >> 2, f = r & 3;
            const int cabs = e.coff + k * g.n + c0 + p;
            const float* src = f == 0 ? e.scale : (f == 1 ? e.shift : e.alpha);
            ereg = f == 3 ? 0.f : (src ? src[cabs] : (f == 1 ? 0.f : 1.f));
        }
        write_tile();
        if (tid < nwts) wl[tid] = wreg;
        if (tid < neps) el[tid] = ereg;
        if (g.stamps && t == (int)blockIdx.x) st1 = __builtin_amdgcn_s_memrealtime();
        __syncthreads();
        if (g.stamps && t == (int)blockIdx.x) st2 = __builtin_amdgcn_s_memrealtime();
        if (PERSIST && t + (int)gridDim.x < g.ntiles) issue_loads(t + gridDim.x);     // next tile's rows fly during this tile's arithmetic

        // ---- compute: item = (plane p, band row ty, strip xs); xs fastest so that lanes are 16 bytes apart
        {
            DwItemCtx cx;
            cx.CP = g.CP; cx.rows_here = min(g.TH, g.Ho - y0); cx.XS = g.XS; cx.PS = PS; cx.RS = g.RS; cx.Wo = g.Wo; cx.hw = hw;
            cx.y0 = y0; cx.mag_xs = g.mag_xs; cx.wt = g.wt; cx.o16 = o16; cx.o8 = o8; cx.has_act = has_act; cx.kstride = kstride;
            cx.ob = reinterpret_cast<char*>(out + ((size_t)img * e.ctot + e.coff + c0) * (size_t)hw);
            cx.out = out; cx.orsrc = orsrc;
            cx.rawb = e.raw ? reinterpret_cast<char*>(e.raw + ((size_t)img * e.ctot + c0) * (size_t)hw) : nullptr;     // (ctot == 4n, coff == 0)
            dw_compute_items<STRIDE, DS>(tile, wl, el, cx, tid, nthr);
        }
        if (PERSIST) __syncthreads();                  // every wave is done reading the tile before the next one is written
    }
    if (g.stamps && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* d = g.stamps + (size_t)blockIdx.x * 4;
        d[0] = st0; d[1] = st1; d[2] = st2; d[3] = __builtin_amdgcn_s_memrealtime();
    }
}

static unsigned magic20(int d) { return ((1u << 20) + (unsigned)d - 1) / (unsigned)d; }
static bool magic_exact(unsigned mag, int dv, int limit) {
    for (int i = 0; i < limit; ++i)
        if ((int)(((unsigned)i * mag) >> 20) != i / dv) return false;
    return true;
}

// ------------------------------------------------------------------------------------------------------------------
// Direct form: no LDS tile, no staging phase.  A pure data-movement kernel of K2's bytes runs in 3-6 us at the 18x30 / 36x60
// levels (tools/ubench/floor.hip: 4.0 us for the 22 MB of a 18x30 launch) where the tiled kernel above takes 11-12 us: its
// load -> LDS -> barrier -> arithmetic -> store phases run one after the other in every workgroup of a one-round launch.  Here
// a lane owns RV vertically adjacent 1x4 output strips of one plane and reads the input rows it needs straight from
// global memory (the rows of a plane are L1 / L2 hits after their first use; each row is read once per output row that uses it)
// with 16-byte BUFFER loads whose out-of-range lanes (rows above / below the plane, the float4 left of column 0 or right of
// the row end) are given an offset beyond the descriptor's range and come back as zeros: the zero padding costs no
// arithmetic.  All loads of an item are issued before the first FMA; contributions are accumulated input row by input row,
// which visits (branch, kernel row) in the same order per accumulator as the tiled kernel: results are bit-identical.
// Weights / epilogue constants of the workgroup's planes still sit in LDS (48 + 16 floats per plane, staged while the loads fly).
struct DdGeom {
    int N, n, H, W, Ho, Wo;
    int CP, TH, bands, cgroups, ntiles;
    int XS;                   // strips of 4 outputs per output row
    unsigned mag_xs;          // exact it / XS for it < CP * TH * XS (verified on the host)
    int wt;
    unsigned in_bytes;        // bytes of x: range of the load descriptor
};

typedef unsigned int dd_u32x4 __attribute__((ext_vector_type(4)));

template <int STRIDE, class DS, int RV, bool PARTIAL>
__global__ __launch_bounds__(512, 2) void eesp_dw_direct_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 DdGeom g, Epi e, float* __restrict__ out) {
    constexpr int MAXD = DS::maxd();
    constexpr int NF = STRIDE == 1 ? 3 : 4;                      // float4s of an input row that a strip's taps touch
    constexpr int NRO = (RV - 1) * STRIDE + 2 * MAXD + 1;         // input rows of an item
    __shared__ __attribute__((aligned(16))) float wl[4 * 48];     // [CP <= 4][branch * 3 + ky][4]
    __shared__ __attribute__((aligned(16))) float el[4 * 16];     // [CP][branch][4]
    const int tid = threadIdx.x, nthr = blockDim.x;
    int L = xcd_remap(blockIdx.x, g.ntiles);
    const int band = L % g.bands;  L /= g.bands;
    const int cg = L % g.cgroups;
    const int img = L / g.cgroups, c0 = cg * g.CP, y0b = band * g.TH;
    const int rows_here = min(g.TH, g.Ho - y0b);
    const int rgs = (rows_here + RV - 1) / RV;                    // row groups of this band
    const int items = g.CP * rgs * g.XS;
    const unsigned mag_rg = ((1u << 20) + (unsigned)rgs - 1) / (unsigned)rgs;
    const int hw = g.Ho * g.Wo;
    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)g.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((size_t)g.N * e.ctot * hw * sizeof(float)), 0x00020000);
    const bool o16 = (g.Wo & 3) == 0;
    const bool has_act = e.alpha != nullptr;
    const size_t kstride = (size_t)g.n * hw * sizeof(float);
    char* ob = reinterpret_cast<char*>(out + ((size_t)img * e.ctot + e.coff + c0) * (size_t)hw);
    const unsigned pbase = (unsigned)((((size_t)img * g.n + c0) * g.H) * (size_t)g.W * sizeof(float));   // < 2^31 (host)
    const unsigned plane_bytes = (unsigned)((size_t)g.H * g.W * sizeof(float));
    const unsigned row_bytes = (unsigned)(g.W * sizeof(float));

    // constants of this workgroup's planes: loads go out first, the LDS writes wait for them while the first item's rows fly
    float wreg = 0.f, ereg = 0.f;
    const int nwts = g.CP * 48, neps = g.CP * 16;
    if (tid < nwts) {
        const int p = tid / 48, r = tid - p * 48, kq = r >> 2, kx = r & 3, k = kq / 3, ky = kq - 3 * k;
        if (kx < 3) wreg = w[((size_t)k * g.n + (c0 + p)) * 9 + ky * 3 + kx];
    }
    if (tid < neps) {
        const int p = tid >> 4, r = tid & 15, k = r >> 2, f = r & 3;
        const int cabs = e.coff + k * g.n + c0 + p;
        const float* src = f == 0 ? e.scale : (f == 1 ? e.shift : e.alpha);
        ereg = f == 3 ? 0.f : (src ? src[cabs] : (f == 1 ? 0.f : 1.f));
    }

    for (int it0 = 0; it0 < items; it0 += nthr) {                 // uniform trip count (the barrier sits in the first trip)
        const bool act = it0 + tid < items;
        const int it = act ? it0 + tid : 0;
        const int t2 = (int)(((unsigned)it * g.mag_xs) >> 20);
        const int xs = it - t2 * g.XS;
        const int p = (int)(((unsigned)t2 * mag_rg) >> 20);
        const int rg = t2 - p * rgs;
        const int yo = y0b + rg * RV;                             // first output row of the item
        const int cc0 = xs * 4 * STRIDE - 4;                      // input column of window element 0
        const int rbase = yo * STRIDE - MAXD;                     // input row of window row 0
        const unsigned off0 = pbase + (unsigned)p * plane_bytes + (unsigned)(rbase * (int)row_bytes + cc0 * 4);   // wraps only where invalid
        bool cv[NF], part[NF];
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int c = cc0 + 4 * i;
            cv[i] = c >= 0 && c < g.W;
            part[i] = PARTIAL && (c + 2 == g.W);                  // W % 4 == 2: the float4 that straddles the row end
        }
        // ---- all loads of the item
        float win[NRO][NF * 4];
#pragma unroll
        for (int ro = 0; ro < NRO; ++ro) {
            const int r = rbase + ro;
            const bool rv = r >= 0 && r < g.H;
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const unsigned off = (rv && cv[i]) ? off0 + (unsigned)ro * row_bytes + 16u * i : 0x80000000u;
                const dd_u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(irsrc, (int)off, 0, 0);
                win[ro][4 * i] = __uint_as_float(q.x); win[ro][4 * i + 1] = __uint_as_float(q.y);
                win[ro][4 * i + 2] = __uint_as_float(q.z); win[ro][4 * i + 3] = __uint_as_float(q.w);
            }
        }
        if (it0 == 0) {
            if (tid < nwts) wl[tid] = wreg;
            if (tid < neps) el[tid] = ereg;
            __syncthreads();
        }
        if (PARTIAL) {
#pragma unroll
            for (int ro = 0; ro < NRO; ++ro)
#pragma unroll
                for (int i = 1; i < NF; ++i) {
                    win[ro][4 * i + 2] = part[i] ? 0.f : win[ro][4 * i + 2];
                    win[ro][4 * i + 3] = part[i] ? 0.f : win[ro][4 * i + 3];
                }
        }
        // ---- arithmetic, input row by input row
        const float4* wp = reinterpret_cast<const float4*>(wl) + p * 12;
        const float4* ep = reinterpret_cast<const float4*>(el) + p * 4;
        float a[RV][4][4];
#pragma unroll
        for (int ry = 0; ry < RV; ++ry)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) a[ry][k][j] = 0.f;
#pragma unroll
        for (int ro = 0; ro < NRO; ++ro) {
#pragma unroll
            for (int ry = 0; ry < RV; ++ry)
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const int d = DS::d(k);
                        if (ry * STRIDE + (ky - 1) * d + MAXD != ro) continue;       // compile time
                        const float4 w4 = wp[k * 3 + ky];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int cj = 4 + j * STRIDE;                           // window index of the centre tap
                            a[ry][k][j] = fmaf(w4.x, win[ro][cj - d], a[ry][k][j]);
                            a[ry][k][j] = fmaf(w4.y, win[ro][cj], a[ry][k][j]);
                            a[ry][k][j] = fmaf(w4.z, win[ro][cj + d], a[ry][k][j]);
                        }
                    }
        }
        // ---- hierarchical feature fusion, folded BN + PReLU, stores
#pragma unroll
        for (int ry = 0; ry < RV; ++ry) {
            const int y = yo + ry;
            const bool yok = act && y < y0b + rows_here;
            const unsigned voff = (unsigned)((((size_t)p * hw) + (size_t)y * g.Wo + xs * 4) * sizeof(float));
            float prev[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 ec = ep[k];
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a[ry][k][j] += prev[j];
                    prev[j] = a[ry][k][j];
                    float q = fmaf(a[ry][k][j], ec.x, ec.y);
                    if (has_act) q = q > 0.f ? q : ec.z * q;
                    v[j] = q;
                }
                if (!yok) continue;
                if (e.raw) {                                      // uniform; training forward (un-sliced destination: same offsets)
                    float* rd = reinterpret_cast<float*>(reinterpret_cast<char*>(e.raw + ((size_t)img * e.ctot + c0) * (size_t)hw) + k * kstride + voff);
                    if (o16) *reinterpret_cast<float4*>(rd) = make_float4(a[ry][k][0], a[ry][k][1], a[ry][k][2], a[ry][k][3]);
                    else if (xs * 4 + 2 < g.Wo) { rd[0] = a[ry][k][0]; rd[1] = a[ry][k][1]; rd[2] = a[ry][k][2]; rd[3] = a[ry][k][3]; }
                    else { rd[0] = a[ry][k][0]; rd[1] = a[ry][k][1]; }
                }
                float* dst = reinterpret_cast<float*>(ob + k * kstride + voff);
                const int boff = (int)(reinterpret_cast<const char*>(dst) - reinterpret_cast<const char*>(out));
                const dd_u32x4 dv = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                if (o16 || xs * 4 + 2 < g.Wo) {                   // (Wo % 4 == 2: the row's last strip holds two pixels)
                    if (g.wt) __builtin_amdgcn_raw_buffer_store_b128(dv, orsrc, boff, 0, 16);
                    else __builtin_amdgcn_raw_buffer_store_b128(dv, orsrc, boff, 0, 0);
                } else {
                    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                    const u32x2 d0 = {dv.x, dv.y};
                    if (g.wt) __builtin_amdgcn_raw_buffer_store_b64(d0, orsrc, boff, 0, 16);
                    else __builtin_amdgcn_raw_buffer_store_b64(d0, orsrc, boff, 0, 0);
                }
            }
        }
    }
}

// Returns MSPL_OK when launched, 1 when the shape is left to the tiled kernel.
template <int STRIDE, class DS>
static int launch_direct(const float* x, const float* w, int N, int n, int H, int W, const Epi& e, float* out, hipStream_t s) {
    static const int dbg = MSPL_TUNE_INT("MSPL_DW_DIRECT", 1);
    static const int dbg_rv = MSPL_TUNE_INT("MSPL_DW_RV", 0);
    static const int dbg_cp = MSPL_TUNE_INT("MSPL_DW_DCP", 0);
    static const int dbg_th = MSPL_TUNE_INT("MSPL_DW_DTH", 0);
    static const int dbg_t = MSPL_TUNE_INT("MSPL_DW_DT", 0);
    // Measured (tools/bench_ops.py k2, batch 16): stride 2 -- 72x120 -> 36x60: 13.6 -> 10.5 us, 36x60 -> 18x30: 9.1 -> 7.4 us,
    // 144x240 -> 72x120: 32.3 -> 30.9 us.  Stride 1 (18x30: 12.0 vs 11.3 us, 36x60: 15.6 vs 12.6 us) stays with the tiled kernel:
    // there every input row is read by 7-9 output rows and the 16-byte loads of all those lanes go through the CU's one
    // 64 B/clk L1 path, which is slower than the tiled kernel's LDS reads (MSPL_DW_DIRECT=2 forces the direct form).
    if (!dbg) return 1;
    if (STRIDE == 1 && dbg < 2) return 1;
    DdGeom g;
    memset(&g, 0, sizeof(g));
    g.N = N; g.n = n; g.H = H; g.W = W;
    g.Ho = (H - 1) / STRIDE + 1;  g.Wo = (W - 1) / STRIDE + 1;
    const bool partial = (W & 3) == 2;
    if ((W & 3) != 0 && !partial) return 1;                       // odd row lengths: tiled kernel
    if ((g.Wo & 1) != 0) return 1;
    const size_t in_bytes = (size_t)N * n * H * W * sizeof(float), out_bytes = (size_t)N * e.ctot * g.Ho * g.Wo * sizeof(float);
    if (in_bytes >= (1ull << 31) || out_bytes >= (1ull << 31)) return 1;
    if ((((uintptr_t)x) & 15) || (((uintptr_t)out) & 15)) return 1;
    g.in_bytes = (unsigned)in_bytes;
    g.XS = ceil_div(g.Wo, 4);
    const int RV = dbg_rv > 0 ? dbg_rv : (STRIDE == 2 && g.Ho >= 36 ? 2 : 1);
    // tile = CP planes x TH output rows with CP * ceil(TH / RV) * XS items, ideally one per thread of a block of T <= 512 threads
    int cp = 0, th = 0, T = 0;
    double best = 1e30;
    for (int c = 1; c <= 4; ++c) {
        if (n % c) continue;
        for (int nb = 1; nb <= g.Ho; ++nb) {
            const int h = ceil_div(ceil_div(g.Ho, nb), RV) * RV;
            if (nb > 1 && ceil_div(ceil_div(g.Ho, nb - 1), RV) * RV == h) continue;
            const int items = c * (h / RV) * g.XS;
            for (int t = 64; t <= 512; t += 64) {
                if (items > t) continue;                          // one item per thread
                const double waste = (double)t / items;
                const int64_t tiles = (int64_t)N * (n / c) * ceil_div(g.Ho, h);
                const double waves = (double)tiles * (t / 64);
                const double starve = waves >= 2048 ? 1.0 : 2048.0 / waves;
                const double halo = (double)((h - 1) * STRIDE + 1 + 2 * DS::maxd()) / (double)((h - 1) * STRIDE + 1);   // L1 misses at band edges
                const double score = waste * starve * (0.8 + 0.2 * halo);
                if (score < best - 1e-9) { best = score; cp = c; th = h; T = t; }
                break;                                            // smallest t that fits
            }
        }
    }
    if (T == 0) return 1;
    if (dbg_cp > 0 && dbg_cp <= 4 && n % std::max(dbg_cp, 1) == 0) cp = dbg_cp;
    if (dbg_th > 0) th = ceil_div(std::min(dbg_th, g.Ho), RV) * RV;
    if (dbg_cp > 0 || dbg_th > 0 || dbg_t > 0) {
        const int items = cp * (th / RV) * g.XS;
        T = dbg_t > 0 ? dbg_t : std::min(512, ceil_div(items, 64) * 64);
        if (T % 64 || T < 64 || T > 512) return 1;
    }
    g.CP = cp; g.TH = th;
    g.bands = ceil_div(g.Ho, th);
    g.cgroups = n / cp;
    const int64_t ntiles = (int64_t)N * g.cgroups * g.bands;
    if (ntiles >= (1ll << 30)) return 1;
    g.ntiles = (int)ntiles;
    g.mag_xs = magic20(g.XS);
    const int max_items = cp * (th / RV) * g.XS;
    if (max_items >= 4096 || !magic_exact(g.mag_xs, g.XS, max_items + 1024)) return 1;
    for (int rg = 1; rg <= th / RV; ++rg)
        if (!magic_exact(magic20(rg), rg, cp * rg + 1024 / g.XS + 2)) return 1;
    static const int dbg_wt = MSPL_TUNE_INT("MSPL_DW_WT", 1);
    g.wt = dbg_wt;
    const dim3 grid((unsigned)ntiles), blk((unsigned)T);
#define MSPL_DD(RVV) do { if (partial) hipLaunchKernelGGL((eesp_dw_direct_kernel<STRIDE, DS, RVV, true>), grid, blk, 0, s, x, w, g, e, out); \
                          else hipLaunchKernelGGL((eesp_dw_direct_kernel<STRIDE, DS, RVV, false>), grid, blk, 0, s, x, w, g, e, out); } while (0)
    if (RV == 2) MSPL_DD(2); else MSPL_DD(1);
#undef MSPL_DD
    MSPL_CHECK_LAUNCH("eesp_dw_hff(direct)");
    return MSPL_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Streaming form, stride 2 (round 3).  The direct form above re-reads every input row through L1 once per output row that uses it:
// 44 16-byte loads per lane for 8 x 4 outputs at 144x240 -> 72x120, ~18 L1 reads per input element.  Alone that launch runs at
// 3.1 TB/s; with three label passes in flight (what the headline measures) the same launch takes 180 us instead of 74 us: the
// CU's one L1 / texture-address path is the resource the co-running kernels fight over (profiles/r03_kernel_stats_inflight3.csv:
// 11.5 % of all kernel time).  Here every input element is loaded ONCE: a lane owns 8 adjacent input columns (4 output columns)
// of one plane and walks down the rows; the 4 halo columns on either side come from the neighbouring lanes (v_mov_dpp wave_shr /
// wave_shl); an input row is consumed the moment it arrives -- it contributes to the (at most five) output rows that use it,
// which are carried in a ring of accumulators -- so there is no row window in registers and no LDS.  Per accumulator the
// contributions arrive by ascending input row = kernel row 0, 1, 2, each as (kx = 0, 1, 2): the order of the other two forms,
// bit-identical results.
//   step t (input rows 2t, 2t+1):  even row 2t  -> output rows t-2 (d=4,ky=2) t-1 (d=2,ky=2) t (ky=1, all d) t+1 (d=2,ky=0) t+2 (d=4,ky=0)
//                                  output row t-2 is complete: hierarchical sums, folded BN + PReLU, four 16-byte stores
//                                  odd row 2t+1 -> output rows t-1 (d=3,ky=2) t (d=1,ky=2) t+1 (d=1,ky=0) t+2 (d=3,ky=0)
// Units (image, row segment) of the SAME channel share a wave (SUB = 64 / (W/8 + 1) of them, one dead lane between two units whose
// out-of-range loads return zeros = the zero padding the DPP exchange needs), so weights and epilogue constants are scalar loads.
// Rows outside the plane / dead lanes: buffer loads with an offset beyond the descriptor's range (zeros, no traffic, no select).
struct S2Geom {
    int N, n, H, W, Ho, Wo;
    int LPR, SLOT, SUB;          // lanes per input row (W / 8), lanes per unit slot (LPR + 1), units per wave
    int SEG, nseg, upc, wpc;     // output rows per unit, units per plane, units and waves per channel
    unsigned total;              // waves
    unsigned in_bytes, out_bytes;
    int wt;
};

__device__ __forceinline__ float s2_from_left(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float s2_from_right(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}

template <class DS, bool WT>
__global__ __launch_bounds__(256) void eesp_dw_stream2_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ escale, const float* __restrict__ eshift,
                                                              const float* __restrict__ ealpha, int ctot, int coff, S2Geom g,
                                                              float* __restrict__ out) {
    static_assert(DS::maxd() <= 4, "accumulator ring of five output rows");
    const int lane = threadIdx.x & 63;
    const unsigned wid = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (wid >= g.total) return;                                   // wave-uniform; no barrier in this kernel
    const int c = wid / g.wpc, wic = wid - c * g.wpc;            // uniform
    const int slot = lane / g.SLOT, cl = lane - slot * g.SLOT;
    const int unit = wic * g.SUB + slot;
    const bool live = slot < g.SUB && cl < g.LPR && unit < g.upc;
    const int uc = live ? unit : 0;
    const int img = uc / g.nseg, sgi = uc - img * g.nseg;
    const int ys = sgi * g.SEG, ye = min(ys + g.SEG, g.Ho);
    // plane constants (wave-uniform: scalar loads)
    float wk[4][9], sc[4], sh[4], al[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int q = 0; q < 9; ++q) wk[k][q] = w[((size_t)k * g.n + c) * 9 + q];
        const int cabs = coff + k * g.n + c;
        sc[k] = escale ? escale[cabs] : 1.f;  sh[k] = eshift ? eshift[cabs] : 0.f;  al[k] = ealpha ? ealpha[cabs] : 1.f;
    }
    const bool has_act = ealpha != nullptr;
    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)g.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)g.out_bytes, 0x00020000);
    const unsigned row_bytes = (unsigned)g.W * 4u, orow_bytes = (unsigned)g.Wo * 4u;
    const unsigned ibase = (unsigned)((((size_t)img * g.n + c) * g.H) * (size_t)g.W * 4u) + (unsigned)cl * 32u;
    const unsigned obase = (unsigned)((((size_t)img * ctot + coff + c) * g.Ho) * (size_t)g.Wo * 4u) + (unsigned)cl * 16u;
    const unsigned okstride = (unsigned)((size_t)g.n * g.Ho * g.Wo * 4u);
    constexpr unsigned OOR = 0x80000000u;

    auto request = [&](int r, dd_u32x4& a, dd_u32x4& b) {           // input row r of this lane's unit: columns 8 cl .. 8 cl + 7
        const unsigned off = (live && r >= 0 && r < g.H) ? ibase + (unsigned)r * row_bytes : OOR;
        a = __builtin_amdgcn_raw_buffer_load_b128(irsrc, (int)off, 0, 0);
        b = __builtin_amdgcn_raw_buffer_load_b128(irsrc, (int)(off + 16u), 0, 0);       // (OOR + 16 is out of range too)
    };
    auto widen = [&](const dd_u32x4& a, const dd_u32x4& b, float (&r)[16]) {
        r[4] = __uint_as_float(a.x); r[5] = __uint_as_float(a.y); r[6] = __uint_as_float(a.z); r[7] = __uint_as_float(a.w);
        r[8] = __uint_as_float(b.x); r[9] = __uint_as_float(b.y); r[10] = __uint_as_float(b.z); r[11] = __uint_as_float(b.w);
        r[0] = s2_from_left(r[8]); r[1] = s2_from_left(r[9]); r[2] = s2_from_left(r[10]); r[3] = s2_from_left(r[11]);
        r[12] = s2_from_right(r[4]); r[13] = s2_from_right(r[5]); r[14] = s2_from_right(r[6]); r[15] = s2_from_right(r[7]);
    };

    float acc[5][4][4];                                           // [output row t-2 .. t+2][branch][column]
#pragma unroll
    for (int q = 0; q < 5; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[q][k][j] = 0.f;

    // the rows of step t + 2 are requested while step t is computed (two steps = 8 KB per wave in flight: with one step the
    // launch ran at 4.5 TB/s, short of bytes in flight)
    dd_u32x4 ea, eb, oa, ob, fa, fb, pa, pb;
    request(2 * (ys - 2), ea, eb);
    request(2 * (ys - 2) + 1, oa, ob);
    request(2 * (ys - 1), fa, fb);
    request(2 * (ys - 1) + 1, pa, pb);
    const int steps = g.SEG + 4;                                  // uniform
    // One step; R = how far the ring has turned: the accumulators of output row t - 2 + q live in acc[(q + R) % 5].  The loop below
    // is unrolled over the ring's period, so "the ring moves on" is a renaming, not 64 register moves per step.
    auto step = [&](auto rc, int i) {
        constexpr int R = decltype(rc)::value % 5, P = decltype(rc)::value & 1;
        const int t = ys - 2 + i;
        float re[16], ro[16];
        // two request buffers taken in turn: the one consumed now is refilled with the rows of step t + 2 (no register copies: a copy
        // of a register with a load in flight waits for the load)
        const int rn = i + 2 < steps ? 2 * t + 4 : -2;            // (steps past the unit's last one request nothing)
        if (P == 0) {
            widen(ea, eb, re);  widen(oa, ob, ro);
            request(rn, ea, eb);  request(rn + 1, oa, ob);
        } else {
            widen(fa, fb, re);  widen(pa, pb, ro);
            request(rn, fa, fb);  request(rn + 1, pa, pb);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- even input row 2t
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int d = DS::d(k), off = d * (ky - 1);
                if (off & 1) continue;                            // compile time
                const int q = (2 - off / 2 + R) % 5;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int cj = 4 + 2 * j;
                    acc[q][k][j] = fmaf(wk[k][ky * 3], re[cj - d], acc[q][k][j]);
                    acc[q][k][j] = fmaf(wk[k][ky * 3 + 1], re[cj], acc[q][k][j]);
                    acc[q][k][j] = fmaf(wk[k][ky * 3 + 2], re[cj + d], acc[q][k][j]);
                }
            }
        // ---- output row t - 2 is complete
        {
            constexpr int q0 = R % 5;
            const int yo = t - 2;
            const bool ok = live && i >= 4 && yo < ye;
            const unsigned o0 = ok ? obase + (unsigned)yo * orow_bytes : OOR;
            float prev[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = acc[q0][k][j] + prev[j];
                    prev[j] = a;
                    float q = fmaf(a, sc[k], sh[k]);
                    if (has_act) q = q > 0.f ? q : al[k] * q;
                    v[j] = q;
                    acc[q0][k][j] = 0.f;                          // becomes output row t + 3 of the next step
                }
                const dd_u32x4 dv = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                const unsigned oo = ok ? o0 + (unsigned)k * okstride : OOR;
                __builtin_amdgcn_raw_buffer_store_b128(dv, orsrc, (int)oo, 0, WT ? 16 : 0);      // 16 = sc1 (write-through, as the other forms)
            }
        }
        // ---- odd input row 2t + 1 (its output rows t-1 .. t+2 are slots 1 .. 4: slot 0 was just retired)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int d = DS::d(k), off = d * (ky - 1);
                if (!(off & 1)) continue;                         // compile time
                const int q = (2 + (1 - off) / 2 + R) % 5;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int cj = 4 + 2 * j;
                    acc[q][k][j] = fmaf(wk[k][ky * 3], ro[cj - d], acc[q][k][j]);
                    acc[q][k][j] = fmaf(wk[k][ky * 3 + 1], ro[cj], acc[q][k][j]);
                    acc[q][k][j] = fmaf(wk[k][ky * 3 + 2], ro[cj + d], acc[q][k][j]);
                }
            }
    };
    // 10 steps per trip = lcm(ring period 5, request buffers 2), all of them unconditional: with a branch per step the compiler's
    // s_waitcnt placement merges the paths and ends up draining the previous step's stores before every step (1.0 us per step instead
    // of 0.5).  Steps past the unit's last one compute on zeros and store nothing; the launcher picks SEG with (SEG + 4) % 10 == 0.
#pragma unroll 1
    for (int i = 0; i < steps; i += 10) {
        step(std::integral_constant<int, 0>(), i);
        step(std::integral_constant<int, 1>(), i + 1);
        step(std::integral_constant<int, 2>(), i + 2);
        step(std::integral_constant<int, 3>(), i + 3);
        step(std::integral_constant<int, 4>(), i + 4);
        step(std::integral_constant<int, 5>(), i + 5);
        step(std::integral_constant<int, 6>(), i + 6);
        step(std::integral_constant<int, 7>(), i + 7);
        step(std::integral_constant<int, 8>(), i + 8);
        step(std::integral_constant<int, 9>(), i + 9);
    }
}

// Returns MSPL_OK when launched, 1 when the shape is left to the other forms.
template <class DS>
static int launch_stream2(const float* x, const float* w, int N, int n, int H, int W, const Epi& e, float* out, hipStream_t s) {
    // 0 off, 1 auto, 2 whenever the shape allows: a per-call launch flag (the forms are bit-identical; tests compare them)
    const int mode = (e.flags & MSPL_LAUNCH_K2_STREAM_OFF) ? 0 : (e.flags & MSPL_LAUNCH_K2_STREAM_FORCE) ? 2 : 1;
    if (!mode || DS::maxd() > 4) return 1;
    if ((W & 7) != 0 || W / 8 > 63 || H < 8) return 1;
    if (e.pre_add || e.residual || e.reinf_r || e.gate || e.raw) return 1;
    S2Geom g;
    memset(&g, 0, sizeof(g));
    g.N = N; g.n = n; g.H = H; g.W = W;
    g.Ho = (H - 1) / 2 + 1;  g.Wo = W / 2;
    const size_t in_bytes = (size_t)N * n * H * W * sizeof(float), out_bytes = (size_t)N * e.ctot * g.Ho * g.Wo * sizeof(float);
    if (in_bytes >= (1ull << 31) || out_bytes >= (1ull << 31)) return 1;
    if ((((uintptr_t)x) & 15) || (((uintptr_t)out) & 15)) return 1;
    g.in_bytes = (unsigned)in_bytes;  g.out_bytes = (unsigned)out_bytes;
    g.LPR = W / 8;  g.SLOT = g.LPR + 1;  g.SUB = 64 / g.SLOT;
    if (g.SUB < 1) return 1;
    // Rows per unit.  The kernel runs 10 steps per loop trip, so SEG + 4 is a multiple of 10 (6, 16, 26, ...); a unit reads 8 halo
    // rows on top of its 2 * SEG, so long units are cheaper -- as long as ~3 waves per CU remain (measured, tools/bench_ops.py k2,
    // MSPL_DW_SSEG sweep: 144x240 n=24 batch 16: SEG 6 / 16 / 26 / 36 = 25.0 / 20.9 / 23.6 / 28.4 us (960 waves at 16), batch 32:
    // 42.5 / 38.2 / 46.3 / 36.5 us (768 waves at 36); 72x120 n=32 batch 16: 9.6 / 15.6 / 21.5 / 27.3 us (768 waves at 6); the direct
    // form: 30.9, 55.7 and 10.5 us).
    static const int dbg_seg = MSPL_TUNE_INT("MSPL_DW_SSEG", 0);
    auto waves_of = [&](int sg) { return (int64_t)n * ceil_div64((int64_t)N * ceil_div(g.Ho, sg), g.SUB); };
    int seg = 6;
    for (int sg = 16; sg - 10 < g.Ho; sg += 10)
        if (waves_of(std::min(sg, g.Ho)) >= 700) seg = sg;
    if (dbg_seg > 0) seg = dbg_seg;
    seg = std::min(seg, g.Ho);
    if (mode == 1 && waves_of(seg) < 700) return 1;              // too few planes: the other forms
    g.SEG = seg;  g.nseg = ceil_div(g.Ho, seg);
    g.upc = N * g.nseg;  g.wpc = ceil_div(g.upc, g.SUB);
    const int64_t waves = (int64_t)n * g.wpc;
    if (waves >= (1ll << 31)) return 1;
    g.total = (unsigned)waves;
    static const int dbg_wt = MSPL_TUNE_INT("MSPL_DW_WT", 1);
    g.wt = dbg_wt;
    if (g.wt) hipLaunchKernelGGL((eesp_dw_stream2_kernel<DS, true>), dim3((unsigned)ceil_div64(waves, 4)), dim3(256), 0, s, x, w, e.scale, e.shift,
                                 e.alpha, e.ctot, e.coff, g, out);
    else hipLaunchKernelGGL((eesp_dw_stream2_kernel<DS, false>), dim3((unsigned)ceil_div64(waves, 4)), dim3(256), 0, s, x, w, e.scale, e.shift,
                            e.alpha, e.ctot, e.coff, g, out);
    MSPL_CHECK_LAUNCH("eesp_dw_hff(streaming, stride 2)");
    return MSPL_OK;
}

template <int STRIDE, class DS>
static int launch(const float* x, const float* w, int N, int n, int H, int W, const Epi& e, float* out,
                  hipStream_t s) {
    constexpr int MAXD = DS::maxd();
    if (STRIDE == 2) {
        const int rc = launch_stream2<DS>(x, w, N, n, H, W, e, out, s);
        if (rc <= 0) return rc;
    }
    {
        const int rc = launch_direct<STRIDE, DS>(x, w, N, n, H, W, e, out, s);
        if (rc <= 0) return rc;
    }
    DwGeom g;
    memset(&g, 0, sizeof(g));
    g.N = N; g.n = n; g.H = H; g.W = W;
    g.Ho = (H - 1) / STRIDE + 1;
    g.Wo = (W - 1) / STRIDE + 1;
    g.XS = ceil_div(g.Wo, 4);
    g.NCH = ceil_div(W, 4);
    // staged row: stride 1: [4 zero | W values | zero fill] >= 4*XS + 8 floats and RS == 4*XS (mod 64 banks), so that lane
    // addresses stay linear across row ends; stride 2: E and O arrays of LSH >= 4*XS + 8 floats each, LSH == XS (mod 16)
    if (STRIDE == 1) {
        int ls = std::max(4 * g.XS + 8, 4 * g.NCH + 8);
        while (((ls - 4 * g.XS) & 63) != 0) ls += 4;
        g.RS = ls;
    } else {
        int lsh = (std::max(4 * g.XS + 8, 2 * g.NCH + 8) + 3) & ~3;
        if ((g.XS & 3) == 0) { while (((lsh - g.XS) & 15) != 0) lsh += 4; }      // (XS % 4 != 0: no exact fit; a few 2-way conflicts)
        g.RS = 2 * lsh;
    }
    static const int dbg_lds = MSPL_TUNE_INT("MSPL_DW_LDS", 0);     // KiB per workgroup
    static const int dbg_cp = MSPL_TUNE_INT("MSPL_DW_CP", 0);
    static const int dbg_th = MSPL_TUNE_INT("MSPL_DW_TH", 0);
    static const int dbg_wgs = MSPL_TUNE_INT("MSPL_DW_WGS", 0);     // persistent workgroups per CU
    static unsigned long long* stamp_buf = nullptr;
    static const int dbg_stamp = MSPL_STAMP_ENV("MSPL_DW_STAMP");
    if (dbg_stamp && !stamp_buf) (void)hipMalloc(&stamp_buf, (size_t)4 * 65536 * sizeof(unsigned long long));
    // Tile = CP planes x one band of TH output rows, worked on by T threads.  Constraints: LDS budget; at most T * DW_PF staging
    // chunks.  Score (lower is better): idle lanes of the item loop (items rarely fill whole rounds of T lanes: 18 x 8 strips
    // = 144 items leave 44 % of 256 lanes idle, 4 such planes fill 192 lanes three times) x halo rows re-read x a penalty for
    // fewer tiles than the chip needs to be busy.
    static const int dbg_thr = MSPL_TUNE_INT("MSPL_DW_THREADS", 0);
    static const int dbg_persist = MSPL_TUNE_INT("MSPL_DW_PERSIST", -1);
    const size_t lds_budget = (size_t)(dbg_lds > 0 ? dbg_lds : 38) * 1024;
    auto rin_of = [&](int th) { return (th - 1) * STRIDE + 1 + 2 * MAXD; };
    auto lds_of = [&](int th, int cp) { return ((size_t)cp * rin_of(th) * g.RS + 16 + (size_t)cp * 64) * sizeof(float); };
    int th = 0, cp = 0, T = 0;
    double best = 1e30;
    for (int c = 1; c <= 4; c *= 2) {
        if (n % c) break;
        for (int nb = 1; nb <= g.Ho; ++nb) {
            const int h = ceil_div(g.Ho, nb);
            if (nb > 1 && ceil_div(g.Ho, nb - 1) == h) continue;          // same band height as the previous count
            if (lds_of(h, c) > lds_budget) continue;
            for (int t = 64; t <= 256; t += 64) {
                if ((int64_t)c * rin_of(h) * g.NCH > (int64_t)t * DW_PF) continue;
                const int items = c * h * g.XS;
                const double waste = (double)(ceil_div(items, t) * t) / items;
                const double halo = (double)rin_of(h) / (double)((h - 1) * STRIDE + 1);
                const int64_t tiles = (int64_t)N * (n / c) * ceil_div(g.Ho, h);
                const double waves = (double)tiles * (t / 64);
                const double starve = waves >= 2048 ? 1.0 : 2048.0 / waves;        // < 2 waves per SIMD: latency shows
                const double rounds = items / (double)t;                             // long per-thread chains of tiny launches
                const double score = waste * (0.5 + 0.5 * halo) * starve * (1.0 + 0.02 * rounds) * (t < 128 ? 1.1 : 1.0);
                if (score < best - 1e-9) { best = score; th = h; cp = c; T = t; }
            }
        }
    }
    MSPL_REQUIRE(T > 0, MSPL_ERR_UNSUPPORTED, "eesp_dw_hff: row of %d floats does not fit the LDS tile", W);
    if (dbg_th > 0 || dbg_cp > 0 || dbg_thr > 0) {        // tuning overrides (checked against the same constraints)
        const int h2 = dbg_th > 0 ? std::min(dbg_th, g.Ho) : th, c2 = dbg_cp > 0 ? dbg_cp : cp, t2 = dbg_thr > 0 ? dbg_thr : T;
        if (c2 <= 4 && n % c2 == 0 && t2 % 64 == 0 && t2 >= 64 && t2 <= (dbg_persist == 1 ? 256 : 1024) && lds_of(h2, c2) <= 64 * 1024 &&
            (int64_t)c2 * rin_of(h2) * g.NCH <= (int64_t)t2 * DW_PF) { th = h2; cp = c2; T = t2; }
    }
    const int bands = ceil_div(g.Ho, th);
    g.TH = th; g.CP = cp;
    g.RIN = rin_of(th);
    g.bands = bands;
    g.cgroups = n / cp;
    g.chunks = cp * g.RIN * g.NCH;
    const int64_t ntiles = (int64_t)N * g.cgroups * g.bands;
    MSPL_REQUIRE(ntiles < (1ll << 30), MSPL_ERR_BAD_SHAPE, "eesp_dw_hff: too many tiles");
    g.ntiles = (int)ntiles;
    g.mag_xs = magic20(g.XS); g.mag_nch = magic20(g.NCH); g.mag_rin = magic20(g.RIN);
    bool mag_ok = cp * th * g.XS < 4096 && g.chunks <= 4096 && g.RIN < 4096 && g.NCH < 4096 &&
                  magic_exact(g.mag_nch, g.NCH, g.chunks) &&
                  magic_exact(g.mag_rin, g.RIN, cp * g.RIN) && magic_exact(g.mag_xs, g.XS, cp * th * g.XS);
    for (int rh = 1; rh <= th && mag_ok; ++rh) mag_ok = magic_exact(magic20(rh), rh, cp * rh);     // (the last band may be shorter)
    MSPL_REQUIRE(mag_ok, MSPL_ERR_UNSUPPORTED, "eesp_dw_hff: tile geometry outside the magic-division range (CP=%d TH=%d XS=%d)", cp, th, g.XS);
    const size_t lds = lds_of(th, cp);
    MSPL_REQUIRE(lds <= 64 * 1024, MSPL_ERR_UNSUPPORTED, "eesp_dw_hff: tile of %zu bytes exceeds LDS", lds);
    // one tile per workgroup unless the launch is many rounds deep; then persistent workgroups with register prefetch
    int per_cu = (int)std::min<size_t>(std::max(1, 3 * 256 / T), (160 * 1024) / lds);
    if (per_cu < 1) per_cu = 1;
    if (dbg_wgs > 0) per_cu = dbg_wgs;
    bool persist = T <= 256 && ntiles > (int64_t)256 * per_cu * 3;
    if (dbg_persist >= 0) persist = dbg_persist != 0 && T <= 256;
    int64_t blocks = ntiles;
    if (persist) {
        blocks = std::min<int64_t>(ntiles, (int64_t)256 * per_cu);
        if (blocks > 8) blocks -= blocks % 8;             // ids b and b + 8 share an XCD: keep the stride a multiple of 8
    }
    g.stamps = (dbg_stamp && blocks <= 65536) ? stamp_buf : nullptr;
    static const int dbg_wt = MSPL_TUNE_INT("MSPL_DW_WT", 1);     // 0: plain stores (A/B aid)
    g.wt = dbg_wt && (size_t)N * e.ctot * g.Ho * g.Wo * sizeof(float) < (1ull << 31);
    if (persist) hipLaunchKernelGGL((eesp_dw_hff_kernel<STRIDE, DS, true>), dim3((unsigned)blocks), dim3(T), lds, s, x, w, g, e, out);
    else hipLaunchKernelGGL((eesp_dw_hff_kernel<STRIDE, DS, false>), dim3((unsigned)blocks), dim3(T), lds, s, x, w, g, e, out);
    MSPL_CHECK_LAUNCH("eesp_dw_hff");
    if (g.stamps) {   // debug only: synchronous dump of the phase timeline (100 MHz ticks)
        (void)hipDeviceSynchronize();
        static unsigned long long host[4 * 65536];
        (void)hipMemcpy(host, stamp_buf, (size_t)blocks * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t3 = 0; double a = 0, b = 0, c = 0;
        for (int64_t i = 0; i < blocks; ++i) { if (host[4*i] < t0) t0 = host[4*i]; if (host[4*i+3] > t3) t3 = host[4*i+3]; a += host[4*i+1]-host[4*i]; b += host[4*i+2]-host[4*i+1]; c += host[4*i+3]-host[4*i+2]; }
        double late = 0; for (int64_t i = 0; i < blocks; ++i) late += host[4*i] - t0;
        fprintf(stderr, "[k2 stamp] blocks=%lld tiles=%lld span=%.2fus  avg: start-delay=%.2fus first load+ldswrite=%.2fus barrier=%.2fus compute+store(+more tiles)=%.2fus\n", (long long)blocks, (long long)ntiles, (t3-t0)/100.0, late/blocks/100.0, a/blocks/100.0, b/blocks/100.0, c/blocks/100.0);
        {   // start / end time distribution (us after the first start), sorted
            static double st[65536], en[65536];
            for (int64_t i = 0; i < blocks; ++i) { st[i] = (host[4*i] - t0) / 100.0; en[i] = (host[4*i+3] - t0) / 100.0; }
            std::sort(st, st + blocks); std::sort(en, en + blocks);
            auto q = [&](double* v, double f) { return v[(int64_t)(f * (blocks - 1))]; };
            fprintf(stderr, "[k2 stamp]   start p10/50/75/90/100 = %.2f %.2f %.2f %.2f %.2f us | end p10/50/90/100 = %.2f %.2f %.2f %.2f us | CP=%d TH=%d T=%d persist=%d RS=%d lds=%zu\n",
                    q(st, .1), q(st, .5), q(st, .75), q(st, .9), q(st, 1.0), q(en, .1), q(en, .5), q(en, .9), q(en, 1.0), g.CP, g.TH, T, (int)persist, g.RS, lds);
        }
    }
    return MSPL_OK;
}


// ------------------------------------------------------------------------------------------------------------------
// K1 + K2 in one launch for the stride-1 EESP blocks whose planes fit LDS (level 4: 16x30 / 18x30).  The reduced tensor never goes
// to memory: a workgroup owns one image and 16 consecutive projection channels (half a group of the grouped 1x1), computes them
// for the WHOLE plane on the matrix cores (v_mfma_f32_16x16x4_f32: 16 rows x 16 pixels x 4 k per instruction; a lane's float4 of x
// feeds four of them, A = the 16 x K weight slab preloaded into K / 4 registers), applies the projection's folded BN + PReLU and
// writes the values into the same zero-haloed LDS tile the tiled kernel stages from global memory; after one barrier the tile is
// handed to the tiled kernel's arithmetic (dw_compute_items).  Removes one launch, the reduced tensor's write and read, and the
// global -> LDS staging phase per block.  nn_layers/eesp.py:60-80.
typedef float fr_f4 __attribute__((ext_vector_type(4)));

struct FrGeom {
    int N, Cin, n, G, K, M, H, W, HW;
    int RS, RIN, PS, XS;
    unsigned mag_w, mag_xs;
    int TH, bands;            // output rows per band, bands per plane (a band re-computes MAXD halo rows of the projection on each side)
    int wt;
    int stop;                 // tuning aid (MSPL_FRONT_STOP): return after phase k
};

template <class DS, int KQ, int SL>   // KQ = K / 4 MFMA steps; SL = 16-row slabs of the group's projection rows (M = 16 SL)
__global__ __launch_bounds__(512) void eesp_proj_dw_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                           const float* __restrict__ pscale, const float* __restrict__ pshift,
                                                           const float* __restrict__ palpha, const float* __restrict__ w,
                                                           FrGeom g, Epi e, float* __restrict__ out) {
    constexpr int MAXD = DS::maxd();
    constexpr int RING = KQ;                               // ALL k-steps of a tile's B in flight per wave (KQ float4 registers): the loads are the latency to hide
    constexpr int CP = 16 * SL;                            // planes of the workgroup = the group's projection channels
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tile = smem;                                   // CP planes * PS (+16 floats of tail pad)
    float* wl = smem + CP * g.PS + 16;                    // [CP][branch*3 + ky][4]
    float* el = wl + CP * 48;                             // [CP][branch][4]
    float* At = el + CP * 16;                             // [CP][K + 4]
    if (g.stop == 3) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, nwaves = nthr >> 6;
    int bid = blockIdx.x;
    const int band = bid % g.bands;  bid /= g.bands;
    const int img = bid / g.G, grp = bid - img * g.G;
    const int c0 = grp * CP;                              // first projection channel of the workgroup (= of the group)
    const int kq = lane >> 4, nl = lane & 15;
    // rows of this band: outputs [r0, r1); the projection is needed on [q0, q1) = [r0 - MAXD, r1 + MAXD) inside the image; tile row 0
    // is image row r0 - MAXD (rows outside the image stay zero)
    const int r0 = band * g.TH, r1 = min(r0 + g.TH, g.H);
    const int q0 = max(r0 - MAXD, 0), q1 = min(r1 + MAXD, g.H);
    const int pbeg = q0 * g.W, pend = q1 * g.W;           // pixel span of the projection (rows are contiguous)
    const int pb4 = pbeg & ~3;                             // tiles start 16-byte aligned (HW % 4 == 0: a lane's float4 never leaves the plane)
    const int ntiles = (pend - pb4 + 63) >> 6;

    // ---- every global load of the prologue goes out first, in one batch (they return in order: a wait in between would stack
    // their latencies): the first tile's B (all k-steps), the K2 constants, the weight slab, the projection's epilogue constants
    const float* xg = x + ((size_t)img * g.Cin + (size_t)grp * g.K + kq) * g.HW;
    const size_t kstep = (size_t)4 * g.HW;                 // floats between MFMA steps
    auto tile_base = [&](int t, bool& pok, int& p4) {
        p4 = pb4 + t * 64 + 4 * nl;
        pok = t < ntiles && p4 < pend;                     // (a lane's 4 pixels may straddle pbeg / pend: masked below)
        return xg + (pok ? p4 : pb4);
    };
    int t = wave, p4 = 0;
    bool pok = false;
    const float* xb = tile_base(t, pok, p4);
    float4 b[RING];
#pragma unroll
    for (int r = 0; r < RING; ++r) b[r] = *reinterpret_cast<const float4*>(xb + r * kstep);
    float wreg[3], ereg = 0.f;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int i = tid + u * nthr;
        wreg[u] = 0.f;
        if (i < CP * 48) {
            const int p = i / 48, r = i - p * 48, kqq = r >> 2, kx = r & 3, k = kqq / 3, ky = kqq - 3 * k;
            if (kx < 3) wreg[u] = w[((size_t)k * g.n + (c0 + p)) * 9 + ky * 3 + kx];
        }
    }
    if (tid < CP * 16) {
        const int p = tid >> 4, r = tid & 15, k = r >> 2, f = r & 3;
        const int cabs = e.coff + k * g.n + c0 + p;
        const float* src = f == 0 ? e.scale : (f == 1 ? e.shift : e.alpha);
        ereg = f == 3 ? 0.f : (src ? src[cabs] : (f == 1 ? 0.f : 1.f));
    }
    // A: the CP x K weight slab, loaded coalesced (float4s) and handed to the lanes through LDS (row stride K + 4: lane (m = nl,
    // k = 4 st + kq) reads bank 4 nl + kq, conflict-free).  Per-lane strided global loads of it cost 8 us here: 16 cache lines per
    // load instruction, 32 instructions per wave.
    const int AS = g.K + 4, kv = g.K >> 2;
    float4 a4[SL];
#pragma unroll
    for (int u = 0; u < SL; ++u) {
        const int i = tid + u * nthr, row = i / kv, c4 = i - row * kv;
        a4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < CP) a4[u] = *reinterpret_cast<const float4*>(wp + ((size_t)(c0 + row)) * g.K + 4 * c4);
    }
    float psc[SL][4], psh[SL][4], pal[SL][4];              // projection epilogue of the lane's rows 16 s + 4 kq + i
#pragma unroll
    for (int sl = 0; sl < SL; ++sl)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ch = c0 + 16 * sl + 4 * kq + i;
            psc[sl][i] = pscale ? pscale[ch] : 1.f;  psh[sl][i] = pshift ? pshift[ch] : 0.f;  pal[sl][i] = palpha ? palpha[ch] : 1.f;
        }
    const bool pact = palpha != nullptr;
    // ---- zero the tile (halo rows / columns stay zero) while the loads fly, then the constants
    {
        const int tot4 = (CP * g.PS + 16) >> 2;
        for (int i = tid; i < tot4; i += nthr) *reinterpret_cast<float4*>(tile + 4 * i) = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 3; ++u)
            if (tid + u * nthr < CP * 48) wl[tid + u * nthr] = wreg[u];
        if (tid < CP * 16) el[tid] = ereg;
#pragma unroll
        for (int u = 0; u < SL; ++u) {
            const int i = tid + u * nthr, row = i / kv, c4 = i - row * kv;
            if (row < CP) *reinterpret_cast<float4*>(At + row * AS + 4 * c4) = a4[u];
        }
    }
    __syncthreads();                                       // zero fill complete before the projection writes
    if (g.stop == 1) return;
    const float* ar = At + nl * AS + kq;

    // ---- projection: wave v handles 64-pixel tiles v, v + nwaves, ... of the span; a tile's B registers are refilled with the
    // wave's next tile as they are used
    for (; t < ntiles; t += nwaves) {
        bool pokn;  int p4n;
        const float* xbn = tile_base(t + nwaves, pokn, p4n);
        fr_f4 acc[SL][4];
#pragma unroll
        for (int sl = 0; sl < SL; ++sl)
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) acc[sl][s2] = (fr_f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < KQ; ++st) {
            const float4 bv = b[st];
#pragma unroll
            for (int sl = 0; sl < SL; ++sl) {
                const float av = ar[sl * 16 * AS + 4 * st];
                acc[sl][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv.x, acc[sl][0], 0, 0, 0);
                acc[sl][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv.y, acc[sl][1], 0, 0, 0);
                acc[sl][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv.z, acc[sl][2], 0, 0, 0);
                acc[sl][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv.w, acc[sl][3], 0, 0, 0);
            }
            b[st] = *reinterpret_cast<const float4*>(xbn + (size_t)st * kstep);
            __builtin_amdgcn_sched_barrier(0);             // keeps the refills where they are
        }
        if (pok) {
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                const int p = p4 + s2;
                if (p < pbeg || p >= pend) continue;
                const int y = (int)(((unsigned)p * g.mag_w) >> 20), xx = p - y * g.W;
#pragma unroll
                for (int sl = 0; sl < SL; ++sl) {
                    float* d = tile + (size_t)(16 * sl + 4 * kq) * g.PS + (y - r0 + MAXD) * g.RS + 4 + xx;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v = fmaf(acc[sl][s2][i], psc[sl][i], psh[sl][i]);
                        if (pact) v = v > 0.f ? v : pal[sl][i] * v;
                        d[(size_t)i * g.PS] = v;
                    }
                }
            }
        }
        xb = xbn;  pok = pokn;  p4 = p4n;
    }
    if (g.stop == 2) return;
    __syncthreads();

    // ---- K2 on the tile
    const int hw = g.HW;
    DwItemCtx cx;
    cx.CP = CP; cx.rows_here = r1 - r0; cx.XS = g.XS; cx.PS = g.PS; cx.RS = g.RS; cx.Wo = g.W; cx.hw = hw; cx.y0 = r0;
    cx.mag_xs = g.mag_xs; cx.wt = g.wt; cx.o16 = (g.W & 3) == 0; cx.o8 = (g.W & 1) == 0; cx.has_act = e.alpha != nullptr;
    cx.kstride = (size_t)g.n * hw * sizeof(float);
    cx.ob = reinterpret_cast<char*>(out + ((size_t)img * e.ctot + e.coff + c0) * (size_t)hw);
    cx.out = out;
    cx.rawb = nullptr;
    cx.orsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((size_t)g.N * e.ctot * hw * sizeof(float)), 0x00020000);
    dw_compute_items<1, DS>(tile, wl, el, cx, tid, nthr);
}

}  // namespace mspl

using namespace mspl;

// x (N,Cin,H,W); wp: the grouped projection's weights (n, Cin/groups); pscale/pshift/palpha (n): its folded BN + PReLU;
// w (4,n,3,3) + dil + ep: as mspl_eesp_dw_hff_fwd (stride 1).  Returns MSPL_ERR_UNSUPPORTED for shapes the fused form does not
// cover (the caller then runs mspl_conv1x1_fwd + mspl_eesp_dw_hff_fwd); mspl_eesp_proj_dw_hff_fits() asks beforehand.
static int eesp_proj_dw_plan(int N, int Cin, int n, int groups, int H, int W, const int32_t* dil, FrGeom& g, size_t& lds) {
    if (N < 1 || groups < 1 || Cin % groups || n % groups || (n & 15)) return 0;
    const int K = Cin / groups, M = n / groups;
    if (!(K == 64 || K == 128) || !(M == 16 || M == 32)) return 0;
    if ((H * W) & 3 || (W & 1) || W > 64 || H > 64) return 0;
    const int key = dil[0] * 1000 + dil[1] * 100 + dil[2] * 10 + dil[3];
    if (key != 1123 && key != 1234) return 0;
    const int maxd = key == 1123 ? 3 : 4;
    memset(&g, 0, sizeof(g));
    g.N = N; g.Cin = Cin; g.n = n; g.G = groups; g.K = K; g.M = M; g.H = H; g.W = W; g.HW = H * W;
    g.XS = ceil_div(W, 4);
    g.RS = 4 * g.XS + 8;                                   // 4 zero columns + the row + zero fill for the right-hand taps
    // bands: enough workgroups for every CU (a launch of 16 images x 8 slabs is 128 workgroups); every band re-computes up to
    // 2 * maxd projection rows
    static const int dbg_bands = MSPL_TUNE_INT("MSPL_FRONT_BANDS", 0);
    g.mag_w = magic20(W); g.mag_xs = magic20(g.XS);
    const int max_bands = 3;                               // (more bands = more of the projection re-computed in the halos than computed once)
    for (int bands = 1; bands <= max_bands; ++bands) {
        if (dbg_bands >= 1 && dbg_bands <= max_bands && bands != dbg_bands) continue;
        g.TH = ceil_div(H, bands);
        g.bands = ceil_div(H, g.TH);
        g.RIN = g.TH + 2 * maxd;
        g.PS = g.RIN * g.RS;
        lds = ((size_t)M * g.PS + 16 + (size_t)M * 64 + (size_t)M * (K + 4)) * sizeof(float);
        const bool enough = (int64_t)N * groups * g.bands >= 256 || g.TH < 2 * maxd + 2 || bands >= 3;
        if (lds <= 128 * 1024 && (enough || dbg_bands)) break;
        if (bands == max_bands) return 0;
    }
    if (lds > 128 * 1024) return 0;
    if (g.HW >= 4096 || M * g.TH * g.XS >= 4096 || !magic_exact(g.mag_w, W, g.HW + 64) || !magic_exact(g.mag_xs, g.XS, M * g.TH * g.XS)) return 0;
    for (int rh = 1; rh <= g.TH; ++rh)
        if (!magic_exact(magic20(rh), rh, M * rh)) return 0;
    return 1;
}

extern "C" int mspl_eesp_proj_dw_hff_fits(int32_t N, int32_t Cin, int32_t n, int32_t groups, int32_t H, int32_t W, const int32_t dil[4],
                                          uint32_t launch_flags) {
    // Measured on the whole label pass: with one batch in flight the fused launch is worth +2 % (it shortens a latency-bound chain);
    // with three launches of 32 images in flight it costs 1.7 % (its workgroups hold 50-100 KB of LDS and 512 threads, which
    // crowds out the other lanes' kernels).  So: used unless the caller asks for throughput launch shapes (MSPL_LAUNCH_THROUGHPUT,
    // per call); MSPL_EESP_FRONT=0 / =2 force it off / on.
    static const int mode = MSPL_TUNE_INT("MSPL_EESP_FRONT", 1);
    if (mode == 0 || !dil || (mode == 1 && (launch_flags & MSPL_LAUNCH_THROUGHPUT))) return 0;
    FrGeom g; size_t lds;
    return eesp_proj_dw_plan(N, Cin, n, groups, H, W, dil, g, lds);
}

extern "C" int mspl_eesp_proj_dw_hff_fwd(const float* x, const float* wp, const float* pscale, const float* pshift, const float* palpha,
                                         const float* w, const int32_t dil[4], int32_t N, int32_t Cin, int32_t n, int32_t groups,
                                         int32_t H, int32_t W, const mspl_epilogue_t* ep, float* out, void* stream) {
    MSPL_REQUIRE(x && wp && w && dil && out, MSPL_ERR_NULL_POINTER, "eesp_proj_dw_hff: null pointer");
    FrGeom g; size_t lds = 0;
    MSPL_REQUIRE(eesp_proj_dw_plan(N, Cin, n, groups, H, W, dil, g, lds), MSPL_ERR_UNSUPPORTED,
                 "eesp_proj_dw_hff: shape N=%d Cin=%d n=%d groups=%d %dx%d is not covered by the fused form", N, Cin, n, groups, H, W);
    if (int rc = check_epi(ep, 4 * n, "eesp_proj_dw_hff")) return rc;
    MSPL_REQUIRE(!ep || (!ep->pre_add && !ep->residual && !ep->reinf_r && !ep->gate), MSPL_ERR_UNSUPPORTED,
                 "eesp_proj_dw_hff: only scale/shift/alpha epilogue terms are supported (br_after_cat)");
    MSPL_REQUIRE((((uintptr_t)x) & 15) == 0 && (((uintptr_t)out) & 15) == 0, MSPL_ERR_UNSUPPORTED, "eesp_proj_dw_hff: unaligned tensors");
    const Epi e = make_epi(ep, 4 * n, H * W);
    MSPL_REQUIRE((size_t)N * e.ctot * H * W * sizeof(float) < (1ull << 31), MSPL_ERR_UNSUPPORTED, "eesp_proj_dw_hff: output too large");
    static const int dbg_wt = MSPL_TUNE_INT("MSPL_DW_WT", 1);
    g.wt = dbg_wt;
    static const int dbg_stop = MSPL_TUNE_INT("MSPL_FRONT_STOP", 0);
    g.stop = dbg_stop;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)(N * groups * g.bands)), blk(512);
    const int key = dil[0] * 1000 + dil[1] * 100 + dil[2] * 10 + dil[3];
    static std::once_flag attr_once;   // dynamic LDS above 64 KiB needs the opt-in (once per process, thread-safe)
    typedef DilSet<1, 1, 2, 3> DS1123;
    typedef DilSet<1, 2, 3, 4> DS1234;
#define MSPL_FR_ALL(OP) OP(DS1123, 32, 2) OP(DS1123, 16, 1) \
                        OP(DS1234, 32, 2) OP(DS1234, 16, 1) \
                        OP(DS1123, 32, 1) OP(DS1123, 16, 2) \
                        OP(DS1234, 32, 1) OP(DS1234, 16, 2)
    std::call_once(attr_once, [] {
#define MSPL_FR_ATTR(DSX, KQX, SLX) (void)hipFuncSetAttribute((const void*)eesp_proj_dw_kernel<DSX, KQX, SLX>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        MSPL_FR_ALL(MSPL_FR_ATTR)
#undef MSPL_FR_ATTR
        (void)hipGetLastError();
    });
    const int kq = g.K / 4, sl = g.M / 16;
#define MSPL_FR_GO(DSX, KQX, SLX) if (key == (DSX::d(0) * 1000 + DSX::d(1) * 100 + DSX::d(2) * 10 + DSX::d(3)) && kq == KQX && sl == SLX) \
        hipLaunchKernelGGL((eesp_proj_dw_kernel<DSX, KQX, SLX>), grid, blk, lds, s, x, wp, pscale, pshift, palpha, w, g, e, out);
    MSPL_FR_ALL(MSPL_FR_GO)
#undef MSPL_FR_GO
#undef MSPL_FR_ALL
    MSPL_CHECK_LAUNCH("eesp_proj_dw_hff");
    return MSPL_OK;
}

extern "C" int mspl_eesp_dw_hff_fwd(const float* x, const float* w, const int32_t dil[4], int32_t stride,
                                    int32_t N, int32_t n, int32_t H, int32_t W,
                                    const mspl_epilogue_t* ep, float* out, void* stream) {
    MSPL_REQUIRE(x && w && dil && out, MSPL_ERR_NULL_POINTER, "eesp_dw_hff: null pointer");
    MSPL_REQUIRE(N > 0 && n > 0 && H > 0 && W > 0, MSPL_ERR_BAD_SHAPE,
                 "eesp_dw_hff: bad shape N=%d n=%d H=%d W=%d", N, n, H, W);
    MSPL_REQUIRE(stride == 1 || stride == 2, MSPL_ERR_UNSUPPORTED, "eesp_dw_hff: stride %d (1 or 2)", stride);
    MSPL_REQUIRE((int64_t)4 * H * W < (1ll << 29), MSPL_ERR_BAD_SHAPE, "eesp_dw_hff: plane %dx%d too large", H, W);
    if (int rc = check_epi(ep, 4 * n, "eesp_dw_hff", true)) return rc;      // raw_out: the training forward keeps the pre-BN sums
    MSPL_REQUIRE(!ep || (!ep->pre_add && !ep->residual && !ep->reinf_r && !ep->gate), MSPL_ERR_UNSUPPORTED,
                 "eesp_dw_hff: only scale/shift/alpha epilogue terms are supported (br_after_cat)");
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const Epi e = make_epi(ep, 4 * n, Ho * Wo);
    hipStream_t s = (hipStream_t)stream;
    const int key = dil[0] * 1000 + dil[1] * 100 + dil[2] * 10 + dil[3];
#define MSPL_DW_CASE(K, A, B, C, D)                                                              \
    if (key == K) {                                                                              \
        return stride == 1 ? launch<1, DilSet<A, B, C, D>>(x, w, N, n, H, W, e, out, s)          \
                           : launch<2, DilSet<A, B, C, D>>(x, w, N, n, H, W, e, out, s);         \
    }
    MSPL_DW_CASE(1234, 1, 2, 3, 4)
    MSPL_DW_CASE(1123, 1, 1, 2, 3)
    MSPL_DW_CASE(1112, 1, 1, 1, 2)
    MSPL_DW_CASE(1111, 1, 1, 1, 1)
#undef MSPL_DW_CASE
    set_error("eesp_dw_hff: unsupported dilation set {%d,%d,%d,%d}", dil[0], dil[1], dil[2], dil[3]);
    return MSPL_ERR_UNSUPPORTED;
}
