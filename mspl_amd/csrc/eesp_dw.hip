// K2 -- EESP split / transform / hierarchical-feature-fusion kernel (the headline HBM-bound kernel).
//
// Reference arithmetic: nn_layers/eesp.py:68-80 -- four CDilated depthwise 3x3 convolutions
// (espnet_utils.py:118-142, padding = dilation) of the SAME reduced tensor, out_k += out_{k-1},
// torch.cat over branches, then br_after_cat (BatchNorm + PReLU, espnet_utils.py:39-60).
//
// MI355X design: one pass over HBM.  A workgroup owns a band of output rows of CP (image, channel)
// planes; it stages the input rows (+ MAXD halo rows, zero-filled borders) into LDS with 16-byte
// coalesced loads, then every thread produces a 1x4 output strip for all four branches from
// register-resident row windows (one aligned LDS row window feeds every tap of every branch), applies
// the prefix sum across branches and the folded BN + PReLU, and writes the four concatenated planes
// with 16-byte stores.  Algorithmic bytes: 4*n*(H*W + 4*Ho*Wo) per image.
#include <stdlib.h>

#include "common.hpp"

namespace mspl {

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // Give each XCD (blocks b, b+8, ... share one) a contiguous chunk of the logical grid so that
    // neighbouring row bands (which share halo rows) hit the same L2.  Bijective for any nwg.
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

template <int D0, int D1, int D2, int D3>
struct DilSet {
    static constexpr int d(int k) { return k == 0 ? D0 : k == 1 ? D1 : k == 2 ? D2 : D3; }
    static constexpr int maxd() { return D3 > D2 ? (D3 > D1 ? (D3 > D0 ? D3 : D0) : (D1 > D0 ? D1 : D0))
                                                 : (D2 > D1 ? (D2 > D0 ? D2 : D0) : (D1 > D0 ? D1 : D0)); }
};

struct DwGeom {
    int N, n, H, W, Ho, Wo;
    int TH;       // output rows per band
    int CP;       // planes per workgroup (same image, consecutive channels)
    int bands;    // ceil(Ho / TH)
    int cgroups;  // n / CP
    int LS;       // LDS row stride in floats: 4 zero columns + W (rounded to 4) + zero fill; chosen so that lane
                  // addresses stay linear (mod 64 banks) across row ends -> conflict-free ds_read_b128
    unsigned mag_xs;   // exact division of an item index by XS: (t * mag) >> 24
    int RIN;      // staged input rows per plane
    int XS;       // output strips per row
    int nocompute; // tuning aid (MSPL_DW_NOCOMPUTE): skip the stencil, keep loads/stores
    unsigned long long* stamps;  // tuning aid (MSPL_DW_STAMP): 4 s_memrealtime stamps per workgroup, or null
};

// One-shot workgroups, branch-sequential compute, conflict-free LDS addressing.
//  * A thread owns OW = 4/STRIDE adjacent output pixels, i.e. always a 12-float input window starting at
//    input column 4*xs - 4 (three ds_read_b128).  Lanes of a wave are exactly 16 bytes apart.
//  * Staged rows carry the horizontal zero padding (4 zero columns left, zero fill right) and the row stride is
//    chosen with STRIDE*LS == 4*XS (mod 64 banks): the lane -> address map then stays linear across row ends and
//    every ds_read_b128 is bank-conflict free (a plain W+8 stride made 60-75% of the LDS cycles conflict cycles).
//  * The four branches are evaluated one after the other (3 row windows each), carrying the hierarchical sum
//    out_k = conv_k + out_{k-1} in OW registers and storing each branch as soon as it is final: ~50 VGPRs, so
//    6-8 waves/SIMD keep loads, LDS reads, FMAs and stores of different tiles overlapped.
template <int STRIDE, class DS>
__global__ __launch_bounds__(256, 6) void eesp_dw_hff_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ w,
                                                             DwGeom g, Epi e, float* __restrict__ out) {
    constexpr int MAXD = DS::maxd();
    constexpr int OW = 4 / STRIDE;                        // outputs per thread (4 or 2)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tile = smem;                                   // CP * RIN * LS (+16 floats of tail pad)
    float* wl = smem + (size_t)g.CP * g.RIN * g.LS + 16;  // CP * 36 (branch, ky, kx)
    float* el = wl + g.CP * 36;                           // CP * 12 (branch, {scale, shift, alpha})

    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int band = bid % g.bands;  bid /= g.bands;
    const int cg = bid % g.cgroups;
    const int img = bid / g.cgroups;
    const int c0 = cg * g.CP;
    const int y0 = band * g.TH;                 // first output row of the band
    const int iy0 = y0 * STRIDE - MAXD;         // input row of LDS row 0
    const int tid = threadIdx.x;
    unsigned long long st0 = 0, st1 = 0, st2 = 0;
    if (g.stamps) st0 = __builtin_amdgcn_s_memrealtime();

    // ---- stage weights and epilogue constants (loads issued before the tile's, written after)
    float wreg = 0.f, ereg = 0.f;
    const int nwts = g.CP * 36, neps = g.CP * 12;
    if (tid < nwts) {
        const int p = tid / 36, r = tid - p * 36, k = r / 9, t = r - k * 9;
        wreg = w[((size_t)k * g.n + (c0 + p)) * 9 + t];
    }
    if (tid < neps) {
        const int p = tid / 12, r = tid - p * 12, k = r / 3, f = r - k * 3;
        const int cabs = e.coff + k * g.n + c0 + p;
        const float* src = f == 0 ? e.scale : (f == 1 ? e.shift : e.alpha);
        ereg = src ? src[cabs] : (f == 1 ? 0.f : 1.f);
    }

    // ---- stage input rows (rows outside the image are zero).  Loads are issued in batches of UL independent
    // 16-byte loads per thread BEFORE any LDS write, so a workgroup has its whole tile in flight at once.
    {
        constexpr int UL = 8;
        const int nvec = g.LS >> 2;
        const int total = g.CP * g.RIN * nvec;
        const bool w4 = (g.W & 3) == 0, w2 = (g.W & 1) == 0;
        for (int base = 0; base < total; base += 256 * UL) {
            float4 v[UL];
            int dsto[UL];
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                const int i = base + u * 256 + tid;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                dsto[u] = -1;
                if (i < total) {
                    const int rr = i / nvec, cv = i - rr * nvec;
                    const int p = rr / g.RIN, r = rr - p * g.RIN;
                    const int iy = iy0 + r;
                    const int col0 = 4 * cv - 4;             // LDS column j <-> input column j - 4
                    dsto[u] = 4 * i;
                    if (iy >= 0 && iy < g.H && col0 >= 0 && col0 < g.W) {
                        const float* src = x + (((size_t)img * g.n + (c0 + p)) * g.H + iy) * (size_t)g.W + col0;
                        if (w4) {
                            v[u] = *reinterpret_cast<const float4*>(src);
                        } else if (w2) {
                            { const float2 a = *reinterpret_cast<const float2*>(src); v[u].x = a.x; v[u].y = a.y; }
                            if (col0 + 2 < g.W) { const float2 a = *reinterpret_cast<const float2*>(src + 2); v[u].z = a.x; v[u].w = a.y; }
                        } else {
                            v[u].x = src[0];
                            if (col0 + 1 < g.W) v[u].y = src[1];
                            if (col0 + 2 < g.W) v[u].z = src[2];
                            if (col0 + 3 < g.W) v[u].w = src[3];
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < UL; ++u)
                if (dsto[u] >= 0) *reinterpret_cast<float4*>(tile + dsto[u]) = v[u];
        }
    }
    if (tid < nwts) wl[tid] = wreg;
    if (tid < neps) el[tid] = ereg;
    if (g.stamps) st1 = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    if (g.stamps) st2 = __builtin_amdgcn_s_memrealtime();

    // ---- compute: item = (plane p, band row ty, strip xs); xs fastest so that lanes are 16 bytes apart
    const int rows_here = min(g.TH, g.Ho - y0);
    const int items = g.CP * rows_here * g.XS;
    const int hw = g.Ho * g.Wo;
    const bool ovec = (g.Wo % OW) == 0;
    const bool has_act = e.alpha != nullptr;
    const size_t kstride = (size_t)g.n * hw * sizeof(float);
    char* ob = reinterpret_cast<char*>(out + ((size_t)img * e.ctot + e.coff + c0) * (size_t)hw);
    const unsigned mag_rows = ((1u << 24) + (unsigned)rows_here - 1) / (unsigned)rows_here;   // uniform
    for (int it = tid; it < items; it += 256) {
        const int t2 = (int)(((unsigned)it * g.mag_xs) >> 24);
        const int xs = it - t2 * g.XS;
        const int p = (int)(((unsigned)t2 * mag_rows) >> 24);
        const int ty = t2 - p * rows_here;
        // window = LDS columns [4*xs, 4*xs + 12) = input columns [4*xs - 4, 4*xs + 8) of staged row ty*STRIDE + MAXD
        const float* lp = tile + ((size_t)p * g.RIN + ty * STRIDE + MAXD) * g.LS + 4 * xs;
        const float* wp = wl + p * 36;
        const float* ep = el + p * 12;
        const int xb = xs * OW;
        // ONE 32-bit lane offset; the branch part of the address is uniform and goes in the scalar base
        const unsigned voff = (unsigned)((((size_t)p * hw) + (size_t)(y0 + ty) * g.Wo + xb) * sizeof(float));
        float prev[OW];
#pragma unroll
        for (int j = 0; j < OW; ++j) prev[j] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int d = DS::d(k);
            if (k > 0 && DS::d(k - 1) == d) continue;       // evaluated together with the first branch of its run
            // branches with the same dilation read the same three rows: the run k .. k+R-1 shares one set of row reads
            // (level 4 uses d = 1,1,2,3: 27 instead of 36 ds_read_b128 per item)
            int R = 1;
#pragma unroll
            for (int q = k + 1; q < 4; ++q) if (DS::d(q) == d && q == k + R) ++R;
            float a[4][OW];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < OW; ++j) a[r][j] = 0.f;
            if (!g.nocompute) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const float* row = lp + (ky - 1) * d * g.LS;
                    float rv[12];
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const float4 q = *reinterpret_cast<const float4*>(row + 4 * i);
                        // all four elements count as used: keeps this one ds_read_b128 (hipcc otherwise narrows the
                        // read to the elements this dilation touches and emits ~2x as many ds_read2_b32)
                        asm volatile("" :: "v"(q.x), "v"(q.y), "v"(q.z), "v"(q.w));
                        rv[4 * i] = q.x; rv[4 * i + 1] = q.y; rv[4 * i + 2] = q.z; rv[4 * i + 3] = q.w;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (r >= R) break;
                        const float w0 = wp[(k + r) * 9 + ky * 3], w1 = wp[(k + r) * 9 + ky * 3 + 1], w2 = wp[(k + r) * 9 + ky * 3 + 2];
#pragma unroll
                        for (int j = 0; j < OW; ++j) {
                            const int ci = 4 + j * STRIDE;
                            a[r][j] = fmaf(w0, rv[ci - d], a[r][j]);
                            a[r][j] = fmaf(w1, rv[ci], a[r][j]);
                            a[r][j] = fmaf(w2, rv[ci + d], a[r][j]);
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (r >= R) break;
                const int kk = k + r;
                // hierarchical feature fusion: out_k = conv_k + out_{k-1}   (nn_layers/eesp.py:72-76)
                const float sc = ep[kk * 3], sh = ep[kk * 3 + 1], al = ep[kk * 3 + 2];
                float v[OW];
#pragma unroll
                for (int j = 0; j < OW; ++j) {
                    a[r][j] += prev[j];
                    prev[j] = a[r][j];
                    float q = fmaf(a[r][j], sc, sh);
                    if (has_act) q = q > 0.f ? q : al * q;
                    v[j] = q;
                }
                float* dst = reinterpret_cast<float*>(ob + kk * kstride + voff);
                if (ovec) {
                    if constexpr (OW == 4) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                    else *reinterpret_cast<float2*>(dst) = make_float2(v[0], v[1]);
                } else {
#pragma unroll
                    for (int j = 0; j < OW; ++j)
                        if (xb + j < g.Wo) dst[j] = v[j];
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // one branch at a time (keeps the register footprint small)
        }
    }
    if (g.stamps && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* d = g.stamps + (size_t)blockIdx.x * 4;
        d[0] = st0; d[1] = st1; d[2] = st2; d[3] = __builtin_amdgcn_s_memrealtime();
    }
}

static int round_up4(int v) { return (v + 3) & ~3; }

template <int STRIDE, class DS>
static int launch(const float* x, const float* w, int N, int n, int H, int W, const Epi& e, float* out,
                  hipStream_t s) {
    constexpr int MAXD = DS::maxd();
    constexpr int OW = 4 / STRIDE;
    DwGeom g;
    memset(&g, 0, sizeof(g));
    g.N = N; g.n = n; g.H = H; g.W = W;
    g.Ho = (H - 1) / STRIDE + 1;
    g.Wo = (W - 1) / STRIDE + 1;
    g.XS = ceil_div(g.Wo, OW);
    g.mag_xs = ((1u << 24) + (unsigned)g.XS - 1) / (unsigned)g.XS;
    // row stride: >= 4 + W4 + 8 (window over-read), multiple of 4, and STRIDE*LS == 4*XS (mod 64 banks)
    g.LS = round_up4(W) + 12;
    {   // exact when 4*XS is a multiple of 4*STRIDE; otherwise the closest slip (4 floats) -- bounded search
        int best_ls = g.LS, best_err = 1 << 30;
        for (int cand = g.LS; cand < g.LS + 64; cand += 4) {
            const int err = (STRIDE * cand - 4 * g.XS) & 63;
            if (err < best_err) { best_err = err; best_ls = cand; }
            if (err == 0) break;
        }
        g.LS = best_ls;
    }
    static const int dbg_nc = getenv("MSPL_DW_NOCOMPUTE") ? atoi(getenv("MSPL_DW_NOCOMPUTE")) : 0;
    static const int dbg_lds = getenv("MSPL_DW_LDS") ? atoi(getenv("MSPL_DW_LDS")) : 0;     // KiB per workgroup
    static const int dbg_cp = getenv("MSPL_DW_CP") ? atoi(getenv("MSPL_DW_CP")) : 0;
    g.nocompute = dbg_nc;
    static unsigned long long* stamp_buf = nullptr;
    static const int dbg_stamp = MSPL_STAMP_ENV("MSPL_DW_STAMP");
    if (dbg_stamp && !stamp_buf) (void)hipMalloc(&stamp_buf, (size_t)4 * 65536 * sizeof(unsigned long long));
    // Tile = CP planes x one band.  Small tiles -> many workgroups per CU in different phases.
    const size_t lds_budget = (size_t)(dbg_lds > 0 ? dbg_lds : 24) * 1024;
    auto rin_of = [&](int th) { return (th - 1) * STRIDE + 1 + 2 * MAXD; };
    int th = g.Ho;
    while (th > 1 && (size_t)rin_of(th) * g.LS * 4 > lds_budget) th = (th + 1) / 2;
    MSPL_REQUIRE((size_t)rin_of(th) * g.LS * 4 + 512 <= 64 * 1024, MSPL_ERR_UNSUPPORTED,
                 "eesp_dw_hff: row of %d floats does not fit the LDS tile", W);
    const int bands = ceil_div(g.Ho, th);
    int best_cp = 1;
    double best = 1e30;
    for (int cp = 1; cp <= 16; cp *= 2) {
        if (n % cp || (size_t)rin_of(th) * g.LS * 4 * cp > lds_budget || cp * 36 > 256) break;
        const int items = cp * th * g.XS;
        const int64_t tiles = (int64_t)N * (n / cp) * bands;
        const double waste = (double)(ceil_div(items, 256) * 256) / items;     // idle lanes in the item loop
        const double starve = tiles >= 2048 ? 1.0 : 2048.0 / (double)tiles;    // too few workgroups to fill the chip
        const double score = waste * starve;
        if (score < best - 1e-9) { best = score; best_cp = cp; }
    }
    int cp = best_cp;
    if (dbg_cp > 0 && n % dbg_cp == 0 && dbg_cp * 36 <= 256) cp = dbg_cp;
    g.TH = th; g.CP = cp;
    g.RIN = rin_of(th);
    g.bands = bands;
    g.cgroups = n / cp;
    const size_t lds = ((size_t)cp * g.RIN * g.LS + 16 + (size_t)cp * 48) * sizeof(float);
    const int64_t blocks = (int64_t)N * g.cgroups * g.bands;
    MSPL_REQUIRE(blocks < (1ll << 31), MSPL_ERR_BAD_SHAPE, "eesp_dw_hff: grid too large");
    g.stamps = (dbg_stamp && blocks <= 65536) ? stamp_buf : nullptr;
    hipLaunchKernelGGL((eesp_dw_hff_kernel<STRIDE, DS>), dim3((unsigned)blocks), dim3(256), lds, s, x, w, g, e, out);
    MSPL_CHECK_LAUNCH("eesp_dw_hff");
    if (g.stamps) {   // debug only: synchronous dump of the phase timeline (100 MHz ticks)
        (void)hipDeviceSynchronize();
        static unsigned long long host[4 * 65536];
        (void)hipMemcpy(host, stamp_buf, (size_t)blocks * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t3 = 0; double a = 0, b = 0, c = 0;
        for (int64_t i = 0; i < blocks; ++i) { if (host[4*i] < t0) t0 = host[4*i]; if (host[4*i+3] > t3) t3 = host[4*i+3]; a += host[4*i+1]-host[4*i]; b += host[4*i+2]-host[4*i+1]; c += host[4*i+3]-host[4*i+2]; }
        double late = 0; for (int64_t i = 0; i < blocks; ++i) late += host[4*i] - t0;
        fprintf(stderr, "[k2 stamp] blocks=%lld span=%.2fus  avg: start-delay=%.2fus load+ldswrite=%.2fus barrier=%.2fus compute+store=%.2fus\n", (long long)blocks, (t3-t0)/100.0, late/blocks/100.0, a/blocks/100.0, b/blocks/100.0, c/blocks/100.0);
    }
    return MSPL_OK;
}

}  // namespace mspl

using namespace mspl;

extern "C" int mspl_eesp_dw_hff_fwd(const float* x, const float* w, const int32_t dil[4], int32_t stride,
                                    int32_t N, int32_t n, int32_t H, int32_t W,
                                    const mspl_epilogue_t* ep, float* out, void* stream) {
    MSPL_REQUIRE(x && w && dil && out, MSPL_ERR_NULL_POINTER, "eesp_dw_hff: null pointer");
    MSPL_REQUIRE(N > 0 && n > 0 && H > 0 && W > 0, MSPL_ERR_BAD_SHAPE,
                 "eesp_dw_hff: bad shape N=%d n=%d H=%d W=%d", N, n, H, W);
    MSPL_REQUIRE(stride == 1 || stride == 2, MSPL_ERR_UNSUPPORTED, "eesp_dw_hff: stride %d (1 or 2)", stride);
    if (int rc = check_epi(ep, 4 * n, "eesp_dw_hff")) return rc;
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
