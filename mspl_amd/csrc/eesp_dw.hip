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
    int LS;       // LDS row stride in floats (multiple of 4)
    int RIN;      // staged input rows per plane
    int XS;       // ceil(Wo / 4) output strips per row
    int txl_log2; // loader: lanes per LDS row = 1 << txl_log2 (>= LS/4)
};

template <int STRIDE, class DS>
__global__ __launch_bounds__(256) void eesp_dw_hff_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ w,
                                                          DwGeom g, Epi e, float* __restrict__ out) {
    constexpr int MAXD = DS::maxd();
    constexpr int NR = (STRIDE == 1) ? 12 : 16;  // row-window floats per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tile = smem;                                   // CP * RIN * LS
    float* wl = smem + (size_t)g.CP * g.RIN * g.LS;       // CP * 36 (branch, ky, kx)

    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int band = bid % g.bands;  bid /= g.bands;
    const int cg = bid % g.cgroups;
    const int img = bid / g.cgroups;
    const int c0 = cg * g.CP;
    const int y0 = band * g.TH;                 // first output row of the band
    const int iy0 = y0 * STRIDE - MAXD;         // input row of LDS row 0
    const int tid = threadIdx.x;

    // ---- stage weights: wl[p][k][ky][kx] = w[k][c0+p][ky][kx]
    for (int i = tid; i < g.CP * 36; i += 256) {
        const int p = i / 36, r = i - p * 36, k = r / 9, t = r - k * 9;
        wl[i] = w[((size_t)k * g.n + (c0 + p)) * 9 + t];
    }

    // ---- stage input rows (zero-filled outside the image).  LDS column j <-> input column j - 4.
    // Loads are issued in batches of UL independent 16-byte loads per thread BEFORE any LDS write, so a
    // workgroup has its whole tile in flight at once instead of one load-latency per loop iteration.
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
                    const int col0 = 4 * cv - 4;
                    dsto[u] = rr * g.LS + 4 * cv;
                    if (iy >= 0 && iy < g.H) {
                        const float* src = x + (((size_t)img * g.n + (c0 + p)) * g.H + iy) * (size_t)g.W;
                        if (w4) {
                            if (col0 >= 0 && col0 < g.W) v[u] = *reinterpret_cast<const float4*>(src + col0);
                        } else if (w2) {
                            if (col0 >= 0 && col0 < g.W) { const float2 a = *reinterpret_cast<const float2*>(src + col0); v[u].x = a.x; v[u].y = a.y; }
                            if (col0 + 2 >= 0 && col0 + 2 < g.W) { const float2 a = *reinterpret_cast<const float2*>(src + col0 + 2); v[u].z = a.x; v[u].w = a.y; }
                        } else {
                            if (col0 >= 0 && col0 < g.W) v[u].x = src[col0];
                            if (col0 + 1 >= 0 && col0 + 1 < g.W) v[u].y = src[col0 + 1];
                            if (col0 + 2 >= 0 && col0 + 2 < g.W) v[u].z = src[col0 + 2];
                            if (col0 + 3 >= 0 && col0 + 3 < g.W) v[u].w = src[col0 + 3];
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < UL; ++u)
                if (dsto[u] >= 0) *reinterpret_cast<float4*>(tile + dsto[u]) = v[u];
        }
    }
    __syncthreads();

    // ---- compute: item = (plane p, band row ty, strip xs)
    const int rows_here = min(g.TH, g.Ho - y0);
    const int items = g.CP * rows_here * g.XS;
    const int hw = g.Ho * g.Wo;
    const bool o4 = (g.Wo & 3) == 0;
    for (int it = tid; it < items; it += 256) {
        const int xs = it % g.XS;
        const int t2 = it / g.XS;
        const int ty = t2 % rows_here;
        const int p = t2 / rows_here;
        const float* lp = tile + (size_t)p * g.RIN * g.LS + (size_t)(ty * STRIDE) * g.LS + xs * 4 * STRIDE;
        float wr[36];
#pragma unroll
        for (int i = 0; i < 36; i += 4) {
            const float4 t = *reinterpret_cast<const float4*>(wl + p * 36 + i);
            wr[i] = t.x; wr[i + 1] = t.y; wr[i + 2] = t.z; wr[i + 3] = t.w;
        }
        float acc[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[k][j] = 0.f;

#pragma unroll
        for (int o = -MAXD; o <= MAXD; ++o) {
            bool used = (o == 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) used = used || (o == DS::d(k)) || (o == -DS::d(k));
            if (!used) continue;
            float rv[NR];
            const float* row = lp + (size_t)(o + MAXD) * g.LS;
#pragma unroll
            for (int i = 0; i < NR; i += 4) {
                const float4 t = *reinterpret_cast<const float4*>(row + i);
                rv[i] = t.x; rv[i + 1] = t.y; rv[i + 2] = t.z; rv[i + 3] = t.w;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                constexpr int dummy = 0; (void)dummy;
                const int d = DS::d(k);
                int ky = -1;
                if (o == -d) ky = 0; else if (o == 0) ky = 1; else if (o == d) ky = 2;
                if (ky < 0) continue;
                const float w0 = wr[k * 9 + ky * 3 + 0], w1 = wr[k * 9 + ky * 3 + 1], w2 = wr[k * 9 + ky * 3 + 2];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ci = 4 + j * STRIDE;
                    acc[k][j] = fmaf(w0, rv[ci - d], acc[k][j]);
                    acc[k][j] = fmaf(w1, rv[ci], acc[k][j]);
                    acc[k][j] = fmaf(w2, rv[ci + d], acc[k][j]);
                }
            }
        }
        // hierarchical feature fusion: out_k += out_{k-1}   (nn_layers/eesp.py:72-76)
#pragma unroll
        for (int k = 1; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[k][j] += acc[k - 1][j];

        const int y = y0 + ty, xb = xs * 4;
        const int pix = y * g.Wo + xb;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cabs = e.coff + k * g.n + c0 + p;
            const EpiCh ec = epi_channel(e, cabs);
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (xb + j < g.Wo) ? epi_apply(e, ec, acc[k][j], img, cabs, pix + j) : 0.f;
            float* dst = out + ((size_t)img * e.ctot + cabs) * (size_t)hw + pix;
            if (o4) {
                *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (xb + j < g.Wo) dst[j] = v[j];
            }
        }
    }
}

static int round_up4(int v) { return (v + 3) & ~3; }

template <int STRIDE, class DS>
static int launch(const float* x, const float* w, int N, int n, int H, int W, const Epi& e, float* out,
                  hipStream_t s) {
    constexpr int MAXD = DS::maxd();
    DwGeom g;
    g.N = N; g.n = n; g.H = H; g.W = W;
    g.Ho = (H - 1) / STRIDE + 1;
    g.Wo = (W - 1) / STRIDE + 1;
    g.XS = ceil_div(g.Wo, 4);
    g.LS = (STRIDE == 1) ? round_up4(W) + 8 : 2 * round_up4(g.Wo) + 8;
    const size_t lds_budget = 40 * 1024;
    auto rin_of = [&](int th) { return (th - 1) * STRIDE + 1 + 2 * MAXD; };
    int th = g.Ho;
    while (th > 1 && (size_t)rin_of(th) * g.LS * 4 > lds_budget) th = (th + 1) / 2;
    MSPL_REQUIRE((size_t)rin_of(th) * g.LS * 4 + 144 <= 64 * 1024, MSPL_ERR_UNSUPPORTED,
                 "eesp_dw_hff: row of %d floats does not fit the LDS tile", W);
    int cp = 1;
    // small planes: several channels per workgroup so that 256 threads have work
    while (cp * 2 <= 8 && n % (cp * 2) == 0 && th * g.XS * cp < 256 &&
           (size_t)rin_of(th) * g.LS * 4 * (cp * 2) <= lds_budget)
        cp *= 2;
    g.TH = th; g.CP = cp;
    g.RIN = rin_of(th);
    g.bands = ceil_div(g.Ho, th);
    g.cgroups = n / cp;
    int lg = 0;
    while ((1 << lg) < (g.LS >> 2) && lg < 8) ++lg;
    g.txl_log2 = lg;
    const size_t lds = ((size_t)cp * g.RIN * g.LS + (size_t)cp * 36) * sizeof(float);
    const int64_t blocks = (int64_t)N * g.cgroups * g.bands;
    MSPL_REQUIRE(blocks < (1ll << 31), MSPL_ERR_BAD_SHAPE, "eesp_dw_hff: grid too large");
    hipLaunchKernelGGL((eesp_dw_hff_kernel<STRIDE, DS>), dim3((unsigned)blocks), dim3(256), lds, s, x, w, g, e, out);
    MSPL_CHECK_LAUNCH("eesp_dw_hff");
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
