// K13 -- dense (groups = 1) 1x1 / dilated 3x3 convolution with many input channels on the fp32 matrix cores:
// the DeepLabv3 ASPP heads (nn_layers/aspp.py:7-99: Conv2d(2048 | 512 | 1280, 256, k = 1 | 3, padding = dilation =
// 6 | 12 | 18, bias=True) + BatchNorm + ReLU).  This is the one MFMA-bound kernel of the path: 29 GMAC per image at
// 1024x512 / OS16, arithmetic intensity far above the 20 FLOP/B ridge.
//
// Implicit GEMM, never materialising im2col:  out[co][p] = sum_tap sum_ci Wp[tap][co][ci] * x[ci][p shifted by tap]
// (zero outside the image).  Weights arrive pre-packed as (taps, Cout, Cin) so that a workgroup's A chunk
// (128 rows x 32 input channels of one tap) is 128 contiguous 128-byte rows.  Workgroup tile: 128 output channels x 64
// consecutive pixels of the flattened (image, y, x) axis; K runs over taps x channel chunks of 32.  Per chunk the next
// A / B chunks are fetched into registers while the current one is multiplied from LDS (v_mfma_f32_32x32x2_f32: exact
// fp32, a k-ordered fmaf chain); a wave owns 32 rows x 64 pixels = two accumulator tiles, so one A read and two B reads
// feed two matrix instructions.  LDS strides (33 / 96 floats) keep both operand reads conflict free.
// Epilogue: per-channel scale / shift (folded BatchNorm, conv bias folded into shift by the caller) + PReLU (alpha = 0
// gives the reference's ReLU), channel-slice destination (the concatenation of aspp.py:48 is never materialised twice).
#include <stdlib.h>

#include "common.hpp"

namespace mspl {

typedef float floatx16c __attribute__((ext_vector_type(16)));

struct DenseConvGeom {
    int N, Cin, Cout, H, W, taps, dil;      // taps = 1 or 9
    int mblocks, ptiles;
};

constexpr int DC_BM = 128, DC_BN = 64, DC_KC = 32;
constexpr int DC_AS = DC_KC + 1;            // A row stride (floats)
constexpr int DC_BS = 96;                   // B row stride (floats): rows k and k+1 land 32 banks apart

__global__ __launch_bounds__(256, 2) void dense_conv_mfma_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                                 DenseConvGeom g, Epi e, float* __restrict__ out) {
    __shared__ float As[DC_BM * DC_AS];
    __shared__ float Bs[DC_KC * DC_BS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, half = lane >> 5;
    const int mb = blockIdx.x % g.mblocks, pt = blockIdx.x / g.mblocks;
    const int m0 = mb * DC_BM;
    const int HW = g.H * g.W;
    const int64_t P = (int64_t)g.N * HW;

    // B loader role: this thread always fetches pixel column bj of channel rows bk0 + 4*u (u = 0..7)
    const int bj = tid & 63, bk0 = tid >> 6;
    const int64_t bp = (int64_t)pt * DC_BN + bj;
    const bool bpok = bp < P;
    const int bn = bpok ? (int)(bp / HW) : 0;
    const int brem = bpok ? (int)(bp - (int64_t)bn * HW) : 0;
    const int by = brem / g.W, bx = brem - by * g.W;
    const float* xn = x + (size_t)bn * g.Cin * HW;
    // A loader role: rows ar0 + 32*u (u = 0..3), 16-byte column ac of the 32-channel chunk
    const int ac = (tid & 7) * 4, ar0 = tid >> 3;

    floatx16c acc[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;

    const int nchunk = g.Cin / DC_KC;                         // Cin % 32 == 0 (launcher)
    const int nstage = g.taps * nchunk;
    float4 areg[4];
    float breg[8];
    auto fetch = [&](int stage) {
        const int tap = stage / nchunk, ci0 = (stage - tap * nchunk) * DC_KC;
        const float* wt = wp + ((size_t)tap * g.Cout + m0) * g.Cin + ci0 + ac;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = ar0 + 32 * u;
            areg[u] = (m0 + r < g.Cout) ? *reinterpret_cast<const float4*>(wt + (size_t)r * g.Cin) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        int dy = 0, dx = 0;
        if (g.taps == 9) { dy = (tap / 3 - 1) * g.dil; dx = (tap % 3 - 1) * g.dil; }
        const int yy = by + dy, xx = bx + dx;
        const bool ok = bpok && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
        const float* src = xn + (size_t)(ci0 + bk0) * HW + (ok ? yy * g.W + xx : 0);
#pragma unroll
        for (int u = 0; u < 8; ++u) breg[u] = ok ? src[(size_t)(4 * u) * HW] : 0.f;
    };
    auto stash = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float* d = As + (ar0 + 32 * u) * DC_AS + ac;
            d[0] = areg[u].x; d[1] = areg[u].y; d[2] = areg[u].z; d[3] = areg[u].w;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) Bs[(bk0 + 4 * u) * DC_BS + bj] = breg[u];
    };

    fetch(0);
    stash();
    __syncthreads();
    const float* ap = As + (wave * 32 + li) * DC_AS + half;
    const float* bpp = Bs + half * DC_BS + li;
    for (int stage = 0; stage < nstage; ++stage) {
        if (stage + 1 < nstage) fetch(stage + 1);            // global -> registers while this chunk is multiplied
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < DC_KC / 2; ++ks) {
            const float a = ap[2 * ks];
            const float b0 = bpp[(2 * ks) * DC_BS], b1 = bpp[(2 * ks) * DC_BS + 32];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
        }
        __syncthreads();
        if (stage + 1 < nstage) stash();
        __syncthreads();
    }

    // epilogue: lane holds pixel column li (+32 for the second tile), rows (r & 3) + 8 * (r >> 2) + 4 * half
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int64_t p = (int64_t)pt * DC_BN + s * 32 + li;
        if (p >= P) continue;
        const int n = (int)(p / HW);
        const int rem = (int)(p - (int64_t)n * HW);
        float* ob = out + ((size_t)n * e.ctot + e.coff) * (size_t)HW + rem;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (co < g.Cout) {
                const int cabs = e.coff + co;
                float v = acc[s][r];
                v = fmaf(v, e.scale ? e.scale[cabs] : 1.f, e.shift ? e.shift[cabs] : 0.f);
                if (e.alpha) v = v > 0.f ? v : e.alpha[cabs] * v;
                ob[(size_t)co * HW] = v;
            }
        }
    }
}

// Wide form for large pixel counts: workgroup tile 128 output channels x 128 pixels, waves 2 x 2, a wave owns 64 rows x
// 64 pixels = four accumulator tiles, so two A reads and two B reads feed four matrix instructions and a chunk carries
// 64 MFMAs per wave between its barriers (half the barrier / LDS overhead per flop of the 64-pixel form).
constexpr int DC_BN2 = 128;
constexpr int DC_BS2 = 160;                 // B row stride: rows k and k+1 again 32 banks apart

__global__ __launch_bounds__(256, 2) void dense_conv_mfma_wide_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                                      DenseConvGeom g, Epi e, float* __restrict__ out) {
    __shared__ float As[DC_BM * DC_AS];
    __shared__ float Bs[DC_KC * DC_BS2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, half = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;                  // wave's 64-row / 64-pixel quadrant
    const int mb = blockIdx.x % g.mblocks, pt = blockIdx.x / g.mblocks;
    const int m0 = mb * DC_BM;
    const int HW = g.H * g.W;
    const int64_t P = (int64_t)g.N * HW;

    const int bj = tid & 127, bk0 = tid >> 7;                 // B loader: pixel column bj, channel rows bk0 + 2*u (u = 0..15)
    const int64_t bp = (int64_t)pt * DC_BN2 + bj;
    const bool bpok = bp < P;
    const int bn = bpok ? (int)(bp / HW) : 0;
    const int brem = bpok ? (int)(bp - (int64_t)bn * HW) : 0;
    const int by = brem / g.W, bx = brem - by * g.W;
    const float* xn = x + (size_t)bn * g.Cin * HW;
    const int ac = (tid & 7) * 4, ar0 = tid >> 3;             // A loader: rows ar0 + 32*u, 16-byte column ac

    floatx16c acc[2][2];
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a2][b2][r] = 0.f;

    const int nchunk = g.Cin / DC_KC;
    const int nstage = g.taps * nchunk;
    float4 areg[4];
    float breg[16];
    auto fetch = [&](int stage) {
        const int tap = stage / nchunk, ci0 = (stage - tap * nchunk) * DC_KC;
        const float* wt = wp + ((size_t)tap * g.Cout + m0) * g.Cin + ci0 + ac;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = ar0 + 32 * u;
            areg[u] = (m0 + r < g.Cout) ? *reinterpret_cast<const float4*>(wt + (size_t)r * g.Cin) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        int dy = 0, dx = 0;
        if (g.taps == 9) { dy = (tap / 3 - 1) * g.dil; dx = (tap % 3 - 1) * g.dil; }
        const int yy = by + dy, xx = bx + dx;
        const bool ok = bpok && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
        const float* src = xn + (size_t)(ci0 + bk0) * HW + (ok ? yy * g.W + xx : 0);
#pragma unroll
        for (int u = 0; u < 16; ++u) breg[u] = ok ? src[(size_t)(2 * u) * HW] : 0.f;
    };
    auto stash = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float* d = As + (ar0 + 32 * u) * DC_AS + ac;
            d[0] = areg[u].x; d[1] = areg[u].y; d[2] = areg[u].z; d[3] = areg[u].w;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) Bs[(bk0 + 2 * u) * DC_BS2 + bj] = breg[u];
    };

    fetch(0);
    stash();
    __syncthreads();
    const float* ap = As + (wm * 64 + li) * DC_AS + half;
    const float* bpp = Bs + half * DC_BS2 + wn * 64 + li;
    for (int stage = 0; stage < nstage; ++stage) {
        if (stage + 1 < nstage) fetch(stage + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < DC_KC / 2; ++ks) {
            const float a0 = ap[2 * ks], a1 = ap[32 * DC_AS + 2 * ks];
            const float b0 = bpp[(2 * ks) * DC_BS2], b1 = bpp[(2 * ks) * DC_BS2 + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
        if (stage + 1 < nstage) stash();
        __syncthreads();
    }

#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2) {
        const int64_t p = (int64_t)pt * DC_BN2 + wn * 64 + b2 * 32 + li;
        if (p >= P) continue;
        const int n = (int)(p / HW);
        const int rem = (int)(p - (int64_t)n * HW);
        float* ob = out + ((size_t)n * e.ctot + e.coff) * (size_t)HW + rem;
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + wm * 64 + a2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (co < g.Cout) {
                    const int cabs = e.coff + co;
                    float v = acc[a2][b2][r];
                    v = fmaf(v, e.scale ? e.scale[cabs] : 1.f, e.shift ? e.shift[cabs] : 0.f);
                    if (e.alpha) v = v > 0.f ? v : e.alpha[cabs] * v;
                    ob[(size_t)co * HW] = v;
                }
            }
    }
}

}  // namespace mspl

using namespace mspl;

extern "C" int mspl_dense_conv_fwd(const float* x, const float* w_packed, int32_t N, int32_t Cin, int32_t Cout, int32_t H,
                                   int32_t W, int32_t ksize, int32_t dilation, const mspl_epilogue_t* ep, float* out,
                                   void* stream) {
    MSPL_REQUIRE(x && w_packed && out, MSPL_ERR_NULL_POINTER, "dense_conv: null pointer");
    MSPL_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, MSPL_ERR_BAD_SHAPE,
                 "dense_conv: bad shape N=%d Cin=%d Cout=%d %dx%d", N, Cin, Cout, H, W);
    MSPL_REQUIRE(ksize == 1 || ksize == 3, MSPL_ERR_UNSUPPORTED, "dense_conv: kernel size %d (1 or 3)", ksize);
    MSPL_REQUIRE(dilation >= 1, MSPL_ERR_BAD_SHAPE, "dense_conv: dilation %d", dilation);
    MSPL_REQUIRE(Cin % DC_KC == 0, MSPL_ERR_UNSUPPORTED, "dense_conv: Cin=%d is not a multiple of %d", Cin, DC_KC);
    MSPL_REQUIRE((((uintptr_t)w_packed) & 15) == 0, MSPL_ERR_UNSUPPORTED, "dense_conv: packed weights must be 16-byte aligned");
    if (int rc = check_epi(ep, Cout, "dense_conv")) return rc;
    MSPL_REQUIRE(!ep || (!ep->pre_add && !ep->residual && !ep->reinf_r && !ep->gate), MSPL_ERR_UNSUPPORTED,
                 "dense_conv: only scale/shift/alpha epilogue terms are supported");
    const Epi e = make_epi(ep, Cout, H * W);
    DenseConvGeom g;
    g.N = N; g.Cin = Cin; g.Cout = Cout; g.H = H; g.W = W; g.taps = ksize * ksize; g.dil = dilation;
    g.mblocks = ceil_div(Cout, DC_BM);
    const int64_t P = (int64_t)N * H * W;
    static const int dbg_wide = MSPL_TUNE_INT("MSPL_DC_WIDE", -1);
    const bool wide = dbg_wide >= 0 ? dbg_wide != 0 : (ceil_div64(P, DC_BN2) * g.mblocks >= 512);     // enough 128-pixel tiles to fill the chip
    const int64_t ptiles = ceil_div64(P, wide ? DC_BN2 : DC_BN);
    MSPL_REQUIRE(ptiles * g.mblocks < (1ll << 31), MSPL_ERR_BAD_SHAPE, "dense_conv: grid too large");
    g.ptiles = (int)ptiles;
    if (wide)
        hipLaunchKernelGGL(dense_conv_mfma_wide_kernel, dim3((unsigned)(ptiles * g.mblocks)), dim3(256), 0, (hipStream_t)stream, x,
                           w_packed, g, e, out);
    else
        hipLaunchKernelGGL(dense_conv_mfma_kernel, dim3((unsigned)(ptiles * g.mblocks)), dim3(256), 0, (hipStream_t)stream, x, w_packed,
                           g, e, out);
    MSPL_CHECK_LAUNCH("dense_conv");
    return MSPL_OK;
}
