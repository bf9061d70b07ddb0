// 1x1 convolution with FEW output channels and a deep reduction (M <= 32, K >= 4M: the pyramid projections 512->16, 64->16) as a
// split-K matrix-core kernel.
// With one 32-row MFMA tile per group the tile-pipelined kernel (conv1x1.hip) has only groups x pixels/32 waves to offer -- 0.5-1
// wave per SIMD at the 18x30 and 36x60 levels, each walking the whole K behind a 4-deep ring: a chain of memory latencies with
// nothing to hide them.  Here the four waves of a workgroup share ONE pixel tile and split K four ways: 4x the waves, a quarter
// of the chain each, all operand loads of a 32-k chunk requested up front, partial tiles summed through LDS, and each wave
// finishes 8 rows of the tile (BN scale/shift + PReLU epilogue, coalesced stores).
//   lane (r = lane & 31, h = lane >> 5):  A element = w[grp*M + r][k + h],  B element = x[grp*K + k + h][pixel r (+32 per sub-tile)]
#include <stdlib.h>

#include "common.hpp"

namespace mspl {

typedef float f32x16s __attribute__((ext_vector_type(16)));

struct SkGeom {
    int N, G, M, K, HW;      // images, groups, cout_g (<= 32), cin_g (% 8 == 0), pixels per plane
    int tiles;               // pixel tiles per image (32 * NSUB pixels each)
    int KS;                  // k per wave = K / 4
};

template <int NSUB, int CH>   // CH = MFMA steps (2 k each) per chunk whose loads are issued together
__global__ __launch_bounds__(256) void conv1x1_splitk_kernel(const float* __restrict__ x, const float* __restrict__ w, SkGeom g, Epi e,
                                                             float* __restrict__ out) {
    __shared__ float red[4][NSUB][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    int b = blockIdx.x;
    const int tile = b % g.tiles;  b /= g.tiles;
    const int grp = b % g.G;
    const int n = b / g.G;
    const int p0 = tile * 32 * NSUB;
    const bool arow = r < g.M;
    // A: row r of this group's weights (rows beyond M read row 0 and are zeroed); B: pixel p0 + r + 32*s (clamped inside the plane)
    const float* wp = w + ((size_t)grp * g.M + (arow ? r : 0)) * g.K + wave * g.KS + h;
    const float* xp = x + ((size_t)n * g.G * g.K + (size_t)grp * g.K + wave * g.KS + h) * (size_t)g.HW;
    int px[NSUB];
#pragma unroll
    for (int s = 0; s < NSUB; ++s) px[s] = min(p0 + r + 32 * s, g.HW - 1);
    f32x16s acc[NSUB];
#pragma unroll
    for (int s = 0; s < NSUB; ++s)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[s][i] = 0.f;
    for (int k0 = 0; k0 < g.KS; k0 += 2 * CH) {                       // KS % (2*CH) == 0 (launcher)
        float a[CH], bv[CH][NSUB];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            a[i] = wp[k0 + 2 * i];
#pragma unroll
            for (int s = 0; s < NSUB; ++s) bv[i][s] = xp[(size_t)(k0 + 2 * i) * g.HW + px[s]];
        }
        __builtin_amdgcn_sched_barrier(0);                            // every load of the chunk leaves before the first MFMA
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const float av = arow ? a[i] : 0.f;
#pragma unroll
            for (int s = 0; s < NSUB; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[i][s], acc[s], 0, 0, 0);
        }
    }
#pragma unroll
    for (int s = 0; s < NSUB; ++s)
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wave][s][i][lane] = acc[s][i];
    __syncthreads();
    // wave w finishes registers 4w .. 4w+3 of every sub-tile: rows 8w + (i & 3) + 4h, column r
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
        const int p = p0 + r + 32 * s;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = 4 * wave + q;
            const float v = (red[0][s][i][lane] + red[1][s][i][lane]) + (red[2][s][i][lane] + red[3][s][i][lane]);
            const int row = 8 * wave + q + 4 * h;
            if (row < g.M && p < g.HW) {
                const int cabs = e.coff + grp * g.M + row;
                float t = v;
                if (e.raw) e.raw[((size_t)n * e.ctot + cabs) * (size_t)g.HW + p] = v;
                t = fmaf(t, e.scale ? e.scale[cabs] : 1.f, e.shift ? e.shift[cabs] : 0.f);
                if (e.alpha) t = t > 0.f ? t : e.alpha[cabs] * t;
                out[((size_t)n * e.ctot + cabs) * (size_t)g.HW + p] = t;
            }
        }
    }
}

// Returns 1 when the shape / epilogue is not this kernel's (the caller continues with the tile-pipelined kernel).
int conv1x1_splitk_try(const float* x, const float* w, int N, int Cin, int Cout, int groups, int HW, const Epi& e, float* out,
                       hipStream_t s) {
    static const int enabled = MSPL_TUNE_INT("MSPL_PW_SPLITK", 1);
    const int M = Cout / groups, K = Cin / groups;
    if (!enabled || M > 32 || K < 64 || (K & 31) != 0) return 1;          // K/4 per wave in chunks of 8 or 16 MFMA steps
    // measured (tools/bench_ops.py conv1x1): wins for the ungrouped deep-and-narrow projections of the pyramid (512->16: 22.1 ->
    // 12.6 us, 64->16: 8.2 -> 6.8 us); loses to the tile-pipelined kernel on the grouped EESP projections (512->128 g4: 11.7 vs 14.3 us)
    if (groups != 1 || K < 4 * M) return 1;
    if (e.pre_add || e.residual || e.reinf_r || e.gate) return 1;
    // worth it where the tile-pipelined kernel starves: fewer than ~2 waves per SIMD there
    const int64_t pipe_waves = (int64_t)N * groups * ceil_div(HW, 32);
    if (pipe_waves > 6000) return 1;
    SkGeom g;
    g.N = N; g.G = groups; g.M = M; g.K = K; g.HW = HW; g.KS = K / 4;
    const int nsub = (int64_t)N * groups * ceil_div(HW, 64) >= 1024 ? 2 : 1;
    g.tiles = ceil_div(HW, 32 * nsub);
    const int64_t blocks = (int64_t)N * groups * g.tiles;
    if (blocks >= (1ll << 31)) return 1;
    const dim3 grid((unsigned)blocks), blk(256);
    const bool ch16 = (g.KS % 32) == 0;                                   // 16 steps = 32 k per chunk when the slice allows
    if (nsub == 2) {
        if (ch16) hipLaunchKernelGGL((conv1x1_splitk_kernel<2, 16>), grid, blk, 0, s, x, w, g, e, out);
        else hipLaunchKernelGGL((conv1x1_splitk_kernel<2, 8>), grid, blk, 0, s, x, w, g, e, out);
    } else {
        if (ch16) hipLaunchKernelGGL((conv1x1_splitk_kernel<1, 16>), grid, blk, 0, s, x, w, g, e, out);
        else hipLaunchKernelGGL((conv1x1_splitk_kernel<1, 8>), grid, blk, 0, s, x, w, g, e, out);
    }
    return 0;
}

}  // namespace mspl
