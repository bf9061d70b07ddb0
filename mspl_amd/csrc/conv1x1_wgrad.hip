// Weight gradient of the grouped 1x1 convolutions (K1 / K3 / decoder projections) on the matrix cores:
//     gw[g, m, k] += sum_{n,p} gy[n, g*M + m, p] * x[n, g*K + k, p]            (an "NT" GEMM whose reduction runs over pixels)
// One wave owns one 32x32 tile of gw for one slice of the (image, pixel) range.  v_mfma_f32_32x32x2_f32 takes A[row][k] from
// lane (row, k = lane>>5) and B[k][col] from lane (col, k = lane>>5): with lane (r, h) loading the 16 bytes at pixel
// p0 + 4h of row r of BOTH operands, element j of the two float4 is a valid (A, B) pair for one MFMA -- no LDS, no transposes,
// eight pixels per iteration, every byte loaded is used.  Partial tiles are combined with float atomics (gw zeroed by the
// launcher, or the parameter's gradient buffer when accumulating).
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "common.hpp"

namespace mspl {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgG {
    int N, G, M, K, P;          // images, groups, cout_g, cin_g, pixels per plane
    int tm, tk;                 // 32-tiles along M and K
    int gpi;                    // 64-pixel groups per image
    int ngroups;                // N * gpi: the reduction range a tile's waves share
    int nslots;                 // waves per tile (multiple of 4: one workgroup = 4 slots of one tile)
};

// One wave: the 64-pixel groups slot, slot + nslots, ... of the flattened (image, pixel group) range of its tile -- several
// images per wave.  (Rounds 1-2: a wave saw ONE image, so the 18x30 planes of level 4 gave every wave a single group of 32 MFMAs
// and every workgroup left with 1024 atomics: 2 M device atomics per launch; now a tile's waves number what fills the chip, not
// what the plane size dictates: 740 -> 590 us per train step over the 33 launches.)  All 16 float4 of a group (8 per operand) are
// requested before the first MFMA, so a wave exposes ONE memory latency per 32 MFMAs.  (A two-stage software pipeline over the
// groups -- next group's requests before this group's MFMAs -- measured no faster: 24 us either way for the 512 -> 512 expansion
// at 18x30, whose fp32 MFMA floor is 6.5 us.)  The four waves of a workgroup hold four slices of the same tile: they are summed
// through LDS and leave with one atomic per tile element.
// rowscale (or null): gw[m, k] += rowscale[m] * sum -- the caller's gy is then the gradient BEFORE a per-output-channel scale (the folded
// BatchNorm scale of a convolution whose pointwise backward kept only the unscaled gradient in memory)
__device__ __forceinline__ void wgrad_tile_body(const float* __restrict__ gy, const float* __restrict__ x, const WgG& g,
                                                float* __restrict__ gw, const int64_t bid, float (&red)[4][16][64],
                                                const float* __restrict__ rowscale = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int bslots = g.nslots >> 2;                                   // workgroups per tile
    const int64_t tile = bid / bslots;
    const int slot = (int)(bid - tile * bslots) * 4 + wave;
    const int grp = (int)(tile / (g.tm * g.tk));
    const int tt = (int)(tile - (int64_t)grp * g.tm * g.tk);
    const int m0 = (tt / g.tk) * 32, k0 = (tt % g.tk) * 32;
    const bool am = m0 + r < g.M, bk = k0 + r < g.K;
    // rows outside the tile read row 0 of the tile (always valid); their products land in elements that are never stored
    const float* ap0 = gy + ((size_t)grp * g.M + m0 + (am ? r : 0)) * (size_t)g.P;
    const float* bp0 = x + ((size_t)grp * g.K + k0 + (bk ? r : 0)) * (size_t)g.P;
    const size_t aimg = (size_t)g.G * g.M * g.P, bimg = (size_t)g.G * g.K * g.P;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q = slot; q < g.ngroups; q += g.nslots) {                  // wave-uniform
        const int n = q / g.gpi, p0 = (q - n * g.gpi) * 64;
        const float* ap = ap0 + (size_t)n * aimg;
        const float* bp = bp0 + (size_t)n * bimg;
        float4 a[8], b[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pc = min(p0 + 8 * i + 4 * h, g.P - 4);            // clamped, unconditional: P % 4 == 0, so a float4 is
            a[i] = *reinterpret_cast<const float4*>(ap + pc);           // entirely inside or outside the plane
            b[i] = *reinterpret_cast<const float4*>(bp + pc);
        }
        // all sixteen requests leave before anything consumes them (hipcc otherwise sinks each load to its first use and the
        // loop degenerates into load -> wait -> mfma)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            // rows / columns outside the tile are discarded at the store; only pixels past the plane must contribute zero
            if (p0 + 8 * i + 4 * h >= g.P) { a[i] = zero; b[i] = zero; }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[i].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[i].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[i].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[i].w, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) red[wave][i][lane] = acc[i];
    __syncthreads();
    // 1024 tile elements, 4 per thread.  Element (i, l): row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5), column = l & 31
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = threadIdx.x + 256 * j;
        const int i = e >> 6, l = e & 63;
        const float v = (red[0][i][l] + red[1][i][l]) + (red[2][i][l] + red[3][i][l]);
        const int row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5), col = l & 31;
        if (m0 + row < g.M && k0 + col < g.K)
            atomicAdd(gw + ((size_t)grp * g.M + m0 + row) * g.K + k0 + col, rowscale ? v * rowscale[grp * g.M + m0 + row] : v);
    }
}

// ---- the same tile through LDS in full 128-byte lines (round 5).  The body above loads its MFMA fragments straight from memory: lane
// (r, h) takes 16 bytes of row r, so ONE load instruction touches 32 rows x 32 bytes -- a quarter of every 128-byte line it opens, and
// four instructions later the line has to be found again.  With loads only (no MFMA, no atomics) the 512 -> 512 expansion in 4 groups at
// 16 x 18x30 (31.5 MB of operands) took 26 us, 1.2 TB/s.  Here a wave fills a private LDS stage with LDS-DMA pieces of 8 rows x 128 bytes
// (global_load_lds_dwordx4: lane l of a piece reads 16 bytes of row l >> 3; destination = piece base + 16 l, so the XOR swizzle that
// keeps the fragment reads conflict-free is applied to the SOURCE chunk: chunk (l & 7) ^ (l >> 3)), two stages of 32 pixels in flight,
// and reads its fragments with ds_read_b128 at the swizzled slot.  Wave-private stages: no barrier in the loop, the wave waits on its
// own vmcnt.  The four waves still hold four pixel slots of one tile and meet in LDS at the end.  (Seating the four waves on the 2x2
// neighbouring tiles of one pixel slot instead -- shared operand rows through L1 -- was neutral on the fragment-load body: 6.17 vs 6.20 ms.)
// Level-4 run of 16 problems (413 MB): 223 -> 158 us; graphed train step 6.12 -> 6.04 ms, same box.
constexpr int WGL_STAGE = 2048;                  // floats per stage and wave: 32 rows x 32 pixels of each operand (2 x 4 pieces of 1 KiB)
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ void wgrad_tile_body_lds(const float* __restrict__ gy, const float* __restrict__ x, const WgG& g,
                                                    float* __restrict__ gw, const int64_t bid, float* smem,
                                                    const float* __restrict__ rowscale) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), r = lane & 31, h = lane >> 5;
    const int bslots = g.nslots >> 2;
    const int64_t tile = bid / bslots;
    const int slot = (int)(bid - tile * bslots) * 4 + wave;
    const int grp = (int)(tile / (g.tm * g.tk));
    const int tt = (int)(tile - (int64_t)grp * g.tm * g.tk);
    const int m0 = (tt / g.tk) * 32, k0 = (tt % g.tk) * 32;
    float* wbuf = smem + wave * 2 * WGL_STAGE;
    // fill side: piece j of an operand = rows 8 j .. 8 j + 7; this lane's row inside a piece and its (swizzled) 16-byte chunk
    const int lr = lane >> 3, ck = (lane & 7) ^ lr;
    const float* arow[4];
    const float* brow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {                // rows outside the tile read its first row; their products are never stored
        const int ra = m0 + 8 * j + lr, rb = k0 + 8 * j + lr;
        arow[j] = gy + ((size_t)grp * g.M + (ra < g.M ? ra : m0)) * (size_t)g.P;
        brow[j] = x + ((size_t)grp * g.K + (rb < g.K ? rb : k0)) * (size_t)g.P;
    }
    const size_t aimg = (size_t)g.G * g.M * g.P, bimg = (size_t)g.G * g.K * g.P;
    // read side: chunk c = 2 i + h of row r sits at slot (r & 7) * 8 + (c ^ (r & 7)) of piece r >> 3
    const int rbase = (r >> 3) * 256 + (r & 7) * 32, rx = r & 7;
    const int nq = (g.ngroups - slot + g.nslots - 1) / g.nslots;        // 64-pixel groups of this wave (slot < nslots <= ngroups)
    const int S = 2 * nq;                                               // stages of 32 pixels
    auto issue = [&](int s) {
        const int q = slot + (s >> 1) * g.nslots;
        const int n = q / g.gpi, p0 = (q - n * g.gpi) * 64 + 32 * (s & 1);
        const int pc = min(p0 + 4 * ck, g.P - 4);                       // clamped: a chunk is entirely inside or outside the plane
        float* dst = wbuf + (s & 1) * WGL_STAGE;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_global_load_lds(arow[j] + (size_t)n * aimg + pc, (lds_ptr_t)(dst + j * 256), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(brow[j] + (size_t)n * bimg + pc, (lds_ptr_t)(dst + 1024 + j * 256), 16, 0, 0);
        }
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    if (S > 0) issue(0);                         // (a slot past the last pixel group has nothing to load: nslots is rounded up to 4)
#pragma unroll 1
    for (int s = 0; s < S; ++s) {
        if (s + 1 < S) {
            issue(s + 1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");            // stage s has landed; the 8 pieces of stage s + 1 stay in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const int q = slot + (s >> 1) * g.nslots;
        const int p0 = (q % g.gpi) * 64 + 32 * (s & 1);
        const float* src = wbuf + (s & 1) * WGL_STAGE + rbase;
        float4 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = 2 * i + h;
            a[i] = *reinterpret_cast<const float4*>(src + ((c ^ rx) << 2));
            b[i] = *reinterpret_cast<const float4*>(src + 1024 + ((c ^ rx) << 2));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (p0 + 8 * i + 4 * h >= g.P) { a[i] = zero; b[i] = zero; }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[i].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[i].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[i].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[i].w, acc, 0, 0, 0);
        }
    }
    __syncthreads();                             // every wave is done with its stages: the first 16 KiB become the reduction buffer
    float (*red)[16][64] = reinterpret_cast<float (*)[16][64]>(smem);
#pragma unroll
    for (int i = 0; i < 16; ++i) red[wave][i][lane] = acc[i];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = threadIdx.x + 256 * j;
        const int i = e >> 6, l = e & 63;
        const float v = (red[0][i][l] + red[1][i][l]) + (red[2][i][l] + red[3][i][l]);
        const int row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5), col = l & 31;
        if (m0 + row < g.M && k0 + col < g.K)
            atomicAdd(gw + ((size_t)grp * g.M + m0 + row) * g.K + k0 + col, rowscale ? v * rowscale[grp * g.M + m0 + row] : v);
    }
}

__global__ __launch_bounds__(256) void conv1x1_wgrad_mfma_kernel(const float* __restrict__ gy, const float* __restrict__ x, WgG g,
                                                                 float* __restrict__ gw) {
    __shared__ float red[4][16][64];
    wgrad_tile_body(gy, x, g, gw, (int64_t)blockIdx.x, red);
}

__global__ __launch_bounds__(256) void conv1x1_wgrad_mfma_lds_kernel(const float* __restrict__ gy, const float* __restrict__ x, WgG g,
                                                                     float* __restrict__ gw) {
    __shared__ __attribute__((aligned(16))) float smem[4 * 2 * WGL_STAGE];
    wgrad_tile_body_lds(gy, x, g, gw, (int64_t)blockIdx.x, smem, nullptr);
}

// Several weight-gradient problems in ONE launch (mspl_conv1x1_wgrad_batch): the grouped 1x1 convolutions of a training step are
// ~33 launches of 8-25 us for <= 7 us of matrix work each, and nothing downstream waits for any of them -- the autograd nodes queue
// them (autograd.WgradQueue) and the queue goes out as a few launches whose grids are the problems' grids back to back.
constexpr int WG_MAXP = 16;
struct WgProb {
    const float* gy;
    const float* x;
    float* gw;
    const float* rowscale;       // per output channel, or null
    WgG g;
    unsigned first;              // first workgroup of this problem
};
struct WgBatch {
    WgProb p[WG_MAXP];
    int n;
};

__global__ __launch_bounds__(256) void conv1x1_wgrad_batch_lds_kernel(WgBatch b) {
    __shared__ __attribute__((aligned(16))) float smem[4 * 2 * WGL_STAGE];          // 64 KiB: two workgroups per CU
    int k = 0;
#pragma unroll 1
    for (int i = 1; i < b.n; ++i)
        if (blockIdx.x >= b.p[i].first) k = i;                          // uniform
    const WgProb& q = b.p[k];
    wgrad_tile_body_lds(q.gy, q.x, q.g, q.gw, (int64_t)(blockIdx.x - q.first), smem, q.rowscale);
}

__global__ __launch_bounds__(256) void conv1x1_wgrad_batch_kernel(WgBatch b) {
    __shared__ float red[4][16][64];
    int k = 0;
#pragma unroll 1
    for (int i = 1; i < b.n; ++i)
        if (blockIdx.x >= b.p[i].first) k = i;                          // uniform
    const WgProb& q = b.p[k];
    wgrad_tile_body(q.gy, q.x, q.g, q.gw, (int64_t)(blockIdx.x - q.first), red, q.rowscale);
}

static bool wgrad_geom(int N, int G, int M, int K, int P, WgG& g, int64_t& blocks, int share) {
    if (M < 8 || K < 8 || (P & 3) != 0) return false;
    g.N = N; g.G = G; g.M = M; g.K = K; g.P = P;
    g.tm = ceil_div(M, 32); g.tk = ceil_div(K, 32);
    const int64_t tiles = (int64_t)G * g.tm * g.tk;
    g.gpi = ceil_div(P, 64);
    if ((int64_t)N * g.gpi >= (1ll << 30)) return false;
    g.ngroups = N * g.gpi;
    // (in a batch the chip is shared: `share` problems aim at 3072 waves together, each at least at a quarter of that)
    const int target = std::max(768, 3072 / std::max(1, share));
    int64_t ns = (target + tiles - 1) / tiles;
    if (ns < 4) ns = 4;
    if (ns > g.ngroups) ns = g.ngroups;
    g.nslots = (int)((ns + 3) & ~3ll);
    blocks = tiles * (g.nslots >> 2);
    return blocks < (1ll << 30);
}

// Returns the number of problems launched from the front of the list (0: the first problem is not eligible for this kernel).
int conv1x1_wgrad_mfma_batch(const float* const* gy, const float* const* x, float* const* gw, const float* const* rowscale, const int* N,
                             const int* G, const int* M, const int* K, const int* P, int nprob, hipStream_t s) {
    WgBatch b;
    memset(&b, 0, sizeof(b));
    int64_t total = 0;
    int n = 0;
    const int share = std::min(nprob, 4);
    for (; n < nprob && n < WG_MAXP; ++n) {
        int64_t blocks;
        if (!wgrad_geom(N[n], G[n], M[n], K[n], P[n], b.p[n].g, blocks, share)) break;
        if (total + blocks >= (1ll << 31)) break;
        b.p[n].gy = gy[n]; b.p[n].x = x[n]; b.p[n].gw = gw[n]; b.p[n].rowscale = rowscale ? rowscale[n] : nullptr;
        b.p[n].first = (unsigned)total;
        total += blocks;
    }
    if (n == 0) return 0;
    b.n = n;
    static const int lds_on = MSPL_TUNE_INT("MSPL_WGRAD_GLDS", 1);
    if (lds_on) hipLaunchKernelGGL(conv1x1_wgrad_batch_lds_kernel, dim3((unsigned)total), dim3(256), 0, s, b);
    else hipLaunchKernelGGL(conv1x1_wgrad_batch_kernel, dim3((unsigned)total), dim3(256), 0, s, b);
    return n;
}

// Called by mspl_conv_bwd_weight for K == 1 (gw already zeroed unless accumulating).  Returns 1 when the shape is left to the
// 16x16 LDS kernel (tiny groups such as the 3-channel image reinforcement, planes that are not a multiple of 4 pixels).
int conv1x1_wgrad_mfma_try(const float* gy, const float* x, int N, int G, int M, int K, int P, float* gw, hipStream_t s) {
    if (M < 8 || K < 8 || (P & 3) != 0) return 1;
    WgG g;
    g.N = N; g.G = G; g.M = M; g.K = K; g.P = P;
    g.tm = ceil_div(M, 32); g.tk = ceil_div(K, 32);
    const int64_t tiles = (int64_t)G * g.tm * g.tk;
    g.gpi = ceil_div(P, 64);
    if ((int64_t)N * g.gpi >= (1ll << 30)) return 1;
    g.ngroups = N * g.gpi;
    // waves per tile: enough to put ~3 waves on every SIMD (MSPL_WGRAD_WAVES, 3072), at least MSPL_WGRAD_MINREP pixel groups per
    // wave when the range allows (each workgroup leaves with 1024 atomics), never more waves than groups
    static const int target = MSPL_TUNE_INT("MSPL_WGRAD_WAVES", 3072);
    static const int minrep = MSPL_TUNE_INT("MSPL_WGRAD_MINREP", 1);
    int64_t ns = (target + tiles - 1) / tiles;
    if (ns * minrep > g.ngroups) ns = g.ngroups / (minrep > 0 ? minrep : 1);
    if (ns < 4) ns = 4;
    if (ns > g.ngroups) ns = g.ngroups;
    g.nslots = (int)((ns + 3) & ~3ll);
    const int64_t blocks = tiles * (g.nslots >> 2);
    if (blocks >= (1ll << 31)) return 1;
    static const int lds_on = MSPL_TUNE_INT("MSPL_WGRAD_GLDS", 1);
    if (lds_on) hipLaunchKernelGGL(conv1x1_wgrad_mfma_lds_kernel, dim3((unsigned)blocks), dim3(256), 0, s, gy, x, g, gw);
    else hipLaunchKernelGGL(conv1x1_wgrad_mfma_kernel, dim3((unsigned)blocks), dim3(256), 0, s, gy, x, g, gw);
    return 0;
}

}  // namespace mspl
