// Backward kernels, the fused uncertainty-weighted loss and Adam for one uest self-training step
// (uest_seg_multi_os.py:958-1089: forward -> PixelwiseKLD -> UncertaintyWeightedSegmentationLoss*20 + kld.mean()
//  -> backward -> torch.optim.Adam), BatchNorm frozen (eval mode, Appendix B-3 of SURVEY.md).
//
// Round-1 form: generic, direct and correct (gather-style data gradients, block-reduced weight gradients,
// atomic scatter for the resampling ops).  They are native HIP behind the same C ABI; the specialised MFMA /
// LDS-tiled versions of the hot ones (1x1 wgrad/dgrad, depthwise dilated dgrad) are the next optimisation step.
#include <stdlib.h>

#include <algorithm>

#include "common.hpp"

namespace mspl {

int conv1x1_wgrad_mfma_try(const float* gy, const float* x, int N, int G, int M, int K, int P, float* gw, hipStream_t s);
int conv1x1_wgrad_mfma_batch(const float* const* gy, const float* const* x, float* const* gw, const float* const* rowscale, const int* N,
                             const int* G, const int* M, const int* K, const int* P, int nprob, hipStream_t s);

struct ConvGeom {
    int N, Cin, Cout, G, cin_g, cout_g, H, W, Ho, Wo, K, stride, dil, pad;
};

// ---- data gradient: gx[n,ci,iy,ix] = sum_{co in group, ky, kx} w[co,ci_g,ky,kx] * gy[n,co,oy,ox]
//      with oy*stride = iy + pad - ky*dil (must divide), same for x.
__global__ __launch_bounds__(256) void conv_bwd_data_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                            ConvGeom g, int accumulate, float* __restrict__ gx,
                                                            int64_t total) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int ix = (int)(idx % g.W);  int64_t t = idx / g.W;
    const int iy = (int)(t % g.H);  t /= g.H;
    const int ci = (int)(t % g.Cin);
    const int n = (int)(t / g.Cin);
    const int grp = ci / g.cin_g, cig = ci - grp * g.cin_g;
    float acc = 0.f;
    for (int ky = 0; ky < g.K; ++ky) {
        const int ty = iy + g.pad - ky * g.dil;
        if (ty < 0 || ty % g.stride) continue;
        const int oy = ty / g.stride;
        if (oy >= g.Ho) continue;
        for (int kx = 0; kx < g.K; ++kx) {
            const int tx = ix + g.pad - kx * g.dil;
            if (tx < 0 || tx % g.stride) continue;
            const int ox = tx / g.stride;
            if (ox >= g.Wo) continue;
            const float* gp = gy + (((size_t)n * g.Cout + (size_t)grp * g.cout_g) * g.Ho + oy) * (size_t)g.Wo + ox;
            const float* wp = w + (((size_t)grp * g.cout_g) * g.cin_g + cig) * (size_t)(g.K * g.K) + ky * g.K + kx;
            for (int co = 0; co < g.cout_g; ++co)
                acc = fmaf(wp[(size_t)co * g.cin_g * g.K * g.K], gp[(size_t)co * g.Ho * g.Wo], acc);
        }
    }
    if (accumulate) gx[idx] += acc; else gx[idx] = acc;
}

// ---- weight gradient, generic: one workgroup per (weight element, pixel chunk); partial sums are combined with
//      float atomics (gw zero-filled by the launcher unless accumulating).
__global__ __launch_bounds__(256) void conv_bwd_weight_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                              ConvGeom g, int chunks, float* __restrict__ gw) {
    int b = blockIdx.x;
    const int chunk = b % chunks;  b /= chunks;
    const int widx = b;
    const int kx = b % g.K;  b /= g.K;
    const int ky = b % g.K;  b /= g.K;
    const int cig = b % g.cin_g;
    const int co = b / g.cin_g;
    const int grp = co / g.cout_g;
    const int ci = grp * g.cin_g + cig;
    const int npix = g.Ho * g.Wo;
    const int64_t total = (int64_t)g.N * npix;
    const int64_t per = (total + chunks - 1) / chunks;
    const int64_t i0 = chunk * per, i1 = min(total, i0 + per);
    float acc = 0.f;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
        const int n = (int)(i / npix), p = (int)(i - (int64_t)n * npix);
        const int oy = p / g.Wo, ox = p - oy * g.Wo;
        const int iy = oy * g.stride - g.pad + ky * g.dil, ix = ox * g.stride - g.pad + kx * g.dil;
        if (iy < 0 || iy >= g.H || ix < 0 || ix >= g.W) continue;
        acc = fmaf(gy[((size_t)n * g.Cout + co) * npix + p], x[(((size_t)n * g.Cin + ci) * g.H + iy) * (size_t)g.W + ix], acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&gw[widx], (part[0] + part[1]) + (part[2] + part[3]));
}

// ---- weight gradient of a 3x3 convolution with few input channels per group (CG = cin_g <= 5: depthwise, the
//      stem, the pyramid merge conv; any stride / dilation): one workgroup per (output channel, pixel chunk); a thread
//      walks output pixels (coalesced gy reads, the x taps hit L1/L2) carrying the CG*9 tap sums in registers, so gy
//      is streamed once instead of once per tap.
template <int CG>
__global__ __launch_bounds__(256) void g3x3_bwd_weight_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                              ConvGeom g, int chunks, float* __restrict__ gw) {
    const int chunk = blockIdx.x % chunks, co = blockIdx.x / chunks;
    const int ci0 = (co / g.cout_g) * CG;
    const int npix = g.Ho * g.Wo;
    const int64_t total = (int64_t)g.N * npix;
    const int64_t per = (total + chunks - 1) / chunks;
    const int64_t i0 = chunk * per, i1 = min(total, i0 + per);
    const size_t plane = (size_t)g.H * g.W;
    float acc[CG * 9];
#pragma unroll
    for (int t = 0; t < CG * 9; ++t) acc[t] = 0.f;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
        const int n = (int)(i / npix), p = (int)(i - (int64_t)n * npix);
        const int oy = p / g.Wo, ox = p - oy * g.Wo;
        const float gv = gy[((size_t)n * g.Cout + co) * npix + p];
        const float* xp = x + ((size_t)n * g.Cin + ci0) * plane;
        const int by = oy * g.stride - g.pad, bx = ox * g.stride - g.pad;
        int offs[9];
        float msk[9];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = by + ky * g.dil;
            const bool oky = iy >= 0 && iy < g.H;
            const int iyc = min(max(iy, 0), g.H - 1);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = bx + kx * g.dil;
                offs[ky * 3 + kx] = iyc * g.W + min(max(ix, 0), g.W - 1);
                msk[ky * 3 + kx] = (oky && ix >= 0 && ix < g.W) ? gv : 0.f;
            }
        }
        // request every tap of this position before the first use (hipcc otherwise emits load -> wait -> fma per tap)
        float xv[CG * 9];
#pragma unroll
        for (int ci = 0; ci < CG; ++ci)
#pragma unroll
            for (int t = 0; t < 9; ++t) xv[ci * 9 + t] = xp[ci * plane + offs[t]];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ci = 0; ci < CG; ++ci)
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[ci * 9 + t] = fmaf(msk[t], xv[ci * 9 + t], acc[ci * 9 + t]);
    }
    __shared__ float part[4][CG * 9];
#pragma unroll
    for (int t = 0; t < CG * 9; ++t) {
        const float v = wave_sum_dpp(acc[t]);                  // total in lane 63
        if ((threadIdx.x & 63) == 63) part[threadIdx.x >> 6][t] = v;
    }
    __syncthreads();
    if (threadIdx.x < CG * 9)
        atomicAdd(&gw[(size_t)co * CG * 9 + threadIdx.x],
                  (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]));
}

// Strip form of the same for stride 1 / dilation 1 / padding 1 on rows of whole 16-byte strips (the EfficientPWConv expansions, the
// pyramid stages and merge convolutions: every stride-1 3x3 of the training steps): a thread takes FOUR adjacent output positions per
// step -- gy as one 16-byte load, each input row as one 16-byte load + its two neighbours -- so a position's tap costs a quarter of a
// load instruction instead of one: the per-position form is bound by the L1 load path (CG * 9 four-byte loads per position:
// 179 us for the 8-channel-group expansion at 16 x 144x240 whose operands are 141 MB).
template <int CG>
__global__ __launch_bounds__(256) void g3x3_bwd_weight_strip_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                    ConvGeom g, int chunks, float* __restrict__ gw) {
    // The cout_g output channels of a group read the SAME input chunk: their workgroups are issued back to back on ONE XCD (workgroup b
    // goes to XCD b % 8), so the chunk is fetched into that XCD's L2 once instead of cout_g times through the Infinity Cache (137 us for
    // the 8-channel-group expansion: 4.6 TB/s of re-reads).
    const int xcd = blockIdx.x & 7, qq = blockIdx.x >> 3;
    const int cog = qq % g.cout_g, pair = (qq / g.cout_g) * 8 + xcd;          // pair = (group, chunk)
    if (pair >= g.G * chunks) return;
    const int chunk = pair % chunks, co = (pair / chunks) * g.cout_g + cog;
    const int ci0 = (co / g.cout_g) * CG;
    const int XS = g.W >> 2, nstrip = g.H * XS;                     // Ho == H, Wo == W
    const int64_t total = (int64_t)g.N * nstrip;
    const int64_t per = (total + chunks - 1) / chunks;
    const int64_t i0 = chunk * per, i1 = min(total, i0 + per);
    const size_t plane = (size_t)g.H * g.W;
    float acc[CG * 9];
#pragma unroll
    for (int t = 0; t < CG * 9; ++t) acc[t] = 0.f;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
        const int n = (int)(i / nstrip), sidx = (int)(i - (int64_t)n * nstrip);
        const int oy = sidx / XS, x0 = (sidx - oy * XS) * 4;
        const float4 g4 = *reinterpret_cast<const float4*>(gy + ((size_t)n * g.Cout + co) * plane + (size_t)oy * g.W + x0);
        const float gvv[4] = {g4.x, g4.y, g4.z, g4.w};
        const float* xp = x + ((size_t)n * g.Cin + ci0) * plane;
        const float ml = x0 > 0 ? 1.f : 0.f, mr = x0 + 4 < g.W ? 1.f : 0.f;
        const int xl = max(x0 - 1, 0), xr = min(x0 + 4, g.W - 1);
        int roff[3];  float rm[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy - 1 + ky;
            rm[ky] = (iy >= 0 && iy < g.H) ? 1.f : 0.f;
            roff[ky] = min(max(iy, 0), g.H - 1) * g.W;
        }
#pragma unroll
        for (int ci = 0; ci < CG; ++ci) {
            // the channel's nine loads first, then the arithmetic
            float4 v[3];  float l[3], r[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float* row = xp + ci * plane + roff[ky];
                v[ky] = *reinterpret_cast<const float4*>(row + x0);
                l[ky] = row[xl];
                r[ky] = row[xr];
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float w6[6] = {l[ky] * (ml * rm[ky]), v[ky].x * rm[ky], v[ky].y * rm[ky], v[ky].z * rm[ky], v[ky].w * rm[ky],
                                     r[ky] * (mr * rm[ky])};
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    float a = acc[ci * 9 + ky * 3 + kx];
#pragma unroll
                    for (int j = 0; j < 4; ++j) a = fmaf(gvv[j], w6[j + kx], a);
                    acc[ci * 9 + ky * 3 + kx] = a;
                }
            }
        }
    }
    __shared__ float part[4][CG * 9];
#pragma unroll
    for (int t = 0; t < CG * 9; ++t) {
        const float v = wave_sum_dpp(acc[t]);                  // total in lane 63
        if ((threadIdx.x & 63) == 63) part[threadIdx.x >> 6][t] = v;
    }
    __syncthreads();
    if (threadIdx.x < CG * 9)
        atomicAdd(&gw[(size_t)co * CG * 9 + threadIdx.x],
                  (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]));
}

// The same for COB consecutive output channels of one group per workgroup: a position's CG * 9 taps are loaded once and serve COB
// output channels (the stem's 3 -> 32 stride-2 gradient loaded every tap 32 times: 424 M four-byte loads through L1, 197 us for
// 87 MB of operands).  Per accumulator the positions arrive in the same order as above: same sums.
template <int CG, int COB>
__global__ __launch_bounds__(256, 3) void g3x3_bwd_weight_cob_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                  ConvGeom g, int chunks, float* __restrict__ gw) {
    const int chunk = blockIdx.x % chunks, co0 = (blockIdx.x / chunks) * COB;
    const int ci0 = (co0 / g.cout_g) * CG;
    const int npix = g.Ho * g.Wo;
    const int64_t total = (int64_t)g.N * npix;
    const int64_t per = (total + chunks - 1) / chunks;
    const int64_t i0 = chunk * per, i1 = min(total, i0 + per);
    const size_t plane = (size_t)g.H * g.W;
    float acc[COB][CG * 9];
#pragma unroll
    for (int c = 0; c < COB; ++c)
#pragma unroll
        for (int t = 0; t < CG * 9; ++t) acc[c][t] = 0.f;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
        const int n = (int)(i / npix), p = (int)(i - (int64_t)n * npix);
        const int oy = p / g.Wo, ox = p - oy * g.Wo;
        float gv[COB];
#pragma unroll
        for (int c = 0; c < COB; ++c) gv[c] = gy[((size_t)n * g.Cout + co0 + c) * npix + p];
        const float* xp = x + ((size_t)n * g.Cin + ci0) * plane;
        const int by = oy * g.stride - g.pad, bx = ox * g.stride - g.pad;
        int offs[9];
        float msk[9];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = by + ky * g.dil;
            const bool oky = iy >= 0 && iy < g.H;
            const int iyc = min(max(iy, 0), g.H - 1);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = bx + kx * g.dil;
                offs[ky * 3 + kx] = iyc * g.W + min(max(ix, 0), g.W - 1);
                msk[ky * 3 + kx] = (oky && ix >= 0 && ix < g.W) ? 1.f : 0.f;
            }
        }
        float xv[CG * 9];
#pragma unroll
        for (int ci = 0; ci < CG; ++ci)
#pragma unroll
            for (int t = 0; t < 9; ++t) xv[ci * 9 + t] = xp[ci * plane + offs[t]];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ci = 0; ci < CG; ++ci)
#pragma unroll
            for (int t = 0; t < 9; ++t) xv[ci * 9 + t] *= msk[t];           // zero padding (the address was clamped)
#pragma unroll
        for (int c = 0; c < COB; ++c)
#pragma unroll
            for (int t = 0; t < CG * 9; ++t) acc[c][t] = fmaf(gv[c], xv[t], acc[c][t]);
    }
    __shared__ float part[4][COB * CG * 9];
#pragma unroll
    for (int c = 0; c < COB; ++c)
#pragma unroll
        for (int t = 0; t < CG * 9; ++t) {
            const float v = wave_sum_dpp(acc[c][t]);           // total in lane 63
            if ((threadIdx.x & 63) == 63) part[threadIdx.x >> 6][c * CG * 9 + t] = v;
        }
    __syncthreads();
    if (threadIdx.x < COB * CG * 9)
        atomicAdd(&gw[(size_t)co0 * CG * 9 + threadIdx.x],
                  (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]));
}

// Stride 2 / padding 1 on rows of whole 32-byte blocks (the stem, 3 -> 32 at 288x480): four adjacent output positions per thread.
// Their 3x3 windows cover input columns 2*ox0 - 1 .. 2*ox0 + 7 of each input row: two aligned 16-byte loads and the left neighbour
// instead of twelve 4-byte loads at a stride of two floats; COB output channels share them.  (The per-position form ran at two, then
// three waves per SIMD with one exposed load round trip per position: 166 -> 135 us for 96 MB of operands.)
template <int CG, int COB>
__global__ __launch_bounds__(256, 3) void g3x3_bwd_weight_s2_strip_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                          ConvGeom g, int chunks, float* __restrict__ gw) {
    const int cgb = g.cout_g / COB;                               // workgroups per (group, chunk) pair: issued back to back on one XCD
    const int xcd = blockIdx.x & 7, qq = blockIdx.x >> 3;
    const int cog = qq % cgb, pair = (qq / cgb) * 8 + xcd;
    if (pair >= g.G * chunks) return;
    const int chunk = pair % chunks, co0 = (pair / chunks) * g.cout_g + cog * COB;
    const int ci0 = (co0 / g.cout_g) * CG;
    const int XS = g.Wo >> 2, nstrip = g.Ho * XS;
    const int64_t total = (int64_t)g.N * nstrip;
    const int64_t per = (total + chunks - 1) / chunks;
    const int64_t i0 = chunk * per, i1 = min(total, i0 + per);
    const size_t plane = (size_t)g.H * g.W, oplane = (size_t)g.Ho * g.Wo;
    float acc[COB][CG * 9];
#pragma unroll
    for (int c = 0; c < COB; ++c)
#pragma unroll
        for (int t = 0; t < CG * 9; ++t) acc[c][t] = 0.f;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
        const int n = (int)(i / nstrip), sidx = (int)(i - (int64_t)n * nstrip);
        const int oy = sidx / XS, ox0 = (sidx - oy * XS) * 4;
        float gv[COB][4];
#pragma unroll
        for (int c = 0; c < COB; ++c) {
            const float4 t4 = *reinterpret_cast<const float4*>(gy + ((size_t)n * g.Cout + co0 + c) * oplane + (size_t)oy * g.Wo + ox0);
            gv[c][0] = t4.x; gv[c][1] = t4.y; gv[c][2] = t4.z; gv[c][3] = t4.w;
        }
        const float* xp = x + ((size_t)n * g.Cin + ci0) * plane + 2 * ox0;
        const float ml = ox0 > 0 ? 1.f : 0.f;
        const int lo = ox0 > 0 ? -1 : 0;
#pragma unroll
        for (int ci = 0; ci < CG; ++ci) {
            float4 a[3], b[3];  float l[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = 2 * oy - 1 + ky;                                  // <= H - 1 (H even); -1 at the top
                const float* row = xp + ci * plane + (size_t)max(iy, 0) * g.W;
                a[ky] = *reinterpret_cast<const float4*>(row);
                b[ky] = *reinterpret_cast<const float4*>(row + 4);
                l[ky] = row[lo];
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float rm = (2 * oy - 1 + ky >= 0) ? 1.f : 0.f;
                const float w9[9] = {l[ky] * (ml * rm), a[ky].x * rm, a[ky].y * rm, a[ky].z * rm, a[ky].w * rm,
                                     b[ky].x * rm, b[ky].y * rm, b[ky].z * rm, b[ky].w * rm};
#pragma unroll
                for (int c = 0; c < COB; ++c)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        float v = acc[c][ci * 9 + ky * 3 + kx];
#pragma unroll
                        for (int j = 0; j < 4; ++j) v = fmaf(gv[c][j], w9[2 * j + kx], v);
                        acc[c][ci * 9 + ky * 3 + kx] = v;
                    }
            }
            __builtin_amdgcn_sched_barrier(0);            // one channel's nine loads in flight at a time (all 27 hoisted: 70+ spilled VGPRs)
        }
    }
    __shared__ float part[4][COB * CG * 9];
#pragma unroll
    for (int c = 0; c < COB; ++c)
#pragma unroll
        for (int t = 0; t < CG * 9; ++t) {
            const float v = wave_sum_dpp(acc[c][t]);           // total in lane 63
            if ((threadIdx.x & 63) == 63) part[threadIdx.x >> 6][c * CG * 9 + t] = v;
        }
    __syncthreads();
    if (threadIdx.x < COB * CG * 9)
        atomicAdd(&gw[(size_t)co0 * CG * 9 + threadIdx.x],
                  (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]));
}

// ---- weight gradient of a grouped 1x1 convolution: gw[co, ci] = sum_{n,p} gy[n,co,p] * x[n,ci,p].
//      A workgroup owns a 16 x 16 tile of one group's (co, ci) pairs and a chunk of pixels; both operand tiles
//      (16 rows x 256 pixels) are staged in LDS with coalesced loads, each thread reduces one (co, ci) pair over the
//      staged pixels (row reads are LDS broadcasts / conflict-free), partial sums combined with float atomics.
__global__ __launch_bounds__(256) void conv1x1_bwd_weight_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                 ConvGeom g, int chunks, int tiles_m, int tiles_k,
                                                                 float* __restrict__ gw) {
    __shared__ float A[16][260];      // gy rows (co), 256 pixels (+4 pad: rows land on different banks)
    __shared__ float B[16][260];      // x rows (ci)
    int b = blockIdx.x;
    const int chunk = b % chunks;  b /= chunks;
    const int tk = b % tiles_k;  b /= tiles_k;
    const int tm = b % tiles_m;
    const int grp = b / tiles_m;
    const int M = g.cout_g, K = g.cin_g, HW = g.Ho * g.Wo;
    const int m0 = tm * 16, k0 = tk * 16;
    const int tid = threadIdx.x;
    const int tmi = tid >> 4, tki = tid & 15;      // this thread's (co, ci) pair inside the tile
    const int64_t total = (int64_t)g.N * HW;
    const int64_t per = ((total + chunks - 1) / chunks + 255) / 256 * 256;
    const int64_t i0 = chunk * per, i1 = min(total, i0 + per);
    float acc = 0.f;
    for (int64_t base = i0; base < i1; base += 256) {
        // stage: thread t loads pixel base+t of 16 rows of each operand (coalesced along pixels).  Row and pixel
        // indices are clamped so all 32 loads are unconditional and issue back to back; masked to zero afterwards.
        const int64_t i = base + tid;
        const bool ok = i < i1;
        const int64_t ic = ok ? i : i1 - 1;
        const int n = (int)(ic / HW), p = (int)(ic - (int64_t)n * HW);
        const float* ga = gy + ((size_t)n * g.Cout + (size_t)grp * M) * HW + p;
        const float* xb = x + ((size_t)n * g.Cin + (size_t)grp * K) * HW + p;
        float ra[16], rb[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            ra[r] = ga[(size_t)min(m0 + r, M - 1) * HW];
            rb[r] = xb[(size_t)min(k0 + r, K - 1) * HW];
        }
        __builtin_amdgcn_sched_barrier(0);          // keep the 32 loads in flight together (no per-load wait)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            A[r][tid] = (ok && m0 + r < M) ? ra[r] : 0.f;
            B[r][tid] = (ok && k0 + r < K) ? rb[r] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int q = 0; q < 256; q += 4) {
            const float4 a4 = *reinterpret_cast<const float4*>(&A[tmi][q]);
            const float4 b4 = *reinterpret_cast<const float4*>(&B[tki][q]);
            acc = fmaf(a4.x, b4.x, acc); acc = fmaf(a4.y, b4.y, acc); acc = fmaf(a4.z, b4.z, acc); acc = fmaf(a4.w, b4.w, acc);
        }
        __syncthreads();
    }
    if (m0 + tmi < M && k0 + tki < K)
        atomicAdd(&gw[((size_t)grp * M + m0 + tmi) * K + k0 + tki], acc);
}

// Batch-statistics BatchNorm (the supervised loop): what follows the channel sums of the affine backward -- d gamma, d beta and the
// coefficients of gz = p * z + q (mspl_bn_batch_stats_bwd_coeffs) -- done by the workgroup that adds a channel's last partial ("last
// block done": atomic adds -> their acknowledgement -> counter), which also hands the two sums and the counter back zeroed (persistent workspace).
struct BnTail {
    unsigned* cnt;            // null: no tail (gscale / gshift are the caller's accumulators)
    unsigned per_channel;     // workgroups per channel
    const float* gamma;  const float* mean;  const float* invstd;
    float inv_m;  int accumulate;
    float* ggamma;  float* gbeta;  float* pc;  float* qc;
};

// ---- affine + PReLU backward.  Forward: z = (c + pre) * scale + shift + res ; y = z > 0 ? z : alpha * z.
// Outputs gz (optional) and gc = gz * scale; per-channel sums into gscale / gshift / galpha (atomics; zeroed by caller).
__global__ __launch_bounds__(256) void affine_prelu_bwd_kernel(const float* __restrict__ c, const float* __restrict__ pre,
                                                               const float* __restrict__ res, const float* __restrict__ gy,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               const float* __restrict__ alpha, int N, int C, int HW, int parts,
                                                               float* __restrict__ gz_out, float* __restrict__ gc_out,
                                                               float* __restrict__ gscale, float* __restrict__ gshift,
                                                               float* __restrict__ galpha, const float* __restrict__ bn_mean,
                                                               const float* __restrict__ bn_inv, BnTail tl) {
    // A workgroup owns one of `parts` contiguous ranges of a channel's N * HW elements (the N planes taken as one sequence): the
    // number of same-address atomics per channel is `parts`, whatever N is.  (One workgroup per (image, channel, chunk) made it
    // N * chunks: at 16 x 16 x 144x240 the 256 atomic chains per channel cost 55 of the kernel's 81 us.)
    const int ch = blockIdx.x / parts, rng = blockIdx.x - ch * parts;
    const float sc = scale ? scale[ch] : 1.f, sh = shift ? shift[ch] : 0.f;
    const bool act = alpha != nullptr;
    const float al = act ? alpha[ch] : 1.f;
    float s_scale = 0.f, s_shift = 0.f, s_alpha = 0.f;
    auto one = [&](float cv, float pv, float rv, float g, float& gz, float& gcv) {
        const float u = cv + pv;
        const float z = fmaf(u, sc, sh) + rv;             // the forward epilogue's expression (epi_apply: fmaf, then + residual): same PReLU branch
        gz = (!act || z > 0.f) ? g : al * g;
        if (act && z <= 0.f) s_alpha += g * z;
        s_scale += gz * u;
        s_shift += gz;
        gcv = gz * sc;
    };
    const bool vec = (HW & 3) == 0;
    const int L = vec ? (HW >> 2) : HW;                    // units (quads or elements) per plane
    const long long total = (long long)N * L;
    const long long per = (total + parts - 1) / parts;
    const long long u0 = (long long)rng * per, u1 = min(total, u0 + per);
    const int step_n = 256 / L, step_q = 256 - step_n * L;
    long long u = u0 + threadIdx.x;
    int n = (int)(u / L), q = (int)(u - (long long)n * L);
    auto advance = [&]() { u += 256; q += step_q; n += step_n; if (q >= L) { q -= L; ++n; } };
    if (vec) {                                 // 16-byte operands, four independent quads per operand in flight per iteration
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        constexpr int UQ = 4;
        while (u < u1) {
            size_t o[UQ];  bool h[UQ];
#pragma unroll
            for (int i = 0; i < UQ; ++i) {
                h[i] = u < u1;
                o[i] = h[i] ? (((size_t)n * C + ch) * (size_t)HW) + 4 * (size_t)q : o[0];
                advance();
            }
            float4 cv[UQ], gv[UQ], pv[UQ], rv[UQ];
#pragma unroll
            for (int i = 0; i < UQ; ++i) {
                cv[i] = *reinterpret_cast<const float4*>(c + o[i]);
                gv[i] = *reinterpret_cast<const float4*>(gy + o[i]);
                pv[i] = pre ? *reinterpret_cast<const float4*>(pre + o[i]) : zero;
                rv[i] = res ? *reinterpret_cast<const float4*>(res + o[i]) : zero;
            }
#pragma unroll
            for (int i = 0; i < UQ; ++i) {
                if (!h[i]) continue;
                float4 gz, gc;
                one(cv[i].x, pv[i].x, rv[i].x, gv[i].x, gz.x, gc.x); one(cv[i].y, pv[i].y, rv[i].y, gv[i].y, gz.y, gc.y);
                one(cv[i].z, pv[i].z, rv[i].z, gv[i].z, gz.z, gc.z); one(cv[i].w, pv[i].w, rv[i].w, gv[i].w, gz.w, gc.w);
                if (gz_out) *reinterpret_cast<float4*>(gz_out + o[i]) = gz;
                if (gc_out) *reinterpret_cast<float4*>(gc_out + o[i]) = gc;
            }
        }
    } else {
        while (u < u1) {
            const size_t o = (((size_t)n * C + ch) * (size_t)HW) + (size_t)q;
            advance();
            float gz, gc;
            one(c[o], pre ? pre[o] : 0.f, res ? res[o] : 0.f, gy[o], gz, gc);
            if (gz_out) gz_out[o] = gz;
            if (gc_out) gc_out[o] = gc;
        }
    }
    float v[3] = {s_scale, s_shift, s_alpha};
    __shared__ float part[3][4];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        v[k] = wave_sum_dpp(v[k]);                            // total in lane 63
        if ((threadIdx.x & 63) == 63) part[k][threadIdx.x >> 6] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t_scale = (part[0][0] + part[0][1]) + (part[0][2] + part[0][3]);
        const float t_shift = (part[1][0] + part[1][1]) + (part[1][2] + part[1][3]);
        // frozen BatchNorm folded into (scale, shift) = (gamma*inv, beta - mean*gamma*inv): the map from (dscale, dshift) to
        // (dgamma, dbeta) is linear, so every workgroup's partial goes straight to the parameters' gradient buffers
        if (gscale) atomicAdd(&gscale[ch], bn_inv ? (t_scale - bn_mean[ch] * t_shift) * bn_inv[ch] : t_scale);
        if (gshift) atomicAdd(&gshift[ch], t_shift);
        if (galpha && act) atomicAdd(&galpha[ch], (part[2][0] + part[2][1]) + (part[2][2] + part[2][3]));
        if (tl.cnt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (atomics acknowledged before the counter: see bn_stats_fused_kernel)
            if (atomicAdd(tl.cnt + ch, 1u) == tl.per_channel - 1) {
                const float dsc = __uint_as_float(atomicExch(reinterpret_cast<unsigned*>(gscale + ch), 0u));      // read and clear
                const float dsh = __uint_as_float(atomicExch(reinterpret_cast<unsigned*>(gshift + ch), 0u));
                tl.cnt[ch] = 0u;
                const float is = tl.invstd[ch], mu = tl.mean[ch];
                const float t = dsc + (-mu) * dsh;
                tl.ggamma[ch] = tl.accumulate ? tl.ggamma[ch] + t * is : t * is;
                tl.gbeta[ch] = tl.accumulate ? tl.gbeta[ch] + dsh : dsh;
                const float p = ((tl.gamma[ch] * t) * (is * is * is)) * (-tl.inv_m);
                tl.pc[ch] = p;
                tl.qc[ch] = (dsh * sc) * (-tl.inv_m) - p * mu;
            }
        }
    }
}

// ---- resampling backward (atomic scatter of the output gradient; gx zeroed by the caller)
struct RsG { int N, C, Hi, Wi, Ho, Wo; float sh, sw; unsigned mHi, mWi, mHo, mWo; };   // m*: ceil(2^32/d) division magics

// n / d for n * d < 2^32 with m = ceil(2^32 / d) (d == 1: m overflows to 0, handled)
__device__ __forceinline__ unsigned udiv_magic(unsigned n, unsigned d, unsigned m) { return d == 1 ? n : __umulhi(n, m); }

// Even planes, rows of whole 16-byte strips: a thread turns a 2 x 5 patch of the output gradient (two float4 + their right neighbours)
// into the 2 x 8 block of the input gradient under it -- even rows / columns are covered by one output, odd ones by two (the gather of
// the kernel below, written out: same order of additions) -- with 16-byte loads and stores instead of one pixel per thread
// (56 -> 15 us at 32 x 128x240).
__global__ __launch_bounds__(256) void avgpool3x3s2_bwd_vec_kernel(const float* __restrict__ gy, RsG g, int XS, float* __restrict__ gx) {
    const int t = blockIdx.z * gridDim.y + blockIdx.y;      // plane n*C + c
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (t >= g.N * g.C || s >= g.Ho * XS) return;
    const int k = s / XS, m0 = (s - k * XS) * 4;             // output row, first output column
    const float* r0 = gy + (size_t)t * g.Ho * g.Wo + (size_t)k * g.Wo + m0;
    const bool hasr = m0 + 4 < g.Wo, hasd = k + 1 < g.Ho;
    const float4 a4 = *reinterpret_cast<const float4*>(r0);
    const float ar = hasr ? r0[4] : 0.f;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);  float br = 0.f;
    if (hasd) { b4 = *reinterpret_cast<const float4*>(r0 + g.Wo);  br = hasr ? r0[g.Wo + 4] : 0.f; }
    const float a[5] = {a4.x, a4.y, a4.z, a4.w, ar}, b[5] = {b4.x, b4.y, b4.z, b4.w, br};
    float e[8], o[8];                                        // input rows 2k (even) and 2k+1 (odd), columns 2 m0 .. 2 m0 + 7
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // the gather visits (oy, ox) in ascending order starting from 0: 0 + g[k][m] (+ g[k][m+1]) (+ g[k+1][m]) (+ g[k+1][m+1])
        e[2 * j] = (0.f + a[j]) * (1.0f / 9.0f);
        o[2 * j] = hasd ? ((0.f + a[j]) + b[j]) * (1.0f / 9.0f) : (0.f + a[j]) * (1.0f / 9.0f);
        const bool hr = j < 3 || hasr;                       // the last odd column of a row has no right neighbour output
        e[2 * j + 1] = hr ? ((0.f + a[j]) + a[j + 1]) * (1.0f / 9.0f) : (0.f + a[j]) * (1.0f / 9.0f);
        o[2 * j + 1] = hasd ? (hr ? ((((0.f + a[j]) + a[j + 1]) + b[j]) + b[j + 1]) * (1.0f / 9.0f) : ((0.f + a[j]) + b[j]) * (1.0f / 9.0f))
                            : (hr ? ((0.f + a[j]) + a[j + 1]) * (1.0f / 9.0f) : (0.f + a[j]) * (1.0f / 9.0f));
    }
    float* d = gx + (size_t)t * g.Hi * g.Wi + (size_t)(2 * k) * g.Wi + 2 * m0;
    *reinterpret_cast<float4*>(d) = make_float4(e[0], e[1], e[2], e[3]);
    *reinterpret_cast<float4*>(d + 4) = make_float4(e[4], e[5], e[6], e[7]);
    *reinterpret_cast<float4*>(d + g.Wi) = make_float4(o[0], o[1], o[2], o[3]);
    *reinterpret_cast<float4*>(d + g.Wi + 4) = make_float4(o[4], o[5], o[6], o[7]);
}

__global__ __launch_bounds__(256) void avgpool3x3s2_bwd_kernel(const float* __restrict__ gy, RsG g, float* __restrict__ gx, int64_t total) {
    const int t = blockIdx.z * gridDim.y + blockIdx.y;      // plane n*C + c
    const int pi = blockIdx.x * 256 + threadIdx.x;
    if (t >= g.N * g.C || pi >= g.Hi * g.Wi) return;
    const int iy = (int)udiv_magic(pi, g.Wi, g.mWi), ix = pi - iy * g.Wi;
    const int64_t idx = (int64_t)t * g.Hi * g.Wi + pi;
    const float* gp = gy + (size_t)t * g.Ho * g.Wo;
    float acc = 0.f;                                         // gather: outputs whose 3x3/s2/p1 window covers (iy, ix)
    for (int oy = max(0, iy / 2); oy <= min(g.Ho - 1, (iy + 1) / 2); ++oy) {
        if (2 * oy - 1 > iy || 2 * oy + 1 < iy) continue;
        for (int ox = max(0, ix / 2); ox <= min(g.Wo - 1, (ix + 1) / 2); ++ox) {
            if (2 * ox - 1 > ix || 2 * ox + 1 < ix) continue;
            acc += gp[(size_t)oy * g.Wo + ox];
        }
    }
    gx[idx] = acc * (1.0f / 9.0f);
}

// Gather form (deterministic, no atomics): one thread per INPUT pixel, loops over the output rows/columns whose
// two bilinear sources include it.  Output y uses source rows floor(y*sh) and +1, so the candidates for input row iy
// are y in [ceil((iy-1)/sh), floor((iy+1)/sh)] (clamped); same along x.
template <int MAXC>
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ gy, RsG g, float* __restrict__ gx, int64_t total) {
    const int t = blockIdx.z * gridDim.y + blockIdx.y;      // plane n*C + c
    const int pi = blockIdx.x * 256 + threadIdx.x;
    if (t >= g.N * g.C || pi >= g.Hi * g.Wi) return;
    const int iy = pi / g.Wi, ix = pi - iy * g.Wi;
    const int64_t idx = (int64_t)t * g.Hi * g.Wi + pi;
    const float* gp = gy + (size_t)t * g.Ho * g.Wo;
    int ylo = 0, yhi = g.Ho - 1, xlo = 0, xhi = g.Wo - 1;
    // outputs whose source floor(o*s) is i-1 or i lie in [ceil((i-1)/s), ceil((i+1)/s) - 1]; one extra on each side covers the
    // float rounding of o*s in the forward (the weights below are evaluated exactly as the forward does, so extras get 0)
    if (g.sh > 0.f) { ylo = max(0, (int)ceilf((float)(iy - 1) / g.sh) - 1); yhi = min(g.Ho - 1, (int)floorf((float)(iy + 1) / g.sh) + 1); }
    if (g.sw > 0.f) { xlo = max(0, (int)ceilf((float)(ix - 1) / g.sw) - 1); xhi = min(g.Wo - 1, (int)floorf((float)(ix + 1) / g.sw) + 1); }
    float acc = 0.f;
    const int nx = xhi - xlo + 1;
    if (nx <= MAXC) {
        // the column weights do not depend on the row: evaluate them once (x2 / x4 up-sampling: 7 / 11 candidates, at most 4
        // of them non-zero), then each candidate row costs one weight evaluation and the loads of the non-zero columns only
        float wxs[MAXC];
#pragma unroll
        for (int j = 0; j < MAXC; ++j) {
            wxs[j] = 0.f;
            if (j < nx) {
                int x0, x1;  float wx0, wx1;
                bilinear_src(g.sw, xlo + j, g.Wi, x0, x1, wx0, wx1);
                wxs[j] = (x0 == ix ? wx0 : 0.f) + (x1 == ix ? wx1 : 0.f);
            }
        }
        for (int oy = ylo; oy <= yhi; ++oy) {
            int y0, y1;  float wy0, wy1;
            bilinear_src(g.sh, oy, g.Hi, y0, y1, wy0, wy1);
            const float wy = (y0 == iy ? wy0 : 0.f) + (y1 == iy ? wy1 : 0.f);
            if (wy == 0.f) continue;
            const float* row = gp + (size_t)oy * g.Wo + xlo;
            float rv[MAXC];
#pragma unroll
            for (int j = 0; j < MAXC; ++j) rv[j] = row[min(j, nx - 1)];      // clamped and unconditional (wxs is 0 past nx):
            __builtin_amdgcn_sched_barrier(0);                                // the row's reads leave as one batch
            float rowacc = 0.f;
#pragma unroll
            for (int j = 0; j < MAXC; ++j) rowacc = fmaf(wxs[j], rv[j], rowacc);
            acc = fmaf(wy, rowacc, acc);
        }
        gx[idx] = acc;
        return;
    }
    for (int oy = ylo; oy <= yhi; ++oy) {
        int y0, y1;  float wy0, wy1;
        bilinear_src(g.sh, oy, g.Hi, y0, y1, wy0, wy1);
        const float wy = (y0 == iy ? wy0 : 0.f) + (y1 == iy ? wy1 : 0.f);
        if (wy == 0.f) continue;
        float rowacc = 0.f;
        for (int ox = xlo; ox <= xhi; ++ox) {
            int x0, x1;  float wx0, wx1;
            bilinear_src(g.sw, ox, g.Wi, x0, x1, wx0, wx1);
            const float wx = (x0 == ix ? wx0 : 0.f) + (x1 == ix ? wx1 : 0.f);
            if (wx != 0.f) rowacc = fmaf(wx, gp[(size_t)oy * g.Wo + ox], rowacc);
        }
        acc = fmaf(wy, rowacc, acc);
    }
    gx[idx] = acc;
}

// Tiled separable form of the same gather (x2 / x4 up-sampling: the decoder merges and the two heads, 9-10 launches per training
// step).  bilinear^T = R_y^T . gy . R_x is separable: a workgroup owns TLH x 64 INPUT pixels of one plane, stages the output-gradient
// rows / columns they can receive from in LDS, (A) contracts the columns -- H[r][j] = sum_k wx[j][k] * G[r][clo[j] + k], the column
// weights evaluated ONCE per workgroup into an LDS table instead of once per thread and candidate -- and (B) contracts the rows of H
// with the row-weight table.  Candidates, weights and the order of the additions are exactly bilinear_bwd_kernel's (zero-weight
// candidates contribute fma(0, finite, acc) = acc), so the results are bit-identical; the per-thread form spent ~250-300 vector
// instructions per input pixel on re-evaluating bilinear_src (146 us for the 72x120 -> 288x480 head at 16 x 13 planes: 0.8 TB/s).
__device__ __forceinline__ int bb_lo(float s, int i) { return s > 0.f ? max(0, (int)ceilf((float)(i - 1) / s) - 1) : 0; }
__device__ __forceinline__ int bb_hi(float s, int i, int O) { return s > 0.f ? min(O - 1, (int)floorf((float)(i + 1) / s) + 1) : O - 1; }

template <int MAXC, int TLH>
__global__ __launch_bounds__(256) void bilinear_bwd_tile_kernel(const float* __restrict__ gy, RsG g, int FH, int FWS, int tiles_x,
                                                                float* __restrict__ gx) {
    constexpr int TLW = 64;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* G = sm;                                      // [FH][FWS]: output-gradient tile, zero past the staged columns
    float* Hh = G + FH * FWS;                           // [FH + MAXC][TLW]: column-contracted rows, zero past the staged rows
    float* WX = Hh + (FH + MAXC) * TLW;                 // [MAXC][TLW] column weights of candidate k of input column j
    float* WY = WX + MAXC * TLW;                        // [TLH][MAXC] row weights
    int* CL = reinterpret_cast<int*>(WY + TLH * MAXC);  // [TLW] first candidate column of input column j, relative to the tile
    int* RL = CL + TLW;                                 // [TLH] first candidate row of input row i, relative to the tile
    const int t = blockIdx.z * gridDim.y + blockIdx.y;  // plane n*C + c
    if (t >= g.N * g.C) return;
    const int txi = blockIdx.x % tiles_x, tyi = blockIdx.x / tiles_x;
    const int iy0 = tyi * TLH, ix0 = txi * TLW;
    const int iyl = min(iy0 + TLH, g.Hi) - 1, ixl = min(ix0 + TLW, g.Wi) - 1;
    const int RY0 = bb_lo(g.sh, iy0), RY1 = bb_hi(g.sh, iyl, g.Ho), CX0 = bb_lo(g.sw, ix0), CX1 = bb_hi(g.sw, ixl, g.Wo);
    const int nrows = min(RY1 - RY0 + 1, FH), ncols = min(CX1 - CX0 + 1, FWS - MAXC);      // (the host sized FH / FWS from the same formulas)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* gp = gy + (size_t)t * g.Ho * g.Wo;
    // ---- stage the gradient tile: a wave per row, lanes over the columns, two rows' loads in flight
    for (int r = wave; r < nrows; r += 8) {
        const float* ra = gp + (size_t)(RY0 + r) * g.Wo + CX0;
        const bool hb = r + 4 < nrows;
        const float* rb = gp + (size_t)(RY0 + (hb ? r + 4 : r)) * g.Wo + CX0;
        for (int c0 = 0; c0 < FWS; c0 += 128) {
            const int ca = c0 + lane, cb = c0 + 64 + lane;
            const float a0 = ca < ncols ? ra[ca] : 0.f, a1 = cb < ncols ? ra[cb] : 0.f;
            const float b0 = (hb && ca < ncols) ? rb[ca] : 0.f, b1 = (hb && cb < ncols) ? rb[cb] : 0.f;
            if (ca < FWS) { G[r * FWS + ca] = a0;  if (hb) G[(r + 4) * FWS + ca] = b0; }
            if (cb < FWS) { G[r * FWS + cb] = a1;  if (hb) G[(r + 4) * FWS + cb] = b1; }
        }
    }
    // rows of H past the staged ones: read (with zero weights) by phase B
    for (int i = tid; i < MAXC * TLW; i += 256) Hh[nrows * TLW + i] = 0.f;
    // ---- weight tables
    for (int task = tid; task < MAXC * TLW; task += 256) {
        const int k = task / TLW, j = task - k * TLW, ix = ix0 + j;
        float w = 0.f;
        int lo = CX0;
        if (ix <= ixl) {
            lo = bb_lo(g.sw, ix);
            const int ox = lo + k;
            if (ox <= bb_hi(g.sw, ix, g.Wo)) {
                int x0, x1;  float wx0, wx1;
                bilinear_src(g.sw, ox, g.Wi, x0, x1, wx0, wx1);
                w = (x0 == ix ? wx0 : 0.f) + (x1 == ix ? wx1 : 0.f);
            }
        }
        WX[task] = w;
        if (k == 0) CL[j] = lo - CX0;
    }
    for (int task = tid; task < TLH * MAXC; task += 256) {
        const int i = task / MAXC, k = task - i * MAXC, iy = iy0 + i;
        float w = 0.f;
        int lo = RY0;
        if (iy <= iyl) {
            lo = bb_lo(g.sh, iy);
            const int oy = lo + k;
            if (oy <= bb_hi(g.sh, iy, g.Ho)) {
                int y0, y1;  float wy0, wy1;
                bilinear_src(g.sh, oy, g.Hi, y0, y1, wy0, wy1);
                w = (y0 == iy ? wy0 : 0.f) + (y1 == iy ? wy1 : 0.f);
            }
        }
        WY[task] = w;
        if (k == 0) RL[i] = lo - RY0;
    }
    __syncthreads();
    // ---- (A) contract the columns: thread = input column j, rows wave, wave + 4, ...
    {
        float wx[MAXC];
#pragma unroll
        for (int k = 0; k < MAXC; ++k) wx[k] = WX[k * TLW + lane];
        const float* gcol = G + CL[lane];
        for (int r = wave; r < nrows; r += 4) {
            const float* row = gcol + r * FWS;
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < MAXC; ++k) acc = fmaf(wx[k], row[k], acc);
            Hh[r * TLW + lane] = acc;
        }
    }
    __syncthreads();
    // ---- (B) contract the rows and store
    for (int i = wave; i < TLH; i += 4) {
        const int iy = iy0 + i, ix = ix0 + lane;
        if (iy > iyl || ix > ixl) continue;
        const float* hcol = Hh + RL[i] * TLW + lane;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < MAXC; ++k) acc = fmaf(WY[i * MAXC + k], hcol[k * TLW], acc);
        gx[(size_t)t * g.Hi * g.Wi + (size_t)iy * g.Wi + ix] = acc;
    }
}

// Large up-sampling ratios (the pyramid's 0.1 branch: 7x12 -> 64x120, 24x24 candidate outputs per input pixel, and only a
// few hundred input pixels per plane): one WAVE per input pixel, lanes over the candidate columns, rows walked together.
__global__ __launch_bounds__(256) void bilinear_bwd_wave_kernel(const float* __restrict__ gy, RsG g, float* __restrict__ gx, int64_t total_in) {
    const int64_t w = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (w >= total_in) return;                                  // wave-uniform
    const int hw = g.Hi * g.Wi;
    const int64_t t = w / hw;
    const int pi = (int)(w - t * hw);
    const int iy = pi / g.Wi, ix = pi - iy * g.Wi;
    const float* gp = gy + (size_t)t * g.Ho * g.Wo;
    int ylo = 0, yhi = g.Ho - 1, xlo = 0, xhi = g.Wo - 1;
    if (g.sh > 0.f) { ylo = max(0, (int)floorf((float)(iy - 1) / g.sh) - 1); yhi = min(g.Ho - 1, (int)ceilf((float)(iy + 1) / g.sh) + 1); }
    if (g.sw > 0.f) { xlo = max(0, (int)floorf((float)(ix - 1) / g.sw) - 1); xhi = min(g.Wo - 1, (int)ceilf((float)(ix + 1) / g.sw) + 1); }
    float acc = 0.f;
    for (int xb = xlo; xb <= xhi; xb += 64) {
        const int ox = xb + lane;
        float wx = 0.f;
        if (ox <= xhi) {
            int x0, x1;  float wx0, wx1;
            bilinear_src(g.sw, ox, g.Wi, x0, x1, wx0, wx1);
            wx = (x0 == ix ? wx0 : 0.f) + (x1 == ix ? wx1 : 0.f);
        }
        const int oxc = min(ox, xhi);
        float col = 0.f;
        for (int oy = ylo; oy <= yhi; ++oy) {
            int y0, y1;  float wy0, wy1;
            bilinear_src(g.sh, oy, g.Hi, y0, y1, wy0, wy1);
            const float wy = (y0 == iy ? wy0 : 0.f) + (y1 == iy ? wy1 : 0.f);
            if (wy == 0.f) continue;                            // wave-uniform
            col = fmaf(wy, gp[(size_t)oy * g.Wo + oxc], col);
        }
        acc = fmaf(wx, col, acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0) gx[w] = acc;
}

// Large ratios, second form: a workgroup per (plane, INPUT row).  The candidate output rows of that input row are reduced
// vertically with coalesced reads (thread = output column), the row of column sums goes through LDS, then thread b < Wi finishes
// input pixel (row, b) over its candidate columns.  Every output row is read about twice (once per input row it feeds) instead of
// once per input PIXEL and candidate column block: 96 -> ~15 us for the 13x24 -> 128x240 map of the pyramid's 0.1 branch.
__global__ __launch_bounds__(256) void bilinear_bwd_rows_kernel(const float* __restrict__ gy, RsG g, float* __restrict__ gx) {
    extern __shared__ float colsum[];                       // Wo floats
    const int iy = blockIdx.x % g.Hi;
    const int t = blockIdx.x / g.Hi;                        // plane
    const float* gp = gy + (size_t)t * g.Ho * g.Wo;
    int ylo = 0, yhi = g.Ho - 1;
    if (g.sh > 0.f) { ylo = max(0, (int)floorf((float)(iy - 1) / g.sh) - 1); yhi = min(g.Ho - 1, (int)ceilf((float)(iy + 1) / g.sh) + 1); }
    for (int ox = threadIdx.x; ox < g.Wo; ox += 256) {
        float col = 0.f;
        for (int oy = ylo; oy <= yhi; ++oy) {
            int y0, y1;  float wy0, wy1;
            bilinear_src(g.sh, oy, g.Hi, y0, y1, wy0, wy1);
            const float wy = (y0 == iy ? wy0 : 0.f) + (y1 == iy ? wy1 : 0.f);
            if (wy == 0.f) continue;                            // uniform
            col = fmaf(wy, gp[(size_t)oy * g.Wo + ox], col);
        }
        colsum[ox] = col;
    }
    __syncthreads();
    for (int ix = threadIdx.x; ix < g.Wi; ix += 256) {
        int xlo = 0, xhi = g.Wo - 1;
        if (g.sw > 0.f) { xlo = max(0, (int)floorf((float)(ix - 1) / g.sw) - 1); xhi = min(g.Wo - 1, (int)ceilf((float)(ix + 1) / g.sw) + 1); }
        float acc = 0.f;
        for (int ox = xlo; ox <= xhi; ++ox) {
            int x0, x1;  float wx0, wx1;
            bilinear_src(g.sw, ox, g.Wi, x0, x1, wx0, wx1);
            const float wx = (x0 == ix ? wx0 : 0.f) + (x1 == ix ? wx1 : 0.f);
            acc = fmaf(wx, colsum[ox], acc);
        }
        gx[((size_t)t * g.Hi + iy) * g.Wi + ix] = acc;
    }
}

// Gather form (deterministic, no atomics, no zero fill): one thread per INPUT pixel; the outputs whose window
// [floor(o*I/O), ceil((o+1)*I/O)) contains it are o in [floor(i*O/I), ceil((i+1)*O/I) - 1] (1-2 per axis when pooling down,
// 2-3 when the "pool" enlarges the map, the 2.0 / 1.5 pyramid scales).
__global__ __launch_bounds__(256) void adaptive_avgpool_bwd_kernel(const float* __restrict__ gy, RsG g, float* __restrict__ gx, int64_t total) {
    const int t = blockIdx.z * gridDim.y + blockIdx.y;      // plane n*C + c
    const int pi = blockIdx.x * 256 + threadIdx.x;
    if (t >= g.N * g.C || pi >= g.Hi * g.Wi) return;
    const float* gp = gy + (size_t)t * g.Ho * g.Wo;
    // divisions by the four (launch-constant) sizes through mul-hi magics: exact while numerator * divisor < 2^32, which the
    // launcher checks; a 32-bit hardware-less division is ~30 instructions, and this kernel does ten of them per pixel
    const unsigned Hi = g.Hi, Wi = g.Wi, Ho = g.Ho, Wo = g.Wo;
    const unsigned uy = udiv_magic(pi, Wi, g.mWi), ux = pi - uy * Wi;
    const int oy0 = (int)udiv_magic(uy * Ho, Hi, g.mHi), oy1 = min(g.Ho - 1, (int)udiv_magic((uy + 1) * Ho + Hi - 1, Hi, g.mHi) - 1);
    const int ox0 = (int)udiv_magic(ux * Wo, Wi, g.mWi), ox1 = min(g.Wo - 1, (int)udiv_magic((ux + 1) * Wo + Wi - 1, Wi, g.mWi) - 1);
    float acc = 0.f;
    for (int oy = oy0; oy <= oy1; ++oy) {
        const int ys = (int)udiv_magic((unsigned)oy * Hi, Ho, g.mHo), ye = (int)udiv_magic(((unsigned)oy + 1) * Hi + Ho - 1, Ho, g.mHo);
        if ((int)uy < ys || (int)uy >= ye) continue;
        for (int ox = ox0; ox <= ox1; ++ox) {
            const int xs = (int)udiv_magic((unsigned)ox * Wi, Wo, g.mWo), xe = (int)udiv_magic(((unsigned)ox + 1) * Wi + Wo - 1, Wo, g.mWo);
            if ((int)ux < xs || (int)ux >= xe) continue;
            acc += gp[(size_t)oy * g.Wo + ox] / (float)((ye - ys) * (xe - xs));
        }
    }
    gx[(size_t)t * g.Hi * g.Wi + pi] = acc;
}

// ---- per-plane dot product: out[n*C + c] = sum_p a[n,c,p] * b[n,c,p]   (gate gradient; b == nullptr: plain sum)
__global__ __launch_bounds__(256) void plane_dot_kernel(const float* __restrict__ a, const float* __restrict__ b, int HW,
                                                        float* __restrict__ out) {
    const size_t base = (size_t)blockIdx.x * HW;
    float s = 0.f;
    if ((HW & 3) == 0 && (((uintptr_t)a | (uintptr_t)b) & 15) == 0) {          // 16-byte operands, four quads per operand in flight
        const float4* a4 = reinterpret_cast<const float4*>(a + base);
        const float4* b4 = b ? reinterpret_cast<const float4*>(b + base) : nullptr;
        const int L = HW >> 2;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        const float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll 4
        for (int i = threadIdx.x; i < L; i += 256) {
            const float4 x = a4[i], y = b4 ? b4[i] : one;
            s0 = fmaf(x.x, y.x, s0); s1 = fmaf(x.y, y.y, s1); s2 = fmaf(x.z, y.z, s2); s3 = fmaf(x.w, y.w, s3);
        }
        s = (s0 + s1) + (s2 + s3);
    } else
    for (int i = threadIdx.x; i < HW; i += 256) s = fmaf(a[base + i], b ? b[base + i] : 1.f, s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

// ---- EfficientPWConv gate backward: gate = sigmoid(W . mean); given ggate (N,Cout):
//      gs = ggate * gate * (1 - gate); gW[co,ci] = sum_n gs[n,co] * mean[n,ci]; gmean[n,ci] = sum_co gs[n,co] * W[co,ci]
template <bool ACC>
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ ggate, const float* __restrict__ gate,
                                                       const float* __restrict__ mean, const float* __restrict__ w, int N,
                                                       int Cin, int Cout, float* __restrict__ gw, float* __restrict__ gmean) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < Cout * Cin) {
        const int co = idx / Cin, ci = idx - co * Cin;
        float s = 0.f;
        // (unrolled: the loads of eight iterations leave together; one iteration at a time the two loops were 16 + Cout dependent L2
        // round trips: 20-31 us per launch for ~1 MB)
#pragma unroll 8
        for (int n = 0; n < N; ++n) {
            const float gt = gate[n * Cout + co];
            s = fmaf(ggate[n * Cout + co] * gt * (1.f - gt), mean[n * Cin + ci], s);
        }
        if (ACC) atomicAdd(&gw[idx], s); else gw[idx] = s;      // ACC: gw is the parameter's gradient buffer (other launches add to it too)
    }
    if (idx < N * Cin) {
        const int n = idx / Cin, ci = idx - n * Cin;
        float s = 0.f;
#pragma unroll 8
        for (int co = 0; co < Cout; ++co) {
            const float gt = gate[n * Cout + co];
            s = fmaf(ggate[n * Cout + co] * gt * (1.f - gt), w[co * Cin + ci], s);
        }
        gmean[idx] = s;
    }
}

// ---- broadcast a per-plane value over the plane (gx = v[n*C+c] * mul), optionally accumulating
__global__ __launch_bounds__(256) void plane_broadcast_kernel(const float* __restrict__ v, int HW, float mul, int accumulate,
                                                              float* __restrict__ gx, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const float val = v[idx / HW] * mul;
    if (accumulate) gx[idx] += val; else gx[idx] = val;
}

// ---- K11: loss = 20 * mean_pix( -w[t] * log_softmax(pred + 0.5 aux)[t] * exp(-kld) ) + mean_pix(kld),
//      kld = KL(softmax(pred) || softmax(aux)) per pixel (NOT detached: gradients flow through it, uest:1020-1023).
//      One thread per pixel: forward terms and the closed-form gradients w.r.t. pred and aux.
__global__ __launch_bounds__(256) void uw_loss_kernel(const float* __restrict__ pred, const float* __restrict__ aux,
                                                      const int64_t* __restrict__ target, const float* __restrict__ cw,
                                                      int N, int C, int HW, float ce_scale, float inv_npix,
                                                      float* __restrict__ loss_acc, float* __restrict__ gpred,
                                                      float* __restrict__ gaux, float* __restrict__ kld_out) {
    // grid-stride over the pixels with a bounded grid: every workgroup ends with ONE atomic on the same address, and a chain of
    // same-address device atomics advances at ~12-25 ns per link (one workgroup per 256 pixels: 7 680 links at 16 x 256x480)
    const int64_t total = (int64_t)N * HW;
    float contrib = 0.f;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int n = (int)(idx / HW), p = (int)(idx - (int64_t)n * HW);
        const float* pp = pred + (size_t)n * C * HW + p;
        const float* ap = aux + (size_t)n * C * HW + p;
        float m1 = -INFINITY, m2 = -INFINITY, mo = -INFINITY;
        for (int c = 0; c < C; ++c) {
            const float a = pp[(size_t)c * HW], b = ap[(size_t)c * HW];
            m1 = fmaxf(m1, a); m2 = fmaxf(m2, b); mo = fmaxf(mo, a + 0.5f * b);
        }
        float s1 = 0.f, s2 = 0.f, so = 0.f;
        for (int c = 0; c < C; ++c) {
            const float a = pp[(size_t)c * HW], b = ap[(size_t)c * HW];
            s1 += expf(a - m1); s2 += expf(b - m2); so += expf(a + 0.5f * b - mo);
        }
        const float l1 = m1 + logf(s1), l2 = m2 + logf(s2), lo = mo + logf(so);
        float kld = 0.f;
        for (int c = 0; c < C; ++c) {
            const float a = pp[(size_t)c * HW], b = ap[(size_t)c * HW];
            const float p1 = expf(a - l1);
            kld += p1 * (a - l1) - p1 * (b - l2);
        }
        const int t = (int)target[idx];
        const float wt = (t >= 0 && t < C) ? cw[t] : 0.f;
        const float ot = (t >= 0 && t < C) ? pp[(size_t)t * HW] + 0.5f * ap[(size_t)t * HW] : 0.f;
        const float nll = -(ot - lo);                      // -log_softmax(o)[t]
        const float u = expf(-kld);
        const float ce = wt * nll * u;                     // per-pixel weighted CE
        contrib += ce_scale * ce * inv_npix + kld * inv_npix;
        if (kld_out) kld_out[idx] = kld;
        if (gpred) {
            // d loss / d kld = (1 - ce_scale * ce) / npix ; d ce / d o_c = wt * u * (softmax(o)_c - [c == t])
            const float gk = (1.f - ce_scale * ce) * inv_npix;
            const float go = ce_scale * wt * u * inv_npix;
            for (int c = 0; c < C; ++c) {
                const float a = pp[(size_t)c * HW], b = ap[(size_t)c * HW];
                const float p1 = expf(a - l1), p2 = expf(b - l2), po = expf(a + 0.5f * b - lo);
                const float d = (a - l1) - (b - l2);       // log p1 - log p2
                const float dk_da = p1 * (d - kld);        // d kld / d pred_c
                const float dk_db = p1 - p2;               // d kld / d aux_c   (= -(p1 - p2) * -1 ... see derivation)
                const float dce = go * (po - (c == t ? 1.f : 0.f));
                gpred[(size_t)n * C * HW + (size_t)c * HW + p] = dce + gk * dk_da;
                gaux[(size_t)n * C * HW + (size_t)c * HW + p] = 0.5f * dce - gk * dk_db;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) contrib += __shfl_down(contrib, o, 64);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = contrib;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss_acc, (part[0] + part[1]) + (part[2] + part[3]));
}

// ---- Adam (torch.optim.Adam semantics: L2 weight decay folded into the gradient, bias correction)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                   float wd, float bc1, float bc2_sqrt) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i] + wd * p[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= (lr / bc1) * (mi / denom);
}

// ---- suffix sum over the 4 branch blocks of an (N, 4n, HW) gradient: out_k = sum_{j >= k} g_j  (HFF backward).
//      Output is branch-major (4, N, n, HW) so that every branch is a contiguous (N, n, HW) tensor.
__global__ __launch_bounds__(256) void hff_suffix_kernel(const float* __restrict__ g, int n, int HW, float* __restrict__ out,
                                                         int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;     // over N * n * HW
    if (idx >= total) return;
    const int64_t plane = (int64_t)n * HW;
    const int64_t img = idx / plane, r = idx - img * plane;
    const float* gp = g + img * 4 * plane + r;
    const float g3 = gp[3 * plane], g2 = gp[2 * plane] + g3, g1 = gp[plane] + g2, g0 = gp[0] + g1;
    out[idx] = g0; out[total + idx] = g1; out[2 * total + idx] = g2; out[3 * total + idx] = g3;
}

// ---- br_after_cat's BatchNorm + PReLU backward and the suffix sum in one pass (the EESP block's backward between conv_1x1_exp and
//      the four depthwise branches, nn_layers/eesp.py:76-80): z (N,4n,HW) = the raw concatenation K2 produced, gy = dL/d(PReLU(BN(z))).
//      gc = gy * (u > 0 ? 1 : alpha) * scale; out_k = sum_{m >= k} gc_m, branch-major (4,N,n,HW); per-channel sums -> d gamma / d beta /
//      d alpha (frozen BatchNorm transform as in affine_prelu_bwd_kernel).  One workgroup per (image, j, chunk): the four channels
//      k*n + j of a pixel are needed together.
__global__ __launch_bounds__(256) void hff_bn_prelu_suffix_kernel(const float* __restrict__ z, const float* __restrict__ gy,
                                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                                  const float* __restrict__ alpha, const float* __restrict__ bn_mean,
                                                                  const float* __restrict__ bn_inv, int n, int HW, int chunks,
                                                                  int64_t total, float* __restrict__ out, float* __restrict__ gscale,
                                                                  float* __restrict__ gshift, float* __restrict__ galpha,
                                                                  const float* __restrict__ stat_p, const float* __restrict__ stat_q) {
    // stat_p / stat_q (4n each, or null): br_after_cat in train(): the statistics path p * z + q joins the direct gradient before the
    // suffix sum (mspl_hff_bn_stat_suffix_bwd; the channel sums were taken by mspl_bn_train_prelu_bwd, gscale / gshift / galpha null)
    const int tix = (int)threadIdx.x, tstep = 256;
    int b = (int)blockIdx.x;
    const int chunk = b % chunks;  b /= chunks;
    const int j = b % n;
    const int img = b / n;
    float sc[4], sh[4], al[4], s_sc[4], s_sh[4], s_al[4];
    const bool act = alpha != nullptr;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ch = k * n + j;
        sc[k] = scale ? scale[ch] : 1.f;  sh[k] = shift ? shift[ch] : 0.f;  al[k] = act ? alpha[ch] : 1.f;
        s_sc[k] = s_sh[k] = s_al[k] = 0.f;
    }
    const bool stat = stat_p != nullptr;
    float pk[4] = {0.f, 0.f, 0.f, 0.f}, qk[4] = {0.f, 0.f, 0.f, 0.f};
    if (stat) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { pk[k] = stat_p[k * n + j];  qk[k] = stat_q[k * n + j]; }
    }
    const size_t in0 = ((size_t)img * 4 * n + j) * (size_t)HW, kin = (size_t)n * HW;
    const size_t out0 = ((size_t)img * n + j) * (size_t)HW;
    auto one = [&](int k, float zv, float g) {
        const float u = fmaf(zv, sc[k], sh[k]);              // K2's own epilogue expression: same PReLU branch
        const bool pos = !act || u > 0.f;
        const float gz = pos ? g : al[k] * g;
        if (!pos) s_al[k] += g * u;
        s_sc[k] += gz * zv;
        s_sh[k] += gz;
        return stat ? fmaf(zv, pk[k], qk[k]) + gz * sc[k] : gz * sc[k];
    };
    if ((HW & 3) == 0) {
        const int q4 = HW >> 2, per = (q4 + chunks - 1) / chunks;
        const int q0 = chunk * per, q1 = min(q4, q0 + per);
        for (int q = q0 + tix; q < q1; q += tstep) {
            float4 zv[4], gv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                zv[k] = reinterpret_cast<const float4*>(z + in0 + k * kin)[q];
                gv[k] = reinterpret_cast<const float4*>(gy + in0 + k * kin)[q];
            }
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 3; k >= 0; --k) {
                acc.x += one(k, zv[k].x, gv[k].x);  acc.y += one(k, zv[k].y, gv[k].y);
                acc.z += one(k, zv[k].z, gv[k].z);  acc.w += one(k, zv[k].w, gv[k].w);
                reinterpret_cast<float4*>(out + (size_t)k * total + out0)[q] = acc;
            }
        }
    } else {
        const int per = (HW + chunks - 1) / chunks;
        const int p0 = chunk * per, p1 = min(HW, p0 + per);
        for (int p = p0 + tix; p < p1; p += tstep) {
            float acc = 0.f;
#pragma unroll
            for (int k = 3; k >= 0; --k) {
                acc += one(k, z[in0 + k * kin + p], gy[in0 + k * kin + p]);
                out[(size_t)k * total + out0 + p] = acc;
            }
        }
    }
    __shared__ float part[12][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float v[3] = {s_sc[k], s_sh[k], s_al[k]};
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            v[t] = wave_sum_dpp(v[t]);                        // total in lane 63
            if ((threadIdx.x & 63) == 63) part[k * 3 + t][threadIdx.x >> 6] = v[t];
        }
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int k = threadIdx.x, ch = k * n + j;
        auto tot = [&](int i) { return (part[i][0] + part[i][1]) + (part[i][2] + part[i][3]); };
        const float t_sc = tot(k * 3), t_sh = tot(k * 3 + 1);
        if (gscale) atomicAdd(&gscale[ch], bn_inv ? (t_sc - bn_mean[ch] * t_sh) * bn_inv[ch] : t_sc);
        if (gshift) atomicAdd(&gshift[ch], t_sh);
        if (galpha && act) atomicAdd(&galpha[ch], tot(k * 3 + 2));
    }
}

static int conv_geom(const char* who, int N, int Cin, int Cout, int groups, int H, int W, int K, int stride, int dil, ConvGeom& g) {
    MSPL_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && groups > 0 && H > 0 && W > 0, MSPL_ERR_BAD_SHAPE, "%s: bad shape", who);
    MSPL_REQUIRE(Cin % groups == 0 && Cout % groups == 0, MSPL_ERR_BAD_SHAPE, "%s: channels not divisible by groups", who);
    MSPL_REQUIRE((K == 1 || K == 3) && (stride == 1 || stride == 2) && dil >= 1, MSPL_ERR_UNSUPPORTED,
                 "%s: kernel %d stride %d dilation %d", who, K, stride, dil);
    g.N = N; g.Cin = Cin; g.Cout = Cout; g.G = groups; g.cin_g = Cin / groups; g.cout_g = Cout / groups;
    g.H = H; g.W = W; g.K = K; g.stride = stride; g.dil = dil; g.pad = dil * (K - 1) / 2;
    g.Ho = (H + 2 * g.pad - dil * (K - 1) - 1) / stride + 1;
    g.Wo = (W + 2 * g.pad - dil * (K - 1) - 1) / stride + 1;
    return MSPL_OK;
}

}  // namespace mspl

using namespace mspl;

extern "C" int mspl_conv_bwd_data(const float* gy, const float* w, int32_t N, int32_t Cin, int32_t Cout, int32_t groups,
                                  int32_t H, int32_t W, int32_t K, int32_t stride, int32_t dilation, int32_t accumulate,
                                  float* gx, void* stream) {
    MSPL_REQUIRE(gy && w && gx, MSPL_ERR_NULL_POINTER, "conv_bwd_data: null pointer");
    ConvGeom g;
    if (int rc = conv_geom("conv_bwd_data", N, Cin, Cout, groups, H, W, K, stride, dilation, g)) return rc;
    const int64_t total = (int64_t)N * Cin * H * W;
    MSPL_REQUIRE(ceil_div64(total, 256) < (1ll << 31), MSPL_ERR_BAD_SHAPE, "conv_bwd_data: grid too large");
    hipLaunchKernelGGL(conv_bwd_data_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream, gy, w, g,
                       accumulate, gx, total);
    MSPL_CHECK_LAUNCH("conv_bwd_data");
    return MSPL_OK;
}

// Weight gradients of several grouped 1x1 convolutions, ACCUMULATED into gw[i] (parameter gradient buffers), in as few launches as
// the problems allow: runs of problems that fit the matrix-core kernel share one launch (csrc/conv1x1_wgrad.hip), the others go
// through mspl_conv_bwd_weight one by one.
extern "C" int mspl_conv_bwd_weight(const float* gy, const float* x, int32_t N, int32_t Cin, int32_t Cout, int32_t groups,
                                    int32_t H, int32_t W, int32_t K, int32_t stride, int32_t dilation, int32_t accumulate,
                                    float* gw, void* stream);

extern "C" int mspl_conv1x1_wgrad_batch(const float* const* gy, const float* const* x, float* const* gw, const float* const* rowscale,
                                        const int32_t* N, const int32_t* Cin, const int32_t* Cout, const int32_t* groups, const int32_t* HW,
                                        int32_t nprob, void* stream) {
    MSPL_REQUIRE(gy && x && gw && N && Cin && Cout && groups && HW, MSPL_ERR_NULL_POINTER, "conv1x1_wgrad_batch: null pointer");
    MSPL_REQUIRE(nprob >= 0 && nprob <= 4096, MSPL_ERR_BAD_SHAPE, "conv1x1_wgrad_batch: %d problems", nprob);
    hipStream_t s = (hipStream_t)stream;
    int G[16], M[16], K[16], P[16], NN[16];
    int at = 0;
    while (at < nprob) {
        int cnt = 0;
        for (; cnt < 16 && at + cnt < nprob; ++cnt) {
            const int i = at + cnt;
            MSPL_REQUIRE(gy[i] && x[i] && gw[i], MSPL_ERR_NULL_POINTER, "conv1x1_wgrad_batch: problem %d has a null pointer", i);
            MSPL_REQUIRE(N[i] > 0 && Cin[i] > 0 && Cout[i] > 0 && groups[i] > 0 && HW[i] > 0 && Cin[i] % groups[i] == 0 &&
                         Cout[i] % groups[i] == 0, MSPL_ERR_BAD_SHAPE, "conv1x1_wgrad_batch: problem %d: N=%d Cin=%d Cout=%d groups=%d HW=%d",
                         i, N[i], Cin[i], Cout[i], groups[i], HW[i]);
            NN[cnt] = N[i]; G[cnt] = groups[i]; M[cnt] = Cout[i] / groups[i]; K[cnt] = Cin[i] / groups[i]; P[cnt] = HW[i];
        }
        const int done = conv1x1_wgrad_mfma_batch(gy + at, x + at, gw + at, rowscale ? rowscale + at : nullptr, NN, G, M, K, P, cnt, s);
        if (done > 0) {
            MSPL_CHECK_LAUNCH("conv1x1_wgrad_batch");
            at += done;
        } else {                 // the first problem of the run is not for the matrix-core kernel
            MSPL_REQUIRE(!(rowscale && rowscale[at]), MSPL_ERR_UNSUPPORTED, "conv1x1_wgrad_batch: problem %d: rowscale needs the matrix-core kernel "
                         "(>= 8 channels per group, HW %% 4 == 0)", at);
            if (int rc = mspl_conv_bwd_weight(gy[at], x[at], N[at], Cin[at], Cout[at], groups[at], 1, HW[at], 1, 1, 1, 1, gw[at], stream)) return rc;
            ++at;
        }
    }
    return MSPL_OK;
}

extern "C" int mspl_conv_bwd_weight(const float* gy, const float* x, int32_t N, int32_t Cin, int32_t Cout, int32_t groups,
                                    int32_t H, int32_t W, int32_t K, int32_t stride, int32_t dilation, int32_t accumulate,
                                    float* gw, void* stream) {
    MSPL_REQUIRE(gy && x && gw, MSPL_ERR_NULL_POINTER, "conv_bwd_weight: null pointer");
    ConvGeom g;
    if (int rc = conv_geom("conv_bwd_weight", N, Cin, Cout, groups, H, W, K, stride, dilation, g)) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int64_t nw = (int64_t)Cout * g.cin_g * K * K;
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(gw, 0, (size_t)nw * sizeof(float), s);      // async memset node: graph-capturable
        MSPL_REQUIRE(e == hipSuccess, MSPL_ERR_HIP, "conv_bwd_weight: memset failed: %s", hipGetErrorString(e));
    }
    const int64_t total = (int64_t)N * g.Ho * g.Wo;
    if (K == 1) {
        static const bool no_mfma = MSPL_TUNE_INT("MSPL_WGRAD_LDS", 0) != 0;      // A/B aid: force the 16x16 LDS kernel
        if (!no_mfma && conv1x1_wgrad_mfma_try(gy, x, N, groups, g.cout_g, g.cin_g, g.Ho * g.Wo, gw, s) == 0) {
            MSPL_CHECK_LAUNCH("conv_bwd_weight(1x1, mfma)");
            return MSPL_OK;
        }
        // (the 16x16 tile is zero padded for groups with fewer channels, e.g. the 3-channel image reinforcement)
        const int tiles_m = ceil_div(g.cout_g, 16), tiles_k = ceil_div(g.cin_g, 16);
        int64_t base = (int64_t)groups * tiles_m * tiles_k;
        int chunks = 1;
        static const int minpx = MSPL_TUNE_INT("MSPL_WGRAD16_MINPX", 256);
        while (base * chunks < 2048 && total / (chunks * 2) >= minpx) chunks *= 2;
        const int64_t blocks = base * chunks;
        MSPL_REQUIRE(blocks < (1ll << 31), MSPL_ERR_BAD_SHAPE, "conv_bwd_weight: grid too large");
        hipLaunchKernelGGL(conv1x1_bwd_weight_kernel, dim3((unsigned)blocks), dim3(256), 0, s, gy, x, g, chunks, tiles_m, tiles_k, gw);
        MSPL_CHECK_LAUNCH("conv_bwd_weight(1x1)");
        return MSPL_OK;
    }
    static const int s2_strip = MSPL_TUNE_INT("MSPL_G3X3_WGRAD_S2STRIP", 1);
    if (s2_strip && K == 3 && g.cin_g == 3 && g.cout_g % 4 == 0 && Cout >= 16 && g.stride == 2 && g.dil == 1 && g.pad == 1 && (g.W & 7) == 0 &&
        (g.H & 1) == 0 && ((((uintptr_t)gy) | ((uintptr_t)x)) & 15) == 0) {
        const int64_t strips = total / 4;
        int chunks = 1;
        while ((int64_t)(Cout / 2) * chunks < 2048 && strips / (chunks * 2) >= 512) chunks *= 2;
        const int64_t pairs8 = ((int64_t)g.G * chunks + 7) & ~7ll;
        hipLaunchKernelGGL((g3x3_bwd_weight_s2_strip_kernel<3, 2>), dim3((unsigned)(pairs8 * (g.cout_g / 2))), dim3(256), 0, s, gy, x, g, chunks, gw);
        MSPL_CHECK_LAUNCH("conv_bwd_weight(3x3 stride 2, few input channels, strips)");
        return MSPL_OK;
    }
    if (K == 3 && g.cin_g == 3 && g.cout_g % 4 == 0 && Cout >= 16) {       // the stem: four output channels per workgroup share the taps
        int chunks = 1;
        while ((int64_t)(Cout / 4) * chunks < 2048 && total / (chunks * 2) >= 2048) chunks *= 2;
        hipLaunchKernelGGL((g3x3_bwd_weight_cob_kernel<3, 4>), dim3((unsigned)(Cout / 4 * chunks)), dim3(256), 0, s, gy, x, g, chunks, gw);
        MSPL_CHECK_LAUNCH("conv_bwd_weight(3x3, few input channels, 4 output channels per workgroup)");
        return MSPL_OK;
    }
    static const int strip_form = MSPL_TUNE_INT("MSPL_G3X3_WGRAD_STRIP", 1);
    if (strip_form && K == 3 && (g.cin_g <= 5 || g.cin_g == 8) && g.stride == 1 && g.dil == 1 && g.pad == 1 && (g.W & 3) == 0 &&
        ((((uintptr_t)gy) | ((uintptr_t)x)) & 15) == 0) {
        const int64_t strips = total / 4;
        int chunks = 1;
        while ((int64_t)Cout * chunks < 2048 && strips / (chunks * 2) >= 1024) chunks *= 2;
        const int64_t pairs8 = ((int64_t)g.G * chunks + 7) & ~7ll;                 // (group, chunk) pairs, padded to whole rounds of the 8 XCDs
        const dim3 grid((unsigned)(pairs8 * g.cout_g)), blk(256);
        switch (g.cin_g) {
            case 1: hipLaunchKernelGGL(g3x3_bwd_weight_strip_kernel<1>, grid, blk, 0, s, gy, x, g, chunks, gw); break;
            case 2: hipLaunchKernelGGL(g3x3_bwd_weight_strip_kernel<2>, grid, blk, 0, s, gy, x, g, chunks, gw); break;
            case 3: hipLaunchKernelGGL(g3x3_bwd_weight_strip_kernel<3>, grid, blk, 0, s, gy, x, g, chunks, gw); break;
            case 4: hipLaunchKernelGGL(g3x3_bwd_weight_strip_kernel<4>, grid, blk, 0, s, gy, x, g, chunks, gw); break;
            case 5: hipLaunchKernelGGL(g3x3_bwd_weight_strip_kernel<5>, grid, blk, 0, s, gy, x, g, chunks, gw); break;
            default: hipLaunchKernelGGL(g3x3_bwd_weight_strip_kernel<8>, grid, blk, 0, s, gy, x, g, chunks, gw); break;
        }
        MSPL_CHECK_LAUNCH("conv_bwd_weight(3x3, few input channels, strips)");
        return MSPL_OK;
    }
    if (K == 3 && (g.cin_g <= 5 || g.cin_g == 8)) {
        int chunks = 1;
        while ((int64_t)Cout * chunks < 4096 && total / (chunks * 2) >= 2048) chunks *= 2;
        const dim3 grid((unsigned)(Cout * chunks)), blk(256);
        switch (g.cin_g) {
            case 1: hipLaunchKernelGGL(g3x3_bwd_weight_kernel<1>, grid, blk, 0, s, gy, x, g, chunks, gw); break;
            case 2: hipLaunchKernelGGL(g3x3_bwd_weight_kernel<2>, grid, blk, 0, s, gy, x, g, chunks, gw); break;
            case 3: hipLaunchKernelGGL(g3x3_bwd_weight_kernel<3>, grid, blk, 0, s, gy, x, g, chunks, gw); break;
            case 4: hipLaunchKernelGGL(g3x3_bwd_weight_kernel<4>, grid, blk, 0, s, gy, x, g, chunks, gw); break;
            case 5: hipLaunchKernelGGL(g3x3_bwd_weight_kernel<5>, grid, blk, 0, s, gy, x, g, chunks, gw); break;
            default: hipLaunchKernelGGL(g3x3_bwd_weight_kernel<8>, grid, blk, 0, s, gy, x, g, chunks, gw); break;
        }
        MSPL_CHECK_LAUNCH("conv_bwd_weight(3x3, few input channels)");
        return MSPL_OK;
    }
    int chunks = 1;
    while (nw * chunks < 2048 && total / (chunks * 2) >= 4096) chunks *= 2;
    const int64_t blocks = nw * chunks;
    MSPL_REQUIRE(blocks < (1ll << 31), MSPL_ERR_BAD_SHAPE, "conv_bwd_weight: grid too large");
    hipLaunchKernelGGL(conv_bwd_weight_kernel, dim3((unsigned)blocks), dim3(256), 0, s, gy, x, g, chunks, gw);
    MSPL_CHECK_LAUNCH("conv_bwd_weight");
    return MSPL_OK;
}

static int affine_prelu_bwd_launch(const float* c, const float* pre_add, const float* residual, const float* gy,
                                   const float* scale, const float* shift, const float* alpha, int32_t N, int32_t C,
                                   int32_t HW, float* gz, float* gc, float* gscale, float* gshift, float* galpha,
                                   const float* bn_mean, const float* bn_inv, void* stream, BnTail tl = BnTail()) {
    MSPL_REQUIRE(c && gy, MSPL_ERR_NULL_POINTER, "affine_prelu_bwd: null pointer");
    MSPL_REQUIRE(N > 0 && C > 0 && HW > 0, MSPL_ERR_BAD_SHAPE, "affine_prelu_bwd: bad shape N=%d C=%d HW=%d", N, C, HW);
    // ~MSPL_AFF_BLOCKS workgroups, at least ~2 units per thread, and as few same-address atomic chains per channel as that allows.
    // 384 (1.5 per CU; four 16-byte loads per operand in flight per thread): inside the training steps, where two micro-batch lanes
    // run side by side, the leaner grid is the faster one (same box, alternating: 768 / 512 / 384 / 256 -> uest step 5.73 / 5.68 /
    // 5.665 / 5.67 ms, supervised 8.73 / 8.68 / 8.65 / 8.68 ms; rounds 2-4 ran 768, tuned on the kernel alone)
    static const int target = MSPL_TUNE_INT("MSPL_AFF_BLOCKS", 384);
    const int64_t units = (int64_t)N * ((HW & 3) == 0 ? HW / 4 : HW);
    int64_t parts = ceil_div64(target, C);
    if (parts > units / 512) parts = units / 512;
    if (parts < 1) parts = 1;
    const int64_t blocks = (int64_t)C * parts;
    MSPL_REQUIRE(blocks < (1ll << 31), MSPL_ERR_BAD_SHAPE, "affine_prelu_bwd: grid too large");
    tl.per_channel = (unsigned)parts;
    hipLaunchKernelGGL(affine_prelu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, c, pre_add, residual, gy,
                       scale, shift, alpha, N, C, HW, (int)parts, gz, gc, gscale, gshift, galpha, bn_mean, bn_inv, tl);
    MSPL_CHECK_LAUNCH("affine_prelu_bwd");
    return MSPL_OK;
}

// ---- DownSampler tail (nn_layers/eesp.py:131-144): y = PReLU(cat[a, b] + reinf) without the concatenation.  a (N,nin,HW): the
// pooled input, b (N,C-nin,HW): the strided EESP branch, reinf (N,C,HW) or null.  Forward: one pass reading the two sources where
// torch.cat + add + PReLU made three; backward: dL/da, dL/db land in two CONTIGUOUS tensors (the slices of one gradient that
// autograd's CatBackward hands out had to be copied before our kernels could take them) and d alpha is reduced on the way.
__global__ __launch_bounds__(256) void down_tail_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            const float* __restrict__ r, const float* __restrict__ alpha, int nin, int C,
                                                            int HW4, float* __restrict__ y) {
    const int plane = blockIdx.z * gridDim.y + blockIdx.y;              // n * C + c
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int n = plane / C, c = plane - n * C;
    if (q >= HW4) return;
    const float4* src = reinterpret_cast<const float4*>(c < nin ? a + ((size_t)n * nin + c) * (size_t)HW4 * 4
                                                                : b + ((size_t)n * (C - nin) + (c - nin)) * (size_t)HW4 * 4);
    float4 v = src[q];
    if (r) {
        const float4 t = reinterpret_cast<const float4*>(r + (size_t)plane * HW4 * 4)[q];
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    const float al = alpha[c];
    v.x = v.x > 0.f ? v.x : al * v.x; v.y = v.y > 0.f ? v.y : al * v.y;
    v.z = v.z > 0.f ? v.z : al * v.z; v.w = v.w > 0.f ? v.w : al * v.w;
    reinterpret_cast<float4*>(y + (size_t)plane * HW4 * 4)[q] = v;
}

__global__ __launch_bounds__(256) void down_tail_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            const float* __restrict__ r, const float* __restrict__ gy,
                                                            const float* __restrict__ alpha, int nin, int C, int HW4, int chunks,
                                                            float* __restrict__ ga, float* __restrict__ gb, float* __restrict__ gr,
                                                            float* __restrict__ galpha) {
    int bid = blockIdx.x;
    const int chunk = bid % chunks;  bid /= chunks;
    const int c = bid % C, n = bid / C;
    const size_t soff = (c < nin ? ((size_t)n * nin + c) : ((size_t)n * (C - nin) + (c - nin))) * (size_t)HW4;
    const float4* src = reinterpret_cast<const float4*>(c < nin ? a : b) + soff;
    float4* gsrc = reinterpret_cast<float4*>(c < nin ? ga : gb) + soff;
    const size_t poff = ((size_t)n * C + c) * (size_t)HW4;
    const float4* r4 = r ? reinterpret_cast<const float4*>(r) + poff : nullptr;
    const float4* g4 = reinterpret_cast<const float4*>(gy) + poff;
    float4* gr4 = gr ? reinterpret_cast<float4*>(gr) + poff : nullptr;
    const float al = alpha[c];
    const int per = (HW4 + chunks - 1) / chunks, q0 = chunk * per, q1 = min(HW4, q0 + per);
    float s_alpha = 0.f;
    auto one = [&](float z, float g) { if (z > 0.f) return g; s_alpha += g * z; return al * g; };
    for (int q = q0 + threadIdx.x; q < q1; q += 256) {
        float4 z = src[q];
        if (r4) { const float4 t = r4[q]; z.x += t.x; z.y += t.y; z.z += t.z; z.w += t.w; }
        const float4 g = g4[q];
        float4 gz;
        gz.x = one(z.x, g.x); gz.y = one(z.y, g.y); gz.z = one(z.z, g.z); gz.w = one(z.w, g.w);
        gsrc[q] = gz;
        if (gr4) gr4[q] = gz;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s_alpha += __shfl_down(s_alpha, o, 64);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s_alpha;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(galpha + c, (part[0] + part[1]) + (part[2] + part[3]));
}

extern "C" int mspl_down_tail_fwd(const float* a, const float* b, const float* reinf, const float* alpha, int32_t N, int32_t nin,
                                  int32_t C, int32_t HW, float* y, void* stream) {
    MSPL_REQUIRE(a && b && alpha && y, MSPL_ERR_NULL_POINTER, "down_tail_fwd: null pointer");
    MSPL_REQUIRE(N > 0 && nin > 0 && nin < C && HW > 0, MSPL_ERR_BAD_SHAPE, "down_tail_fwd: bad shape N=%d nin=%d C=%d HW=%d", N, nin, C, HW);
    MSPL_REQUIRE((HW & 3) == 0 && ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)reinf | (uintptr_t)y)) & 15) == 0, MSPL_ERR_UNSUPPORTED,
                 "down_tail_fwd: planes must be a multiple of 4 pixels and 16-byte aligned");
    const int64_t planes = (int64_t)N * C;
    MSPL_REQUIRE(planes < 65535ll * 65535ll, MSPL_ERR_BAD_SHAPE, "down_tail_fwd: too many planes");
    const int gy = planes < 65535 ? (int)planes : 65535;
    MSPL_REQUIRE(planes % gy == 0 || planes < 65535, MSPL_ERR_UNSUPPORTED, "down_tail_fwd: plane count %lld", (long long)planes);
    const dim3 grid((unsigned)ceil_div(HW / 4, 256), (unsigned)gy, (unsigned)(planes / gy));
    hipLaunchKernelGGL(down_tail_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, a, b, reinf, alpha, nin, C, HW / 4, y);
    MSPL_CHECK_LAUNCH("down_tail_fwd");
    return MSPL_OK;
}

extern "C" int mspl_down_tail_bwd(const float* a, const float* b, const float* reinf, const float* gy, const float* alpha, int32_t N,
                                  int32_t nin, int32_t C, int32_t HW, float* ga, float* gb, float* greinf, float* galpha, void* stream) {
    MSPL_REQUIRE(a && b && gy && alpha && ga && gb && galpha, MSPL_ERR_NULL_POINTER, "down_tail_bwd: null pointer");
    MSPL_REQUIRE((reinf == nullptr) == (greinf == nullptr), MSPL_ERR_NULL_POINTER, "down_tail_bwd: reinf and greinf go together");
    MSPL_REQUIRE(N > 0 && nin > 0 && nin < C && HW > 0, MSPL_ERR_BAD_SHAPE, "down_tail_bwd: bad shape N=%d nin=%d C=%d HW=%d", N, nin, C, HW);
    MSPL_REQUIRE((HW & 3) == 0 && ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)reinf | (uintptr_t)gy | (uintptr_t)ga | (uintptr_t)gb |
                                     (uintptr_t)greinf)) & 15) == 0, MSPL_ERR_UNSUPPORTED,
                 "down_tail_bwd: planes must be a multiple of 4 pixels and 16-byte aligned");
    int chunks = 1;
    while ((int64_t)N * C * chunks < 4096 && HW / 4 / (chunks * 2) >= 512) chunks *= 2;
    const int64_t blocks = (int64_t)N * C * chunks;
    MSPL_REQUIRE(blocks < (1ll << 31), MSPL_ERR_BAD_SHAPE, "down_tail_bwd: grid too large");
    hipLaunchKernelGGL(down_tail_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, reinf, gy, alpha, nin, C, HW / 4,
                       chunks, ga, gb, greinf, galpha);
    MSPL_CHECK_LAUNCH("down_tail_bwd");
    return MSPL_OK;
}

extern "C" int mspl_affine_prelu_bwd(const float* c, const float* pre_add, const float* residual, const float* gy,
                                     const float* scale, const float* shift, const float* alpha, int32_t N, int32_t C,
                                     int32_t HW, float* gz, float* gc, float* gscale, float* gshift, float* galpha,
                                     void* stream) {
    return affine_prelu_bwd_launch(c, pre_add, residual, gy, scale, shift, alpha, N, C, HW, gz, gc, gscale, gshift, galpha, nullptr,
                                   nullptr, stream);
}

extern "C" int mspl_bn_train_prelu_bwd(const float* z, const float* residual, const float* gy, const float* scale, const float* shift,
                                       const float* alpha, const float* gamma, const float* mean, const float* invstd, int32_t N, int32_t C,
                                       int32_t HW, float* gres, float* gc, void* ws_zeroed, int32_t accumulate, float* ggamma, float* gbeta,
                                       float* galpha, float* p, float* q, void* stream) {
    MSPL_REQUIRE(scale && shift && gamma && mean && invstd && ws_zeroed && ggamma && gbeta && p && q, MSPL_ERR_NULL_POINTER,
                 "bn_train_prelu_bwd: null pointer");
    MSPL_REQUIRE(((uintptr_t)ws_zeroed & 7) == 0, MSPL_ERR_BAD_SHAPE, "bn_train_prelu_bwd: workspace must be 8-byte aligned");
    // the backward's part of the BatchNorm's persistent workspace (mspl_bn_fused_workspace_bytes): behind the forward's 2C doubles + C counters
    float* sums = reinterpret_cast<float*>(static_cast<char*>(ws_zeroed) + (size_t)C * 20 + ((8 - ((size_t)C * 20) % 8) % 8));
    BnTail tl = BnTail();
    tl.cnt = reinterpret_cast<unsigned*>(sums + 2 * (size_t)C);
    tl.gamma = gamma; tl.mean = mean; tl.invstd = invstd;
    tl.inv_m = (float)(1.0 / ((double)N * (double)HW)); tl.accumulate = accumulate;
    tl.ggamma = ggamma; tl.gbeta = gbeta; tl.pc = p; tl.qc = q;
    return affine_prelu_bwd_launch(z, nullptr, residual, gy, scale, shift, alpha, N, C, HW, gres, gc, sums, sums + C, galpha, nullptr, nullptr,
                                   stream, tl);
}

extern "C" int mspl_bn_prelu_bwd(const float* c, const float* pre_add, const float* residual, const float* gy, const float* scale,
                                 const float* shift, const float* alpha, const float* bn_mean, const float* bn_inv, int32_t N,
                                 int32_t C, int32_t HW, float* gz, float* gc, float* ggamma, float* gbeta, float* galpha,
                                 void* stream) {
    MSPL_REQUIRE(scale && shift && bn_mean && bn_inv, MSPL_ERR_NULL_POINTER, "bn_prelu_bwd: null pointer");
    return affine_prelu_bwd_launch(c, pre_add, residual, gy, scale, shift, alpha, N, C, HW, gz, gc, ggamma, gbeta, galpha, bn_mean,
                                   bn_inv, stream);
}

static int rs_geom(const char* who, const float* gy, float* gx, int N, int C, int Hi, int Wi, int Ho, int Wo, RsG& g) {
    MSPL_REQUIRE(gy && gx, MSPL_ERR_NULL_POINTER, "%s: null pointer", who);
    MSPL_REQUIRE(N > 0 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, MSPL_ERR_BAD_SHAPE, "%s: bad shape", who);
    g.N = N; g.C = C; g.Hi = Hi; g.Wi = Wi; g.Ho = Ho; g.Wo = Wo;
    g.sh = bilinear_scale(Hi, Ho); g.sw = bilinear_scale(Wi, Wo);
    auto magic = [](int d) { return (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); };
    g.mHi = magic(Hi); g.mWi = magic(Wi); g.mHo = magic(Ho); g.mWo = magic(Wo);
    return MSPL_OK;
}

extern "C" int mspl_avgpool3x3s2_bwd(const float* gy, int32_t N, int32_t C, int32_t H, int32_t W, float* gx, void* stream) {
    RsG g;
    if (int rc = rs_geom("avgpool3x3s2_bwd", gy, gx, N, C, H, W, (H - 1) / 2 + 1, (W - 1) / 2 + 1, g)) return rc;
    const int64_t total = (int64_t)N * C * H * W;
    MSPL_REQUIRE((int64_t)H * W * W < (1ll << 32), MSPL_ERR_BAD_SHAPE, "avgpool3x3s2_bwd: plane too large for 32-bit index arithmetic");
    const int planes = N * C, gyd = planes < 65535 ? planes : 65535;
    if ((H & 1) == 0 && (W & 7) == 0 && ((((uintptr_t)gy) | ((uintptr_t)gx)) & 15) == 0) {      // whole 16-byte strips on both sides
        const int XS = g.Wo / 4;
        hipLaunchKernelGGL(avgpool3x3s2_bwd_vec_kernel, dim3((unsigned)ceil_div(g.Ho * XS, 256), (unsigned)gyd, (unsigned)ceil_div(planes, gyd)),
                           dim3(256), 0, (hipStream_t)stream, gy, g, XS, gx);
        MSPL_CHECK_LAUNCH("avgpool3x3s2_bwd(16-byte strips)");
        return MSPL_OK;
    }
    hipLaunchKernelGGL(avgpool3x3s2_bwd_kernel, dim3((unsigned)ceil_div(H * W, 256), (unsigned)gyd, (unsigned)ceil_div(planes, gyd)), dim3(256), 0,
                       (hipStream_t)stream, gy, g, gx, total);
    MSPL_CHECK_LAUNCH("avgpool3x3s2_bwd");
    return MSPL_OK;
}

/* Gather form: gx is overwritten. */
extern "C" int mspl_bilinear_bwd(const float* gy, int32_t N, int32_t C, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo,
                                 float* gx, void* stream) {
    RsG g;
    if (int rc = rs_geom("bilinear_bwd", gy, gx, N, C, Hi, Wi, Ho, Wo, g)) return rc;
    const int64_t total = (int64_t)N * C * Hi * Wi;
    // candidate window per input pixel is 2/scale + 3 wide: beyond 12 columns a wave per input pixel is the better shape
    const bool wide = g.sw <= 0.f || 2.0f / g.sw + 3.0f > 12.0f;
    static const int rows_form = MSPL_TUNE_INT("MSPL_BILINEAR_BWD_ROWS", 1);
    if (wide && rows_form && Wo <= 8192 && (int64_t)N * C * Hi < (1ll << 31)) {
        hipLaunchKernelGGL(bilinear_bwd_rows_kernel, dim3((unsigned)((int64_t)N * C * Hi)), dim3(256), (size_t)Wo * sizeof(float),
                           (hipStream_t)stream, gy, g, gx);
        MSPL_CHECK_LAUNCH("bilinear_bwd(rows)");
        return MSPL_OK;
    }
    if (wide && total * 64 < (1ll << 31) * 256ll) {
        hipLaunchKernelGGL(bilinear_bwd_wave_kernel, dim3((unsigned)ceil_div64(total * 64, 256)), dim3(256), 0, (hipStream_t)stream, gy, g,
                           gx, total);
        MSPL_CHECK_LAUNCH("bilinear_bwd(wave)");
        return MSPL_OK;
    }
    const int planes = N * C, gyd = planes < 65535 ? planes : 65535;
    static const int tile_form = MSPL_TUNE_INT("MSPL_BILINEAR_BWD_TILE", 1);
    if (tile_form && g.sh > 0.f && g.sw > 0.f) {
        // candidate windows (rows and columns) of at most 8 (x2) or 12 (x4): the tiled separable form
        const float cw = 2.0f / g.sw + 3.0f, rw = 2.0f / g.sh + 3.0f;
        const int maxc = (cw <= 8.0f && rw <= 8.0f) ? 8 : ((cw <= 12.0f && rw <= 12.0f) ? 12 : 0);
        if (maxc) {
            const int TLH = maxc == 8 ? 8 : 4, TLW = 64;
            const int tiles_x = ceil_div(Wi, TLW), tiles_y = ceil_div(Hi, TLH);
            auto lo = [](float sc, int i) { const int v = (int)ceilf((float)(i - 1) / sc) - 1; return v > 0 ? v : 0; };
            auto hi = [](float sc, int i, int O) { const int v = (int)floorf((float)(i + 1) / sc) + 1; return v < O - 1 ? v : O - 1; };
            int FH = 1, FW = 1;
            for (int ty = 0; ty < tiles_y; ++ty) {
                const int i0 = ty * TLH, il = (i0 + TLH < Hi ? i0 + TLH : Hi) - 1;
                FH = std::max(FH, hi(g.sh, il, Ho) - lo(g.sh, i0) + 1);
            }
            for (int tx = 0; tx < tiles_x; ++tx) {
                const int i0 = tx * TLW, il = (i0 + TLW < Wi ? i0 + TLW : Wi) - 1;
                FW = std::max(FW, hi(g.sw, il, Wo) - lo(g.sw, i0) + 1);
            }
            const int FWS = ((FW + maxc + 3) & ~3) | 4;            // row stride: staged columns + the candidates' over-read, not a multiple of 8 floats
            const size_t lds = ((size_t)FH * FWS + (size_t)(FH + maxc) * TLW + (size_t)maxc * TLW + (size_t)TLH * maxc + TLW + TLH) * sizeof(float);
            if (lds <= 64 * 1024 && (int64_t)tiles_x * tiles_y < (1ll << 31)) {
                const dim3 tgrid((unsigned)(tiles_x * tiles_y), (unsigned)gyd, (unsigned)ceil_div(planes, gyd));
                if (maxc == 8) hipLaunchKernelGGL((bilinear_bwd_tile_kernel<8, 8>), tgrid, dim3(256), lds, (hipStream_t)stream, gy, g, FH, FWS, tiles_x, gx);
                else hipLaunchKernelGGL((bilinear_bwd_tile_kernel<12, 4>), tgrid, dim3(256), lds, (hipStream_t)stream, gy, g, FH, FWS, tiles_x, gx);
                MSPL_CHECK_LAUNCH("bilinear_bwd(tiled)");
                return MSPL_OK;
            }
        }
    }
    const dim3 grid((unsigned)ceil_div(Hi * Wi, 256), (unsigned)gyd, (unsigned)ceil_div(planes, gyd));
    if (g.sw > 0.f && 2.0f / g.sw + 3.0f <= 8.0f)          // x2 up-sampling (the decoder, the heads): at most 7 candidate columns
        hipLaunchKernelGGL(bilinear_bwd_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, gy, g, gx, total);
    else
        hipLaunchKernelGGL(bilinear_bwd_kernel<12>, grid, dim3(256), 0, (hipStream_t)stream, gy, g, gx, total);
    MSPL_CHECK_LAUNCH("bilinear_bwd");
    return MSPL_OK;
}

/* Gather form: gx is overwritten. */
extern "C" int mspl_adaptive_avgpool_bwd(const float* gy, int32_t N, int32_t C, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo,
                                         float* gx, void* stream) {
    RsG g;
    if (int rc = rs_geom("adaptive_avgpool_bwd", gy, gx, N, C, Hi, Wi, Ho, Wo, g)) return rc;
    // numerators reach (Hi+1)*Ho + Hi resp. Hi*Wi; the magics are exact while numerator * divisor < 2^32
    MSPL_REQUIRE(((int64_t)(Hi + 1) * Ho + Hi) * (Hi > Ho ? Hi : Ho) < (1ll << 32) && ((int64_t)(Wi + 1) * Wo + Wi) * (Wi > Wo ? Wi : Wo) < (1ll << 32) &&
                     (int64_t)Hi * Wi * Wi < (1ll << 32),
                 MSPL_ERR_BAD_SHAPE, "adaptive_avgpool_bwd: map too large for the 32-bit window arithmetic");
    const int64_t total = (int64_t)N * C * Hi * Wi;
    const int planes = N * C, gyd = planes < 65535 ? planes : 65535;
    hipLaunchKernelGGL(adaptive_avgpool_bwd_kernel, dim3((unsigned)ceil_div(Hi * Wi, 256), (unsigned)gyd, (unsigned)ceil_div(planes, gyd)),
                       dim3(256), 0, (hipStream_t)stream, gy, g, gx, total);
    MSPL_CHECK_LAUNCH("adaptive_avgpool_bwd");
    return MSPL_OK;
}

extern "C" int mspl_plane_dot(const float* a, const float* b, int32_t planes, int32_t HW, float* out, void* stream) {
    MSPL_REQUIRE(a && out, MSPL_ERR_NULL_POINTER, "plane_dot: null pointer");
    MSPL_REQUIRE(planes > 0 && HW > 0, MSPL_ERR_BAD_SHAPE, "plane_dot: bad shape");
    hipLaunchKernelGGL(plane_dot_kernel, dim3((unsigned)planes), dim3(256), 0, (hipStream_t)stream, a, b, HW, out);
    MSPL_CHECK_LAUNCH("plane_dot");
    return MSPL_OK;
}

extern "C" int mspl_plane_broadcast(const float* v, int32_t planes, int32_t HW, float mul, int32_t accumulate, float* gx,
                                    void* stream) {
    MSPL_REQUIRE(v && gx, MSPL_ERR_NULL_POINTER, "plane_broadcast: null pointer");
    MSPL_REQUIRE(planes > 0 && HW > 0, MSPL_ERR_BAD_SHAPE, "plane_broadcast: bad shape");
    const int64_t total = (int64_t)planes * HW;
    hipLaunchKernelGGL(plane_broadcast_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream, v, HW, mul,
                       accumulate, gx, total);
    MSPL_CHECK_LAUNCH("plane_broadcast");
    return MSPL_OK;
}

extern "C" int mspl_gap_gate_bwd(const float* ggate, const float* gate, const float* mean, const float* w, int32_t N,
                                 int32_t Cin, int32_t Cout, float* gw, float* gmean, void* stream) {
    MSPL_REQUIRE(ggate && gate && mean && w && gw && gmean, MSPL_ERR_NULL_POINTER, "gap_gate_bwd: null pointer");
    MSPL_REQUIRE(N > 0 && Cin > 0 && Cout > 0, MSPL_ERR_BAD_SHAPE, "gap_gate_bwd: bad shape");
    const int n = max(Cout * Cin, N * Cin);
    hipLaunchKernelGGL(gate_bwd_kernel<false>, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, ggate, gate, mean, w, N,
                       Cin, Cout, gw, gmean);
    MSPL_CHECK_LAUNCH("gap_gate_bwd");
    return MSPL_OK;
}

extern "C" int mspl_gap_gate_bwd_accum(const float* ggate, const float* gate, const float* mean, const float* w, int32_t N,
                                       int32_t Cin, int32_t Cout, float* gw, float* gmean, void* stream) {
    MSPL_REQUIRE(ggate && gate && mean && w && gw && gmean, MSPL_ERR_NULL_POINTER, "gap_gate_bwd_accum: null pointer");
    MSPL_REQUIRE(N > 0 && Cin > 0 && Cout > 0, MSPL_ERR_BAD_SHAPE, "gap_gate_bwd_accum: bad shape");
    const int n = max(Cout * Cin, N * Cin);
    hipLaunchKernelGGL(gate_bwd_kernel<true>, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, ggate, gate, mean, w, N,
                       Cin, Cout, gw, gmean);
    MSPL_CHECK_LAUNCH("gap_gate_bwd_accum");
    return MSPL_OK;
}

/* loss_acc (1 float, device) is ACCUMULATED into (caller zeroes).  gpred/gaux/kld_out may be NULL (forward only). */
static int uw_loss_launch(const float* pred, const float* aux, const int64_t* target, const float* class_weights, int32_t N, int32_t C,
                          int32_t HW, float ce_scale, float out_scale, float* loss_acc, float* gpred, float* gaux, float* kld_out,
                          void* stream) {
    MSPL_REQUIRE(pred && aux && target && class_weights && loss_acc, MSPL_ERR_NULL_POINTER, "uw_loss: null pointer");
    MSPL_REQUIRE((gpred == nullptr) == (gaux == nullptr), MSPL_ERR_NULL_POINTER, "uw_loss: gpred and gaux go together");
    MSPL_REQUIRE(N > 0 && C > 0 && HW > 0, MSPL_ERR_BAD_SHAPE, "uw_loss: bad shape N=%d C=%d HW=%d", N, C, HW);
    const int64_t total = (int64_t)N * HW;
    // (measured, one atomic per workgroup: 16 x 5 x 256x480 two heads 156 / 105 / 95 / 92 / 112 us at 512 / 1024 / 2048 / 4096 / 7680
    // workgroups; the 4-image micro-batches of the graphed step 32 / 28 / 36 us at 512 / 1024 / 1920)
    static const int max_blocks = MSPL_TUNE_INT("MSPL_LOSS_BLOCKS", 1024);
    const int64_t blocks = std::min<int64_t>(ceil_div64(total, 256), max_blocks);
    hipLaunchKernelGGL(uw_loss_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pred, aux, target,
                       class_weights, N, C, HW, ce_scale, out_scale / (float)total, loss_acc, gpred, gaux, kld_out);
    MSPL_CHECK_LAUNCH("uw_loss");
    return MSPL_OK;
}

extern "C" int mspl_uw_loss_fwd_bwd(const float* pred, const float* aux, const int64_t* target, const float* class_weights,
                                    int32_t N, int32_t C, int32_t HW, float ce_scale, float* loss_acc, float* gpred,
                                    float* gaux, float* kld_out, void* stream) {
    return uw_loss_launch(pred, aux, target, class_weights, N, C, HW, ce_scale, 1.0f, loss_acc, gpred, gaux, kld_out, stream);
}

extern "C" int mspl_uw_loss_scaled_fwd_bwd(const float* pred, const float* aux, const int64_t* target, const float* class_weights,
                                           int32_t N, int32_t C, int32_t HW, float ce_scale, float out_scale, float* loss_acc,
                                           float* gpred, float* gaux, float* kld_out, void* stream) {
    return uw_loss_launch(pred, aux, target, class_weights, N, C, HW, ce_scale, out_scale, loss_acc, gpred, gaux, kld_out, stream);
}

extern "C" int mspl_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                              float eps, float weight_decay, int32_t step, void* stream) {
    MSPL_REQUIRE(p && g && m && v, MSPL_ERR_NULL_POINTER, "adam_step: null pointer");
    MSPL_REQUIRE(n >= 0 && step >= 1, MSPL_ERR_BAD_SHAPE, "adam_step: n=%lld step=%d", (long long)n, step);
    if (n == 0) return MSPL_OK;
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1,
                       beta2, eps, weight_decay, bc1, bc2s);
    MSPL_CHECK_LAUNCH("adam_step");
    return MSPL_OK;
}

// Transposed (and, for 3x3, spatially flipped) copies of many convolution weights in ONE launch: the weights of the data-gradient
// convolutions of a training step (autograd.ConvFn.backward runs the data gradient of a stride-1 convolution on the forward
// kernels).  Per weight this was a permute copy (+ a flip) of a few KB issued from ATen inside the backward chain: 81 + 33
// launches of the ~1500 of a step.  seg: per weight {src, dst} pointers and {groups, cin_g, cout_g, k}; blk: per block its
// segment and the index of its first element inside it.
namespace mspl {
struct WtSeg { const float* src; float* dst; int32_t G, cin_g, cout_g, k; int32_t numel, pad; };

__global__ __launch_bounds__(256) void transpose_weights_kernel(const WtSeg* __restrict__ seg, const int2* __restrict__ blk) {
    const int2 b = blk[blockIdx.x];
    const WtSeg sg = seg[b.x];
    const int i = b.y + threadIdx.x;                   // destination index: (((g * cin_g + ci) * cout_g + co) * k + ky) * k + kx
    if (i >= sg.numel) return;
    const int kk = sg.k * sg.k;
    const int t = i / kk, r = i - t * kk;
    const int co = t % sg.cout_g, t2 = t / sg.cout_g;
    const int ci = t2 % sg.cin_g, g = t2 / sg.cin_g;
    const int rs = kk - 1 - r;                          // flip both spatial axes (identity for k = 1)
    sg.dst[i] = sg.src[(((size_t)g * sg.cout_g + co) * sg.cin_g + ci) * kk + rs];
}
}  // namespace mspl

extern "C" int mspl_transpose_weights(const void* seg_table, const void* block_table, int32_t nblocks, void* stream) {
    MSPL_REQUIRE(seg_table && block_table, MSPL_ERR_NULL_POINTER, "transpose_weights: null pointer");
    MSPL_REQUIRE(nblocks >= 0, MSPL_ERR_BAD_SHAPE, "transpose_weights: nblocks=%d", nblocks);
    if (nblocks == 0) return MSPL_OK;
    hipLaunchKernelGGL(mspl::transpose_weights_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream,
                       (const mspl::WtSeg*)seg_table, (const int2*)block_table);
    MSPL_CHECK_LAUNCH("transpose_weights");
    return MSPL_OK;
}

// out = sum of up to 8 equally shaped tensors in one launch (the gradient of a tensor with several consumers: autograd would add
// them pairwise, n - 1 launches that read 2 and write 1 tensor each).  Summation order: ((s0 + s1) + s2) + ... like the pairwise adds.
namespace mspl {
struct SumSrcs { const float* p[8]; };

__global__ __launch_bounds__(256) void sum_n_kernel(SumSrcs srcs, int n, int64_t count4, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count4) return;
    float4 a = reinterpret_cast<const float4*>(srcs.p[0])[i];
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        if (k >= n) break;
        const float4 b = reinterpret_cast<const float4*>(srcs.p[k])[i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    reinterpret_cast<float4*>(out)[i] = a;
}
}  // namespace mspl

extern "C" int mspl_sum_n(const float* const* srcs, int32_t n, int64_t count, float* out, void* stream) {
    MSPL_REQUIRE(srcs && out, MSPL_ERR_NULL_POINTER, "sum_n: null pointer");
    MSPL_REQUIRE(n >= 1 && n <= 8 && count >= 0 && (count & 3) == 0, MSPL_ERR_BAD_SHAPE, "sum_n: n=%d count=%lld (1..8 tensors, count %% 4 == 0)",
                 n, (long long)count);
    mspl::SumSrcs s;
    for (int k = 0; k < 8; ++k) {
        s.p[k] = k < n ? srcs[k] : srcs[0];
        MSPL_REQUIRE(s.p[k] && (((uintptr_t)s.p[k]) & 15) == 0, MSPL_ERR_NULL_POINTER, "sum_n: source %d is NULL or not 16-byte aligned", k);
    }
    MSPL_REQUIRE((((uintptr_t)out) & 15) == 0, MSPL_ERR_UNSUPPORTED, "sum_n: unaligned destination");
    if (count == 0) return MSPL_OK;
    hipLaunchKernelGGL(mspl::sum_n_kernel, dim3((unsigned)ceil_div64(count / 4, 256)), dim3(256), 0, (hipStream_t)stream, s, n, count / 4, out);
    MSPL_CHECK_LAUNCH("sum_n");
    return MSPL_OK;
}

namespace mspl {
// sum of n tensors + a per-plane constant (mul * pc[plane]): the gradient of a global average pool is constant over its plane, so the
// sum of an encoder output's gradients takes it as N * C values instead of a full-size tensor somebody wrote only to have it added
__global__ __launch_bounds__(256) void sum_n_planes_kernel(SumSrcs srcs, int n, const float* __restrict__ pc, float mul, int HW4,
                                                           float* __restrict__ out) {
    const int plane = blockIdx.z * gridDim.y + blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= HW4) return;
    const int64_t i = (int64_t)plane * HW4 + q;
    const float c = mul * pc[plane];
    float4 a = reinterpret_cast<const float4*>(srcs.p[0])[i];
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        if (k >= n) break;
        const float4 b = reinterpret_cast<const float4*>(srcs.p[k])[i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    a.x += c; a.y += c; a.z += c; a.w += c;
    reinterpret_cast<float4*>(out)[i] = a;
}
}  // namespace mspl

extern "C" int mspl_sum_n_planes(const float* const* srcs, int32_t n, const float* plane_const, float mul, int32_t planes, int32_t HW,
                                 float* out, void* stream) {
    MSPL_REQUIRE(srcs && out && plane_const, MSPL_ERR_NULL_POINTER, "sum_n_planes: null pointer");
    MSPL_REQUIRE(n >= 1 && n <= 8 && planes > 0 && HW > 0 && (HW & 3) == 0, MSPL_ERR_BAD_SHAPE,
                 "sum_n_planes: n=%d planes=%d HW=%d (1..8 tensors, HW %% 4 == 0)", n, planes, HW);
    mspl::SumSrcs s;
    for (int k = 0; k < 8; ++k) {
        s.p[k] = k < n ? srcs[k] : srcs[0];
        MSPL_REQUIRE(s.p[k] && (((uintptr_t)s.p[k]) & 15) == 0, MSPL_ERR_NULL_POINTER, "sum_n_planes: source %d is NULL or not 16-byte aligned", k);
    }
    MSPL_REQUIRE((((uintptr_t)out) & 15) == 0, MSPL_ERR_UNSUPPORTED, "sum_n_planes: unaligned destination");
    const int gy = planes < 65535 ? planes : 65535;
    hipLaunchKernelGGL(mspl::sum_n_planes_kernel, dim3((unsigned)ceil_div(HW / 4, 256), (unsigned)gy, (unsigned)ceil_div(planes, gy)), dim3(256), 0,
                       (hipStream_t)stream, s, n, plane_const, mul, HW / 4, out);
    MSPL_CHECK_LAUNCH("sum_n_planes");
    return MSPL_OK;
}

extern "C" int mspl_hff_suffix_sum(const float* g, int32_t N, int32_t n, int32_t HW, float* out, void* stream) {
    MSPL_REQUIRE(g && out, MSPL_ERR_NULL_POINTER, "hff_suffix_sum: null pointer");
    MSPL_REQUIRE(N > 0 && n > 0 && HW > 0, MSPL_ERR_BAD_SHAPE, "hff_suffix_sum: bad shape");
    const int64_t total = (int64_t)N * n * HW;
    hipLaunchKernelGGL(hff_suffix_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream, g, n, HW, out, total);
    MSPL_CHECK_LAUNCH("hff_suffix_sum");
    return MSPL_OK;
}

extern "C" int mspl_hff_bn_prelu_suffix_bwd(const float* z, const float* gy, const float* scale, const float* shift, const float* alpha,
                                            const float* bn_mean, const float* bn_inv, int32_t N, int32_t n, int32_t HW, float* out,
                                            float* gscale, float* gshift, float* galpha, void* stream) {
    MSPL_REQUIRE(z && gy && out, MSPL_ERR_NULL_POINTER, "hff_bn_prelu_suffix_bwd: null pointer");
    MSPL_REQUIRE((bn_mean == nullptr) == (bn_inv == nullptr), MSPL_ERR_NULL_POINTER, "hff_bn_prelu_suffix_bwd: mean / inv must come together");
    MSPL_REQUIRE(N > 0 && n > 0 && HW > 0, MSPL_ERR_BAD_SHAPE, "hff_bn_prelu_suffix_bwd: bad shape");
    const int64_t total = (int64_t)N * n * HW;
    int chunks = 1;
    while ((int64_t)N * n * chunks < 2048 && HW / (chunks * 2) >= 1024) chunks *= 2;
    const int64_t blocks = (int64_t)N * n * chunks;
    MSPL_REQUIRE(blocks < (1ll << 31), MSPL_ERR_BAD_SHAPE, "hff_bn_prelu_suffix_bwd: grid too large");
    hipLaunchKernelGGL(hff_bn_prelu_suffix_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, z, gy, scale, shift, alpha,
                       bn_mean, bn_inv, n, HW, chunks, total, out, gscale, gshift, galpha, (const float*)nullptr, (const float*)nullptr);
    MSPL_CHECK_LAUNCH("hff_bn_prelu_suffix_bwd");
    return MSPL_OK;
}

extern "C" int mspl_hff_bn_stat_suffix_bwd(const float* z, const float* gy, const float* scale, const float* shift, const float* alpha,
                                           const float* stat_p, const float* stat_q, int32_t N, int32_t n, int32_t HW, float* out,
                                           void* stream) {
    MSPL_REQUIRE(z && gy && out && scale && shift && stat_p && stat_q, MSPL_ERR_NULL_POINTER, "hff_bn_stat_suffix_bwd: null pointer");
    MSPL_REQUIRE(N > 0 && n > 0 && HW > 0, MSPL_ERR_BAD_SHAPE, "hff_bn_stat_suffix_bwd: bad shape");
    const int64_t total = (int64_t)N * n * HW;
    int chunks = 1;
    while ((int64_t)N * n * chunks < 2048 && HW / (chunks * 2) >= 1024) chunks *= 2;
    const int64_t blocks = (int64_t)N * n * chunks;
    MSPL_REQUIRE(blocks < (1ll << 31), MSPL_ERR_BAD_SHAPE, "hff_bn_stat_suffix_bwd: grid too large");
    hipLaunchKernelGGL(hff_bn_prelu_suffix_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, z, gy, scale, shift, alpha,
                       (const float*)nullptr, (const float*)nullptr, n, HW, chunks, total, out, (float*)nullptr, (float*)nullptr,
                       (float*)nullptr, stat_p, stat_q);
    MSPL_CHECK_LAUNCH("hff_bn_stat_suffix_bwd");
    return MSPL_OK;
}
