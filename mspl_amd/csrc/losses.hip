// Stand-alone loss kernels behind the reference's loss classes (loss_fns/segmentation_loss.py), forward and backward:
//   PixelwiseKLD.forward                       :181-189   kld[n,p] = sum_c softmax(d1)_c * (log_softmax(d1)_c - log_softmax(d2)_c)
//   UncertaintyWeightedSegmentationLoss.forward :155-175   mean_{n,p}( w[t] * -log_softmax(pred)[t] * exp(-u) )   (mean over ALL pixels)
//   SegmentationLoss (loss_type 'ce')          :11-52     nn.CrossEntropyLoss(weight, ignore_index): sum(w[t]*nll) / sum_{valid}(w[t])
// The fused K11 kernel (train.hip) covers the exact uest combination in one pass; these serve callers that compose
// the modules themselves.  One thread per pixel, classes streamed with a running log-sum-exp; NCHW fp32, int64 targets.
#include <algorithm>

#include "common.hpp"

namespace mspl {

struct Lse { float m, s; };
__device__ __forceinline__ void lse_push(Lse& a, float v) {
    if (v > a.m) { a.s = a.s * expf(a.m - v) + 1.f; a.m = v; } else { a.s += expf(v - a.m); }
}

__global__ __launch_bounds__(256) void kld_fwd_kernel(const float* __restrict__ d1, const float* __restrict__ d2, int C, int HW,
                                                      float* __restrict__ kld, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int n = (int)(idx / HW), p = (int)(idx - (int64_t)n * HW);
    const float* a = d1 + (size_t)n * C * HW + p;
    const float* b = d2 + (size_t)n * C * HW + p;
    Lse l1{-INFINITY, 0.f}, l2{-INFINITY, 0.f};
    for (int c = 0; c < C; ++c) { lse_push(l1, a[(size_t)c * HW]); lse_push(l2, b[(size_t)c * HW]); }
    const float z1 = l1.m + logf(l1.s), z2 = l2.m + logf(l2.s);
    float k = 0.f;
    for (int c = 0; c < C; ++c) {
        const float av = a[(size_t)c * HW] - z1, bv = b[(size_t)c * HW] - z2;
        const float p1 = expf(av);
        k += p1 * av - p1 * bv;
    }
    kld[idx] = k;
}

// d kld / d d1_c = p1_c * ((log p1_c - log p2_c) - kld);   d kld / d d2_c = p2_c - p1_c
__global__ __launch_bounds__(256) void kld_bwd_kernel(const float* __restrict__ d1, const float* __restrict__ d2,
                                                      const float* __restrict__ gk, int C, int HW, float* __restrict__ g1,
                                                      float* __restrict__ g2, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int n = (int)(idx / HW), p = (int)(idx - (int64_t)n * HW);
    const size_t base = (size_t)n * C * HW + p;
    const float* a = d1 + base;
    const float* b = d2 + base;
    Lse l1{-INFINITY, 0.f}, l2{-INFINITY, 0.f};
    for (int c = 0; c < C; ++c) { lse_push(l1, a[(size_t)c * HW]); lse_push(l2, b[(size_t)c * HW]); }
    const float z1 = l1.m + logf(l1.s), z2 = l2.m + logf(l2.s);
    float k = 0.f;
    for (int c = 0; c < C; ++c) {
        const float av = a[(size_t)c * HW] - z1, bv = b[(size_t)c * HW] - z2;
        const float p1 = expf(av);
        k += p1 * av - p1 * bv;
    }
    const float g = gk[idx];
    for (int c = 0; c < C; ++c) {
        const float av = a[(size_t)c * HW] - z1, bv = b[(size_t)c * HW] - z2;
        const float p1 = expf(av), p2 = expf(bv);
        if (g1) g1[base + (size_t)c * HW] = g * p1 * ((av - bv) - k);
        if (g2) g2[base + (size_t)c * HW] = g * (p2 - p1);
    }
}

// sums[0] += sum w[t] * nll * exp(-u);  sums[1] += sum_{t valid} w[t]
__global__ __launch_bounds__(256) void wce_fwd_kernel(const float* __restrict__ pred, const int64_t* __restrict__ target,
                                                      const float* __restrict__ u, const float* __restrict__ cw, int ignore,
                                                      int C, int HW, float* __restrict__ sums, int64_t total) {
    // grid-stride with a bounded grid: a workgroup ends with two atomics on the same two addresses, and such a chain advances at
    // ~12-25 ns per link (one workgroup per 256 pixels: 8 640 links, 231 us at 16 x 13 x 288x480; 92 us with 2 048 workgroups)
    float num = 0.f, den = 0.f;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int n = (int)(idx / HW), p = (int)(idx - (int64_t)n * HW);
        const int64_t t = target[idx];
        if (t >= 0 && t < C && t != ignore) {
            const float* a = pred + (size_t)n * C * HW + p;
            Lse l{-INFINITY, 0.f};
            for (int c = 0; c < C; ++c) lse_push(l, a[(size_t)c * HW]);
            const float nll = (l.m + logf(l.s)) - a[(size_t)t * HW];
            const float w = cw ? cw[t] : 1.f;
            num += w * nll * (u ? expf(-u[idx]) : 1.f);
            den += w;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { num += __shfl_down(num, o, 64); den += __shfl_down(den, o, 64); }
    __shared__ float part[4][2];
    if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6][0] = num; part[threadIdx.x >> 6][1] = den; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&sums[0], (part[0][0] + part[1][0]) + (part[2][0] + part[3][0]));
        atomicAdd(&sums[1], (part[0][1] + part[1][1]) + (part[2][1] + part[3][1]));
    }
}

// scale = g[0] * (den ? 1 / den[0] : inv_npix);  gpred_c = scale * w * e^{-u} * (softmax_c - [c == t]);  gu = -scale * w * nll * e^{-u}
__global__ __launch_bounds__(256) void wce_bwd_kernel(const float* __restrict__ pred, const int64_t* __restrict__ target,
                                                      const float* __restrict__ u, const float* __restrict__ cw, int ignore,
                                                      int C, int HW, const float* __restrict__ g, const float* __restrict__ den,
                                                      float inv_npix, float* __restrict__ gpred, float* __restrict__ gu,
                                                      int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int n = (int)(idx / HW), p = (int)(idx - (int64_t)n * HW);
    const size_t base = (size_t)n * C * HW + p;
    const int64_t t = target[idx];
    const bool valid = t >= 0 && t < C && t != ignore;
    const float scale = g[0] * (den ? 1.0f / den[0] : inv_npix);
    const float w = valid ? (cw ? cw[t] : 1.f) : 0.f;
    const float eu = u ? expf(-u[idx]) : 1.f;
    const float* a = pred + base;
    Lse l{-INFINITY, 0.f};
    for (int c = 0; c < C; ++c) lse_push(l, a[(size_t)c * HW]);
    const float z = l.m + logf(l.s);
    const float k = scale * w * eu;
    if (gpred)
        for (int c = 0; c < C; ++c)
            gpred[base + (size_t)c * HW] = k * (expf(a[(size_t)c * HW] - z) - ((int64_t)c == t ? 1.f : 0.f));
    if (gu) gu[idx] = valid ? -k * (z - a[(size_t)t * HW]) : 0.f;
}

}  // namespace mspl

using namespace mspl;

extern "C" int mspl_pixelwise_kld_fwd(const float* d1, const float* d2, int32_t N, int32_t C, int32_t HW, float* kld, void* stream) {
    MSPL_REQUIRE(d1 && d2 && kld, MSPL_ERR_NULL_POINTER, "pixelwise_kld: null pointer");
    MSPL_REQUIRE(N > 0 && C > 0 && HW > 0, MSPL_ERR_BAD_SHAPE, "pixelwise_kld: bad shape N=%d C=%d HW=%d", N, C, HW);
    const int64_t total = (int64_t)N * HW;
    hipLaunchKernelGGL(kld_fwd_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream, d1, d2, C, HW, kld, total);
    MSPL_CHECK_LAUNCH("pixelwise_kld_fwd");
    return MSPL_OK;
}

extern "C" int mspl_pixelwise_kld_bwd(const float* d1, const float* d2, const float* gkld, int32_t N, int32_t C, int32_t HW,
                                      float* gd1, float* gd2, void* stream) {
    MSPL_REQUIRE(d1 && d2 && gkld && (gd1 || gd2), MSPL_ERR_NULL_POINTER, "pixelwise_kld_bwd: null pointer");
    MSPL_REQUIRE(N > 0 && C > 0 && HW > 0, MSPL_ERR_BAD_SHAPE, "pixelwise_kld_bwd: bad shape N=%d C=%d HW=%d", N, C, HW);
    const int64_t total = (int64_t)N * HW;
    hipLaunchKernelGGL(kld_bwd_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream, d1, d2, gkld, C, HW,
                       gd1, gd2, total);
    MSPL_CHECK_LAUNCH("pixelwise_kld_bwd");
    return MSPL_OK;
}

extern "C" int mspl_weighted_ce_fwd(const float* pred, const int64_t* target, const float* u_weight, const float* class_weights,
                                    int32_t ignore_index, int32_t N, int32_t C, int32_t HW, float* sums, void* stream) {
    MSPL_REQUIRE(pred && target && sums, MSPL_ERR_NULL_POINTER, "weighted_ce: null pointer");
    MSPL_REQUIRE(N > 0 && C > 0 && HW > 0, MSPL_ERR_BAD_SHAPE, "weighted_ce: bad shape N=%d C=%d HW=%d", N, C, HW);
    const int64_t total = (int64_t)N * HW;
    // (measured at 16 x 13 x 288x480: 143 / 96 / 92 / 128 / 231 us at 512 / 1024 / 2048 / 4096 / 8640 workgroups)
    static const int max_blocks = MSPL_TUNE_INT("MSPL_LOSS_BLOCKS", 2048);
    const int64_t blocks = std::min<int64_t>(ceil_div64(total, 256), max_blocks);
    hipLaunchKernelGGL(wce_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pred, target,
                       u_weight, class_weights, ignore_index, C, HW, sums, total);
    MSPL_CHECK_LAUNCH("weighted_ce_fwd");
    return MSPL_OK;
}

extern "C" int mspl_weighted_ce_bwd(const float* pred, const int64_t* target, const float* u_weight, const float* class_weights,
                                    int32_t ignore_index, int32_t N, int32_t C, int32_t HW, const float* g, const float* den,
                                    float* gpred, float* gu, void* stream) {
    MSPL_REQUIRE(pred && target && g && (gpred || gu), MSPL_ERR_NULL_POINTER, "weighted_ce_bwd: null pointer");
    MSPL_REQUIRE(N > 0 && C > 0 && HW > 0, MSPL_ERR_BAD_SHAPE, "weighted_ce_bwd: bad shape N=%d C=%d HW=%d", N, C, HW);
    MSPL_REQUIRE(!(gu && !u_weight), MSPL_ERR_NULL_POINTER, "weighted_ce_bwd: gu without u_weight");
    const int64_t total = (int64_t)N * HW;
    hipLaunchKernelGGL(wce_bwd_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream, pred, target,
                       u_weight, class_weights, ignore_index, C, HW, g, den, 1.0f / (float)total, gpred, gu, total);
    MSPL_CHECK_LAUNCH("weighted_ce_bwd");
    return MSPL_OK;
}
