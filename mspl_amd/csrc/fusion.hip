// RGB-D fusion gate blend (nn_layers/fusion_gate.py:26-47), forward and backward:
//   trainable:      w = sigmoid(z),  out = rgb * w + depth * (1 - w)        (z = conv_1x1(cat(rgb, depth)), computed by conv1x1)
//   not trainable:  out = rgb + depth                                       (z == NULL)
// Pure streaming: 3 reads + 1 write of 4 B per element, 16-byte accesses, grid-stride free (one float4 per thread).
#include "common.hpp"

namespace mspl {

__device__ __forceinline__ float sigmoidf(float z) { return 1.0f / (1.0f + expf(-z)); }

__global__ __launch_bounds__(256) void fusion_gate_kernel(const float* __restrict__ z, const float* __restrict__ rgb,
                                                          const float* __restrict__ depth, float* __restrict__ out,
                                                          int64_t count) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= count) return;
    if (i + 4 <= count) {
        const float4 r = *reinterpret_cast<const float4*>(rgb + i);
        const float4 d = *reinterpret_cast<const float4*>(depth + i);
        float4 o;
        if (z) {
            const float4 q = *reinterpret_cast<const float4*>(z + i);
            const float w0 = sigmoidf(q.x), w1 = sigmoidf(q.y), w2 = sigmoidf(q.z), w3 = sigmoidf(q.w);
            o.x = r.x * w0 + d.x * (1.0f - w0);
            o.y = r.y * w1 + d.y * (1.0f - w1);
            o.z = r.z * w2 + d.z * (1.0f - w2);
            o.w = r.w * w3 + d.w * (1.0f - w3);
        } else {
            o.x = r.x + d.x; o.y = r.y + d.y; o.z = r.z + d.z; o.w = r.w + d.w;
        }
        *reinterpret_cast<float4*>(out + i) = o;
    } else {
        for (int64_t j = i; j < count; ++j) {
            if (z) {
                const float w = sigmoidf(z[j]);
                out[j] = rgb[j] * w + depth[j] * (1.0f - w);
            } else {
                out[j] = rgb[j] + depth[j];
            }
        }
    }
}

// g_rgb = g*w, g_depth = g*(1-w), g_z = g*(rgb-depth)*w*(1-w)
__global__ __launch_bounds__(256) void fusion_gate_bwd_kernel(const float* __restrict__ z, const float* __restrict__ rgb,
                                                              const float* __restrict__ depth, const float* __restrict__ gy,
                                                              float* __restrict__ gz, float* __restrict__ grgb,
                                                              float* __restrict__ gdepth, int64_t count) {
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    for (int64_t j = i0; j < i0 + 4 && j < count; ++j) {
        const float g = gy[j];
        if (z) {
            const float w = sigmoidf(z[j]);
            grgb[j] = g * w;
            gdepth[j] = g * (1.0f - w);
            gz[j] = g * (rgb[j] - depth[j]) * (w * (1.0f - w));
        } else {
            grgb[j] = g;
            gdepth[j] = g;
        }
    }
}

}  // namespace mspl

using namespace mspl;

extern "C" int mspl_fusion_gate_fwd(const float* z, const float* rgb, const float* depth, int64_t count, float* out,
                                    void* stream) {
    MSPL_REQUIRE(rgb && depth && out, MSPL_ERR_NULL_POINTER, "fusion_gate: null pointer");
    MSPL_REQUIRE(count > 0, MSPL_ERR_BAD_SHAPE, "fusion_gate: bad element count %lld", (long long)count);
    MSPL_REQUIRE(((uintptr_t)rgb | (uintptr_t)depth | (uintptr_t)out | (uintptr_t)z) % 16 == 0, MSPL_ERR_BAD_SHAPE,
                 "fusion_gate: operands must be 16-byte aligned");
    const int64_t nthr = (count + 3) / 4;
    hipLaunchKernelGGL(fusion_gate_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z, rgb,
                       depth, out, count);
    MSPL_CHECK_LAUNCH("fusion_gate");
    return MSPL_OK;
}

extern "C" int mspl_fusion_gate_bwd(const float* z, const float* rgb, const float* depth, const float* gy, int64_t count,
                                    float* gz, float* grgb, float* gdepth, void* stream) {
    MSPL_REQUIRE(gy && grgb && gdepth, MSPL_ERR_NULL_POINTER, "fusion_gate_bwd: null pointer");
    MSPL_REQUIRE(!z || (rgb && depth && gz), MSPL_ERR_NULL_POINTER, "fusion_gate_bwd: trainable gate needs rgb, depth and gz");
    MSPL_REQUIRE(count > 0, MSPL_ERR_BAD_SHAPE, "fusion_gate_bwd: bad element count %lld", (long long)count);
    const int64_t nthr = (count + 3) / 4;
    hipLaunchKernelGGL(fusion_gate_bwd_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z,
                       rgb, depth, gy, gz, grgb, gdepth, count);
    MSPL_CHECK_LAUNCH("fusion_gate_bwd");
    return MSPL_OK;
}
