// Micro-benchmark: does v_pk_fma_f32 double the fp32 FMA rate of a wave64 on gfx950?  (scalar v_fma_f32 vs packed, 8 waves/SIMD)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int PK>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float2v x[8];
    for (int i = 0; i < 8; ++i) { x[i].x = threadIdx.x * 0.001f + i; x[i].y = threadIdx.x * 0.002f - i; }
    float2v av = {a, a * 0.5f}, bv = {b, b * 0.25f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (PK) {
                x[i] = __builtin_elementwise_fma(x[i], av, bv);
            } else {
                x[i].x = __builtin_fmaf(x[i].x, av.x, bv.x);
                x[i].y = __builtin_fmaf(x[i].y, av.y, bv.y);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, 2048 * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096;
    for (int pk = 0; pk < 2; ++pk) for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (pk) hipLaunchKernelGGL(k<1>, dim3(2048), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
        else hipLaunchKernelGGL(k<0>, dim3(2048), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fma = 2048.0 * 256 * iters * 16;
        printf("%s: %.3f ms  %.1f TFLOP/s (fp32 FMA = 2 flop)\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ", ms, 2 * fma / ms / 1e9);
    }
    return 0;
}
