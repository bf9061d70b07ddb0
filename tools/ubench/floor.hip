// Floor of a kernel that moves K2's bytes and does nothing else: reads `in` floats once (16-byte loads), writes `out` floats
// once (16-byte stores, plain or write-through), launched back to back from a hipGraph like the K2 micro-benchmark.
// What fraction of 8 TB/s can ANY launch of that size reach on this box?   hipcc --offload-arch=gfx950 -O3 floor.hip -o floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

template <int REP, bool WT>
__global__ __launch_bounds__(256) void move_kernel(const float4* __restrict__ in, float4* __restrict__ out, int n4, int ostride4) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 v = in[i];
#pragma unroll
    for (int k = 0; k < REP; ++k) {
        float4 o = make_float4(v.x + k, v.y, v.z, v.w);
        if (WT) {
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 dv = {__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w)};
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x7fffffff, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(dv, r, (int)(((size_t)k * ostride4 + i) * 16), 0, 16);
        } else {
            out[(size_t)k * ostride4 + i] = o;
        }
    }
}

__global__ void empty_kernel(int) {}

static double time_graph(hipStream_t s, hipGraphExec_t ge, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<double> t;
    for (int r = 0; r < 7; ++r) {
        hipEventRecord(e0, s);
        hipGraphLaunch(ge, s);
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        t.push_back(ms * 1e3 / iters);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

template <class F>
static double bench(hipStream_t s, F launch, int iters = 20) {
    for (int i = 0; i < 3; ++i) launch();
    hipStreamSynchronize(s);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < iters; ++i) launch();
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    const double us = time_graph(s, ge, iters);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return us;
}

int main() {
    hipStream_t s; hipStreamCreate(&s);
    float4 *in, *out;
    hipMalloc(&in, 256 << 20); hipMalloc(&out, 512 << 20);
    hipMemset(in, 0, 256 << 20);
    printf("empty kernel (1 block): %.2f us per launch\n", bench(s, [&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, 0); }));
    printf("empty kernel (2048 blocks of 256): %.2f us per launch\n", bench(s, [&] { hipLaunchKernelGGL(empty_kernel, dim3(2048), dim3(256), 0, s, 0); }));
    struct Shape { const char* name; long in_floats; int rep; };   // batch 16
    const Shape shapes[] = {{"L2_0 s2 n=24 144x240 (out = in)", 16L * 24 * 144 * 240, 1}, {"L3_0 s2 n=32 72x120 (out = in)", 16L * 32 * 72 * 120, 1},
                            {"L3 s1 n=64 36x60 (out = 4 in)", 16L * 64 * 36 * 60, 4}, {"L4_0 s2 n=64 36x60 (out = in)", 16L * 64 * 36 * 60, 1},
                            {"L4 s1 n=128 18x30 (out = 4 in)", 16L * 128 * 18 * 30, 4}};
    for (const Shape& sh : shapes) {
        const int n4 = (int)(sh.in_floats / 4);
        const dim3 grid((n4 + 255) / 256), blk(256);
        const double bytes = sh.in_floats * 4.0 * (1 + sh.rep);
        double a, b;
        if (sh.rep == 4) {
            a = bench(s, [&] { hipLaunchKernelGGL((move_kernel<4, false>), grid, blk, 0, s, in, out, n4, n4); });
            b = bench(s, [&] { hipLaunchKernelGGL((move_kernel<4, true>), grid, blk, 0, s, in, out, n4, n4); });
        } else {
            a = bench(s, [&] { hipLaunchKernelGGL((move_kernel<1, false>), grid, blk, 0, s, in, out, n4, n4); });
            b = bench(s, [&] { hipLaunchKernelGGL((move_kernel<1, true>), grid, blk, 0, s, in, out, n4, n4); });
        }
        printf("%-36s %6.1f MB  plain stores %6.2f us (%.2f of 8 TB/s)   write-through %6.2f us (%.2f of 8 TB/s)\n", sh.name, bytes / 1e6, a,
               bytes / a / 8e6, b, bytes / b / 8e6);
    }
    return 0;
}
