// Backward of K2 (the four dilated depthwise 3x3 branches of an EESP block with hierarchical feature fusion,
// nn_layers/eesp.py:72-86) as TWO launches instead of eight: the data gradient of all four branches in one pass over the input
// plane, the weight gradients of all four branches in one pass over the output plane.  Both read the suffix-summed output
// gradient G_k = sum_{j>=k} gy_j (mspl_hff_suffix_sum; the HFF adds make branch k feed every later block).
//   gx[n,c,iy,ix]   = sum_k sum_{ky,kx} w_k[c,ky,kx] * G_k[n,c,(iy + d_k - ky*d_k)/s, (ix + d_k - kx*d_k)/s]     (exact divisions only)
//   gw_k[c,ky,kx]  += sum_{n,oy,ox}     G_k[n,c,oy,ox] * x[n,c, oy*s - d_k + ky*d_k, ox*s - d_k + kx*d_k]
// Streaming, HBM/L2-bound: per input pixel 36 gathered reads that hit L1/L2 (each G element is used 9 times) and one write.
#include "common.hpp"

namespace mspl {

struct DwBwdG {
    int N, n, H, W, Ho, Wo, stride;
    int dil[4];
};

// grid (ceil(H*W/256), N*n): one thread per input pixel of one plane
__global__ __launch_bounds__(256) void eesp_dw_bwd_data_kernel(const float* __restrict__ gs, const float* __restrict__ w4, DwBwdG g,
                                                               float* __restrict__ gx) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= g.H * g.W) return;
    const int plane = blockIdx.y;                         // img * n + c
    const int c = plane % g.n;
    const int iy = p / g.W, ix = p - iy * g.W;
    const size_t opl = (size_t)g.Ho * g.Wo;
    const size_t branch = (size_t)g.N * g.n * opl;        // gs is branch-major (4, N, n, Ho*Wo)
    const float* gp = gs + (size_t)plane * opl;
    const int sh = g.stride - 1;                          // stride 1 | 2
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int d = g.dil[k];
        const float* wk = w4 + ((size_t)k * g.n + c) * 9;
        const float* gk = gp + (size_t)k * branch;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = iy + d - ky * d;
            const int oy = ty >> sh;
            const bool vy = ty >= 0 && (ty & sh) == 0 && oy < g.Ho;
            const int oyc = min(max(oy, 0), g.Ho - 1);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = ix + d - kx * d;
                const int ox = tx >> sh;
                const bool v = vy && tx >= 0 && (tx & sh) == 0 && ox < g.Wo;
                const float gv = gk[(size_t)oyc * g.Wo + min(max(ox, 0), g.Wo - 1)];     // clamped, unconditional: loads batch
                acc = fmaf(v ? wk[ky * 3 + kx] : 0.f, gv, acc);
            }
        }
    }
    gx[(size_t)plane * g.H * g.W + p] = acc;
}

struct GwPtrs { float* p[4]; };

// grid (n * chunks): one workgroup per (channel, chunk of the N*Ho*Wo output positions); 36 tap sums per thread
__global__ __launch_bounds__(256) void eesp_dw_bwd_weight_kernel(const float* __restrict__ gs, const float* __restrict__ x, DwBwdG g,
                                                                 int chunks, GwPtrs gw) {
    const int chunk = blockIdx.x % chunks, c = blockIdx.x / chunks;
    const int npix = g.Ho * g.Wo;
    const int64_t total = (int64_t)g.N * npix;
    const int64_t per = (total + chunks - 1) / chunks;
    const int64_t i0 = chunk * per, i1 = min(total, i0 + per);
    const size_t plane = (size_t)g.H * g.W;
    const size_t branch = (size_t)g.N * g.n * npix;
    float acc[36];
#pragma unroll
    for (int t = 0; t < 36; ++t) acc[t] = 0.f;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
        const int img = (int)(i / npix), p = (int)(i - (int64_t)img * npix);
        const int oy = p / g.Wo, ox = p - oy * g.Wo;
        const float* gp = gs + ((size_t)img * g.n + c) * npix + p;
        const float* xp = x + ((size_t)img * g.n + c) * plane;
        // all 40 reads of this position are requested before the first use (clamped, unconditional; hipcc otherwise issues
        // load -> wait -> fma per tap and the loop runs at one memory latency per tap)
        float gv[4], xv[36];
        bool ok[36];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int d = g.dil[k];
            gv[k] = gp[(size_t)k * branch];
            const int by = oy * g.stride - d, bx = ox * g.stride - d;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = by + ky * d;
                const bool oky = iy >= 0 && iy < g.H;
                const int iyc = min(max(iy, 0), g.H - 1);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ix = bx + kx * d;
                    xv[k * 9 + ky * 3 + kx] = xp[(size_t)iyc * g.W + min(max(ix, 0), g.W - 1)];
                    ok[k * 9 + ky * 3 + kx] = oky && ix >= 0 && ix < g.W;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 36; ++t) acc[t] = fmaf(ok[t] ? gv[t / 9] : 0.f, xv[t], acc[t]);
    }
    __shared__ float part[4][36];
#pragma unroll
    for (int t = 0; t < 36; ++t) {
        float v = acc[t];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][t] = v;
    }
    __syncthreads();
    if (threadIdx.x < 36) {
        const int k = threadIdx.x / 9, t = threadIdx.x - k * 9;
        atomicAdd(gw.p[k] + (size_t)c * 9 + t, (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]));
    }
}

}  // namespace mspl

using namespace mspl;

// gs: suffix-summed output gradient, branch-major (4,N,n,Ho,Wo) (mspl_hff_suffix_sum); x: (N,n,H,W); w4: (4,n,3,3).
// gx (N,n,H,W) is overwritten (NULL: skipped).  gw[k] (n,3,3) each: ACCUMULATED into (caller zeroes, or passes the parameters'
// gradient buffers); gw == NULL skips the weight gradient.
extern "C" int mspl_eesp_dw_bwd(const float* gs, const float* x, const float* w4, const int32_t* dil, int32_t stride, int32_t N,
                                int32_t n, int32_t H, int32_t W, float* gx, float* const* gw, void* stream) {
    MSPL_REQUIRE(gs && dil && (gx == nullptr || w4) && (gw == nullptr || x), MSPL_ERR_NULL_POINTER, "eesp_dw_bwd: null pointer");
    MSPL_REQUIRE(N > 0 && n > 0 && H > 0 && W > 0 && (stride == 1 || stride == 2), MSPL_ERR_BAD_SHAPE,
                 "eesp_dw_bwd: bad shape N=%d n=%d %dx%d stride=%d", N, n, H, W, stride);
    DwBwdG g;
    g.N = N; g.n = n; g.H = H; g.W = W; g.stride = stride;
    g.Ho = (H - 1) / stride + 1; g.Wo = (W - 1) / stride + 1;
    for (int k = 0; k < 4; ++k) {
        MSPL_REQUIRE(dil[k] >= 1, MSPL_ERR_UNSUPPORTED, "eesp_dw_bwd: dilation %d", dil[k]);
        g.dil[k] = dil[k];
    }
    hipStream_t s = (hipStream_t)stream;
    if (gx) {
        MSPL_REQUIRE((int64_t)N * n <= 65535, MSPL_ERR_BAD_SHAPE, "eesp_dw_bwd: too many planes (%lld)", (long long)N * n);
        hipLaunchKernelGGL(eesp_dw_bwd_data_kernel, dim3((unsigned)ceil_div(H * W, 256), (unsigned)(N * n)), dim3(256), 0, s, gs, w4, g, gx);
        MSPL_CHECK_LAUNCH("eesp_dw_bwd(data)");
    }
    if (gw) {
        GwPtrs ptrs;
        for (int k = 0; k < 4; ++k) {
            MSPL_REQUIRE(gw[k], MSPL_ERR_NULL_POINTER, "eesp_dw_bwd: gw[%d] is NULL", k);
            ptrs.p[k] = gw[k];
        }
        const int64_t total = (int64_t)N * g.Ho * g.Wo;
        int chunks = 1;
        while ((int64_t)n * chunks < 2048 && total / (chunks * 2) >= 1024) chunks *= 2;
        hipLaunchKernelGGL(eesp_dw_bwd_weight_kernel, dim3((unsigned)(n * chunks)), dim3(256), 0, s, gs, x, g, chunks, ptrs);
        MSPL_CHECK_LAUNCH("eesp_dw_bwd(weight)");
    }
    return MSPL_OK;
}
