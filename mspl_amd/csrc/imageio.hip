// Image / label input transforms on the device: the loader step in front of the hot path.
//   data_loader/segmentation/greenhouse.py:216-222      val_transforms = Resize(size) -> Normalize() | Tensorize()
//   transforms/segmentation/data_transforms.py:191-212  Resize: PIL BILINEAR for rgb/depth, PIL NEAREST for labels
//   transforms/segmentation/data_transforms.py:15-46    to_tensor (/255) and normalize ((x - MEAN) / STD)
// The reference resizes with Pillow on the host, one image at a time, with workers=0 (uest_seg_multi_os.py:577).  Here a
// batch of decoded uint8 images is resized and normalised by two launches.  The arithmetic is Pillow's own
// (libImaging/Resample.c): triangle filter whose support grows with the down-scale factor, coefficients normalised and
// rounded to 22-bit fixed point, horizontal pass first, uint8 rounding between the passes -- bit-exact by construction,
// the coefficient tables are built on the host in double precision by the same sequence of operations.
// Byte work, HBM-bound and tiny (8 MB in, 24 MB out per batch of 16): one thread per output pixel, coalesced stores.
#include <cmath>
#include <vector>

#include "common.hpp"

namespace mspl {

constexpr int RS_PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ int clip8(int v) {
    v >>= RS_PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// src (N,Hs,Ws,C) -> dst (N,Hs,W,C), uint8.  One thread per (row, xo); all C channels.
template <int C>
__global__ __launch_bounds__(256) void resample_h_u8_kernel(const uint8_t* __restrict__ src, int rows, int Ws, int W,
                                                            const int* __restrict__ xb, const int* __restrict__ xk, int kx,
                                                            uint8_t* __restrict__ dst) {
    const int xo = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;
    if (xo >= W || r >= rows) return;
    const int x0 = xb[2 * xo], cnt = xb[2 * xo + 1];
    const uint8_t* s = src + ((size_t)r * Ws + x0) * C;
    const int* k = xk + (size_t)xo * kx;
    int acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 1 << (RS_PRECISION_BITS - 1);
    for (int j = 0; j < cnt; ++j) {
        const int kv = k[j];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] += (int)s[j * C + c] * kv;
    }
    uint8_t* d = dst + ((size_t)r * W + xo) * C;
#pragma unroll
    for (int c = 0; c < C; ++c) d[c] = (uint8_t)clip8(acc[c]);
}

// tmp (N,Hs,W,C) uint8 -> out (N,C,H,W) fp32: vertical pass, /255, optional (v - mean) / std, optional mirror.
template <int C>
__global__ __launch_bounds__(256) void resample_v_norm_kernel(const uint8_t* __restrict__ tmp, int Hs, int H, int W,
                                                              const int* __restrict__ yb, const int* __restrict__ yk, int ky,
                                                              const float* __restrict__ mean, const float* __restrict__ stdv,
                                                              const uint8_t* __restrict__ flip, float* __restrict__ out) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int yo = blockIdx.y, n = blockIdx.z;
    if (x >= W) return;
    const int y0 = yb[2 * yo], cnt = yb[2 * yo + 1];
    const int* k = yk + (size_t)yo * ky;
    const uint8_t* s = tmp + (((size_t)n * Hs + y0) * W + x) * C;
    int acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 1 << (RS_PRECISION_BITS - 1);
    for (int j = 0; j < cnt; ++j) {
        const int kv = k[j];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] += (int)s[(size_t)j * W * C + c] * kv;
    }
    const int xd = (flip && flip[n]) ? W - 1 - x : x;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        float v = (float)clip8(acc[c]) / 255.0f;                  // to_tensor: true division, as ATen's div(255)
        if (mean) v = (v - mean[c]) / stdv[c];                    // normalize: sub_ then div_
        out[(((size_t)n * C + c) * H + yo) * W + xd] = v;
    }
}

// labels: src (N,Hs,Ws) uint8 -> out (N,H,W) int64, nearest (index tables from the host), optional mirror.
__global__ __launch_bounds__(256) void resize_nearest_label_kernel(const uint8_t* __restrict__ src, int Hs, int Ws, int H, int W,
                                                                   const int* __restrict__ yi, const int* __restrict__ xi,
                                                                   const uint8_t* __restrict__ flip, int64_t* __restrict__ out) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int yo = blockIdx.y, n = blockIdx.z;
    if (x >= W) return;
    const int xd = (flip && flip[n]) ? W - 1 - x : x;
    out[((size_t)n * H + yo) * W + xd] = src[((size_t)n * Hs + yi[yo]) * Ws + xi[x]];
}

static inline double bilinear_filter(double x) {
    if (x < 0.0) x = -x;
    return x < 1.0 ? 1.0 - x : 0.0;
}

}  // namespace mspl

using namespace mspl;

// Taps per output sample for a given size change (Resample.c precompute_coeffs: ksize).
extern "C" int mspl_resample_ksize(int32_t in_size, int32_t out_size) {
    if (in_size <= 0 || out_size <= 0) return MSPL_ERR_BAD_SHAPE;
    double filterscale = (double)in_size / out_size;
    if (filterscale < 1.0) filterscale = 1.0;
    return (int)ceil(1.0 * filterscale) * 2 + 1;
}

// Host-side table builder: Pillow's precompute_coeffs + normalize_coeffs_8bpc for BILINEAR over the full box.
// bounds: (out,2) = (first source index, tap count); kk: (out, ksize) 22-bit fixed-point weights, zero padded.
extern "C" int mspl_resample_coeffs(int32_t in_size, int32_t out_size, int32_t* bounds, int32_t* kk) {
    MSPL_REQUIRE(bounds && kk, MSPL_ERR_NULL_POINTER, "resample_coeffs: null pointer");
    MSPL_REQUIRE(in_size > 0 && out_size > 0, MSPL_ERR_BAD_SHAPE, "resample_coeffs: bad sizes %d -> %d", in_size, out_size);
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    const double ss = 1.0 / filterscale;
    std::vector<double> w((size_t)ksize);
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            w[x] = bilinear_filter((x + xmin - center + 0.5) * ss);
            ww += w[x];
        }
        int32_t* k = kk + (size_t)xx * ksize;
        for (int x = 0; x < ksize; ++x) k[x] = 0;
        for (int x = 0; x < xmax; ++x) {
            const double v = ww != 0.0 ? w[x] / ww : w[x];
            k[x] = v < 0 ? (int)(-0.5 + v * (1 << RS_PRECISION_BITS)) : (int)(0.5 + v * (1 << RS_PRECISION_BITS));
        }
        bounds[2 * xx] = xmin;
        bounds[2 * xx + 1] = xmax;
    }
    return MSPL_OK;
}

// Source index per destination index for PIL NEAREST (Geometry.c ImagingScaleAffine: a running double sum).
extern "C" int mspl_nearest_index(int32_t in_size, int32_t out_size, int32_t* idx) {
    MSPL_REQUIRE(idx, MSPL_ERR_NULL_POINTER, "nearest_index: null pointer");
    MSPL_REQUIRE(in_size > 0 && out_size > 0, MSPL_ERR_BAD_SHAPE, "nearest_index: bad sizes %d -> %d", in_size, out_size);
    const double a0 = (double)in_size / out_size;
    double xo = a0 * 0.5;
    for (int x = 0; x < out_size; ++x) {
        int i = xo >= 0 ? (int)xo : 0;
        idx[x] = i < in_size ? i : in_size - 1;
        xo += a0;
    }
    return MSPL_OK;
}

extern "C" int mspl_preprocess_u8_fwd(const uint8_t* src, int32_t N, int32_t Hs, int32_t Ws, int32_t C, int32_t H, int32_t W,
                                      const int32_t* xb, const int32_t* xk, int32_t kx, const int32_t* yb, const int32_t* yk,
                                      int32_t ky, const float* mean, const float* stdv, const uint8_t* flip, uint8_t* tmp,
                                      float* out, void* stream) {
    MSPL_REQUIRE(src && yb && yk && out, MSPL_ERR_NULL_POINTER, "preprocess: null pointer");
    MSPL_REQUIRE(N > 0 && Hs > 0 && Ws > 0 && H > 0 && W > 0 && ky > 0, MSPL_ERR_BAD_SHAPE,
                 "preprocess: bad shape N=%d %dx%d -> %dx%d", N, Hs, Ws, H, W);
    MSPL_REQUIRE(C == 1 || C == 3, MSPL_ERR_UNSUPPORTED, "preprocess: %d channels (1 = depth, 3 = RGB)", C);
    MSPL_REQUIRE((mean == nullptr) == (stdv == nullptr), MSPL_ERR_NULL_POINTER, "preprocess: mean and std go together");
    MSPL_REQUIRE(Ws == W || (xb && xk && tmp && kx > 0), MSPL_ERR_NULL_POINTER,
                 "preprocess: a width change needs the horizontal tables and the (N,Hs,W,C) uint8 workspace");
    MSPL_REQUIRE(H <= 65535 && N <= 65535 && (int64_t)N * Hs <= 0x7fffffff, MSPL_ERR_BAD_SHAPE, "preprocess: grid too large");
    hipStream_t s = (hipStream_t)stream;
    const uint8_t* mid = src;
    if (Ws != W) {                                                // Pillow skips a pass that does not change the size
        const int rows = N * Hs;
        for (int r0 = 0; r0 < rows; r0 += 65535) {
            const int nr = rows - r0 < 65535 ? rows - r0 : 65535;
            dim3 grid((unsigned)ceil_div(W, 256), (unsigned)nr);
            if (C == 3)
                hipLaunchKernelGGL((resample_h_u8_kernel<3>), grid, dim3(256), 0, s, src + (size_t)r0 * Ws * 3, nr, Ws, W, xb, xk,
                                   kx, tmp + (size_t)r0 * W * 3);
            else
                hipLaunchKernelGGL((resample_h_u8_kernel<1>), grid, dim3(256), 0, s, src + (size_t)r0 * Ws, nr, Ws, W, xb, xk, kx,
                                   tmp + (size_t)r0 * W);
        }
        MSPL_CHECK_LAUNCH("preprocess(horizontal)");
        mid = tmp;
    }
    dim3 grid((unsigned)ceil_div(W, 256), (unsigned)H, (unsigned)N);
    if (C == 3)
        hipLaunchKernelGGL((resample_v_norm_kernel<3>), grid, dim3(256), 0, s, mid, Hs, H, W, yb, yk, ky, mean, stdv, flip, out);
    else
        hipLaunchKernelGGL((resample_v_norm_kernel<1>), grid, dim3(256), 0, s, mid, Hs, H, W, yb, yk, ky, mean, stdv, flip, out);
    MSPL_CHECK_LAUNCH("preprocess(vertical)");
    return MSPL_OK;
}

extern "C" int mspl_resize_label_fwd(const uint8_t* src, int32_t N, int32_t Hs, int32_t Ws, int32_t H, int32_t W,
                                     const int32_t* yi, const int32_t* xi, const uint8_t* flip, int64_t* out, void* stream) {
    MSPL_REQUIRE(src && yi && xi && out, MSPL_ERR_NULL_POINTER, "resize_label: null pointer");
    MSPL_REQUIRE(N > 0 && Hs > 0 && Ws > 0 && H > 0 && W > 0 && H <= 65535 && N <= 65535, MSPL_ERR_BAD_SHAPE,
                 "resize_label: bad shape N=%d %dx%d -> %dx%d", N, Hs, Ws, H, W);
    dim3 grid((unsigned)ceil_div(W, 256), (unsigned)H, (unsigned)N);
    hipLaunchKernelGGL(resize_nearest_label_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, Hs, Ws, H, W, yi, xi, flip, out);
    MSPL_CHECK_LAUNCH("resize_label");
    return MSPL_OK;
}
