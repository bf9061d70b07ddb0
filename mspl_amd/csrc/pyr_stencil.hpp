// Separable-stencil view of the up-sampled EfficientPyrPool branches, shared by the streaming forward (pyrpool_stream.hip) and the
// streaming branch backward (pyrpool_train.hip).
//
//   t = adaptive_avg_pool2d(dw3x3(bilinear_up(x)))  ==  sum_ky sum_kx w[ky][kx] * A_ky x G_kx^T
// with banded, position-dependent matrices A_k (rows) / G_k (columns): entry (p, q) of A_k multiplies x[p - R + q], R = (T - 1) / 2.
#pragma once
#include <algorithm>

#include "common.hpp"

namespace mspl {

__device__ __forceinline__ int p3_ada_s(int o, int I, int O) { return (int)(((unsigned)o * (unsigned)I) / (unsigned)O); }
__device__ __forceinline__ int p3_ada_e(int o, int I, int O) { return (int)((((unsigned)(o + 1)) * (unsigned)I + O - 1) / (unsigned)O); }

// Stencil coefficients of one output position p (row or column) and one kernel offset k of an up branch with T taps:
// acc[q] multiplies x[p - R + q], R = (T - 1) / 2.  Identical to p2_fill_up_tables (pyrpool_sep.hip).  A branch of the map's own size
// (S == I) gives acc[k] = 1: the plain 3x3 tap.
template <int T>
__device__ __forceinline__ void p3_coeffs(int p, int k, int I, int S, float sc, float (&acc)[5]) {
    constexpr int R = (T - 1) / 2;
#pragma unroll
    for (int q = 0; q < 5; ++q) acc[q] = 0.f;
    if (p < 0 || p >= I) return;
    const int us = p3_ada_s(p, S, I), ue = p3_ada_e(p, S, I);
    const float inv = 1.0f / (float)(ue - us);
    for (int uu = us; uu < ue; ++uu) {
        const int v = uu + k - 1;
        if (v < 0 || v >= S) continue;          // zero padding of the 3x3 on the up-sampled grid
        int ia, ib;  float w0, w1;
        bilinear_src(sc, v, I, ia, ib, w0, w1);
        const int ta = ia - (p - R), tb = ib - (p - R);
#pragma unroll
        for (int q = 0; q < T; ++q) {
            if (q == ta) acc[q] += w0 * inv;
            if (q == tb) acc[q] += w1 * inv;
        }
    }
}

__device__ __forceinline__ float p3_from_left(float v) {    // value of lane - 1 (0 for lane 0)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float p3_from_right(float v) {   // value of lane + 1 (0 for lane 63)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}

// Host twin of bilinear_src's index part (same fp32 operations; -ffp-contract=off).
static inline void p3_host_bilinear_idx(float scale, int dst, int in_size, int& i0, int& i1) {
    const float real = scale * (float)dst;
    int idx = (int)floorf(real);
    if (idx > in_size - 1) idx = in_size - 1;
    i0 = idx;
    i1 = idx + ((idx < in_size - 1) ? 1 : 0);
}

// Smallest R such that every source of output p lies in [p-R, p+R] (one dimension); large when unsupported.
static inline int p3_stencil_radius(int I, int S) {
    const float sc = bilinear_scale(I, S);
    int R = 0;
    for (int p = 0; p < I; ++p) {
        const int us = (int)(((int64_t)p * S) / I), ue = (int)((((int64_t)p + 1) * S + I - 1) / I);
        for (int v = std::max(us - 1, 0); v <= std::min(ue, S - 1); ++v) {
            int a, b;
            p3_host_bilinear_idx(sc, v, I, a, b);
            R = std::max(R, std::max(p - a, b - p));
        }
    }
    return R;
}


}  // namespace mspl
