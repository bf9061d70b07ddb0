// NIDLoss (loss_fns/segmentation_loss.py:54-144): normalised information distance between the grey-level histogram of the
// camera image and the (soft-arg-max) label histogram.  This file holds the two heavy steps; the (K x C) entropy arithmetic
// after them is done by the caller on tiny tensors.
//   soft_arg_max (:124-141)        lab = sum_j j * exp((A_j - max A)*500) / (sum exp + 1e-12)
//   get_probabilities (:76-101)    P_c[k,p] = sum_b  sig((g - mu_k + L/2)/bw_c) - sig((g - mu_k - L/2)/bw_c),  g = (r+g+b)/3,
//                                  P_l[c,p] = sum_b  sig((lab - c + 1/2)/bw_l) - sig((lab - c - 1/2)/bw_l)   (rows c < min(C,K):
//                                  the reference fills P_l inside its loop over the K image bins),
//                                  joint = P_c P_l^T / norm, p_c = rowsum(P_c)/norm, p_l = rowsum(P_l)/norm, norm = B*H*W
// Forward: one thread per pixel POSITION p (the batch sum happens before the outer product, as in the reference); per-thread
// columns of P_c / P_l live in LDS, then each wave reduces its share of the K*C products over the workgroup's 256
// positions; per-workgroup partials are summed in double by a second kernel (no float atomics: deterministic).
// Backward (w.r.t. the label logits only; the camera image carries no gradient in the reference's use): same staging,
// G_l[c,p] = sum_k GJ[k,c] P_c[k,p] + Gpl[c], then d lab and the soft-arg-max Jacobian per image.
#include "common.hpp"

namespace mspl {

constexpr int NID_MAXB = 32;      // image bins and label bins are both limited to 32 (the scripts use 16 / 32 and <= 21)

__device__ __forceinline__ float sigm(float u) { return 1.0f / (1.0f + expf(-u)); }

struct NidG {
    int B, C, K, Cl, P;           // Cl = min(C, K): label bins that are ever filled
    float bw_c, bw_l, beta;
};

__device__ __forceinline__ float soft_arg_max(const float* __restrict__ a, int C, size_t stride, float beta, float eps) {
    float m = a[0];
    for (int j = 1; j < C; ++j) m = fmaxf(m, a[j * stride]);
    float s = 0.f, t = 0.f;
    for (int j = 0; j < C; ++j) {
        const float e = expf((a[j * stride] - m) * beta);
        s += e;
        t += (e * (float)j);
    }
    return t / (s + eps);
}

__device__ __forceinline__ void stage_columns(const float* __restrict__ cam, const float* __restrict__ lab, const NidG& g, int p,
                                              float* As, float* Ls, bool with_labels) {
    const int t = threadIdx.x;
    for (int k = 0; k < g.K; ++k) As[k * 256 + t] = 0.f;
    if (with_labels)
        for (int c = 0; c < g.Cl; ++c) Ls[c * 256 + t] = 0.f;
    if (p >= g.P) return;
    const float Lc = 1.0f / (float)g.K;
    for (int b = 0; b < g.B; ++b) {
        const float* cp = cam + (size_t)b * 3 * g.P + p;
        const float gray = ((cp[0] + cp[g.P]) + cp[2 * (size_t)g.P]) / 3.0f;
        for (int k = 0; k < g.K; ++k) {
            const float mu = Lc * ((float)k + 0.5f);
            As[k * 256 + t] += sigm((gray - mu + Lc / 2) / g.bw_c) - sigm((gray - mu - Lc / 2) / g.bw_c);
        }
        if (with_labels) {
            const float v = soft_arg_max(lab + (size_t)b * g.C * g.P + p, g.C, (size_t)g.P, g.beta, 1e-12f);
            for (int c = 0; c < g.Cl; ++c)
                Ls[c * 256 + t] += sigm((v - (float)c + 0.5f) / g.bw_l) - sigm((v - (float)c - 0.5f) / g.bw_l);
        }
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// partial: (blocks, K*Cl + K + Cl)
__global__ __launch_bounds__(256) void nid_hist_kernel(const float* __restrict__ cam, const float* __restrict__ lab, NidG g,
                                                       float* __restrict__ partial) {
    extern __shared__ float lds[];
    float* As = lds;
    float* Ls = lds + g.K * 256;
    stage_columns(cam, lab, g, blockIdx.x * 256 + threadIdx.x, As, Ls, true);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int E = g.K * g.Cl + g.K + g.Cl;
    float* out = partial + (size_t)blockIdx.x * E;
    for (int q = wave; q < E; q += 4) {
        float s = 0.f;
        if (q < g.K * g.Cl) {
            const float* a = As + (q / g.Cl) * 256;
            const float* l = Ls + (q % g.Cl) * 256;
#pragma unroll
            for (int j = 0; j < 4; ++j) s = fmaf(a[lane + 64 * j], l[lane + 64 * j], s);
        } else {
            const float* a = q < g.K * g.Cl + g.K ? As + (q - g.K * g.Cl) * 256 : Ls + (q - g.K * g.Cl - g.K) * 256;
#pragma unroll
            for (int j = 0; j < 4; ++j) s += a[lane + 64 * j];
        }
        s = wave_sum(s);
        if (lane == 0) out[q] = s;
    }
}

__global__ __launch_bounds__(256) void nid_reduce_kernel(const float* __restrict__ partial, int blocks, int E, double inv_norm,
                                                         float* __restrict__ out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    double s = 0.0;
    for (int b = 0; b < blocks; ++b) s += (double)partial[(size_t)b * E + e];
    out[e] = (float)(s * inv_norm);
}

// gj: (K, Cl) = dL/d joint, gpl: (Cl) = dL/d p_l, both already divided by norm.  glab: (B, C, P) overwritten.
__global__ __launch_bounds__(256) void nid_bwd_kernel(const float* __restrict__ cam, const float* __restrict__ lab, NidG g,
                                                      const float* __restrict__ gj, const float* __restrict__ gpl,
                                                      float* __restrict__ glab) {
    extern __shared__ float lds[];
    float* As = lds;
    float* Gl = lds + g.K * 256;
    float* Gj = Gl + g.Cl * 256;
    const int t = threadIdx.x, p = blockIdx.x * 256 + t;
    for (int i = t; i < g.K * g.Cl; i += 256) Gj[i] = gj[i];
    stage_columns(cam, lab, g, p, As, Gl, false);
    __syncthreads();
    if (p >= g.P) return;
    for (int c = 0; c < g.Cl; ++c) {
        float s = gpl[c];
        for (int k = 0; k < g.K; ++k) s = fmaf(Gj[k * g.Cl + c], As[k * 256 + t], s);
        Gl[c * 256 + t] = s;
    }
    for (int b = 0; b < g.B; ++b) {
        const float* a = lab + (size_t)b * g.C * g.P + p;
        float* ga = glab + (size_t)b * g.C * g.P + p;
        float m = a[0];
        for (int j = 1; j < g.C; ++j) m = fmaxf(m, a[(size_t)j * g.P]);
        float s = 0.f, tt = 0.f;
        for (int j = 0; j < g.C; ++j) {
            const float e = expf((a[(size_t)j * g.P] - m) * g.beta);
            s += e;
            tt += e * (float)j;
        }
        const float S = s + 1e-12f;
        const float v = tt / S;
        float dv = 0.f;                                              // dL / d lab
        for (int c = 0; c < g.Cl; ++c) {
            const float s1 = sigm((v - (float)c + 0.5f) / g.bw_l), s2 = sigm((v - (float)c - 0.5f) / g.bw_l);
            dv = fmaf(Gl[c * 256 + t], (s1 * (1.f - s1) - s2 * (1.f - s2)) / g.bw_l, dv);
        }
        for (int j = 0; j < g.C; ++j) {
            const float e = expf((a[(size_t)j * g.P] - m) * g.beta);
            ga[(size_t)j * g.P] = dv * g.beta * (e / S) * ((float)j - v);
        }
    }
}

static int nid_geom(NidG& g, int B, int C, int H, int W, int K, float bw_c, float bw_l, const char* who) {
    MSPL_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && K > 0, MSPL_ERR_BAD_SHAPE, "%s: bad shape B=%d C=%d %dx%d K=%d", who, B, C, H, W, K);
    MSPL_REQUIRE(K <= NID_MAXB && C <= NID_MAXB, MSPL_ERR_UNSUPPORTED, "%s: at most %d image bins and label bins (got %d, %d)", who,
                 NID_MAXB, K, C);
    MSPL_REQUIRE(bw_c > 0.f && bw_l > 0.f, MSPL_ERR_BAD_SHAPE, "%s: bandwidths must be positive", who);
    MSPL_REQUIRE((size_t)(K + (C < K ? C : K)) * 1024 + (size_t)K * C * 4 <= 65536, MSPL_ERR_UNSUPPORTED,
                 "%s: K=%d image bins x C=%d label bins exceed the 64 KiB staging budget", who, K, C);
    g.B = B; g.C = C; g.K = K; g.Cl = C < K ? C : K; g.P = H * W; g.bw_c = bw_c; g.bw_l = bw_l; g.beta = 500.0f;
    return MSPL_OK;
}

}  // namespace mspl

using namespace mspl;

extern "C" int64_t mspl_nid_workspace_floats(int32_t C, int32_t H, int32_t W, int32_t K) {
    if (C <= 0 || H <= 0 || W <= 0 || K <= 0) return MSPL_ERR_BAD_SHAPE;
    const int Cl = C < K ? C : K;
    return (int64_t)ceil_div(H * W, 256) * (K * Cl + K + Cl);
}

extern "C" int mspl_nid_hist_fwd(const float* camera, const float* label, int32_t B, int32_t C, int32_t H, int32_t W, int32_t K,
                                 float bw_camera, float bw_label, float* ws, float* out, void* stream) {
    MSPL_REQUIRE(camera && label && ws && out, MSPL_ERR_NULL_POINTER, "nid_hist: null pointer");
    NidG g;
    if (int rc = nid_geom(g, B, C, H, W, K, bw_camera, bw_label, "nid_hist")) return rc;
    const int blocks = ceil_div(g.P, 256), E = g.K * g.Cl + g.K + g.Cl;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(nid_hist_kernel, dim3((unsigned)blocks), dim3(256), (size_t)(g.K + g.Cl) * 256 * sizeof(float), s, camera, label, g, ws);
    MSPL_CHECK_LAUNCH("nid_hist");
    hipLaunchKernelGGL(nid_reduce_kernel, dim3((unsigned)ceil_div(E, 256)), dim3(256), 0, s, ws, blocks, E, 1.0 / ((double)g.P * B), out);
    MSPL_CHECK_LAUNCH("nid_hist(reduce)");
    return MSPL_OK;
}

extern "C" int mspl_nid_hist_bwd(const float* camera, const float* label, int32_t B, int32_t C, int32_t H, int32_t W, int32_t K,
                                 float bw_camera, float bw_label, const float* gjoint, const float* gpl, float* glabel, void* stream) {
    MSPL_REQUIRE(camera && label && gjoint && gpl && glabel, MSPL_ERR_NULL_POINTER, "nid_hist_bwd: null pointer");
    NidG g;
    if (int rc = nid_geom(g, B, C, H, W, K, bw_camera, bw_label, "nid_hist_bwd")) return rc;
    const size_t lds = ((size_t)(g.K + g.Cl) * 256 + (size_t)g.K * g.Cl) * sizeof(float);
    hipLaunchKernelGGL(nid_bwd_kernel, dim3((unsigned)ceil_div(g.P, 256)), dim3(256), lds, (hipStream_t)stream, camera, label, g, gjoint,
                       gpl, glabel);
    MSPL_CHECK_LAUNCH("nid_hist_bwd");
    return MSPL_OK;
}
