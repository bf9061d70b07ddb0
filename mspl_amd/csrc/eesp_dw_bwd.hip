// Backward of K2 (the four dilated depthwise 3x3 branches of an EESP block with hierarchical feature fusion,
// nn_layers/eesp.py:72-86) as TWO launches instead of eight: the data gradient of all four branches in one pass over the input
// plane, the weight gradients of all four branches in one pass over the output plane.  Both read the suffix-summed output
// gradient G_k = sum_{j>=k} gy_j (mspl_hff_suffix_sum; the HFF adds make branch k feed every later block).
//   gx[n,c,iy,ix]   = sum_k sum_{ky,kx} w_k[c,ky,kx] * G_k[n,c,(iy + d_k - ky*d_k)/s, (ix + d_k - kx*d_k)/s]     (exact divisions only)
//   gw_k[c,ky,kx]  += sum_{n,oy,ox}     G_k[n,c,oy,ox] * x[n,c, oy*s - d_k + ky*d_k, ox*s - d_k + kx*d_k]
// Streaming, HBM/L2-bound: per input pixel 36 gathered reads that hit L1/L2 (each G element is used 9 times) and one write.
#include <stdlib.h>

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

// Stride 2 on even planes (the three DownSampler EESPs: the largest launches of this file): a thread owns a 2 x 2 quad of INPUT pixels.
// Which taps reach an input pixel is then a matter of parity, known per quad position: with an odd dilation the even/even pixel takes
// the centre tap only, even/odd and odd/even two taps, odd/odd four; with an even dilation all nine taps land on the even/even pixel.
// 36 loads per quad instead of 36 masked loads per pixel (three quarters of them multiplied by zero), same taps in the same order
// per pixel: bit-identical to eesp_dw_bwd_data_kernel.
__global__ __launch_bounds__(256) void eesp_dw_bwd_data_s2_kernel(const float* __restrict__ gs, const float* __restrict__ w4, DwBwdG g,
                                                                  float* __restrict__ gx) {
    const int QW = g.W >> 1, QH = g.H >> 1;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= QH * QW) return;
    const int plane = blockIdx.y;                         // img * n + c
    const int c = plane % g.n;
    const int qy = t / QW, qx = t - qy * QW;
    const size_t opl = (size_t)g.Ho * g.Wo;
    const size_t branch = (size_t)g.N * g.n * opl;
    const float* gp = gs + (size_t)plane * opl;
    float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;     // input pixels (2qy, 2qx), (2qy, 2qx+1), (2qy+1, 2qx), (2qy+1, 2qx+1)
    auto G = [&](const float* gk, int oy, int ox, float& m) {
        m = (oy >= 0 && oy < g.Ho && ox >= 0 && ox < g.Wo) ? 1.f : 0.f;
        return gk[(size_t)min(max(oy, 0), g.Ho - 1) * g.Wo + min(max(ox, 0), g.Wo - 1)];
    };
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int d = g.dil[k];
        const float* wk = w4 + ((size_t)k * g.n + c) * 9;
        const float* gk = gp + (size_t)k * branch;
        if (d & 1) {                                      // uniform
            const int hp = (1 + d) >> 1, hm = (1 - d) / 2;        // output offsets of taps 0 and 2 for an odd input coordinate
            float m[9];
            const float v11 = G(gk, qy, qx, m[0]);
            const float v10 = G(gk, qy, qx + hp, m[1]), v12 = G(gk, qy, qx + hm, m[2]);
            const float v01 = G(gk, qy + hp, qx, m[3]), v21 = G(gk, qy + hm, qx, m[4]);
            const float v00 = G(gk, qy + hp, qx + hp, m[5]), v02 = G(gk, qy + hp, qx + hm, m[6]);
            const float v20 = G(gk, qy + hm, qx + hp, m[7]), v22 = G(gk, qy + hm, qx + hm, m[8]);
            a00 = fmaf(wk[4] * m[0], v11, a00);
            a01 = fmaf(wk[3] * m[1], v10, a01);  a01 = fmaf(wk[5] * m[2], v12, a01);
            a10 = fmaf(wk[1] * m[3], v01, a10);  a10 = fmaf(wk[7] * m[4], v21, a10);
            a11 = fmaf(wk[0] * m[5], v00, a11);  a11 = fmaf(wk[2] * m[6], v02, a11);
            a11 = fmaf(wk[6] * m[7], v20, a11);  a11 = fmaf(wk[8] * m[8], v22, a11);
        } else {
            const int h = d >> 1;
            float v[9], m[9];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) v[ky * 3 + kx] = G(gk, qy + h * (1 - ky), qx + h * (1 - kx), m[ky * 3 + kx]);
#pragma unroll
            for (int i = 0; i < 9; ++i) a00 = fmaf(wk[i] * m[i], v[i], a00);
        }
    }
    float* o = gx + (size_t)plane * g.H * g.W + (size_t)(2 * qy) * g.W + 2 * qx;
    *reinterpret_cast<float2*>(o) = make_float2(a00, a01);
    *reinterpret_cast<float2*>(o + g.W) = make_float2(a10, a11);
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
        v = wave_sum_dpp(v);                                  // total in lane 63
        if ((threadIdx.x & 63) == 63) part[threadIdx.x >> 6][t] = v;
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
        static const int s2_form = MSPL_TUNE_INT("MSPL_DW_BWD_S2", 1);
        if (s2_form && stride == 2 && (H & 1) == 0 && (W & 1) == 0 && (((uintptr_t)gx) & 7) == 0)
            hipLaunchKernelGGL(eesp_dw_bwd_data_s2_kernel, dim3((unsigned)ceil_div((H / 2) * (W / 2), 256), (unsigned)(N * n)), dim3(256), 0, s, gs,
                               w4, g, gx);
        else
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

namespace mspl {

// ------------------------------------------------------------------------------------------------ stride-1 blocks, one launch
// The whole backward between conv_1x1_exp's data gradient and proj_1x1's BatchNorm of a stride-1 EESP block in ONE kernel:
//     gc_k  = gy_k * (u_k > 0 ? 1 : alpha_k) * scale_k,  u_k = z_k * scale_k + shift_k        br_after_cat's BatchNorm + PReLU backward
//     G_k   = sum_{m >= k} gc_m                                                                the HFF suffix sum
//     gx    = sum_k sum_taps w_k[tap] * G_k[. - d_k * (tap - 1)]                               data gradient of the four branches
//     gw_k[tap] += sum G_k[.] * x[. + d_k * (tap - 1)]                                         their weight gradients
//     d gamma / d beta / d alpha of br_after_cat
// (four launches before: mspl_hff_bn_prelu_suffix_bwd, which wrote the 4n-channel suffix-summed gradient, and the two kernels
// above, which read it back: 64 us per level-4 block for ~50 MB.)  A workgroup owns a band of rows of one (image, channel j):
// the four branch planes' G_k and the block input x are staged in LDS with a zero halo of the largest dilation (the band's halo
// rows are recomputed from gy / z, they are not exchanged), every thread then takes pixels of the band: 36 + 36 multiply-adds from
// LDS per pixel.  Nothing but gy, z and x is read, nothing but gx is written.
constexpr int FB_MAXD = 4;

struct FbGeom {
    int N, n, H, W;
    int BH, bands;             // rows per band, bands per plane
    int WT;                    // LDS row stride: W + 2 * FB_MAXD
    int dil[4];
};

// Optional tail: proj_1x1's BatchNorm + PReLU backward applied to the data gradient before it is written (c = the projection's bare
// convolution result): gx then IS dL/dc, and the three per-channel sums of channel j are this workgroup's own.
struct FbProj {
    const float* c;                                  // (N, n, H, W) or null: plain data gradient
    const float *scale, *shift, *alpha, *mean, *inv; // n each (alpha / mean / inv nullable)
    float *gscale, *gshift, *galpha;                 // n each, accumulated
};

// Batch-statistics BatchNorm (the supervised loop): the statistics path of br_after_cat, gz += p * z + q per channel (the coefficients
// come from mspl_bn_train_prelu_bwd's sums), joins the direct gradient before the suffix sum.  p null: frozen statistics, nothing added.
struct FbStat {
    const float *p, *q;                              // 4n each, or null
};

__global__ __launch_bounds__(256) void eesp_bwd_fused_kernel(const float* __restrict__ z, const float* __restrict__ gy,
                                                             const float* __restrict__ x, const float* __restrict__ w4,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             const float* __restrict__ alpha, const float* __restrict__ bn_mean,
                                                             const float* __restrict__ bn_inv, FbGeom g, float* __restrict__ gx,
                                                             GwPtrs gw, float* __restrict__ gscale, float* __restrict__ gshift,
                                                             float* __restrict__ galpha, FbProj pj, FbStat st) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float red[4][15];
    int b = blockIdx.x;
    const int band = b % g.bands;  b /= g.bands;
    const int j = b % g.n;
    const int img = b / g.n;
    const int H = g.H, W = g.W, WT = g.WT;
    const int y0 = band * g.BH, y1 = min(y0 + g.BH, H);
    const int BHT = g.BH + 2 * FB_MAXD;
    const int tile = BHT * WT;
    float* GS = smem;                       // [4][BHT][WT]
    float* XS = smem + 4 * tile;            // [BHT][WT]
    const int tid = threadIdx.x;
    const size_t pl = (size_t)H * W;
    const float* xp = x + ((size_t)img * g.n + j) * pl;
    float sc[4], sh[4], al[4];
    const bool act = alpha != nullptr;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ch = k * g.n + j;
        sc[k] = scale ? scale[ch] : 1.f;  sh[k] = shift ? shift[ch] : 0.f;  al[k] = act ? alpha[ch] : 1.f;
    }
    const bool stat = st.p != nullptr;
    float pk[4] = {0.f, 0.f, 0.f, 0.f}, qk[4] = {0.f, 0.f, 0.f, 0.f};
    if (stat) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { pk[k] = st.p[k * g.n + j];  qk[k] = st.q[k * g.n + j]; }
    }
    float s_sc[4] = {0.f, 0.f, 0.f, 0.f}, s_sh[4] = {0.f, 0.f, 0.f, 0.f}, s_al[4] = {0.f, 0.f, 0.f, 0.f};
    // ---- stage: G_k and x on the band + halo, zero outside the image
    const size_t in0 = ((size_t)img * 4 * g.n + j) * pl, kin = (size_t)g.n * pl;
    // (only positions inside the image are loaded: for an 18x30 plane the zero halo is 45 % of the tile)
    {
        const int nq = (5 * tile) >> 2;
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t = tid; t < nq; t += 256) reinterpret_cast<float4*>(smem)[t] = zero4;
        for (int t = 4 * nq + tid; t < 5 * tile; t += 256) smem[t] = 0.f;
    }
    __syncthreads();
    const int ya = max(y0 - FB_MAXD, 0), yb = min(y1 + FB_MAXD, H);           // image rows staged: [ya, yb)
    const int nin = (yb - ya) * W;
    for (int t = tid; t < nin; t += 256) {
        const int ri = t / W, xx = t - ri * W;
        const int y = ya + ri;
        const size_t o = (size_t)y * W + xx;
        float zv[4], gv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { zv[k] = z[in0 + k * kin + o]; gv[k] = gy[in0 + k * kin + o]; }
        const float xr = xp[o];
        __builtin_amdgcn_sched_barrier(0);
        const bool own = y >= y0 && y < y1;
        float gk[4], run = 0.f;
#pragma unroll
        for (int k = 3; k >= 0; --k) {
            const float u = fmaf(zv[k], sc[k], sh[k]);          // K2's own epilogue expression: same PReLU branch
            const bool pos = !act || u > 0.f;
            const float gz = pos ? gv[k] : al[k] * gv[k];
            if (own) {
                if (!pos) s_al[k] += gv[k] * u;
                s_sc[k] += gz * zv[k];
                s_sh[k] += gz;
            }
            run += stat ? fmaf(zv[k], pk[k], qk[k]) + gz * sc[k] : gz * sc[k];      // (same expression as mspl_bn_train_prelu_bwd_apply)
            gk[k] = run;
        }
        const int lt = (y - (y0 - FB_MAXD)) * WT + xx + FB_MAXD;
#pragma unroll
        for (int k = 0; k < 4; ++k) GS[k * tile + lt] = gk[k];
        XS[lt] = xr;
    }
    __syncthreads();
    // ---- data gradient: the band's pixels
    float wk[4][9];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int q = 0; q < 9; ++q) wk[k][q] = w4[((size_t)k * g.n + j) * 9 + q];
    const int npx = (y1 - y0) * W;
    float* op = gx + ((size_t)img * g.n + j) * pl + (size_t)y0 * W;
    const float* cp = pj.c ? pj.c + ((size_t)img * g.n + j) * pl + (size_t)y0 * W : nullptr;
    const float psc = (cp && pj.scale) ? pj.scale[j] : 1.f, psh = (cp && pj.shift) ? pj.shift[j] : 0.f;
    const bool pact = cp && pj.alpha;
    const float pal = pact ? pj.alpha[j] : 1.f;
    float t_sc = 0.f, t_sh = 0.f, t_al = 0.f;
    for (int p = tid; p < npx; p += 256) {
        const int ry = p / W, cx = p - ry * W;
        const int base = (ry + FB_MAXD) * WT + cx + FB_MAXD;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int d = g.dil[k];
            const float* G = GS + k * tile + base;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) acc = fmaf(wk[k][ky * 3 + kx], G[-((ky - 1) * d * WT + (kx - 1) * d)], acc);
        }
        if (cp) {
            const float cv = cp[p];
            const float u = cv * psc + psh;
            const bool pos = !pact || u > 0.f;
            const float gz = pos ? acc : pal * acc;
            if (!pos) t_al += acc * u;
            t_sc += gz * cv;
            t_sh += gz;
            acc = gz * psc;
        }
        op[p] = acc;
    }
    // ---- weight gradients: thread = (tap, row group) -- ONE running sum per thread over every 7th ROW of the band, so the reduction is
    // over 7 partial sums per tap instead of a 36-value shuffle reduction across 256 threads, and the inner loop walks a row with
    // nothing but two LDS reads and a multiply-add per pixel (every 7th PIXEL instead -- a division-free but branchy column / row
    // carry per step -- was ~10 instructions per pixel: 39 -> 33.5 us per level-4 block at batch 16; with the DPP sums and the 16-byte zero fill 26.3 us)
    const int tap = tid % 36, pgrp = tid / 36;                  // threads 252..255: no tap
    float wsum = 0.f;
    if (pgrp < 7) {
        const int k = tap / 9, q = tap - k * 9, ky = q / 3, kx = q - ky * 3;
        const int d = g.dil[k];
        const int off = (ky - 1) * d * WT + (kx - 1) * d;
        const float* G = GS + k * tile + FB_MAXD * WT + FB_MAXD;
        const float* X = XS + FB_MAXD * WT + FB_MAXD + off;
        float w0 = 0.f, w1 = 0.f;
        for (int ry = pgrp; ry < y1 - y0; ry += 7) {
            const float* gr = G + ry * WT;
            const float* xr = X + ry * WT;
            int cx = 0;
#pragma unroll 4
            for (; cx + 1 < W; cx += 2) { w0 = fmaf(gr[cx], xr[cx], w0);  w1 = fmaf(gr[cx + 1], xr[cx + 1], w1); }
            if (cx < W) w0 = fmaf(gr[cx], xr[cx], w0);
        }
        wsum = w0 + w1;
    }
    // ---- reductions: 12 BatchNorm / PReLU sums by wave shuffles, the 36 x 7 weight-gradient partials through LDS
    __shared__ float wred[7][36];
    if (pgrp < 7) wred[pgrp][tap] = wsum;
    float v[15];
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k * 3] = s_sc[k];  v[k * 3 + 1] = s_sh[k];  v[k * 3 + 2] = s_al[k]; }
    v[12] = t_sc;  v[13] = t_sh;  v[14] = t_al;
#pragma unroll
    for (int i = 0; i < 15; ++i) {
        const float t = wave_sum_dpp(v[i]);                 // total in lane 63
        if ((tid & 63) == 63) red[tid >> 6][i] = t;
    }
    __syncthreads();
    auto tot = [&](int i) { return (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]); };
    if (tid < 36) {
        const int k = tid / 9, q = tid - k * 9;
        const float t = ((wred[0][tid] + wred[1][tid]) + (wred[2][tid] + wred[3][tid])) + ((wred[4][tid] + wred[5][tid]) + wred[6][tid]);
        atomicAdd(&gw.p[k][(size_t)j * 9 + q], t);
    } else if (tid < 40) {
        const int k = tid - 36, ch = k * g.n + j;
        const float t_sc = tot(k * 3), t_sh = tot(k * 3 + 1);
        if (gscale) atomicAdd(&gscale[ch], bn_inv ? (t_sc - bn_mean[ch] * t_sh) * bn_inv[ch] : t_sc);
        if (gshift) atomicAdd(&gshift[ch], t_sh);
        if (galpha && act) atomicAdd(&galpha[ch], tot(k * 3 + 2));
    } else if (tid == 40 && pj.c) {
        const float a_sc = tot(12), a_sh = tot(13);
        if (pj.gscale) atomicAdd(&pj.gscale[j], pj.inv ? (a_sc - pj.mean[j] * a_sh) * pj.inv[j] : a_sc);
        if (pj.gshift) atomicAdd(&pj.gshift[j], a_sh);
        if (pj.galpha && pact) atomicAdd(&pj.galpha[j], tot(14));
    }
}

}  // namespace mspl

using namespace mspl;

static size_t fb_plan(int N, int n, int H, int W, const int32_t* dil, FbGeom& g) {
    if (N <= 0 || n <= 0 || H <= 0 || W <= 0 || !dil) return 0;
    for (int k = 0; k < 4; ++k)
        if (dil[k] < 1 || dil[k] > FB_MAXD) return 0;
    g.N = N; g.n = n; g.H = H; g.W = W;
    g.WT = W + 2 * FB_MAXD;
    for (int k = 0; k < 4; ++k) g.dil[k] = dil[k];
    // band height: the whole plane when the five haloed tiles fit 48 KB, else the largest band that does (>= 8 rows)
    int bh = H;
    while (bh > 8 && (size_t)5 * (bh + 2 * FB_MAXD) * g.WT * sizeof(float) > 48 * 1024) bh = (bh + 1) / 2;
    g.BH = bh;
    g.bands = ceil_div(H, bh);
    const size_t lds = (size_t)5 * (bh + 2 * FB_MAXD) * g.WT * sizeof(float);
    return lds <= 64 * 1024 ? lds : 0;
}

extern "C" int mspl_eesp_bwd_fused_fits(int32_t N, int32_t n, int32_t H, int32_t W, const int32_t* dil) {
    FbGeom g;
    return fb_plan(N, n, H, W, dil, g) > 0 && (int64_t)N * n * g.bands < (1ll << 31);
}

static int eesp_bwd_fused_launch(const float* z, const float* gy, const float* x, const float* w4, const int32_t* dil,
                                 const float* scale, const float* shift, const float* alpha, const float* bn_mean,
                                 const float* bn_inv, int32_t N, int32_t n, int32_t H, int32_t W, float* gx, float* const* gw,
                                 float* gscale, float* gshift, float* galpha, const float* proj_c, const float* proj_scale,
                                 const float* proj_shift, const float* proj_alpha, const float* proj_mean, const float* proj_inv,
                                 float* g_proj_scale, float* g_proj_shift, float* g_proj_alpha, const float* stat_p, const float* stat_q,
                                 void* stream) {
    MSPL_REQUIRE(z && gy && x && w4 && dil && gx && gw, MSPL_ERR_NULL_POINTER, "eesp_bwd_fused: null pointer");
    MSPL_REQUIRE((stat_p == nullptr) == (stat_q == nullptr), MSPL_ERR_NULL_POINTER, "eesp_bwd_fused: statistics coefficients p / q must come together");
    MSPL_REQUIRE((proj_mean == nullptr) == (proj_inv == nullptr), MSPL_ERR_NULL_POINTER, "eesp_bwd_fused: proj mean / inv must come together");
    MSPL_REQUIRE((bn_mean == nullptr) == (bn_inv == nullptr), MSPL_ERR_NULL_POINTER, "eesp_bwd_fused: mean / inv must come together");
    FbGeom g;
    const size_t lds = fb_plan(N, n, H, W, dil, g);
    MSPL_REQUIRE(lds > 0, MSPL_ERR_UNSUPPORTED, "eesp_bwd_fused: N=%d n=%d %dx%d is not covered (dilations 1..4, row of %d floats in LDS)", N, n,
                 H, W, W + 2 * FB_MAXD);
    GwPtrs gp;
    for (int k = 0; k < 4; ++k) {
        MSPL_REQUIRE(gw[k], MSPL_ERR_NULL_POINTER, "eesp_bwd_fused: weight gradient %d is null", k);
        gp.p[k] = gw[k];
    }
    const int64_t blocks = (int64_t)N * n * g.bands;
    MSPL_REQUIRE(blocks < (1ll << 31), MSPL_ERR_BAD_SHAPE, "eesp_bwd_fused: grid too large");
    FbProj pj;
    pj.c = proj_c; pj.scale = proj_scale; pj.shift = proj_shift; pj.alpha = proj_alpha; pj.mean = proj_mean; pj.inv = proj_inv;
    pj.gscale = g_proj_scale; pj.gshift = g_proj_shift; pj.galpha = g_proj_alpha;
    FbStat st;
    st.p = stat_p; st.q = stat_q;
    hipLaunchKernelGGL(eesp_bwd_fused_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, z, gy, x, w4, scale, shift, alpha,
                       bn_mean, bn_inv, g, gx, gp, gscale, gshift, galpha, pj, st);
    MSPL_CHECK_LAUNCH("eesp_bwd_fused");
    return MSPL_OK;
}

extern "C" int mspl_eesp_bwd_fused(const float* z, const float* gy, const float* x, const float* w4, const int32_t* dil,
                                   const float* scale, const float* shift, const float* alpha, const float* bn_mean,
                                   const float* bn_inv, int32_t N, int32_t n, int32_t H, int32_t W, float* gx, float* const* gw,
                                   float* gscale, float* gshift, float* galpha, const float* proj_c, const float* proj_scale,
                                   const float* proj_shift, const float* proj_alpha, const float* proj_mean, const float* proj_inv,
                                   float* g_proj_scale, float* g_proj_shift, float* g_proj_alpha, void* stream) {
    return eesp_bwd_fused_launch(z, gy, x, w4, dil, scale, shift, alpha, bn_mean, bn_inv, N, n, H, W, gx, gw, gscale, gshift, galpha, proj_c,
                                 proj_scale, proj_shift, proj_alpha, proj_mean, proj_inv, g_proj_scale, g_proj_shift, g_proj_alpha, nullptr,
                                 nullptr, stream);
}

extern "C" int mspl_eesp_bwd_fused_bnstat(const float* z, const float* gy, const float* x, const float* w4, const int32_t* dil,
                                          const float* scale, const float* shift, const float* alpha, const float* stat_p,
                                          const float* stat_q, int32_t N, int32_t n, int32_t H, int32_t W, float* gx, float* const* gw,
                                          void* stream) {
    MSPL_REQUIRE(stat_p && stat_q && scale && shift, MSPL_ERR_NULL_POINTER, "eesp_bwd_fused_bnstat: null pointer");
    return eesp_bwd_fused_launch(z, gy, x, w4, dil, scale, shift, alpha, nullptr, nullptr, N, n, H, W, gx, gw, nullptr, nullptr, nullptr,
                                 nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, stat_p, stat_q, stream);
}
