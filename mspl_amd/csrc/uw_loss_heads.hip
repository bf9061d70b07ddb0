// K11 at head resolution: the uest loss of uest_seg_multi_os.py:1020-1023 taken from the two decoder outputs BEFORE their bilinear
// up-sampling (model/segmentation/espdnet_ue.py:301-302: main at H/2 x W/2, auxiliary at H/4 x W/4, both interpolated to H x W with
// align_corners=True).
//
// The three-step form (two bilinear launches, then mspl_uw_loss_scaled_fwd_bwd) writes the two full-size logit tensors and reads them
// back: 75 + 47 us of a 6.7 ms step at 16 x 5 x 288x480.  Here a workgroup owns a band of TH rows x 256 columns of the label map: it
// stages the patches of both low-resolution maps that band interpolates from in LDS and evaluates the up-sampled logits per pixel
// with the reference's own expression (wy0 * (wx0 * a + wx1 * b) + wy1 * (...): the forward is the one mspl_resize_bilinear
// computes), the loss terms and the closed-form gradients of uw_loss_kernel (train.hip).  The gradients leave at label resolution
// (N,C,H,W), for mspl_bilinear_bwd: the full-size logits never exist.
//
// (Measured and not kept, round 5: the transposed interpolation inside this kernel as ds_add_f32 into LDS accumulators of the patches'
// shape -- 40 LDS float atomics per pixel, conflict-free by construction: 463 us against 69 us without them at 16 x 5 x 256x480.  LDS
// float atomics retire at ~0.3 lanes per clock and CU; the global atomic flush of the patches was free.)
#include <algorithm>
#include <mutex>

#include "common.hpp"

namespace mspl {

struct UhGeom {
    int N, C, H, W;
    int Hm, Wm, Ha, Wa;             // main / auxiliary maps
    float shm, swm, sha, swa;       // bilinear scales (align_corners=True)
    int TH, tiles_x, tiles_y;       // band height; 256-column tiles per row; bands per image
    int MR, MC, AR, AC;             // LDS patch capacities (rows, columns)
    unsigned total;                 // tiles
    float ce_scale, inv_npix;
};

constexpr int UH_TW = 256;

// exp / log of the softmax terms on the transcendental unit (v_exp_f32 / v_log_f32 with one multiply: ~1 ulp on arguments <= 0 resp. sums in
// [1, C]; expf / logf expand to ~10 instructions each and the pixel loop holds 3 C + 1 of them: ~1030 -> ~820 vector instructions in the kernel, 69 -> 43 us at 16 x 5 x 256x480)
__device__ __forceinline__ float uh_exp(float x) { return __expf(x); }
__device__ __forceinline__ float uh_log(float x) { return __logf(x); }

// CM: class capacity of the register arrays; EXACT: C == CM (no per-class predicates).  Up to 8 classes the three exponentials of a
// class are kept from the sums to the gradients; beyond, they are recomputed (5 x CM registers spill at CM = 20).
template <int CM, bool EXACT>
__global__ __launch_bounds__(256) void uw_loss_heads_kernel(const float* __restrict__ mainp, const float* __restrict__ auxp,
                                                            const int64_t* __restrict__ target, const float* __restrict__ cw,
                                                            UhGeom g, float* __restrict__ loss_acc, float* __restrict__ gpred,
                                                            float* __restrict__ gaux) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int C = EXACT ? CM : g.C;
    constexpr bool KEEP = CM <= 8;
    constexpr int CK = KEEP ? CM : 1;
    const int msz = g.MR * g.MC, asz = g.AR * g.AC;
    float* LM = sm;                      // [C][MR][MC] main logits
    float* LA = LM + C * msz;            // [C][AR][AC] auxiliary logits
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float contrib = 0.f;
    for (unsigned tile = blockIdx.x; tile < g.total; tile += gridDim.x) {
        unsigned b = tile;
        const int txi = b % g.tiles_x;  b /= g.tiles_x;
        const int tyi = b % g.tiles_y;
        const int img = b / g.tiles_y;
        const int y0 = tyi * g.TH, x0 = txi * UH_TW;
        const int rows = min(g.TH, g.H - y0), cols = min(UH_TW, g.W - x0);
        // patches: first source row / column of the band's first pixel .. second source of its last
        int my0, my1, mx0, mx1, ay0, ay1, ax0, ax1, t0, t1;  float f0, f1;
        bilinear_src(g.shm, y0, g.Hm, my0, t1, f0, f1);              bilinear_src(g.shm, y0 + rows - 1, g.Hm, t0, my1, f0, f1);
        bilinear_src(g.swm, x0, g.Wm, mx0, t1, f0, f1);              bilinear_src(g.swm, x0 + cols - 1, g.Wm, t0, mx1, f0, f1);
        bilinear_src(g.sha, y0, g.Ha, ay0, t1, f0, f1);              bilinear_src(g.sha, y0 + rows - 1, g.Ha, t0, ay1, f0, f1);
        bilinear_src(g.swa, x0, g.Wa, ax0, t1, f0, f1);              bilinear_src(g.swa, x0 + cols - 1, g.Wa, t0, ax1, f0, f1);
        const int nmr = min(my1 - my0 + 1, g.MR), nmc = min(mx1 - mx0 + 1, g.MC);     // (the host sized MR.. from the same rule)
        const int nar = min(ay1 - ay0 + 1, g.AR), nac = min(ax1 - ax0 + 1, g.AC);
        // ---- stage both patches: a wave per (channel, row), lanes over the columns
        for (int cr = wave; cr < C * nmr; cr += 4) {
            const int c = cr / nmr, r = cr - c * nmr;
            const float* src = mainp + (((size_t)img * C + c) * g.Hm + my0 + r) * (size_t)g.Wm + mx0;
            for (int x = lane; x < nmc; x += 64) LM[c * msz + r * g.MC + x] = src[x];
        }
        for (int cr = wave; cr < C * nar; cr += 4) {
            const int c = cr / nar, r = cr - c * nar;
            const float* src = auxp + (((size_t)img * C + c) * g.Ha + ay0 + r) * (size_t)g.Wa + ax0;
            for (int x = lane; x < nac; x += 64) LA[c * asz + r * g.AC + x] = src[x];
        }
        __syncthreads();
        for (int r = wave; r < rows; r += 4) {
            const int y = y0 + r;
            int ya, yb;  float wy0m, wy1m, wy0a, wy1a;
            bilinear_src(g.shm, y, g.Hm, ya, yb, wy0m, wy1m);
            const int rm0 = (ya - my0) * g.MC - mx0, rm1 = (yb - my0) * g.MC - mx0;
            bilinear_src(g.sha, y, g.Ha, ya, yb, wy0a, wy1a);
            const int ra0 = (ya - ay0) * g.AC - ax0, ra1 = (yb - ay0) * g.AC - ax0;
            const size_t rowoff = ((size_t)img * g.H + y) * (size_t)g.W;
            const size_t goff = ((size_t)img * C * g.H + y) * (size_t)g.W;
            // a wave takes the row's 256 columns as four runs of 64 (coalesced label loads and gradient stores), one pixel per lane
            // at a time: unrolled, the four pixels' softmax terms are all live at once (205 VGPRs at C = 5, spills beyond)
#pragma unroll 1
            for (int j = 0; j < 4; ++j) {
                const int xs = x0 + j * 64 + lane;
                if (xs >= g.W) continue;
                const int t = (int)target[rowoff + xs];
                int xa, xb;  float wx0m, wx1m, wx0a, wx1a;
                bilinear_src(g.swm, xs, g.Wm, xa, xb, wx0m, wx1m);
                const int m00 = rm0 + xa, m01 = rm0 + xb, m10 = rm1 + xa, m11 = rm1 + xb;
                bilinear_src(g.swa, xs, g.Wa, xa, xb, wx0a, wx1a);
                const int a00 = ra0 + xa, a01 = ra0 + xb, a10 = ra1 + xa, a11 = ra1 + xb;
                float a[CM], bv[CM];
                float m1 = -INFINITY, m2 = -INFINITY, mo = -INFINITY;
#pragma unroll
                for (int c = 0; c < CM; ++c) {
                    if (EXACT || c < C) {
                        const float* pm = LM + c * msz;
                        const float* pa = LA + c * asz;
                        const float topm = wx0m * pm[m00] + wx1m * pm[m01], botm = wx0m * pm[m10] + wx1m * pm[m11];
                        const float topa = wx0a * pa[a00] + wx1a * pa[a01], bota = wx0a * pa[a10] + wx1a * pa[a11];
                        a[c] = wy0m * topm + wy1m * botm;
                        bv[c] = wy0a * topa + wy1a * bota;
                        m1 = fmaxf(m1, a[c]);  m2 = fmaxf(m2, bv[c]);  mo = fmaxf(mo, a[c] + 0.5f * bv[c]);
                    }
                }
                float e1[CK], e2[CK], eo[CK];
                float s1 = 0.f, s2 = 0.f, so = 0.f;
#pragma unroll
                for (int c = 0; c < CM; ++c) {
                    if (EXACT || c < C) {
                        const float x1 = uh_exp(a[c] - m1), x2 = uh_exp(bv[c] - m2), xo = uh_exp(a[c] + 0.5f * bv[c] - mo);
                        if (KEEP) { e1[KEEP ? c : 0] = x1;  e2[KEEP ? c : 0] = x2;  eo[KEEP ? c : 0] = xo; }
                        s1 += x1;  s2 += x2;  so += xo;
                    }
                }
                const float l1 = m1 + uh_log(s1), l2 = m2 + uh_log(s2), lo = mo + uh_log(so);
                const float r1 = __builtin_amdgcn_rcpf(s1), r2 = __builtin_amdgcn_rcpf(s2), ro = __builtin_amdgcn_rcpf(so);
                const bool tin = t >= 0 && t < C;
                float kld = 0.f, ot = 0.f;
#pragma unroll
                for (int c = 0; c < CM; ++c) {
                    if (EXACT || c < C) {
                        const float p1 = KEEP ? e1[KEEP ? c : 0] * r1 : uh_exp(a[c] - m1) * r1;
                        kld += p1 * (a[c] - l1) - p1 * (bv[c] - l2);
                        if (c == t) ot = a[c] + 0.5f * bv[c];
                    }
                }
                const float wt = tin ? cw[t] : 0.f;
                const float nll = -(ot - lo);
                const float u = uh_exp(-kld);
                const float ce = wt * nll * u;
                contrib += g.ce_scale * ce * g.inv_npix + kld * g.inv_npix;
                const float gk = (1.f - g.ce_scale * ce) * g.inv_npix;
                const float go = g.ce_scale * wt * u * g.inv_npix;
                float* dp = gpred + goff + xs;
                float* da = gaux + goff + xs;
                const size_t hw = (size_t)g.H * g.W;
#pragma unroll
                for (int c = 0; c < CM; ++c) {
                    if (EXACT || c < C) {
                        const float p1 = (KEEP ? e1[KEEP ? c : 0] : uh_exp(a[c] - m1)) * r1, p2 = (KEEP ? e2[KEEP ? c : 0] : uh_exp(bv[c] - m2)) * r2;
                        const float po = (KEEP ? eo[KEEP ? c : 0] : uh_exp(a[c] + 0.5f * bv[c] - mo)) * ro;
                        const float d = (a[c] - l1) - (bv[c] - l2);
                        const float dce = go * (po - (c == t ? 1.f : 0.f));
                        dp[c * hw] = dce + gk * (p1 * (d - kld));
                        da[c * hw] = 0.5f * dce - gk * (p1 - p2);
                    }
                }
            }
        }
        __syncthreads();                 // the next tile's staging overwrites the patches
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) contrib += __shfl_down(contrib, o, 64);
    __shared__ float part[4];
    if (lane == 0) part[wave] = contrib;
    __syncthreads();
    if (tid == 0) atomicAdd(loss_acc, (part[0] + part[1]) + (part[2] + part[3]));
}

// The device's source rule on the host (one fp32 product, floor, clamp: the same values), for exact patch sizes.
static void uh_src(float scale, int dst, int in_size, int& i0, int& i1) {
    const float real = scale * (float)dst;
    int idx = (int)std::floor(real);
    if (idx > in_size - 1) idx = in_size - 1;
    i0 = idx;
    i1 = idx + ((idx < in_size - 1) ? 1 : 0);
}
// most source rows (columns) any band of `n` pixels starting at a multiple of n touches
static int uh_span(float s, int n, int out_size, int in_size) {
    int most = 1;
    for (int p = 0; p < out_size; p += n) {
        int a0, a1, b0, b1;
        uh_src(s, p, in_size, a0, a1);
        uh_src(s, std::min(p + n, out_size) - 1, in_size, b0, b1);
        most = std::max(most, b1 - a0 + 1);
    }
    return most;
}

}  // namespace mspl

using namespace mspl;

// (a predicated 20-class form for the counts in between spills 6 600 registers: those take the three-step form)
extern "C" int mspl_uw_loss_heads_supported(int32_t C) { return (C >= 1 && C <= 8) || C == 13 || C == 20; }

extern "C" int mspl_uw_loss_heads_fwd_bwd(const float* main_lo, const float* aux_lo, const int64_t* target, const float* class_weights,
                                          int32_t N, int32_t C, int32_t Hm, int32_t Wm, int32_t Ha, int32_t Wa, int32_t H, int32_t W,
                                          float ce_scale, float out_scale, float* loss_acc, float* gpred, float* gaux, void* stream) {
    MSPL_REQUIRE(main_lo && aux_lo && target && class_weights && loss_acc && gpred && gaux, MSPL_ERR_NULL_POINTER, "uw_loss_heads: null pointer");
    MSPL_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && Hm > 0 && Wm > 0 && Ha > 0 && Wa > 0, MSPL_ERR_BAD_SHAPE,
                 "uw_loss_heads: bad shape N=%d C=%d main %dx%d aux %dx%d labels %dx%d", N, C, Hm, Wm, Ha, Wa, H, W);
    MSPL_REQUIRE(mspl_uw_loss_heads_supported(C), MSPL_ERR_UNSUPPORTED, "uw_loss_heads: C=%d classes (built for 1..8, 13, 20: use the up-sampled form)", C);
    MSPL_REQUIRE(Hm <= H && Wm <= W && Ha <= H && Wa <= W, MSPL_ERR_UNSUPPORTED, "uw_loss_heads: heads larger than the label map");
    MSPL_REQUIRE((int64_t)N * C * H * W < (1ll << 40), MSPL_ERR_BAD_SHAPE, "uw_loss_heads: too large");
    UhGeom g;
    g.N = N; g.C = C; g.H = H; g.W = W; g.Hm = Hm; g.Wm = Wm; g.Ha = Ha; g.Wa = Wa;
    g.shm = bilinear_scale(Hm, H); g.swm = bilinear_scale(Wm, W); g.sha = bilinear_scale(Ha, H); g.swa = bilinear_scale(Wa, W);
    g.tiles_x = ceil_div(W, UH_TW);
    // band height: 8 rows (two per wave) when that still gives ~3 workgroups per CU, else 4; halved while the patches outgrow 64 KB
    // of LDS (four workgroups per CU at C = 5), 128 KB at most (C = 20)
    size_t lds = 0;
    auto size_for = [&](int th) {
        g.TH = th;
        g.MR = uh_span(g.shm, th, H, Hm);  g.MC = uh_span(g.swm, UH_TW, W, Wm) | 1;
        g.AR = uh_span(g.sha, th, H, Ha);  g.AC = uh_span(g.swa, UH_TW, W, Wa) | 1;
        lds = (size_t)C * ((size_t)g.MR * g.MC + (size_t)g.AR * g.AC) * sizeof(float);
    };
    int th = ((int64_t)N * ceil_div(H, 8) * g.tiles_x >= 768) ? 8 : 4;
    for (size_for(th); lds > 64 * 1024 && th > 1; size_for(th)) th >>= 1;
    MSPL_REQUIRE(lds <= 128 * 1024, MSPL_ERR_UNSUPPORTED, "uw_loss_heads: patches of %zu bytes do not fit", lds);
    g.tiles_y = ceil_div(H, g.TH);
    const int64_t tiles = (int64_t)N * g.tiles_y * g.tiles_x;
    MSPL_REQUIRE(tiles < (1ll << 31), MSPL_ERR_BAD_SHAPE, "uw_loss_heads: too many tiles");
    g.total = (unsigned)tiles;
    g.ce_scale = ce_scale;
    g.inv_npix = out_scale / (float)((int64_t)N * H * W);
    // one same-address atomic per workgroup on the loss: a bounded grid walks the tiles (train.hip, uw_loss_launch)
    const unsigned blocks = (unsigned)std::min<int64_t>(tiles, 1536);
    hipStream_t s = (hipStream_t)stream;
    static std::once_flag once;
    static bool attr_ok = false;
    std::call_once(once, [] {
        attr_ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&uw_loss_heads_kernel<13, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess &&
                  hipFuncSetAttribute(reinterpret_cast<const void*>(&uw_loss_heads_kernel<20, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess;
    });
    MSPL_REQUIRE(lds <= 64 * 1024 || (attr_ok && C > 8), MSPL_ERR_UNSUPPORTED, "uw_loss_heads: patches of %zu bytes do not fit", lds);
#define MSPL_UH(CMv, EX) hipLaunchKernelGGL((uw_loss_heads_kernel<CMv, EX>), dim3(blocks), dim3(256), lds, s, main_lo, aux_lo, target, class_weights, g, loss_acc, gpred, gaux)
    if (C == 5) MSPL_UH(5, true);
    else if (C <= 8) MSPL_UH(8, false);
    else if (C == 13) MSPL_UH(13, true);
    else MSPL_UH(20, true);
#undef MSPL_UH
    MSPL_CHECK_LAUNCH("uw_loss_heads");
    return MSPL_OK;
}
