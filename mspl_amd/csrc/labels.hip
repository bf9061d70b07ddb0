// K8/K9 label epilogue and K10 cross-source label merge -- the integer end of the pseudo-label pass.
//
// label_epilogue: model/segmentation/espdnet_ue.py:301-302 (final bilinear, align_corners=True) fused
//   with uest_seg_multi_os.py:685-691 (get_output: pred + 0.5*aux -> Softmax2d; PixelwiseKLD,
//   loss_fns/segmentation_loss.py:181-189) and :903-912 (np.argmax over classes, first max wins; id LUT).
//   Full-resolution logits are never written unless the caller asks for them.
// merge_labels: uest_seg_multi_os.py:695-718 (merge_outputs) + :919-921 (class histogram).  Pure integer,
//   S bytes in + 1 byte out per pixel; bit-exact contract.
#include "common.hpp"

namespace mspl {

struct LeGeom {
    int N, C, Hm, Wm, Ha, Wa, H, W;
    float shm, swm, sha, swa;
};

// One thread per output pixel, ONE pass over the classes with running (online) maxima, so nothing but a
// handful of scalars lives in registers whatever C is:
//   argmax_c o_c                      (strict '>' keeps the first maximum, like np.argmax)
//   lse(main), lse(aux)               (running max + rescaled sum)
//   E_p1[main - aux]                  (rescaled with lse(main)'s running max)
//   KL(main || aux) = E_p1[main - aux] - lse(main) + lse(aux)
// Softmax probabilities, when requested, take a second pass (get_output's drop-in form only).
__global__ __launch_bounds__(256) void label_epilogue_kernel(const float* __restrict__ mainp,
                                                             const float* __restrict__ auxp, LeGeom g,
                                                             const uint8_t* __restrict__ lut,
                                                             uint8_t* __restrict__ labels, float* __restrict__ prob,
                                                             float* __restrict__ kld, float* __restrict__ main_up,
                                                             float* __restrict__ aux_up, int64_t total) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % g.W);  idx /= g.W;
    const int y = (int)(idx % g.H);
    const int n = (int)(idx / g.H);
    const size_t hw = (size_t)g.H * g.W;
    const size_t pix = (size_t)y * g.W + x;

    int my0, my1, mx0, mx1;  float mwy0, mwy1, mwx0, mwx1;
    bilinear_src(g.shm, y, g.Hm, my0, my1, mwy0, mwy1);
    bilinear_src(g.swm, x, g.Wm, mx0, mx1, mwx0, mwx1);
    const size_t mplane = (size_t)g.Hm * g.Wm;
    const float* mb = mainp + (size_t)n * g.C * mplane;
    const int m00 = my0 * g.Wm + mx0, m01 = my0 * g.Wm + mx1, m10 = my1 * g.Wm + mx0, m11 = my1 * g.Wm + mx1;

    int a00 = 0, a01 = 0, a10 = 0, a11 = 0;  float awy0 = 0.f, awy1 = 0.f, awx0 = 0.f, awx1 = 0.f;
    size_t aplane = 0;
    const float* ab = nullptr;
    if (auxp) {
        int ay0, ay1, ax0, ax1;
        bilinear_src(g.sha, y, g.Ha, ay0, ay1, awy0, awy1);
        bilinear_src(g.swa, x, g.Wa, ax0, ax1, awx0, awx1);
        aplane = (size_t)g.Ha * g.Wa;
        ab = auxp + (size_t)n * g.C * aplane;
        a00 = ay0 * g.Wa + ax0; a01 = ay0 * g.Wa + ax1; a10 = ay1 * g.Wa + ax0; a11 = ay1 * g.Wa + ax1;
    }
    auto interp_main = [&](int c) {
        const float* p = mb + c * mplane;
        const float top = mwx0 * p[m00] + mwx1 * p[m01];
        const float bot = mwx0 * p[m10] + mwx1 * p[m11];
        return mwy0 * top + mwy1 * bot;
    };
    auto interp_aux = [&](int c) {
        const float* q = ab + c * aplane;
        const float top = awx0 * q[a00] + awx1 * q[a01];
        const float bot = awx0 * q[a10] + awx1 * q[a11];
        return awy0 * top + awy1 * bot;
    };

    float omax = -INFINITY;  int best = 0;
    float M1 = -INFINITY, S1 = 0.f, T1 = 0.f;   // lse(main) state and sum exp(m - M1) * (m - a)
    float M2 = -INFINITY, S2 = 0.f;             // lse(aux) state
#pragma unroll 2
    for (int c = 0; c < g.C; ++c) {
        const float m = interp_main(c);
        const float a = ab ? interp_aux(c) : 0.f;
        const float o = m + 0.5f * a;
        if (o > omax) { omax = o; best = c; }
        if (main_up) main_up[((size_t)n * g.C + c) * hw + pix] = m;
        if (aux_up && ab) aux_up[((size_t)n * g.C + c) * hw + pix] = a;
        if (kld && ab) {
            if (m > M1) { const float f = expf(M1 - m); S1 *= f; T1 *= f; M1 = m; }
            const float e1 = expf(m - M1);
            S1 += e1;
            T1 = fmaf(e1, m - a, T1);
            if (a > M2) { S2 *= expf(M2 - a); M2 = a; }
            S2 += expf(a - M2);
        }
    }
    if (labels) labels[(size_t)n * hw + pix] = lut ? lut[best] : (uint8_t)best;
    if (kld) kld[(size_t)n * hw + pix] = ab ? (T1 / S1 - (M1 + logf(S1)) + (M2 + logf(S2))) : 0.f;
    if (prob) {
        float s = 0.f;
        for (int c = 0; c < g.C; ++c) s += expf(interp_main(c) + 0.5f * (ab ? interp_aux(c) : 0.f) - omax);
        const float inv = 1.0f / s;
        for (int c = 0; c < g.C; ++c)
            prob[((size_t)n * g.C + c) * hw + pix] = expf(interp_main(c) + 0.5f * (ab ? interp_aux(c) : 0.f) - omax) * inv;
    }
}

// Label pass form (labels and / or the KL map only, C <= CMAX): one thread per output pixel, a workgroup per piece of
// one output row, so the row interpolation is uniform; the 2*C interpolated logits are kept in registers (all their loads
// are independent and issue together), then max / exp-sum passes run on registers: one exp per class and head instead of
// the online form's rescaling exps, and no 64-bit index arithmetic.
template <int CMAX>
__global__ __launch_bounds__(256) void label_epilogue_reg_kernel(const float* __restrict__ mainp, const float* __restrict__ auxp,
                                                                 LeGeom g, const uint8_t* __restrict__ lut,
                                                                 uint8_t* __restrict__ labels, float* __restrict__ kld,
                                                                 unsigned long long* __restrict__ hist, int ncls) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    // XCD-aware row order: workgroups are dealt to the 8 XCDs round-robin by linear id, and neighbouring output rows read the
    // same source rows -- in natural order every source row was fetched into four different L2s (PMC: 189 MB read for 36 MB of
    // logits).  Row slot s = blockIdx.y maps to row (s % 4) * ceil(H/4) + s / 4, so each XCD (pair) walks one contiguous quarter.
    const int rq = (g.H + 3) >> 2;
    const int y = (int)(blockIdx.y & 3) * rq + (int)(blockIdx.y >> 2), n = blockIdx.z;
    if (y >= g.H) return;                                                    // uniform: the whole workgroup leaves
    // class histogram of the labels this workgroup writes (the single-source pass: uest_seg_multi_os.py:785-815 counts the
    // label map it has just produced -- no separate merge launch): one ballot per class and wave, LDS sum, one 64-bit atomic
    // per non-empty class and workgroup.  Labels >= ncls are not counted.
    __shared__ unsigned int sh_hist[32];
    if (hist) {
        if (threadIdx.x < 32) sh_hist[threadIdx.x] = 0;
        __syncthreads();
    }
    int my_label = -1;
    if (x < g.W) {
    int my0, my1, mx0, mx1;  float mwy0, mwy1, mwx0, mwx1;
    bilinear_src(g.shm, y, g.Hm, my0, my1, mwy0, mwy1);                      // uniform
    bilinear_src(g.swm, x, g.Wm, mx0, mx1, mwx0, mwx1);
    const int mplane = g.Hm * g.Wm;
    const float* mr0 = mainp + (size_t)n * g.C * mplane + my0 * g.Wm;
    const float* mr1 = mainp + (size_t)n * g.C * mplane + my1 * g.Wm;
    float m[CMAX], a[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
        m[c] = -INFINITY;
        if (c < g.C) {
            const float top = mwx0 * mr0[c * mplane + mx0] + mwx1 * mr0[c * mplane + mx1];
            const float bot = mwx0 * mr1[c * mplane + mx0] + mwx1 * mr1[c * mplane + mx1];
            m[c] = mwy0 * top + mwy1 * bot;
        }
    }
    if (auxp) {
        int ay0, ay1, ax0, ax1;  float awy0, awy1, awx0, awx1;
        bilinear_src(g.sha, y, g.Ha, ay0, ay1, awy0, awy1);                  // uniform
        bilinear_src(g.swa, x, g.Wa, ax0, ax1, awx0, awx1);
        const int aplane = g.Ha * g.Wa;
        const float* ar0 = auxp + (size_t)n * g.C * aplane + ay0 * g.Wa;
        const float* ar1 = auxp + (size_t)n * g.C * aplane + ay1 * g.Wa;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            a[c] = -INFINITY;
            if (c < g.C) {
                const float top = awx0 * ar0[c * aplane + ax0] + awx1 * ar0[c * aplane + ax1];
                const float bot = awx0 * ar1[c * aplane + ax0] + awx1 * ar1[c * aplane + ax1];
                a[c] = awy0 * top + awy1 * bot;
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < CMAX; ++c) a[c] = c < g.C ? 0.f : -INFINITY;
    }
    const size_t pix = ((size_t)n * g.H + y) * g.W + x;
    if (labels) {
        float omax = -INFINITY;  int best = 0;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            if (c < g.C) { const float o = m[c] + 0.5f * a[c]; if (o > omax) { omax = o; best = c; } }   // first maximum wins
        }
        my_label = lut ? lut[best] : best;
        labels[pix] = (uint8_t)my_label;
    }
    if (kld) {
        float k = 0.f;
        if (auxp) {
            float M1 = -INFINITY, M2 = -INFINITY;
#pragma unroll
            for (int c = 0; c < CMAX; ++c) { M1 = fmaxf(M1, m[c]); M2 = fmaxf(M2, a[c]); }
            float S1 = 0.f, T1 = 0.f, S2 = 0.f;
#pragma unroll
            for (int c = 0; c < CMAX; ++c) {
                if (c < g.C) {
                    const float e1 = __expf(m[c] - M1);      // v_exp_f32 path: arguments are <= 0, relative error ~1e-7
                    S1 += e1;
                    T1 = fmaf(e1, m[c] - a[c], T1);
                    S2 += __expf(a[c] - M2);
                }
            }
            k = T1 / S1 - (M1 + __logf(S1)) + (M2 + __logf(S2));
        }
        kld[pix] = k;
    }
    }   // x < W
    if (hist) {
        for (int c = 0; c < ncls; ++c) {
            const unsigned long long b = __ballot(my_label == c);
            if ((threadIdx.x & 63) == 0 && b) atomicAdd(&sh_hist[c], (unsigned)__popcll(b));
        }
        __syncthreads();
        if ((int)threadIdx.x < ncls && sh_hist[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)sh_hist[threadIdx.x]);
    }
}

struct MergeSrc {
    const uint8_t* p[8];
};

// 16 pixels per thread (one 16-byte load per source), S and the class bound compile-time so the vote is a
// short unrolled compare tree; per-thread histogram in registers, wave + LDS reduction, one 64-bit atomic per
// class per workgroup.
template <int S, int NCLS>
__global__ __launch_bounds__(256) void merge_labels_kernel(MergeSrc src, int64_t npix, int ncls, int thresh,
                                                           int fill, uint8_t* __restrict__ out,
                                                           unsigned long long* __restrict__ hist, int vec_ok) {
    uint32_t h[NCLS];
#pragma unroll
    for (int c = 0; c < NCLS; ++c) h[c] = 0;
    const int64_t nchunks = (npix + 15) >> 4;
    for (int64_t ch = (int64_t)blockIdx.x * 256 + threadIdx.x; ch < nchunks; ch += (int64_t)gridDim.x * 256) {
        const int64_t p0 = ch << 4;
        const int cnt = (int)((npix - p0) < 16 ? (npix - p0) : 16);
        uint32_t words[S][4];
        if (vec_ok && cnt == 16) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const uint4 v = *reinterpret_cast<const uint4*>(src.p[s] + p0);
                words[s][0] = v.x; words[s][1] = v.y; words[s][2] = v.z; words[s][3] = v.w;
            }
        } else {
#pragma unroll
            for (int s = 0; s < S; ++s) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint32_t wv = 0;
                    for (int b = 0; b < 4; ++b) {
                        const int i = q * 4 + b;
                        if (i < cnt) wv |= (uint32_t)src.p[s][p0 + i] << (8 * b);
                    }
                    words[s][q] = wv;
                }
            }
        }
        uint32_t res[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t r = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                uint32_t lab[S];
#pragma unroll
                for (int s = 0; s < S; ++s) lab[s] = (words[s][q] >> (8 * b)) & 0xFFu;
                int best = 0, bestc = -1;
#pragma unroll
                for (int c = 0; c < NCLS; ++c) {
                    if (c < ncls) {
                        int cn = 0;
#pragma unroll
                        for (int s = 0; s < S; ++s) cn += (lab[s] == (uint32_t)c);
                        if (cn > bestc) { bestc = cn; best = c; }     // first max (np.argmax)
                    }
                }
                const uint32_t m = bestc < thresh ? (uint32_t)fill : (uint32_t)best;
                r |= m << (8 * b);
                if (q * 4 + b < cnt) {
#pragma unroll
                    for (int c = 0; c < NCLS; ++c) h[c] += (m == (uint32_t)c);
                }
            }
            res[q] = r;
        }
        if (vec_ok && cnt == 16) {
            *reinterpret_cast<uint4*>(out + p0) = make_uint4(res[0], res[1], res[2], res[3]);
        } else {
            for (int i = 0; i < cnt; ++i) out[p0 + i] = (uint8_t)(res[i >> 2] >> (8 * (i & 3)));
        }
    }
    if (hist) {
        __shared__ uint32_t sh[NCLS];
        if (threadIdx.x < NCLS) sh[threadIdx.x] = 0;
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NCLS; ++c) {
            uint32_t v = h[c];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            if ((threadIdx.x & 63) == 0 && v) atomicAdd(&sh[c], v);
        }
        __syncthreads();
        if (threadIdx.x < ncls && sh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)sh[threadIdx.x]);
    }
}

template <int S>
static void launch_merge(int ncls, dim3 grid, hipStream_t st, const MergeSrc& ms, int64_t npix, int thresh, int fill,
                         uint8_t* out, unsigned long long* hist, int vec_ok) {
    if (ncls <= 8)
        hipLaunchKernelGGL((merge_labels_kernel<S, 8>), grid, dim3(256), 0, st, ms, npix, ncls, thresh, fill, out, hist, vec_ok);
    else if (ncls <= 16)
        hipLaunchKernelGGL((merge_labels_kernel<S, 16>), grid, dim3(256), 0, st, ms, npix, ncls, thresh, fill, out, hist, vec_ok);
    else        // 20 / 21-class self-label passes (Cityscapes / Pascal source models relabelling their own domain)
        hipLaunchKernelGGL((merge_labels_kernel<S, 32>), grid, dim3(256), 0, st, ms, npix, ncls, thresh, fill, out, hist, vec_ok);
}

// MIOU.get_iou (utilities/metrics/segmentation_miou.py:13-44) without the host round trip: per pixel
//   p = uint8(argmax_c logits) + 1,  t = uint8(target) + 1   (uint8 arithmetic: 255 wraps to 0 = ignored)
//   p = t > 0 ? p : 0;  i = (p == t) ? p : 0;  three K-bin histograms over the values 1..K (torch.histc(min=1, max=K)).
// Integer work: LDS histograms per workgroup, one 64-bit atomic per bin and workgroup.  hist = [inter | pred | mask].
__global__ __launch_bounds__(256) void miou_areas_kernel(const float* __restrict__ logits, const uint8_t* __restrict__ labels,
                                                         const int64_t* __restrict__ target, int C, int HW, int K,
                                                         int64_t total, unsigned long long* __restrict__ hist) {
    __shared__ unsigned int h[3 * 64];
    for (int i = threadIdx.x; i < 3 * K; i += 256) h[i] = 0;
    __syncthreads();
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        unsigned p;
        if (logits) {
            const int64_t n = idx / HW;
            const float* lp = logits + (size_t)n * C * HW + (idx - n * HW);
            float best = lp[0];  int bi = 0;
            for (int c = 1; c < C; ++c) { const float v = lp[(size_t)c * HW]; if (v > best) { best = v; bi = c; } }
            p = (unsigned)bi;
        } else {
            p = labels[idx];
        }
        p = (p + 1u) & 255u;
        const unsigned t = ((unsigned)(target[idx] & 255) + 1u) & 255u;
        if (t == 0) p = 0;
        const unsigned in = (p == t) ? p : 0u;
        if (in >= 1 && in <= (unsigned)K) atomicAdd(&h[in - 1], 1u);
        if (p >= 1 && p <= (unsigned)K) atomicAdd(&h[K + p - 1], 1u);
        if (t >= 1 && t <= (unsigned)K) atomicAdd(&h[2 * K + t - 1], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * K; i += 256)
        if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

}  // namespace mspl

using namespace mspl;

static int label_epilogue_impl(const float* mainp, const float* aux, int32_t N, int32_t C,
                               int32_t Hm, int32_t Wm, int32_t Ha, int32_t Wa, int32_t H, int32_t W,
                               const uint8_t* lut, uint8_t* labels, float* prob, float* kld,
                               float* main_up, float* aux_up, unsigned long long* hist, int32_t ncls, void* stream);

extern "C" int mspl_label_epilogue_fwd(const float* mainp, const float* aux, int32_t N, int32_t C,
                                       int32_t Hm, int32_t Wm, int32_t Ha, int32_t Wa, int32_t H, int32_t W,
                                       const uint8_t* lut, uint8_t* labels, float* prob, float* kld,
                                       float* main_up, float* aux_up, void* stream) {
    return label_epilogue_impl(mainp, aux, N, C, Hm, Wm, Ha, Wa, H, W, lut, labels, prob, kld, main_up, aux_up, nullptr, 0, stream);
}

extern "C" int mspl_label_epilogue_hist_fwd(const float* mainp, const float* aux, int32_t N, int32_t C,
                                            int32_t Hm, int32_t Wm, int32_t Ha, int32_t Wa, int32_t H, int32_t W,
                                            const uint8_t* lut, uint8_t* labels, float* kld, unsigned long long* hist,
                                            int32_t num_classes, void* stream) {
    MSPL_REQUIRE(labels && hist, MSPL_ERR_NULL_POINTER, "label_epilogue_hist: labels and hist are required");
    MSPL_REQUIRE(num_classes >= 1 && num_classes <= 32, MSPL_ERR_UNSUPPORTED, "label_epilogue_hist: %d classes (1..32)", num_classes);
    MSPL_REQUIRE(C <= 24, MSPL_ERR_UNSUPPORTED, "label_epilogue_hist: %d logit channels (the fused form holds <= 24 in registers; "
                 "use label_epilogue + merge_labels)", C);
    return label_epilogue_impl(mainp, aux, N, C, Hm, Wm, Ha, Wa, H, W, lut, labels, nullptr, kld, nullptr, nullptr, hist, num_classes,
                               stream);
}

static int label_epilogue_impl(const float* mainp, const float* aux, int32_t N, int32_t C,
                               int32_t Hm, int32_t Wm, int32_t Ha, int32_t Wa, int32_t H, int32_t W,
                               const uint8_t* lut, uint8_t* labels, float* prob, float* kld,
                               float* main_up, float* aux_up, unsigned long long* hist, int32_t ncls, void* stream) {
    MSPL_REQUIRE(mainp, MSPL_ERR_NULL_POINTER, "label_epilogue: null main logits");
    MSPL_REQUIRE(labels || prob || kld || main_up || aux_up, MSPL_ERR_NULL_POINTER, "label_epilogue: no output requested");
    MSPL_REQUIRE(N > 0 && C > 0 && Hm > 0 && Wm > 0 && H > 0 && W > 0 && (!aux || (Ha > 0 && Wa > 0)),
                 MSPL_ERR_BAD_SHAPE, "label_epilogue: bad shape N=%d C=%d main=%dx%d aux=%dx%d out=%dx%d",
                 N, C, Hm, Wm, Ha, Wa, H, W);
    MSPL_REQUIRE(C <= 255, MSPL_ERR_UNSUPPORTED, "label_epilogue: %d classes do not fit a uint8 label", C);
    LeGeom g;
    g.N = N; g.C = C; g.Hm = Hm; g.Wm = Wm; g.Ha = Ha; g.Wa = Wa; g.H = H; g.W = W;
    g.shm = bilinear_scale(Hm, H); g.swm = bilinear_scale(Wm, W);
    g.sha = aux ? bilinear_scale(Ha, H) : 0.f; g.swa = aux ? bilinear_scale(Wa, W) : 0.f;
    const int64_t total = (int64_t)N * H * W;
    MSPL_REQUIRE(ceil_div64(total, 256) < (1ll << 31), MSPL_ERR_BAD_SHAPE, "label_epilogue: grid too large");
    if (!prob && !main_up && !aux_up && C <= 24 && H <= 65535 && N <= 65535 &&
        (int64_t)N * C * Hm * Wm < (1ll << 31) && (int64_t)N * C * (int64_t)Ha * Wa < (1ll << 31)) {
        const dim3 grid((unsigned)ceil_div(W, 256), (unsigned)(4 * ceil_div(H, 4)), (unsigned)N);     // row slots: see the kernel
        if (C <= 8) hipLaunchKernelGGL(label_epilogue_reg_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, mainp, aux, g, lut, labels, kld, hist, ncls);
        else if (C <= 16) hipLaunchKernelGGL(label_epilogue_reg_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, mainp, aux, g, lut, labels, kld, hist, ncls);
        else hipLaunchKernelGGL(label_epilogue_reg_kernel<24>, grid, dim3(256), 0, (hipStream_t)stream, mainp, aux, g, lut, labels, kld, hist, ncls);
        MSPL_CHECK_LAUNCH("label_epilogue");
        return MSPL_OK;
    }
    MSPL_REQUIRE(!hist, MSPL_ERR_BAD_SHAPE, "label_epilogue_hist: shape outside the fused form (N=%d C=%d H=%d)", N, C, H);
    hipLaunchKernelGGL(label_epilogue_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       mainp, aux, g, lut, labels, prob, kld, main_up, aux_up, total);
    MSPL_CHECK_LAUNCH("label_epilogue");
    return MSPL_OK;
}

extern "C" int mspl_merge_labels_fwd(const uint8_t* const* src, int32_t S, int64_t npix, int32_t num_classes,
                                     int32_t thresh, int32_t fill, uint8_t* out, unsigned long long* hist,
                                     void* stream) {
    MSPL_REQUIRE(src && out, MSPL_ERR_NULL_POINTER, "merge_labels: null pointer");
    MSPL_REQUIRE(S >= 1 && S <= 8, MSPL_ERR_UNSUPPORTED, "merge_labels: %d sources (1..8)", S);
    MSPL_REQUIRE(num_classes >= 1 && num_classes <= 32, MSPL_ERR_UNSUPPORTED, "merge_labels: %d classes (1..32)", num_classes);
    MSPL_REQUIRE(fill >= 0 && fill <= 255, MSPL_ERR_BAD_SHAPE, "merge_labels: fill %d", fill);
    MSPL_REQUIRE(npix >= 0, MSPL_ERR_BAD_SHAPE, "merge_labels: negative pixel count");
    if (npix == 0) return MSPL_OK;   // empty input: nothing to write, histogram untouched
    MergeSrc ms;
    int vec_ok = (((uintptr_t)out) & 15) == 0;
    for (int s = 0; s < 8; ++s) {
        ms.p[s] = s < S ? src[s] : nullptr;
        if (s < S) {
            MSPL_REQUIRE(src[s], MSPL_ERR_NULL_POINTER, "merge_labels: source %d is null", s);
            vec_ok = vec_ok && ((((uintptr_t)src[s]) & 15) == 0);
        }
    }
    const int64_t nchunks = (npix + 15) >> 4;
    int64_t blocks = ceil_div64(nchunks, 256);
    if (blocks > 16384) blocks = 16384;   // grid-stride beyond that
    dim3 grid((unsigned)blocks);
    hipStream_t st = (hipStream_t)stream;
    switch (S) {
        case 1: launch_merge<1>(num_classes, grid, st, ms, npix, thresh, fill, out, hist, vec_ok); break;
        case 2: launch_merge<2>(num_classes, grid, st, ms, npix, thresh, fill, out, hist, vec_ok); break;
        case 3: launch_merge<3>(num_classes, grid, st, ms, npix, thresh, fill, out, hist, vec_ok); break;
        case 4: launch_merge<4>(num_classes, grid, st, ms, npix, thresh, fill, out, hist, vec_ok); break;
        case 5: launch_merge<5>(num_classes, grid, st, ms, npix, thresh, fill, out, hist, vec_ok); break;
        case 6: launch_merge<6>(num_classes, grid, st, ms, npix, thresh, fill, out, hist, vec_ok); break;
        case 7: launch_merge<7>(num_classes, grid, st, ms, npix, thresh, fill, out, hist, vec_ok); break;
        default: launch_merge<8>(num_classes, grid, st, ms, npix, thresh, fill, out, hist, vec_ok); break;
    }
    MSPL_CHECK_LAUNCH("merge_labels");
    return MSPL_OK;
}

extern "C" int mspl_miou_areas_fwd(const float* logits, const uint8_t* labels, const int64_t* target, int32_t N, int32_t C,
                                   int32_t HW, int32_t num_classes, unsigned long long* hist, void* stream) {
    MSPL_REQUIRE((logits != nullptr) != (labels != nullptr), MSPL_ERR_NULL_POINTER, "miou_areas: pass logits OR labels");
    MSPL_REQUIRE(target && hist, MSPL_ERR_NULL_POINTER, "miou_areas: null pointer");
    MSPL_REQUIRE(N > 0 && HW > 0 && (!logits || C > 0), MSPL_ERR_BAD_SHAPE, "miou_areas: bad shape N=%d C=%d HW=%d", N, C, HW);
    MSPL_REQUIRE(num_classes >= 1 && num_classes <= 64, MSPL_ERR_UNSUPPORTED, "miou_areas: %d classes (1..64)", num_classes);
    const int64_t total = (int64_t)N * HW;
    int64_t blocks = ceil_div64(total, 256 * 8);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(miou_areas_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, logits, labels, target, C, HW,
                       num_classes, total, hist);
    MSPL_CHECK_LAUNCH("miou_areas");
    return MSPL_OK;
}
