// K8/K9 label epilogue and K10 cross-source label merge -- the integer end of the pseudo-label pass.
//
// label_epilogue: model/segmentation/espdnet_ue.py:301-302 (final bilinear, align_corners=True) fused
//   with uest_seg_multi_os.py:685-691 (get_output: pred + 0.5*aux -> Softmax2d; PixelwiseKLD,
//   loss_fns/segmentation_loss.py:181-189) and :903-912 (np.argmax over classes, first max wins; id LUT).
//   Full-resolution logits are never written unless the caller asks for them.
// merge_labels: uest_seg_multi_os.py:695-718 (merge_outputs) + :919-921 (class histogram).  Pure integer,
//   S bytes in + 1 byte out per pixel; bit-exact contract.
#include <stdlib.h>

#include <algorithm>

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
                                                                 uint8_t* __restrict__ labels, float* __restrict__ kld) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    // XCD-aware row order: workgroups are dealt to the 8 XCDs round-robin by linear id, and neighbouring output rows read the
    // same source rows -- in natural order every source row was fetched into four different L2s (PMC: 189 MB read for 36 MB of
    // logits).  Row slot s = blockIdx.y maps to row (s % 4) * ceil(H/4) + s / 4, so each XCD (pair) walks one contiguous quarter.
    const int rq = (g.H + 3) >> 2;
    const int y = (int)(blockIdx.y & 3) * rq + (int)(blockIdx.y >> 2), n = blockIdx.z;
    if (x >= g.W || y >= g.H) return;
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
        labels[pix] = lut ? lut[best] : (uint8_t)best;
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
}

// Label pass form, LDS staged (the fast path): a workgroup owns LE_RB output rows x 256 output columns of one image.  The
// source rows those pixels interpolate from (both heads, all classes, the column range of the 256 pixels + one replicated
// column) are staged once in LDS with coalesced loads; a thread then produces the LE_RB pixels of its column from LDS
// (2 ds_read2_b32 per class, head and pixel instead of 4 scattered 4-byte global loads: the register form above is bound by the
// texture-address path, 104 wave-loads per pixel).  Arithmetic and its order are exactly the register form's, so the labels
// and the KL map are bit-identical.  Optionally the class histogram of the labels written: per-thread compare -> wave ballot
// -> LDS -> one row of 32 partial counts per workgroup in `hist_ws` (summed by label_hist_reduce_kernel; same-address
// device atomics from ~10^4 workgroups serialise: 73 us measured for 1.2e5 of them).
constexpr int LE_RB = 4;

struct LeStage {
    int nrm, nra;          // staged source rows per band (max over bands), main / aux
    int wms, was;          // staged columns per row incl. the replicated one (max over column blocks)
    int ncls;              // histogram bins (0: no histogram)
    unsigned magc;         // ceil(65536 / C): pair / C == (pair * magc) >> 16 for pair < 4096
    int vec;               // rows 16-byte aligned: stage with float4 chunks (wms / was are multiples of 4, columns start at a multiple of 4)
    unsigned magqm, magqa; // exact divisions by wms / 4 and was / 4: (k * mag) >> 20 (verified on the host)
};

// EXACT: the class count equals CMAX (5 / 13 / 20, the path's heads): the per-class predicates `c < C` fold away (they were two
// thirds of this kernel's 950 vector instructions per pixel).
// WMS / WAS > 0: the staged row strides are these compile-time constants (480-pixel-wide outputs: 132 / 68), so the per-class LDS
// offsets c * stride become immediate offsets of the ds_read instructions instead of one v_add_u32 per read (100 of the 590
// vector instructions per pixel).
template <int CMAX, bool EXACT, int WMS, int WAS>
__global__ __launch_bounds__(256) void label_epilogue_lds_kernel(const float* __restrict__ mainp, const float* __restrict__ auxp,
                                                                 LeGeom g, LeStage st, const uint8_t* __restrict__ lut,
                                                                 uint8_t* __restrict__ labels, float* __restrict__ kld,
                                                                 unsigned int* __restrict__ hist_ws) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* ml = smem;                                          // [nrm][C][wms]
    float* al = smem + (size_t)st.nrm * g.C * st.wms;          // [nra][C][was]
    unsigned int* sh_hist = reinterpret_cast<unsigned int*>(al + (size_t)st.nra * g.C * st.was);   // [32]
    const int tid = threadIdx.x;
    const int xb = blockIdx.x * 256;
    // XCD-aware band order (see the register form): band slot s -> band (s % 4) * ceil(nbands/4) + s / 4
    const int nbands = (g.H + LE_RB - 1) / LE_RB;
    const int bq = (nbands + 3) >> 2;
    const int band = (int)(blockIdx.y & 3) * bq + (int)(blockIdx.y >> 2), n = blockIdx.z;
    const unsigned blk = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (band >= nbands) {                                      // uniform: a padding slot of the band order
        if (hist_ws && tid < 32) hist_ws[(size_t)blk * 32 + tid] = 0u;
        return;
    }
    const int y0 = band * LE_RB;
    const int ylast = min(y0 + LE_RB, g.H) - 1, xlast = min(xb + 256, g.W) - 1;
    // uniform source ranges of this tile
    int rm_lo, rm_hi, cm_lo, cm_hi, t0, t1;  float f0, f1;
    bilinear_src(g.shm, y0, g.Hm, rm_lo, t1, f0, f1);
    bilinear_src(g.shm, ylast, g.Hm, t0, rm_hi, f0, f1);
    bilinear_src(g.swm, xb, g.Wm, cm_lo, t1, f0, f1);
    bilinear_src(g.swm, xlast, g.Wm, t0, cm_hi, f0, f1);
    int ra_lo = 0, ra_hi = -1, ca_lo = 0, ca_hi = -1;
    if (auxp) {
        bilinear_src(g.sha, y0, g.Ha, ra_lo, t1, f0, f1);
        bilinear_src(g.sha, ylast, g.Ha, t0, ra_hi, f0, f1);
        bilinear_src(g.swa, xb, g.Wa, ca_lo, t1, f0, f1);
        bilinear_src(g.swa, xlast, g.Wa, t0, ca_hi, f0, f1);
    }
    if (hist_ws && tid < 32) sh_hist[tid] = 0;

    // ---- stage.  Column j of a staged row <-> source column cmA + j, cmA = cm_lo rounded down to a multiple of 4.
    const int cmA = st.vec ? (cm_lo & ~3) : cm_lo, caA = st.vec ? (ca_lo & ~3) : ca_lo;
    {
        const int mplane = g.Hm * g.Wm, aplane = g.Ha * g.Wa;
        const float* mb = mainp + (size_t)n * g.C * mplane;
        const float* ab = auxp ? auxp + (size_t)n * g.C * aplane : nullptr;
        const int nm = (rm_hi - rm_lo + 1) * g.C, na = auxp ? (ra_hi - ra_lo + 1) * g.C : 0;
        if (st.vec) {
            // 16-byte chunks.  A wave stages PPW (row, class) pairs per step: lane -> (pair slot = lane / CPQ, chunk q = lane % CPQ)
            // is computed ONCE (CPQ = chunks per pair: wms / 4 resp. was / 4), the step loop then costs one magic division of a
            // uniform pair index per pair -- the flat per-chunk decode this replaces was ~700 of the kernel's 3300 vector
            // instructions per thread.  Four steps' loads are requested before their LDS writes.
            const int wavei = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
#pragma unroll
            for (int head = 0; head < 2; ++head) {
                const bool ism = head == 0;
                const int cq = ism ? (st.wms >> 2) : (st.was >> 2), npair = ism ? nm : na;
                if (npair == 0) continue;
                const int ppw = cq <= 64 ? 64 / cq : 1;                    // pairs per wave step (cq <= 64: checked by the launcher)
                const int slot = (int)(((unsigned)lane * (ism ? st.magqm : st.magqa)) >> 20), q = lane - slot * cq;
                const int col = (ism ? cmA : caA) + 4 * q;
                const bool lane_ok = slot < ppw && col < (ism ? g.Wm : g.Wa) && col <= (ism ? cm_hi : ca_hi);
                const float* base = ism ? mb + (size_t)rm_lo * g.Wm + col : ab + (size_t)ra_lo * g.Wa + col;
                const int plane = ism ? mplane : aplane, rowlen = ism ? g.Wm : g.Wa, wst = ism ? st.wms : st.was;
                float* dbase = (ism ? ml : al) + 4 * q;
                constexpr int UL = 4;
                for (int p0 = wavei * ppw + slot; p0 - slot < npair; p0 += 4 * ppw * UL) {
                    float4 v[UL];  int dst[UL];
#pragma unroll
                    for (int u = 0; u < UL; ++u) {
                        const int pair = p0 + u * 4 * ppw;
                        dst[u] = -1;  v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (lane_ok && pair < npair) {
                            const int row = (int)(((unsigned)pair * st.magc) >> 16), cc = pair - row * g.C;
                            v[u] = *reinterpret_cast<const float4*>(base + (size_t)cc * plane + row * rowlen);
                            dst[u] = pair * wst;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < UL; ++u)
                        if (dst[u] >= 0) *reinterpret_cast<float4*>(dbase + dst[u]) = v[u];
                }
            }
        } else {
            // 4-byte form (rows not 16-byte aligned): (row, class) pairs round-robin over the two halves of the workgroup; a
            // half's 128 lanes take the pair's columns j = lane and lane + 128; 8 pairs per thread in flight
            const int half = __builtin_amdgcn_readfirstlane(tid >> 7), lane = tid & 127;
            const int wm = cm_hi - cm_lo + 1, wa = ca_hi - ca_lo + 1;
            constexpr int UL = 8;
            for (int base = half; base < nm + na; base += 2 * UL) {
                float v[UL][2];  int dst[UL][2];
#pragma unroll
                for (int u = 0; u < UL; ++u) {
                    const int it = base + 2 * u;                                   // uniform per half
                    dst[u][0] = dst[u][1] = -1;  v[u][0] = v[u][1] = 0.f;
                    if (it < nm + na) {
                        const bool ism = it < nm;
                        const int pair = ism ? it : it - nm;
                        const int row = (int)(((unsigned)pair * st.magc) >> 16), c = pair - row * g.C;
                        const int w = ism ? wm : wa;
                        const float* src = ism ? mb + (size_t)c * mplane + (rm_lo + row) * g.Wm + cm_lo
                                               : ab + (size_t)c * aplane + (ra_lo + row) * g.Wa + ca_lo;
                        const int d0 = ism ? pair * st.wms : (int)(al - ml) + pair * st.was;
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            const int j = lane + 128 * h2;
                            if (j < w) { v[u][h2] = src[j];  dst[u][h2] = d0 + j; }
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < UL; ++u) {
                    if (dst[u][0] >= 0) ml[dst[u][0]] = v[u][0];
                    if (dst[u][1] >= 0) ml[dst[u][1]] = v[u][1];
                }
            }
        }
    }
    __syncthreads();

    const int wms_c = WMS > 0 ? WMS : st.wms, was_c = WAS > 0 ? WAS : st.was;
    // From here on a*b + c may contract to one fused multiply-add (the library is otherwise built with -ffp-contract=off): the
    // interpolation and the exp-sum arithmetic lose a third of their vector instructions; results move by one rounding of the
    // product (the reference's own ATen kernels are built with contraction on), labels are decided exactly as before wherever the
    // top-2 margin exceeds rounding level.
    const int x = xb + tid;
    unsigned int cnt_mask_lo = 0;      // labels of this thread's LE_RB pixels, 8 bits each (0xff = none)
    if (x < g.W) {
#pragma clang fp contract(fast)
        int mx0, mx1, ax0 = 0, ax1 = 0;  float mwx0, mwx1, awx0 = 0.f, awx1 = 0.f;
        bilinear_src(g.swm, x, g.Wm, mx0, mx1, mwx0, mwx1);
        if (auxp) bilinear_src(g.swa, x, g.Wa, ax0, ax1, awx0, awx1);
        const float* mcol = ml + (mx0 - cmA);
        const float* acol = al + (ax0 - caA);
        const int mdx = mx1 - mx0, adx = ax1 - ax0;             // 1, or 0 at the clamped right edge (the reference reads column i1)
        // Horizontally interpolated source rows are kept ACROSS the band's output rows: consecutive output rows interpolate between
        // the same two source rows, or the lower one of the pair becomes the upper one (x2 / x4 up-sampling: ~3.5 + ~3 row
        // interpolations per band instead of 8 + 8; 2 LDS reads and 2 vector instructions per class each).  Which rows a step
        // needs is uniform (scalar branches); same expressions, same values as interpolating both rows for every output row.
        float mtop[CMAX], mbot[CMAX], atop[CMAX], abot[CMAX];
        int mkt = -1, mkb = -1, akt = -1, akb = -1;                     // source rows held in (mtop, mbot) / (atop, abot)
#pragma unroll
        for (int c = 0; c < CMAX; ++c) { mtop[c] = mbot[c] = atop[c] = abot[c] = 0.f; }
#pragma unroll 1
        for (int r = 0; r < LE_RB; ++r) {
            const int y = y0 + r;
            unsigned lab8 = 0xffu;
            if (y < g.H) {                                              // uniform
                int my0, my1;  float mwy0, mwy1;
                bilinear_src(g.shm, y, g.Hm, my0, my1, mwy0, mwy1);     // uniform
                if (my0 != mkt) {
                    if (my0 == mkb) {
#pragma unroll
                        for (int c = 0; c < CMAX; ++c) mtop[c] = mbot[c];
                    } else {
                        const float* mr0 = mcol + (my0 - rm_lo) * (EXACT ? CMAX : g.C) * wms_c;
                        const float* mr0b = mr0 + mdx;                  // second source column (the same at the clamped edge)
#pragma unroll
                        for (int c = 0; c < CMAX; ++c)
                            if (EXACT || c < g.C) mtop[c] = mwx0 * mr0[c * wms_c] + mwx1 * mr0b[c * wms_c];
                    }
                    mkt = my0;
                }
                if (my1 != mkb) {
                    if (my1 == mkt) {
#pragma unroll
                        for (int c = 0; c < CMAX; ++c) mbot[c] = mtop[c];
                    } else {
                        const float* mr1 = mcol + (my1 - rm_lo) * (EXACT ? CMAX : g.C) * wms_c;
                        const float* mr1b = mr1 + mdx;
#pragma unroll
                        for (int c = 0; c < CMAX; ++c)
                            if (EXACT || c < g.C) mbot[c] = mwx0 * mr1[c * wms_c] + mwx1 * mr1b[c * wms_c];
                    }
                    mkb = my1;
                }
                float m[CMAX], a[CMAX];
#pragma unroll
                for (int c = 0; c < CMAX; ++c) m[c] = (EXACT || c < g.C) ? mwy0 * mtop[c] + mwy1 * mbot[c] : -INFINITY;
                if (auxp) {
                    int ay0, ay1;  float awy0, awy1;
                    bilinear_src(g.sha, y, g.Ha, ay0, ay1, awy0, awy1);
                    if (ay0 != akt) {
                        if (ay0 == akb) {
#pragma unroll
                            for (int c = 0; c < CMAX; ++c) atop[c] = abot[c];
                        } else {
                            const float* ar0 = acol + (ay0 - ra_lo) * (EXACT ? CMAX : g.C) * was_c;
                            const float* ar0b = ar0 + adx;
#pragma unroll
                            for (int c = 0; c < CMAX; ++c)
                                if (EXACT || c < g.C) atop[c] = awx0 * ar0[c * was_c] + awx1 * ar0b[c * was_c];
                        }
                        akt = ay0;
                    }
                    if (ay1 != akb) {
                        if (ay1 == akt) {
#pragma unroll
                            for (int c = 0; c < CMAX; ++c) abot[c] = atop[c];
                        } else {
                            const float* ar1 = acol + (ay1 - ra_lo) * (EXACT ? CMAX : g.C) * was_c;
                            const float* ar1b = ar1 + adx;
#pragma unroll
                            for (int c = 0; c < CMAX; ++c)
                                if (EXACT || c < g.C) abot[c] = awx0 * ar1[c * was_c] + awx1 * ar1b[c * was_c];
                        }
                        akb = ay1;
                    }
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) a[c] = (EXACT || c < g.C) ? awy0 * atop[c] + awy1 * abot[c] : -INFINITY;
                } else {
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) a[c] = (EXACT || c < g.C) ? 0.f : -INFINITY;
                }
                const size_t pix = ((size_t)n * g.H + y) * g.W + x;
                if (labels) {
                    float omax = -INFINITY;  int best = 0;
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) {
                        if (EXACT || c < g.C) { const float o = m[c] + 0.5f * a[c]; if (o > omax) { omax = o; best = c; } }   // first maximum wins
                    }
                    lab8 = lut ? (unsigned)lut[best] : (unsigned)best;
                    labels[pix] = (uint8_t)lab8;
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
                            if (EXACT || c < g.C) {
                                const float e1 = __expf(m[c] - M1);
                                S1 += e1;
                                T1 = fmaf(e1, m[c] - a[c], T1);
                                S2 += __expf(a[c] - M2);
                            }
                        }
                        k = T1 / S1 - (M1 + __logf(S1)) + (M2 + __logf(S2));
                    }
                    kld[pix] = k;
                }
            }
            cnt_mask_lo |= lab8 << (8 * r);
        }
    } else {
        cnt_mask_lo = 0xffffffffu;
    }
    if (hist_ws) {
        // counts per class over the wave (4 pixels per lane), then LDS, then this workgroup's row of the workspace
        // (one compare per (class, row) and lane; the count over the wave is a population count of the compare's lane mask on the
        // scalar unit -- the per-lane counts + six-step shuffle reduction this replaces were ~450 of the kernel's ~2 650 vector
        // instructions per wave)
        for (int c = 0; c < st.ncls; ++c) {
            unsigned cnt = 0;                                                        // uniform
#pragma unroll
            for (int r = 0; r < LE_RB; ++r)
                cnt += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(((cnt_mask_lo >> (8 * r)) & 0xffu) == (unsigned)c));
            if ((tid & 63) == 0 && cnt) atomicAdd(&sh_hist[c], cnt);
        }
        __syncthreads();
        if (tid < 32) hist_ws[(size_t)blk * 32 + tid] = tid < st.ncls ? sh_hist[tid] : 0u;
    }
}

// Sum the per-workgroup partial histograms (rows of 32 counts) into the caller's 64-bit bins: LE_RED workgroups, each sums a
// slice of the rows (8 independent coalesced loads per thread in flight) and adds its 32 sums with one atomic per bin.
constexpr int LE_RED = 16;
__global__ __launch_bounds__(256) void label_hist_reduce_kernel(const unsigned int* __restrict__ ws, int nrows, int ncls,
                                                                unsigned long long* __restrict__ hist) {
    __shared__ unsigned int part[8][32];
    const int c = threadIdx.x & 31, sub = threadIdx.x >> 5;                 // 8 rows per step and workgroup
    unsigned int s = 0;
    const int step = 8 * LE_RED;
    int r = blockIdx.x * 8 + sub;
    for (; r + 7 * step < nrows; r += 8 * step) {
        unsigned int v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ws[(size_t)(r + u * step) * 32 + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; r < nrows; r += step) s += ws[(size_t)r * 32 + c];
    part[sub][c] = s;
    __syncthreads();
    if (threadIdx.x < 32 && (int)threadIdx.x < ncls) {
        unsigned long long t = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += part[i][threadIdx.x];
        if (t) atomicAdd(&hist[threadIdx.x], t);
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

// Host twin of bilinear_src's index part (same fp32 operations; the build uses -ffp-contract=off).
static void le_host_idx(float scale, int dst, int in_size, int& i0, int& i1) {
    const float real = scale * (float)dst;
    int idx = (int)floorf(real);
    if (idx > in_size - 1) idx = in_size - 1;
    i0 = idx;
    i1 = idx + ((idx < in_size - 1) ? 1 : 0);
}

// Largest number of source positions any tile of `tile` consecutive output positions touches; with align4 the first position
// is rounded down to a multiple of 4 and the count up to a multiple of 4 (16-byte staging chunks).
static int le_span(int in_size, int out_size, int tile, bool align4) {
    const float sc = bilinear_scale(in_size, out_size);
    int best = 1;
    for (int o0 = 0; o0 < out_size; o0 += tile) {
        const int o1 = std::min(o0 + tile, out_size) - 1;
        int a0, a1, b0, b1;
        le_host_idx(sc, o0, in_size, a0, a1);
        le_host_idx(sc, o1, in_size, b0, b1);
        if (align4) a0 &= ~3;
        int cnt = b1 - a0 + 1;
        if (align4) cnt = (cnt + 3) & ~3;
        best = std::max(best, cnt);
    }
    return best;
}

static int64_t le_blocks(int N, int H, int W) { return (int64_t)ceil_div(W, 256) * (4 * ceil_div(ceil_div(H, LE_RB), 4)) * N; }

static int label_epilogue_impl(const float* mainp, const float* aux, int32_t N, int32_t C,
                               int32_t Hm, int32_t Wm, int32_t Ha, int32_t Wa, int32_t H, int32_t W,
                               const uint8_t* lut, uint8_t* labels, float* prob, float* kld,
                               float* main_up, float* aux_up, unsigned long long* hist, int32_t ncls, void* workspace,
                               int64_t workspace_bytes, void* stream) {
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
    hipStream_t st_ = (hipStream_t)stream;
    const bool label_form = !prob && !main_up && !aux_up && C <= 24 && H <= 65535 * LE_RB / 4 && N <= 65535 &&
                            (int64_t)N * C * Hm * Wm < (1ll << 31) && (int64_t)N * C * (int64_t)Ha * Wa < (1ll << 31);
    if (label_form) {
        // LDS-staged form when the tile's source rows fit (they do for the x2 / x4 heads of the path); else the register form
        LeStage st;
        st.vec = (Wm % 4 == 0) && (!aux || Wa % 4 == 0) && (((uintptr_t)mainp) & 15) == 0 && (((uintptr_t)aux) & 15) == 0;
        st.nrm = le_span(Hm, H, LE_RB, false);  st.wms = le_span(Wm, W, 256, st.vec);
        st.nra = aux ? le_span(Ha, H, LE_RB, false) : 0;  st.was = aux ? le_span(Wa, W, 256, st.vec) : 4;
        if (!st.vec) { st.wms = (st.wms + 3) & ~3; st.was = (st.was + 3) & ~3; }      // keep the second head's base 16-byte aligned
        st.ncls = hist ? ncls : 0;
        st.magc = (65536u + (unsigned)C - 1) / (unsigned)C;
        st.magqm = ((1u << 20) + (unsigned)(st.wms >> 2) - 1) / (unsigned)(st.wms >> 2);
        st.magqa = ((1u << 20) + (unsigned)(st.was >> 2) - 1) / (unsigned)(st.was >> 2);
        {   // exactness of the two magic divisions over the ranges used (else: 4-byte staging, which needs neither)
            bool ok = true;                                        // lane / (chunks per pair) for lane < 64
            for (int k = 0; k < 64 && ok; ++k) ok = (int)(((unsigned)k * st.magqm) >> 20) == k / (st.wms >> 2);
            for (int k = 0; k < 64 && ok; ++k) ok = (int)(((unsigned)k * st.magqa) >> 20) == k / (st.was >> 2);
            if (!ok) st.vec = 0;
        }
        const size_t lds = ((size_t)st.nrm * C * st.wms + (size_t)st.nra * C * st.was + 32) * sizeof(float);
        static const int dbg_reg = MSPL_TUNE_INT("MSPL_LE_REG", 0);     // tuning aid: force the register form
        if (lds <= 128 * 1024 && (st.nrm + st.nra) * C < 4096 && st.wms <= 256 && st.was <= 256 && (!st.vec || (st.wms <= 256 && st.was <= 256)) && !(dbg_reg && !hist)) {
            const dim3 grid((unsigned)ceil_div(W, 256), (unsigned)(4 * ceil_div(ceil_div(H, LE_RB), 4)), (unsigned)N);
            unsigned int* ws = nullptr;
            if (hist) {
                const int64_t need = le_blocks(N, H, W) * 32 * (int64_t)sizeof(unsigned int);
                MSPL_REQUIRE(workspace && workspace_bytes >= need, MSPL_ERR_BAD_SHAPE,
                             "label_epilogue_hist: workspace of %lld bytes, need %lld (mspl_label_epilogue_hist_workspace_bytes)",
                             (long long)workspace_bytes, (long long)need);
                ws = (unsigned int*)workspace;
            }
            static const bool big_lds = [] {          // up to 128 KB of staged rows (the default cap is 64 KB): every instantiation launched below
                bool ok = true;
#define MSPL_LE_ATTR(CM, EX, A, B) ok = hipFuncSetAttribute((const void*)label_epilogue_lds_kernel<CM, EX, A, B>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess && ok
                MSPL_LE_ATTR(5, true, 0, 0);  MSPL_LE_ATTR(5, true, 132, 68);
                MSPL_LE_ATTR(13, true, 0, 0); MSPL_LE_ATTR(13, true, 132, 68);
                MSPL_LE_ATTR(20, true, 0, 0); MSPL_LE_ATTR(20, true, 132, 68);
                MSPL_LE_ATTR(8, false, 0, 0); MSPL_LE_ATTR(16, false, 0, 0); MSPL_LE_ATTR(24, false, 0, 0);
#undef MSPL_LE_ATTR
                (void)hipGetLastError();
                return ok;
            }();
            MSPL_REQUIRE(big_lds || lds <= 64 * 1024, MSPL_ERR_HIP, "label_epilogue: could not raise the dynamic LDS limit");
#define MSPL_LE_LAUNCH(CM, EX) do { if (wide480) hipLaunchKernelGGL((label_epilogue_lds_kernel<CM, EX, 132, 68>), grid, dim3(256), lds, st_, mainp, aux, g, st, lut, labels, kld, ws); \
                                   else hipLaunchKernelGGL((label_epilogue_lds_kernel<CM, EX, 0, 0>), grid, dim3(256), lds, st_, mainp, aux, g, st, lut, labels, kld, ws); } while (0)
            const bool wide480 = (C == 5 || C == 13 || C == 20) && st.wms == 132 && (!aux || st.was == 68);     // the x2 / x4 heads of 480-pixel-wide images
            if (C == 5) MSPL_LE_LAUNCH(5, true);
            else if (C == 13) MSPL_LE_LAUNCH(13, true);
            else if (C == 20) MSPL_LE_LAUNCH(20, true);
            else if (C <= 8) { if (wide480) return MSPL_ERR_UNSUPPORTED; hipLaunchKernelGGL((label_epilogue_lds_kernel<8, false, 0, 0>), grid, dim3(256), lds, st_, mainp, aux, g, st, lut, labels, kld, ws); }
            else if (C <= 16) hipLaunchKernelGGL((label_epilogue_lds_kernel<16, false, 0, 0>), grid, dim3(256), lds, st_, mainp, aux, g, st, lut, labels, kld, ws);
            else hipLaunchKernelGGL((label_epilogue_lds_kernel<24, false, 0, 0>), grid, dim3(256), lds, st_, mainp, aux, g, st, lut, labels, kld, ws);
#undef MSPL_LE_LAUNCH
            MSPL_CHECK_LAUNCH("label_epilogue");
            if (hist) {
                hipLaunchKernelGGL(label_hist_reduce_kernel, dim3(LE_RED), dim3(256), 0, st_, ws, (int)le_blocks(N, H, W), ncls, hist);
                MSPL_CHECK_LAUNCH("label_epilogue(histogram)");
            }
            return MSPL_OK;
        }
        MSPL_REQUIRE(!hist, MSPL_ERR_UNSUPPORTED, "label_epilogue_hist: tile does not fit the LDS-staged form (N=%d C=%d %dx%d -> %dx%d)",
                     N, C, Hm, Wm, H, W);
        const dim3 grid((unsigned)ceil_div(W, 256), (unsigned)(4 * ceil_div(H, 4)), (unsigned)N);     // row slots: see the kernel
        MSPL_REQUIRE(H <= 65535, MSPL_ERR_BAD_SHAPE, "label_epilogue: %d rows", H);
        if (C <= 8) hipLaunchKernelGGL(label_epilogue_reg_kernel<8>, grid, dim3(256), 0, st_, mainp, aux, g, lut, labels, kld);
        else if (C <= 16) hipLaunchKernelGGL(label_epilogue_reg_kernel<16>, grid, dim3(256), 0, st_, mainp, aux, g, lut, labels, kld);
        else hipLaunchKernelGGL(label_epilogue_reg_kernel<24>, grid, dim3(256), 0, st_, mainp, aux, g, lut, labels, kld);
        MSPL_CHECK_LAUNCH("label_epilogue");
        return MSPL_OK;
    }
    MSPL_REQUIRE(!hist, MSPL_ERR_UNSUPPORTED, "label_epilogue_hist: shape outside the label-pass form (N=%d C=%d H=%d)", N, C, H);
    hipLaunchKernelGGL(label_epilogue_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, st_,
                       mainp, aux, g, lut, labels, prob, kld, main_up, aux_up, total);
    MSPL_CHECK_LAUNCH("label_epilogue");
    return MSPL_OK;
}

extern "C" int mspl_label_epilogue_fwd(const float* mainp, const float* aux, int32_t N, int32_t C,
                                       int32_t Hm, int32_t Wm, int32_t Ha, int32_t Wa, int32_t H, int32_t W,
                                       const uint8_t* lut, uint8_t* labels, float* prob, float* kld,
                                       float* main_up, float* aux_up, void* stream) {
    return label_epilogue_impl(mainp, aux, N, C, Hm, Wm, Ha, Wa, H, W, lut, labels, prob, kld, main_up, aux_up, nullptr, 0, nullptr, 0,
                               stream);
}

extern "C" int64_t mspl_label_epilogue_hist_workspace_bytes(int32_t N, int32_t H, int32_t W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    return le_blocks(N, H, W) * 32 * (int64_t)sizeof(unsigned int);
}

extern "C" int mspl_label_epilogue_hist_fits(int32_t N, int32_t C, int32_t Hm, int32_t Wm, int32_t Ha, int32_t Wa, int32_t H, int32_t W) {
    // the conditions of the LDS-staged label form in label_epilogue_impl (the only form that accumulates the histogram)
    if (N <= 0 || C <= 0 || C > 24 || Hm <= 0 || Wm <= 0 || H <= 0 || W <= 0 || Ha < 0 || Wa < 0) return 0;
    const bool aux = Ha > 0 && Wa > 0;
    if (!(H <= 65535 * LE_RB / 4 && N <= 65535 && (int64_t)N * C * Hm * Wm < (1ll << 31) && (int64_t)N * C * (int64_t)Ha * Wa < (1ll << 31))) return 0;
    const bool vec = (Wm % 4 == 0) && (!aux || Wa % 4 == 0);      // (pointer alignment can only shrink the tile: unaligned = 4-byte staging)
    const int nrm = le_span(Hm, H, LE_RB, false), nra = aux ? le_span(Ha, H, LE_RB, false) : 0;
    int wms = le_span(Wm, W, 256, vec), was = aux ? le_span(Wa, W, 256, vec) : 4;
    if (!vec) { wms = (wms + 3) & ~3; was = (was + 3) & ~3; }
    const int wms4 = (le_span(Wm, W, 256, true) + 3) & ~3, was4 = aux ? (le_span(Wa, W, 256, true) + 3) & ~3 : 4;   // worst case of either staging
    wms = std::max(wms, wms4); was = std::max(was, was4);
    const size_t lds = ((size_t)nrm * C * wms + (size_t)nra * C * was + 32) * sizeof(float);
    return lds <= 128 * 1024 && (nrm + nra) * C < 4096 && wms <= 256 && was <= 256;
}

extern "C" int mspl_label_epilogue_hist_fwd(const float* mainp, const float* aux, int32_t N, int32_t C,
                                            int32_t Hm, int32_t Wm, int32_t Ha, int32_t Wa, int32_t H, int32_t W,
                                            const uint8_t* lut, uint8_t* labels, float* kld, unsigned long long* hist,
                                            int32_t num_classes, void* workspace, int64_t workspace_bytes, void* stream) {
    MSPL_REQUIRE(labels && hist, MSPL_ERR_NULL_POINTER, "label_epilogue_hist: labels and hist are required");
    MSPL_REQUIRE(num_classes >= 1 && num_classes <= 32, MSPL_ERR_UNSUPPORTED, "label_epilogue_hist: %d classes (1..32)", num_classes);
    MSPL_REQUIRE(C <= 24, MSPL_ERR_UNSUPPORTED, "label_epilogue_hist: %d logit channels (the fused form holds <= 24 in registers; "
                 "use label_epilogue + merge_labels)", C);
    return label_epilogue_impl(mainp, aux, N, C, Hm, Wm, Ha, Wa, H, W, lut, labels, nullptr, kld, nullptr, nullptr, hist, num_classes,
                               workspace, workspace_bytes, stream);
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
