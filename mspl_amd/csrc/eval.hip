// Evaluation step epilogue: what val_seg_ue (utilities/train_eval_seg.py:249-324) and the body of test()
// (uest_seg_multi_os.py:1150-1200) do per batch AFTER the network, in one pass over the low-resolution heads:
//     o      = upsample(main) + aux_weight * upsample(aux)          bilinear, align_corners (espdnet_ue.py:301-302; `out + 0.5 * aux`)
//     loss   = CrossEntropyLoss(weight, ignore_index)(o, target)    = sum_valid w[t] * (lse(o) - o[t]) / sum_valid w[t]
//     areas  = MIOU.get_iou(o, target)                              argmax, the reference's uint8 +1 arithmetic, three histograms
// The reference writes both heads at full resolution (2 * C * H * W floats per image), adds them with an ATen kernel, runs the
// loss, copies prediction and target to the host and calls torch.histc three times.  Here full-resolution logits never exist.
#include "common.hpp"

namespace mspl {

constexpr int EV_RB = 16;       // output rows per workgroup (fewer workgroups -> fewer same-address atomics)

struct EvGeom {
    int N, C, Hm, Wm, Ha, Wa, H, W;
    float shm, swm, sha, swa, aw;
    int ignore_index, K;
};

__global__ __launch_bounds__(256) void eval_epilogue_kernel(const float* __restrict__ mainp, const float* __restrict__ auxp,
                                                            const int64_t* __restrict__ target, const float* __restrict__ cw,
                                                            EvGeom g, double* __restrict__ sums, unsigned long long* __restrict__ areas,
                                                            uint8_t* __restrict__ labels) {
    __shared__ unsigned int hsh[3 * 64];
    __shared__ double red[2][4];
    for (int i = threadIdx.x; i < 3 * g.K; i += 256) hsh[i] = 0;
    __syncthreads();
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.z;
    const int y0 = blockIdx.y * EV_RB;
    const size_t hw = (size_t)g.H * g.W;
    const int mplane = g.Hm * g.Wm, aplane = g.Ha * g.Wa;
    double s_loss = 0.0, s_w = 0.0;
    if (x < g.W) {
        int mx0, mx1;  float mwx0, mwx1;
        bilinear_src(g.swm, x, g.Wm, mx0, mx1, mwx0, mwx1);
        int ax0 = 0, ax1 = 0;  float awx0 = 0.f, awx1 = 0.f;
        if (auxp) bilinear_src(g.swa, x, g.Wa, ax0, ax1, awx0, awx1);
        const float* mb = mainp + (size_t)n * g.C * mplane;
        const float* ab = auxp ? auxp + (size_t)n * g.C * aplane : nullptr;
        for (int y = y0; y < min(y0 + EV_RB, g.H); ++y) {
            int my0, my1;  float mwy0, mwy1;
            bilinear_src(g.shm, y, g.Hm, my0, my1, mwy0, mwy1);                      // uniform
            int ay0 = 0, ay1 = 0;  float awy0 = 0.f, awy1 = 0.f;
            if (ab) bilinear_src(g.sha, y, g.Ha, ay0, ay1, awy0, awy1);
            const int m00 = my0 * g.Wm + mx0, m01 = my0 * g.Wm + mx1, m10 = my1 * g.Wm + mx0, m11 = my1 * g.Wm + mx1;
            const int a00 = ay0 * g.Wa + ax0, a01 = ay0 * g.Wa + ax1, a10 = ay1 * g.Wa + ax0, a11 = ay1 * g.Wa + ax1;
            const size_t pix = (size_t)n * hw + (size_t)y * g.W + x;
            const int64_t t64 = target[pix];
            float omax = -INFINITY, S = 0.f, ot = 0.f;
            int best = 0;
#pragma unroll 2
            for (int c = 0; c < g.C; ++c) {
                const float* p = mb + c * mplane;
                float o = mwy0 * (mwx0 * p[m00] + mwx1 * p[m01]) + mwy1 * (mwx0 * p[m10] + mwx1 * p[m11]);
                if (ab) {
                    const float* q = ab + c * aplane;
                    const float a = awy0 * (awx0 * q[a00] + awx1 * q[a01]) + awy1 * (awx0 * q[a10] + awx1 * q[a11]);
                    o = o + g.aw * a;
                }
                if (o > omax) { S = S * expf(omax - o) + 1.f; omax = o; best = c; }      // strict '>': first maximum, like torch.max
                else S += expf(o - omax);
                if ((int64_t)c == t64) ot = o;
            }
            if (labels) labels[pix] = (uint8_t)best;
            if (t64 != (int64_t)g.ignore_index && t64 >= 0 && t64 < g.C) {
                const float wt = cw ? cw[t64] : 1.f;
                s_loss += (double)(wt * ((omax + logf(S)) - ot));
                s_w += (double)wt;
            }
            // MIOU.get_iou in the reference's uint8 arithmetic (segmentation_miou.py:28-41): +1, 255 wraps to 0 = ignored
            unsigned p8 = ((unsigned)best + 1u) & 255u;
            const unsigned t8 = ((unsigned)(t64 & 255) + 1u) & 255u;
            if (t8 == 0) p8 = 0;
            const unsigned in8 = (p8 == t8) ? p8 : 0u;
            if (in8 >= 1 && in8 <= (unsigned)g.K) atomicAdd(&hsh[in8 - 1], 1u);
            if (p8 >= 1 && p8 <= (unsigned)g.K) atomicAdd(&hsh[g.K + p8 - 1], 1u);
            if (t8 >= 1 && t8 <= (unsigned)g.K) atomicAdd(&hsh[2 * g.K + t8 - 1], 1u);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s_loss += __shfl_down(s_loss, o, 64); s_w += __shfl_down(s_w, o, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s_loss; red[1][threadIdx.x >> 6] = s_w; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&sums[0], (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
        atomicAdd(&sums[1], (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
    }
    for (int i = threadIdx.x; i < 3 * g.K; i += 256)
        if (hsh[i]) atomicAdd(&areas[i], (unsigned long long)hsh[i]);
}

// AverageMeter.update(loss.item(), n) of the loop (train_eval_seg.py:297) on the device: acc[0] += (sums[0] / sums[1]) * n,
// acc[1] += n, then the batch sums are cleared for the next batch.  One thread.
__global__ void eval_batch_finalize_kernel(double* __restrict__ sums, double* __restrict__ acc, double nimg) {
    const double l = sums[1] != 0.0 ? sums[0] / sums[1] : 0.0 / 0.0;        // CrossEntropyLoss of an all-ignored batch is NaN
    acc[0] += l * nimg;
    acc[1] += nimg;
    acc[2] = l;                                                           // this batch's loss (what `loss.item()` was)
    sums[0] = 0.0;
    sums[1] = 0.0;
}

}  // namespace mspl

using namespace mspl;

extern "C" int mspl_eval_epilogue_fwd(const float* mainp, const float* aux, const int64_t* target, const float* class_weights,
                                      int32_t N, int32_t C, int32_t Hm, int32_t Wm, int32_t Ha, int32_t Wa, int32_t H, int32_t W,
                                      float aux_weight, int32_t ignore_index, int32_t miou_classes, double* loss_sums,
                                      unsigned long long* areas, uint8_t* labels, void* stream) {
    MSPL_REQUIRE(mainp && target && loss_sums && areas, MSPL_ERR_NULL_POINTER, "eval_epilogue: null pointer");
    MSPL_REQUIRE(N > 0 && C > 0 && Hm > 0 && Wm > 0 && H > 0 && W > 0 && (!aux || (Ha > 0 && Wa > 0)), MSPL_ERR_BAD_SHAPE,
                 "eval_epilogue: bad shape N=%d C=%d main=%dx%d aux=%dx%d out=%dx%d", N, C, Hm, Wm, Ha, Wa, H, W);
    MSPL_REQUIRE(C <= 255, MSPL_ERR_UNSUPPORTED, "eval_epilogue: %d classes do not fit the reference's uint8 prediction", C);
    MSPL_REQUIRE(miou_classes >= 1 && miou_classes <= 64, MSPL_ERR_UNSUPPORTED, "eval_epilogue: %d MIOU classes (1..64)", miou_classes);
    MSPL_REQUIRE(N <= 65535 && ceil_div(H, EV_RB) <= 65535, MSPL_ERR_BAD_SHAPE, "eval_epilogue: grid too large");
    MSPL_REQUIRE((int64_t)C * Hm * Wm < (1ll << 31) && (int64_t)C * (int64_t)Ha * Wa < (1ll << 31), MSPL_ERR_BAD_SHAPE,
                 "eval_epilogue: head too large for 32-bit plane offsets");
    EvGeom g;
    g.N = N; g.C = C; g.Hm = Hm; g.Wm = Wm; g.Ha = aux ? Ha : 0; g.Wa = aux ? Wa : 0; g.H = H; g.W = W;
    g.shm = bilinear_scale(Hm, H); g.swm = bilinear_scale(Wm, W);
    g.sha = aux ? bilinear_scale(Ha, H) : 0.f; g.swa = aux ? bilinear_scale(Wa, W) : 0.f;
    g.aw = aux_weight; g.ignore_index = ignore_index; g.K = miou_classes;
    const dim3 grid((unsigned)ceil_div(W, 256), (unsigned)ceil_div(H, EV_RB), (unsigned)N);
    hipLaunchKernelGGL(eval_epilogue_kernel, grid, dim3(256), 0, (hipStream_t)stream, mainp, aux, target, class_weights, g, loss_sums,
                       areas, labels);
    MSPL_CHECK_LAUNCH("eval_epilogue");
    return MSPL_OK;
}

extern "C" int mspl_eval_batch_finalize(double* loss_sums, double* acc, int32_t batch_images, void* stream) {
    MSPL_REQUIRE(loss_sums && acc, MSPL_ERR_NULL_POINTER, "eval_batch_finalize: null pointer");
    MSPL_REQUIRE(batch_images > 0, MSPL_ERR_BAD_SHAPE, "eval_batch_finalize: %d images", batch_images);
    hipLaunchKernelGGL(eval_batch_finalize_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, loss_sums, acc, (double)batch_images);
    MSPL_CHECK_LAUNCH("eval_batch_finalize");
    return MSPL_OK;
}
