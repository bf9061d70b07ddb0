// Pieces of the supervised loop (train_segmentation.py / utilities/train_eval_seg.py) that the frozen-BN uest
// step does not need:
//   bn_batch_stats   nn.BatchNorm2d in train(): per-channel mean / biased variance over (N,H,W) + the running-statistics
//                    update (momentum, unbiased variance), model.train() at utilities/train_eval_seg.py:174
//   sgd_step         torch.optim.SGD(momentum, weight_decay) on a flat fp32 buffer, train_segmentation.py:253
// Both are HBM streaming passes (4 B read per element for the statistics; 12 B read + 8 B written per parameter for SGD).
#include "common.hpp"

namespace mspl {

// grid (chunks, N, C): one workgroup sums a slice of one plane, shifted by k_c = z[0,c,0] so that E[(x-k)^2] - E[x-k]^2 does
// not cancel when |mean| >> std; fp32 per thread (<= a few hundred terms), double across threads and workgroups.
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ z, int C, int HW, int per_block,
                                                       double* __restrict__ ws) {
    const int c = blockIdx.z, n = blockIdx.y;
    const float k = z[(size_t)c * HW];
    const float* p = z + ((size_t)n * C + c) * (size_t)HW;
    const int lo = blockIdx.x * per_block;
    const int hi = min(HW, lo + per_block);
    float s1 = 0.f, s2 = 0.f;
    if ((HW & 3) == 0 && (per_block & 3) == 0) {
        for (int i = lo + threadIdx.x * 4; i < hi; i += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(p + i);
            const float a = v.x - k, b = v.y - k, cc = v.z - k, d = v.w - k;
            s1 += (a + b) + (cc + d);
            s2 += (a * a + b * b) + (cc * cc + d * d);
        }
    } else {
        for (int i = lo + threadIdx.x; i < hi; i += 256) {
            const float a = p[i] - k;
            s1 += a;
            s2 += a * a;
        }
    }
    double d1 = s1, d2 = s2;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        d1 += __shfl_down(d1, o, 64);
        d2 += __shfl_down(d2, o, 64);
    }
    __shared__ double part[8];
    if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6] = d1; part[4 + (threadIdx.x >> 6)] = d2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(ws + 2 * c, (part[0] + part[1]) + (part[2] + part[3]));
        atomicAdd(ws + 2 * c + 1, (part[4] + part[5]) + (part[6] + part[7]));
    }
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ z, const double* __restrict__ ws, int C, int HW,
                                                          double M, float eps, float momentum, float* __restrict__ running_mean,
                                                          float* __restrict__ running_var, float* __restrict__ mean,
                                                          float* __restrict__ invstd, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ scale,
                                                          float* __restrict__ shift) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double k = z[(size_t)c * HW];
    const double e1 = ws[2 * c] / M, e2 = ws[2 * c + 1] / M;
    const double mu = k + e1;
    double var = e2 - e1 * e1;
    if (var < 0.0) var = 0.0;
    const float mf = (float)mu, isf = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = mf;
    invstd[c] = isf;
    if (scale) {                                           // the fold the affine/PReLU kernel applies: (gamma * invstd, beta - mean * gamma * invstd)
        const float sc = gamma[c] * isf;
        scale[c] = sc;
        shift[c] = beta[c] + (-mf) * sc;
    }
    if (running_mean) {
        const double unbiased = M > 1.0 ? var * (M / (M - 1.0)) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
}

// Statistics + finalize in ONE launch: the workgroup that adds the last partial of a channel (device-scope counter; the classic
// "last block done" pattern with the partials carried by atomics: adds -> wait for their acknowledgement -> counter) finishes that channel -- mean, 1/std, the fold, the running-statistics
// update -- and hands the channel's workspace back zeroed, so the caller keeps ONE persistent workspace per BatchNorm (zeroed once) and
// neither a memset nor a finalize launch runs per call.  ws: [2C doubles | C counters].
__global__ __launch_bounds__(256) void bn_stats_fused_kernel(const float* __restrict__ z, int N, int C, int HW, unsigned per_channel,
                                                             double* __restrict__ ws, unsigned* __restrict__ cnt, double M, float eps,
                                                             float momentum, float* __restrict__ running_mean,
                                                             float* __restrict__ running_var, float* __restrict__ mean,
                                                             float* __restrict__ invstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ scale,
                                                             float* __restrict__ shift, long long* __restrict__ nbt) {
    // grid (parts, C): a workgroup owns one of `parts` contiguous ranges of the channel's N * HW elements (the N planes taken as one
    // sequence), so a channel sees `parts` atomic chains whatever N is (N * chunks before: 25 of the 33 us at 16 x 16 x 144x240)
    const int c = blockIdx.y, rng = blockIdx.x, parts = gridDim.x;
    const float k = z[(size_t)c * HW];
    const bool vec = (HW & 3) == 0;
    const int L = vec ? (HW >> 2) : HW;
    const long long total = (long long)N * L;
    const long long per = (total + parts - 1) / parts;
    const long long u0 = (long long)rng * per, u1 = min(total, u0 + per);
    const int step_n = 256 / L, step_q = 256 - step_n * L;
    long long u = u0 + threadIdx.x;
    int n = (int)(u / L), q = (int)(u - (long long)n * L);
    auto advance = [&]() { u += 256; q += step_q; n += step_n; if (q >= L) { q -= L; ++n; } };
    float s1 = 0.f, s2 = 0.f;
    if (vec) {
        auto acc4 = [&](const float4& v) {
            const float a = v.x - k, b = v.y - k, cc = v.z - k, d = v.w - k;
            s1 += (a + b) + (cc + d);
            s2 += (a * a + b * b) + (cc * cc + d * d);
        };
        while (u + 768 < u1) {                      // four 16-byte loads in flight per thread
            const float4* p0 = reinterpret_cast<const float4*>(z + ((size_t)n * C + c) * (size_t)HW) + q;  advance();
            const float4* p1 = reinterpret_cast<const float4*>(z + ((size_t)n * C + c) * (size_t)HW) + q;  advance();
            const float4* p2 = reinterpret_cast<const float4*>(z + ((size_t)n * C + c) * (size_t)HW) + q;  advance();
            const float4* p3 = reinterpret_cast<const float4*>(z + ((size_t)n * C + c) * (size_t)HW) + q;  advance();
            const float4 v0 = *p0, v1 = *p1, v2 = *p2, v3 = *p3;
            acc4(v0); acc4(v1); acc4(v2); acc4(v3);
        }
        while (u < u1) {
            const float4 v = reinterpret_cast<const float4*>(z + ((size_t)n * C + c) * (size_t)HW)[q];
            advance();
            acc4(v);
        }
    } else {
        while (u < u1) {
            const float a = z[((size_t)n * C + c) * (size_t)HW + q] - k;
            advance();
            s1 += a;
            s2 += a * a;
        }
    }
    double d1 = s1, d2 = s2;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        d1 += __shfl_down(d1, o, 64);
        d2 += __shfl_down(d2, o, 64);
    }
    __shared__ double part[8];
    if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6] = d1; part[4 + (threadIdx.x >> 6)] = d2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(ws + 2 * c, (part[0] + part[1]) + (part[2] + part[3]));
        atomicAdd(ws + 2 * c + 1, (part[4] + part[5]) + (part[6] + part[7]));
        // the sums travel by device-scope atomics (performed at the memory side): waiting for them to be acknowledged orders them before
        // the counter atomic -- no __threadfence(), whose L2 write-back costs microseconds per workgroup on this chip (measured: the
        // iteration went from 15 to 28 ms with two fences per workgroup)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (atomicAdd(cnt + c, 1u) == per_channel - 1) {                 // every other workgroup of this channel has added
            // read-and-clear with atomics (they execute at the memory side, after every add above)
            const double e1 = __longlong_as_double((long long)atomicExch(reinterpret_cast<unsigned long long*>(ws + 2 * c), 0ull)) / M;
            const double e2 = __longlong_as_double((long long)atomicExch(reinterpret_cast<unsigned long long*>(ws + 2 * c + 1), 0ull)) / M;
            cnt[c] = 0u;
            const double mu = (double)k + e1;
            double var = e2 - e1 * e1;
            if (var < 0.0) var = 0.0;
            const float mf = (float)mu, isf = (float)(1.0 / sqrt(var + (double)eps));
            mean[c] = mf;
            invstd[c] = isf;
            const float sc = gamma[c] * isf;
            scale[c] = sc;
            shift[c] = beta[c] + (-mf) * sc;
            if (running_mean) {
                const double unbiased = M > 1.0 ? var * (M / (M - 1.0)) : var;
                running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
                running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
            }
            if (nbt && c == 0) *nbt += 1;                               // nn.BatchNorm2d's num_batches_tracked (one writer)
        }
    }
}

// Backward of the batch statistics' dependence on z (autograd.BNBatchStatsFn): with t = d scale - mean * d shift per channel,
//   d gamma = t * invstd,   gz = p * z + q  with  p = -gamma * t * invstd^3 / M,  q = -d shift * scale / M - p * mean.
// One launch of C threads instead of eight ATen launches on C-element tensors per BatchNorm (~110 BatchNorms per iteration).
__global__ __launch_bounds__(256) void bn_stats_bwd_coeffs_kernel(const float* __restrict__ gsc, const float* __restrict__ gsh,
                                                                  const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd, const float* __restrict__ scale,
                                                                  int C, float inv_m, int accumulate, float* __restrict__ ggamma,
                                                                  float* __restrict__ gbeta, float* __restrict__ pc,
                                                                  float* __restrict__ qc) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float is = invstd[c], mu = mean[c];
    const float t = gsc[c] + (-mu) * gsh[c];
    // accumulate: ggamma / gbeta are the parameters' own gradient buffers (one writer per element: plain read-modify-write)
    ggamma[c] = accumulate ? ggamma[c] + t * is : t * is;
    if (gbeta) gbeta[c] = accumulate ? gbeta[c] + gsh[c] : gsh[c];
    const float p = ((gamma[c] * t) * (is * is * is)) * (-inv_m);
    pc[c] = p;
    qc[c] = (gsh[c] * scale[c]) * (-inv_m) - p * mu;
}

// torch.optim.SGD: g += wd*p;  buf = first ? g : momentum*buf + g;  p -= lr*buf      (dampening 0, no Nesterov)
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                  int64_t n, float lr, float momentum, float wd, int first) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float gi = g[i];
    const float pi = p[i];
    if (wd != 0.f) gi = gi + wd * pi;
    float b = gi;
    if (momentum != 0.f) {
        b = first ? gi : momentum * buf[i] + gi;
        buf[i] = b;
    }
    p[i] = pi - lr * b;
}

// Batch-statistics path of a BatchNorm over a concatenation whose gradient is kept BRANCH-major (the pyramid body):
//   g[i][n][c][.] += p[i*P + c] * z[n][i*P + c][.] + q[i*P + c]        g: (nb, N, P, HW), z: (N, nb*P, HW)
__global__ __launch_bounds__(256) void bn_stats_path_add_kernel(float* __restrict__ g, const float* __restrict__ z,
                                                                const float* __restrict__ p, const float* __restrict__ q, int N, int P,
                                                                int nb, int HW4) {
    const long long plane = (long long)blockIdx.z * gridDim.y + blockIdx.y;      // (i * N + n) * P + c
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (plane >= (long long)nb * N * P || t >= HW4) return;
    const int c = (int)(plane % P);
    const long long in_ = plane / P;
    const int n = (int)(in_ % N), i = (int)(in_ / N);
    const int ch = i * P + c;
    const float pc = p[ch], qc = q[ch];
    float4* gp = reinterpret_cast<float4*>(g) + plane * HW4 + t;
    const float4 zv = reinterpret_cast<const float4*>(z)[((long long)n * nb * P + ch) * HW4 + t];
    float4 gv = *gp;
    gv.x += fmaf(pc, zv.x, qc); gv.y += fmaf(pc, zv.y, qc); gv.z += fmaf(pc, zv.z, qc); gv.w += fmaf(pc, zv.w, qc);
    *gp = gv;
}

// ---- small planes (levels 3-5: N * HW <= ~40 k values per channel): the whole BatchNorm + PReLU node of a channel in ONE workgroup and
// ONE launch each way.  Forward: shifted sums of the channel (fp32 per thread, double across the workgroup), mean / invstd / fold /
// running statistics, then the apply pass over the same values (L2 hits); backward: the three channel sums, (d gamma, d beta, p, q),
// then gz = p * z + q + direct gradient.  No atomics, no workspace, no second launch: at these sizes the two-launch forms are four
// launch latencies for ~2 us of data movement each (45 of the supervised iteration's 70 BatchNorms).
__device__ __forceinline__ double bn_block_sum(double v, double* part) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();                                   // part may still be read from a previous call
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += part[w];
    return t;
}

__global__ __launch_bounds__(1024) void bn_train_small_fwd_kernel(const float* __restrict__ z, const float* __restrict__ res,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 const float* __restrict__ alpha, int N, int C, int HW, float eps,
                                                                 float momentum, float* __restrict__ running_mean,
                                                                 float* __restrict__ running_var, long long* __restrict__ nbt,
                                                                 float* __restrict__ mean, float* __restrict__ invstd,
                                                                 float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ y) {
    __shared__ double part[16];
    __shared__ float fold[2];
    const int c = blockIdx.x, tid = threadIdx.x;
    const bool vec = (HW & 3) == 0;
    const int L = vec ? (HW >> 2) : HW, total = N * L;
    const int NT = blockDim.x;                  // 1024, or 256 for the smallest channels
    const int step_n = NT / L, step_q = NT - step_n * L;
    const float k = z[(size_t)c * HW];
    float s1 = 0.f, s2 = 0.f;
    {
        int u = tid, n = u / L, q = u - n * L;
        auto advance = [&]() { u += NT; q += step_q; n += step_n; if (q >= L) { q -= L; ++n; } };
        if (vec) {
            auto acc4 = [&](const float4& v) {
                const float a = v.x - k, b = v.y - k, cc = v.z - k, d = v.w - k;
                s1 += (a + b) + (cc + d);
                s2 += (a * a + b * b) + (cc * cc + d * d);
            };
            while (u + 3 * NT < total) {
                const float4* p0 = reinterpret_cast<const float4*>(z + ((size_t)n * C + c) * (size_t)HW) + q;  advance();
                const float4* p1 = reinterpret_cast<const float4*>(z + ((size_t)n * C + c) * (size_t)HW) + q;  advance();
                const float4* p2 = reinterpret_cast<const float4*>(z + ((size_t)n * C + c) * (size_t)HW) + q;  advance();
                const float4* p3 = reinterpret_cast<const float4*>(z + ((size_t)n * C + c) * (size_t)HW) + q;  advance();
                const float4 v0 = *p0, v1 = *p1, v2 = *p2, v3 = *p3;
                acc4(v0); acc4(v1); acc4(v2); acc4(v3);
            }
            while (u < total) {
                const float4 v = reinterpret_cast<const float4*>(z + ((size_t)n * C + c) * (size_t)HW)[q];
                advance();
                acc4(v);
            }
        } else {
            while (u < total) {
                const float a = z[((size_t)n * C + c) * (size_t)HW + q] - k;
                advance();
                s1 += a;
                s2 += a * a;
            }
        }
    }
    const double t1 = bn_block_sum((double)s1, part), t2 = bn_block_sum((double)s2, part);
    if (tid == 0) {
        const double M = (double)N * (double)HW;
        const double e1 = t1 / M, e2 = t2 / M;
        const double mu = (double)k + e1;
        double var = e2 - e1 * e1;
        if (var < 0.0) var = 0.0;
        const float mf = (float)mu, isf = (float)(1.0 / sqrt(var + (double)eps));
        mean[c] = mf;
        invstd[c] = isf;
        const float sc = gamma[c] * isf, sh = beta[c] + (-mf) * sc;
        scale[c] = sc;
        shift[c] = sh;
        fold[0] = sc;  fold[1] = sh;
        if (running_mean) {
            const double unbiased = M > 1.0 ? var * (M / (M - 1.0)) : var;
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
        }
        if (nbt && c == 0) *nbt += 1;
    }
    __syncthreads();
    const float sc = fold[0], sh = fold[1];
    const bool act = alpha != nullptr;
    const float al = act ? alpha[c] : 1.f;
    auto one = [&](float zv, float rv) {
        float v = fmaf(zv, sc, sh);
        v += rv;
        return (!act || v > 0.f) ? v : al * v;
    };
    int u = tid, n = u / L, q = u - n * L;
    auto advance = [&]() { u += NT; q += step_q; n += step_n; if (q >= L) { q -= L; ++n; } };
    if (vec) {
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        while (u < total) {
            const size_t oa = ((size_t)n * C + c) * (size_t)HW + 4 * (size_t)q;
            advance();
            const bool hb = u < total;
            const size_t ob = hb ? ((size_t)n * C + c) * (size_t)HW + 4 * (size_t)q : oa;
            advance();
            const float4 za = *reinterpret_cast<const float4*>(z + oa), zb = *reinterpret_cast<const float4*>(z + ob);
            const float4 ra = res ? *reinterpret_cast<const float4*>(res + oa) : zero, rb = res ? *reinterpret_cast<const float4*>(res + ob) : zero;
            *reinterpret_cast<float4*>(y + oa) = make_float4(one(za.x, ra.x), one(za.y, ra.y), one(za.z, ra.z), one(za.w, ra.w));
            if (hb) *reinterpret_cast<float4*>(y + ob) = make_float4(one(zb.x, rb.x), one(zb.y, rb.y), one(zb.z, rb.z), one(zb.w, rb.w));
        }
    } else {
        while (u < total) {
            const size_t o = ((size_t)n * C + c) * (size_t)HW + q;
            advance();
            y[o] = one(z[o], res ? res[o] : 0.f);
        }
    }
}

__global__ __launch_bounds__(1024) void bn_train_small_bwd_kernel(const float* __restrict__ z, const float* __restrict__ res,
                                                                 const float* __restrict__ gy, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, const float* __restrict__ alpha,
                                                                 const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, int N, int C, int HW, float inv_m,
                                                                 int accumulate, float* __restrict__ gz, float* __restrict__ gres,
                                                                 float* __restrict__ ggamma, float* __restrict__ gbeta,
                                                                 float* __restrict__ galpha) {
    __shared__ float part[3][16];
    __shared__ float coef[2];
    const int c = blockIdx.x, tid = threadIdx.x;
    const bool vec = (HW & 3) == 0;
    const int L = vec ? (HW >> 2) : HW, total = N * L;
    const int NT = blockDim.x;                  // 1024, or 256 for the smallest channels
    const int step_n = NT / L, step_q = NT - step_n * L;
    const float sc = scale[c], sh = shift[c];
    const bool act = alpha != nullptr;
    const float al = act ? alpha[c] : 1.f;
    float s_scale = 0.f, s_shift = 0.f, s_alpha = 0.f;
    auto direct = [&](float zv, float rv, float g) {           // gradient of the pre-activation, as affine_prelu_bwd_kernel
        const float u = fmaf(zv, sc, sh) + rv;          // the forward's expression (one fused multiply-add, then the residual): same PReLU branch
        return (!act || u > 0.f) ? g : al * g;
    };
    auto sums = [&](float zv, float rv, float g) {
        const float u = fmaf(zv, sc, sh) + rv;          // the forward's expression (one fused multiply-add, then the residual): same PReLU branch
        const float gzv = (!act || u > 0.f) ? g : al * g;
        if (act && u <= 0.f) s_alpha += g * u;
        s_scale += gzv * zv;
        s_shift += gzv;
    };
    {
        int u = tid, n = u / L, q = u - n * L;
        auto advance = [&]() { u += NT; q += step_q; n += step_n; if (q >= L) { q -= L; ++n; } };
        if (vec) {
            const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
            while (u < total) {
                const size_t oa = ((size_t)n * C + c) * (size_t)HW + 4 * (size_t)q;
                advance();
                const bool hb = u < total;
                const size_t ob = hb ? ((size_t)n * C + c) * (size_t)HW + 4 * (size_t)q : oa;
                advance();
                const float4 za = *reinterpret_cast<const float4*>(z + oa), ga = *reinterpret_cast<const float4*>(gy + oa);
                const float4 zb = *reinterpret_cast<const float4*>(z + ob), gb = *reinterpret_cast<const float4*>(gy + ob);
                const float4 ra = res ? *reinterpret_cast<const float4*>(res + oa) : zero, rb = res ? *reinterpret_cast<const float4*>(res + ob) : zero;
                sums(za.x, ra.x, ga.x); sums(za.y, ra.y, ga.y); sums(za.z, ra.z, ga.z); sums(za.w, ra.w, ga.w);
                if (hb) { sums(zb.x, rb.x, gb.x); sums(zb.y, rb.y, gb.y); sums(zb.z, rb.z, gb.z); sums(zb.w, rb.w, gb.w); }
            }
        } else {
            while (u < total) {
                const size_t o = ((size_t)n * C + c) * (size_t)HW + q;
                advance();
                sums(z[o], res ? res[o] : 0.f, gy[o]);
            }
        }
    }
    float v[3] = {s_scale, s_shift, s_alpha};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        v[i] = wave_sum_dpp(v[i]);                            // total in lane 63
        if ((tid & 63) == 63) part[i][tid >> 6] = v[i];
    }
    __syncthreads();
    if (tid == 0) {
        float dsc = 0.f, dsh = 0.f, dal = 0.f;
        for (int w = 0; w < (NT >> 6); ++w) { dsc += part[0][w];  dsh += part[1][w];  dal += part[2][w]; }
        const float is = invstd[c], mu = mean[c];
        const float t = dsc + (-mu) * dsh;
        ggamma[c] = accumulate ? ggamma[c] + t * is : t * is;
        gbeta[c] = accumulate ? gbeta[c] + dsh : dsh;
        if (galpha && act) galpha[c] += dal;
        const float p = ((gamma[c] * t) * (is * is * is)) * (-inv_m);
        coef[0] = p;
        coef[1] = (dsh * sc) * (-inv_m) - p * mu;
    }
    __syncthreads();
    const float pc = coef[0], qc = coef[1];
    int u = tid, n = u / L, q = u - n * L;
    auto advance = [&]() { u += NT; q += step_q; n += step_n; if (q >= L) { q -= L; ++n; } };
    if (vec) {
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        while (u < total) {
            const size_t oa = ((size_t)n * C + c) * (size_t)HW + 4 * (size_t)q;
            advance();
            const float4 za = *reinterpret_cast<const float4*>(z + oa), ga = *reinterpret_cast<const float4*>(gy + oa);
            const float4 ra = res ? *reinterpret_cast<const float4*>(res + oa) : zero;
            const float4 d = make_float4(direct(za.x, ra.x, ga.x), direct(za.y, ra.y, ga.y), direct(za.z, ra.z, ga.z), direct(za.w, ra.w, ga.w));
            if (gres) *reinterpret_cast<float4*>(gres + oa) = d;
            *reinterpret_cast<float4*>(gz + oa) = make_float4(fmaf(za.x, pc, qc) + d.x * sc, fmaf(za.y, pc, qc) + d.y * sc,
                                                              fmaf(za.z, pc, qc) + d.z * sc, fmaf(za.w, pc, qc) + d.w * sc);
        }
    } else {
        while (u < total) {
            const size_t o = ((size_t)n * C + c) * (size_t)HW + q;
            advance();
            const float d = direct(z[o], res ? res[o] : 0.f, gy[o]);
            if (gres) gres[o] = d;
            gz[o] = fmaf(z[o], pc, qc) + d * sc;
        }
    }
}

// Second pass of the batch-statistics BatchNorm + PReLU backward without a residual: gz = p * z + q + gc with the direct gradient
// gc = (u > 0 ? gy : alpha * gy) * scale, u = z * scale + shift, RECOMPUTED from (z, gy) instead of read back -- the first pass
// (mspl_bn_train_prelu_bwd with gc = NULL) then only reads: five tensor passes per BatchNorm instead of six.  Same operations in the
// same order as the two-kernel form (affine_prelu_bwd_kernel's gc, then mspl_pointwise_fwd's p * z + q + gc).
template <bool VEC>
__global__ __launch_bounds__(256) void bn_train_bwd_apply_kernel(const float* __restrict__ z, const float* __restrict__ gy,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 const float* __restrict__ alpha, const float* __restrict__ p,
                                                                 const float* __restrict__ q, int C, int L, float* __restrict__ gz) {
    const long long plane = (long long)blockIdx.z * gridDim.y + blockIdx.y;      // n * C + c
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= L) return;
    const int c = (int)(plane % C);
    const float sc = scale[c], sh = shift[c], pc = p[c], qc = q[c];
    const bool act = alpha != nullptr;
    const float al = act ? alpha[c] : 1.f;
    auto one = [&](float zv, float g) {
        const float u = fmaf(zv, sc, sh);               // the forward's expression
        const float gzv = (!act || u > 0.f) ? g : al * g;
        return fmaf(zv, pc, qc) + gzv * sc;
    };
    if (VEC) {
        const size_t o = (size_t)plane * L + t;
        const float4 zv = reinterpret_cast<const float4*>(z)[o], g = reinterpret_cast<const float4*>(gy)[o];
        reinterpret_cast<float4*>(gz)[o] = make_float4(one(zv.x, g.x), one(zv.y, g.y), one(zv.z, g.z), one(zv.w, g.w));
    } else {
        const size_t o = (size_t)plane * L + t;
        gz[o] = one(z[o], gy[o]);
    }
}

// merge_layer.0 .. merge_layer.2's convolution from the KEPT branch values (the batch-statistics pyramid node: the branch values are
// computed once, their statistics give merge_layer.0's fold, and this kernel finishes the forward without evaluating the branches a
// second time):  out[n][c] = sum_i conv3x3( PReLU(zcat[n][i*P + c] * scale[i*P + c] + shift[i*P + c]), merge_w[c][i] ), zero padding
// applied AFTER the activation (efficient_pyramid_pool.py:51-58: BR, Shuffle(groups = nb), grouped 3x3 with groups = P).
// A workgroup owns a TH x TW tile of one (image, channel) plane; all nb halo tiles are fetched into registers up front, then pass
// through a double-buffered LDS tile (one barrier per branch).
template <int TW, int TH>
__global__ __launch_bounds__(256) void pyr_merge_fwd_kernel(const float* __restrict__ zcat, const float* __restrict__ sc,
                                                            const float* __restrict__ sh, const float* __restrict__ al,
                                                            const float* __restrict__ mw, int P, int nb, int h, int w, int tiles_x,
                                                            int tiles_y, float* __restrict__ out) {
    static_assert(TW * TH == 1024 && TW % 4 == 0, "256 threads x 4 pixels");
    constexpr int LW = TW + 4, BH = TH + 2, BW = TW + 2, NE = BH * BW, PER = (NE + 255) / 256, MAXB = 5;
    __shared__ __attribute__((aligned(16))) float tile[2][BH * LW + 4];
    int bid = blockIdx.x;
    const int txi = bid % tiles_x;  bid /= tiles_x;
    const int tyi = bid % tiles_y;  bid /= tiles_y;
    const int c = bid % P, n = bid / P;
    const int y0 = tyi * TH, x0 = txi * TW;
    const int tid = threadIdx.x;
    const size_t hw = (size_t)h * w;
    int goff[PER], loff[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int e = tid + 256 * k, r = e / BW, q = e - r * BW;
        const int py = y0 - 1 + r, px = x0 - 1 + q;
        loff[k] = e < NE ? r * LW + q : -1;
        goff[k] = (e < NE && py >= 0 && py < h && px >= 0 && px < w) ? py * w + px : -1;
    }
    float v[MAXB][PER];
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        if (i >= nb) break;
        const float* pl = zcat + ((size_t)n * nb * P + (size_t)i * P + c) * hw;
#pragma unroll
        for (int k = 0; k < PER; ++k) v[i][k] = goff[k] >= 0 ? pl[goff[k]] : 0.f;
    }
    const int ty = tid / (TW / 4), xs = tid - ty * (TW / 4);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        if (i >= nb) break;
        const int ch = i * P + c;
        const float s0 = sc[ch], s1 = sh[ch], a0 = al[ch];
        float* buf = tile[i & 1];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            float b = fmaf(v[i][k], s0, s1);
            b = b > 0.f ? b : a0 * b;
            if (loff[k] >= 0) buf[loff[k]] = goff[k] >= 0 ? b : 0.f;
        }
        __syncthreads();
        const float* wm = mw + ((size_t)c * nb + i) * 9;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const float* row = buf + (ty + ky) * LW + xs * 4;
            const float4 a = *reinterpret_cast<const float4*>(row);
            const float2 b2 = *reinterpret_cast<const float2*>(row + 4);
            const float rv[6] = {a.x, a.y, a.z, a.w, b2.x, b2.y};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = fmaf(wm[ky * 3 + 0], rv[j], acc[j]);
                acc[j] = fmaf(wm[ky * 3 + 1], rv[j + 1], acc[j]);
                acc[j] = fmaf(wm[ky * 3 + 2], rv[j + 2], acc[j]);
            }
        }
    }
    const int y = y0 + ty, xb = x0 + xs * 4;
    if (y >= h || xb >= w) return;
    float* dst = out + ((size_t)n * P + c) * hw + (size_t)y * w + xb;
    if ((w & 3) == 0) {
        *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (xb + j < w) dst[j] = acc[j];
    }
}

}  // namespace mspl

using namespace mspl;

static int bn_batch_stats_impl(const float* z, int32_t N, int32_t C, int32_t HW, float eps, float momentum, float* running_mean,
                               float* running_var, const float* gamma, const float* beta, double* ws, float* mean, float* invstd,
                               float* scale, float* shift, void* stream);

extern "C" int mspl_bn_batch_stats_fwd(const float* z, int32_t N, int32_t C, int32_t HW, float eps, float momentum,
                                       float* running_mean, float* running_var, double* ws, float* mean, float* invstd,
                                       void* stream) {
    return bn_batch_stats_impl(z, N, C, HW, eps, momentum, running_mean, running_var, nullptr, nullptr, ws, mean, invstd, nullptr,
                               nullptr, stream);
}

extern "C" int mspl_bn_batch_stats_fold_fwd(const float* z, int32_t N, int32_t C, int32_t HW, float eps, float momentum,
                                            float* running_mean, float* running_var, const float* gamma, const float* beta,
                                            double* ws, float* mean, float* invstd, float* scale, float* shift, void* stream) {
    MSPL_REQUIRE(gamma && beta && scale && shift, MSPL_ERR_NULL_POINTER, "bn_batch_stats_fold: null pointer");
    return bn_batch_stats_impl(z, N, C, HW, eps, momentum, running_mean, running_var, gamma, beta, ws, mean, invstd, scale, shift, stream);
}

extern "C" int mspl_bn_train_small_fits(int32_t N, int32_t C, int32_t HW) {
    return N > 0 && C > 0 && HW > 0 && (int64_t)N * HW <= 40960 && C <= 65535;
}

extern "C" int mspl_bn_train_small_fwd(const float* z, const float* residual, const float* gamma, const float* beta, const float* alpha,
                                       int32_t N, int32_t C, int32_t HW, float eps, float momentum, float* running_mean,
                                       float* running_var, int64_t* num_batches_tracked, float* mean, float* invstd, float* scale,
                                       float* shift, float* y, void* stream) {
    MSPL_REQUIRE(z && gamma && beta && mean && invstd && scale && shift && y, MSPL_ERR_NULL_POINTER, "bn_train_small_fwd: null pointer");
    MSPL_REQUIRE((running_mean == nullptr) == (running_var == nullptr), MSPL_ERR_NULL_POINTER,
                 "bn_train_small_fwd: running_mean and running_var go together");
    MSPL_REQUIRE(mspl_bn_train_small_fits(N, C, HW), MSPL_ERR_UNSUPPORTED, "bn_train_small_fwd: N=%d C=%d HW=%d is not a small-plane shape", N, C, HW);
    MSPL_REQUIRE((HW & 3) != 0 || ((((uintptr_t)z) | ((uintptr_t)residual) | ((uintptr_t)y)) & 15) == 0, MSPL_ERR_BAD_SHAPE,
                 "bn_train_small_fwd: operands must be 16-byte aligned");
    // 1024 threads from MSPL_BN_SMALL_WIDE units (quads, or floats when HW % 4 != 0) per channel on, 256 below
    static const int wide_f = MSPL_TUNE_INT("MSPL_BN_SMALL_WIDE", 2048);
    const int nt_f = (int64_t)N * ((HW & 3) == 0 ? HW / 4 : HW) >= wide_f ? 1024 : 256;
    hipLaunchKernelGGL(bn_train_small_fwd_kernel, dim3((unsigned)C), dim3(nt_f), 0, (hipStream_t)stream, z, residual, gamma, beta, alpha, N, C,
                       HW, eps, momentum, running_mean, running_var, reinterpret_cast<long long*>(num_batches_tracked), mean, invstd, scale,
                       shift, y);
    MSPL_CHECK_LAUNCH("bn_train_small_fwd");
    return MSPL_OK;
}

extern "C" int mspl_bn_train_small_bwd(const float* z, const float* residual, const float* gy, const float* scale, const float* shift,
                                       const float* alpha, const float* gamma, const float* mean, const float* invstd, int32_t N, int32_t C,
                                       int32_t HW, int32_t accumulate, float* gz, float* gres, float* ggamma, float* gbeta, float* galpha,
                                       void* stream) {
    MSPL_REQUIRE(z && gy && scale && shift && gamma && mean && invstd && gz && ggamma && gbeta, MSPL_ERR_NULL_POINTER,
                 "bn_train_small_bwd: null pointer");
    MSPL_REQUIRE((gres == nullptr) || residual, MSPL_ERR_NULL_POINTER, "bn_train_small_bwd: gres without residual");
    MSPL_REQUIRE(mspl_bn_train_small_fits(N, C, HW), MSPL_ERR_UNSUPPORTED, "bn_train_small_bwd: N=%d C=%d HW=%d is not a small-plane shape", N, C, HW);
    MSPL_REQUIRE((HW & 3) != 0 || ((((uintptr_t)z) | ((uintptr_t)residual) | ((uintptr_t)gy) | ((uintptr_t)gz) | ((uintptr_t)gres)) & 15) == 0,
                 MSPL_ERR_BAD_SHAPE, "bn_train_small_bwd: operands must be 16-byte aligned");
    static const int wide_b = MSPL_TUNE_INT("MSPL_BN_SMALL_WIDE", 2048);
    const int nt_b = (int64_t)N * ((HW & 3) == 0 ? HW / 4 : HW) >= wide_b ? 1024 : 256;
    hipLaunchKernelGGL(bn_train_small_bwd_kernel, dim3((unsigned)C), dim3(nt_b), 0, (hipStream_t)stream, z, residual, gy, scale, shift, alpha,
                       gamma, mean, invstd, N, C, HW, (float)(1.0 / ((double)N * (double)HW)), accumulate, gz, gres, ggamma, gbeta, galpha);
    MSPL_CHECK_LAUNCH("bn_train_small_bwd");
    return MSPL_OK;
}

extern "C" int mspl_bn_train_prelu_bwd_apply(const float* z, const float* gy, const float* scale, const float* shift, const float* alpha,
                                            const float* p, const float* q, int32_t N, int32_t C, int32_t HW, float* gz, void* stream) {
    MSPL_REQUIRE(z && gy && scale && shift && p && q && gz, MSPL_ERR_NULL_POINTER, "bn_train_prelu_bwd_apply: null pointer");
    MSPL_REQUIRE(N > 0 && C > 0 && HW > 0, MSPL_ERR_BAD_SHAPE, "bn_train_prelu_bwd_apply: bad shape N=%d C=%d HW=%d", N, C, HW);
    const bool vec = (HW & 3) == 0 && ((((uintptr_t)z) | ((uintptr_t)gy) | ((uintptr_t)gz)) & 15) == 0;
    const int L = vec ? HW / 4 : HW;
    const int64_t planes = (int64_t)N * C;
    MSPL_REQUIRE(planes <= 65535ll * 65535ll, MSPL_ERR_BAD_SHAPE, "bn_train_prelu_bwd_apply: too many planes");
    const int gy_ = planes < 65535 ? (int)planes : 65535;
    const dim3 grid((unsigned)ceil_div(L, 256), (unsigned)gy_, (unsigned)ceil_div64(planes, gy_));
    if (vec) hipLaunchKernelGGL(bn_train_bwd_apply_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, z, gy, scale, shift, alpha, p, q, C, L, gz);
    else hipLaunchKernelGGL(bn_train_bwd_apply_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, z, gy, scale, shift, alpha, p, q, C, L, gz);
    MSPL_CHECK_LAUNCH("bn_train_prelu_bwd_apply");
    return MSPL_OK;
}

extern "C" int mspl_pyrpool_merge_fwd(const float* zcat, int32_t N, int32_t P, int32_t h, int32_t w, int32_t nb, const float* br_scale,
                                      const float* br_shift, const float* br_alpha, const float* merge_w, float* out, void* stream) {
    MSPL_REQUIRE(zcat && br_scale && br_shift && br_alpha && merge_w && out, MSPL_ERR_NULL_POINTER, "pyrpool_merge_fwd: null pointer");
    MSPL_REQUIRE(N > 0 && P > 0 && h > 0 && w > 0 && (int64_t)h * w < (1ll << 31), MSPL_ERR_BAD_SHAPE,
                 "pyrpool_merge_fwd: bad shape N=%d P=%d %dx%d", N, P, h, w);
    MSPL_REQUIRE(nb >= 1 && nb <= 5, MSPL_ERR_UNSUPPORTED, "pyrpool_merge_fwd: %d branches (1..5)", nb);
    MSPL_REQUIRE((w & 3) != 0 || (((uintptr_t)out) & 15) == 0, MSPL_ERR_BAD_SHAPE, "pyrpool_merge_fwd: out must be 16-byte aligned");
    const bool narrow = w <= 32;
    const int TW = narrow ? 32 : 64, TH = narrow ? 32 : 16;
    const int tx = ceil_div(w, TW), ty = ceil_div(h, TH);
    const int64_t blocks = (int64_t)N * P * tx * ty;
    MSPL_REQUIRE(blocks < (1ll << 31), MSPL_ERR_BAD_SHAPE, "pyrpool_merge_fwd: grid too large");
    if (narrow)
        hipLaunchKernelGGL((pyr_merge_fwd_kernel<32, 32>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, zcat, br_scale, br_shift,
                           br_alpha, merge_w, P, nb, h, w, tx, ty, out);
    else
        hipLaunchKernelGGL((pyr_merge_fwd_kernel<64, 16>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, zcat, br_scale, br_shift,
                           br_alpha, merge_w, P, nb, h, w, tx, ty, out);
    MSPL_CHECK_LAUNCH("pyrpool_merge_fwd");
    return MSPL_OK;
}

extern "C" int mspl_bn_stats_path_add(float* g, const float* z, const float* p, const float* q, int32_t N, int32_t P, int32_t nb,
                                      int32_t HW, void* stream) {
    MSPL_REQUIRE(g && z && p && q, MSPL_ERR_NULL_POINTER, "bn_stats_path_add: null pointer");
    MSPL_REQUIRE(N > 0 && P > 0 && nb > 0 && HW > 0 && (HW & 3) == 0, MSPL_ERR_BAD_SHAPE, "bn_stats_path_add: bad shape N=%d P=%d nb=%d HW=%d",
                 N, P, nb, HW);
    MSPL_REQUIRE((((uintptr_t)g | (uintptr_t)z) & 15) == 0, MSPL_ERR_BAD_SHAPE, "bn_stats_path_add: operands must be 16-byte aligned");
    const int64_t planes = (int64_t)nb * N * P;
    MSPL_REQUIRE(planes <= 65535ll * 65535ll, MSPL_ERR_BAD_SHAPE, "bn_stats_path_add: too many planes");
    const int gy = planes < 65535 ? (int)planes : 65535;
    const dim3 grid((unsigned)ceil_div(HW / 4, 256), (unsigned)gy, (unsigned)ceil_div64(planes, gy));
    hipLaunchKernelGGL(bn_stats_path_add_kernel, grid, dim3(256), 0, (hipStream_t)stream, g, z, p, q, N, P, nb, HW / 4);
    MSPL_CHECK_LAUNCH("bn_stats_path_add");
    return MSPL_OK;
}

extern "C" int64_t mspl_bn_fused_workspace_bytes(int32_t C) {
    // forward: 2C doubles + C counters; backward (mspl_bn_train_prelu_bwd): 2C floats + C counters, placed behind it
    return (int64_t)C * (16 + 4 + 8 + 4) + 64;
}

extern "C" int mspl_bn_batch_stats_fused_fwd(const float* z, int32_t N, int32_t C, int32_t HW, float eps, float momentum,
                                             float* running_mean, float* running_var, const float* gamma, const float* beta,
                                             void* ws_zeroed, float* mean, float* invstd, float* scale, float* shift,
                                             int64_t* num_batches_tracked, void* stream) {
    MSPL_REQUIRE(z && ws_zeroed && mean && invstd && gamma && beta && scale && shift, MSPL_ERR_NULL_POINTER, "bn_batch_stats_fused: null pointer");
    MSPL_REQUIRE((running_mean == nullptr) == (running_var == nullptr), MSPL_ERR_NULL_POINTER,
                 "bn_batch_stats_fused: running_mean and running_var go together");
    MSPL_REQUIRE(N > 0 && C > 0 && HW > 0 && N <= 65535 && C <= 65535, MSPL_ERR_BAD_SHAPE, "bn_batch_stats_fused: bad shape N=%d C=%d HW=%d",
                 N, C, HW);
    MSPL_REQUIRE(((uintptr_t)ws_zeroed & 7) == 0, MSPL_ERR_BAD_SHAPE, "bn_batch_stats_fused: workspace must be 8-byte aligned");
    static const int target = MSPL_TUNE_INT("MSPL_BN_BLOCKS", 768);
    const int64_t units = (int64_t)N * ((HW & 3) == 0 ? HW / 4 : HW);
    int64_t parts = ceil_div64(target, C);
    if (parts > units / 512) parts = units / 512;
    if (parts < 1) parts = 1;
    if (parts > 65535) parts = 65535;
    double* sums = static_cast<double*>(ws_zeroed);
    unsigned* cnt = reinterpret_cast<unsigned*>(sums + 2 * (size_t)C);
    hipLaunchKernelGGL(bn_stats_fused_kernel, dim3((unsigned)parts, (unsigned)C), dim3(256), 0, (hipStream_t)stream, z, N, C, HW,
                       (unsigned)parts, sums, cnt, (double)N * (double)HW, eps, momentum, running_mean, running_var, mean,
                       invstd, gamma, beta, scale, shift, reinterpret_cast<long long*>(num_batches_tracked));
    MSPL_CHECK_LAUNCH("bn_batch_stats_fused");
    return MSPL_OK;
}

extern "C" int mspl_bn_batch_stats_bwd_coeffs(const float* gscale, const float* gshift, const float* gamma, const float* mean,
                                              const float* invstd, const float* scale, int32_t C, double M, int32_t accumulate,
                                              float* ggamma, float* gbeta, float* p, float* q, void* stream) {
    MSPL_REQUIRE(gscale && gshift && gamma && mean && invstd && scale && ggamma && p && q, MSPL_ERR_NULL_POINTER,
                 "bn_batch_stats_bwd_coeffs: null pointer");
    MSPL_REQUIRE(C > 0 && M > 0, MSPL_ERR_BAD_SHAPE, "bn_batch_stats_bwd_coeffs: C=%d M=%g", C, M);
    hipLaunchKernelGGL(bn_stats_bwd_coeffs_kernel, dim3((unsigned)ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, gscale, gshift,
                       gamma, mean, invstd, scale, C, (float)(1.0 / M), accumulate, ggamma, gbeta, p, q);
    MSPL_CHECK_LAUNCH("bn_batch_stats_bwd_coeffs");
    return MSPL_OK;
}

static int bn_batch_stats_impl(const float* z, int32_t N, int32_t C, int32_t HW, float eps, float momentum, float* running_mean,
                               float* running_var, const float* gamma, const float* beta, double* ws, float* mean, float* invstd,
                               float* scale, float* shift, void* stream) {
    MSPL_REQUIRE(z && ws && mean && invstd, MSPL_ERR_NULL_POINTER, "bn_batch_stats: null pointer");
    MSPL_REQUIRE((running_mean == nullptr) == (running_var == nullptr), MSPL_ERR_NULL_POINTER,
                 "bn_batch_stats: running_mean and running_var go together");
    MSPL_REQUIRE(N > 0 && C > 0 && HW > 0 && N <= 65535 && C <= 65535, MSPL_ERR_BAD_SHAPE, "bn_batch_stats: bad shape N=%d C=%d HW=%d",
                 N, C, HW);
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(ws, 0, sizeof(double) * 2 * (size_t)C, s) != hipSuccess) {
        MSPL_REQUIRE(false, MSPL_ERR_HIP, "bn_batch_stats: hipMemsetAsync failed");
    }
    // slices of >= 4096 elements, and enough workgroups (~2048) to fill the chip when N*C is small
    int chunks = ceil_div(2048, N * C);
    if (chunks < 1) chunks = 1;
    int per_block = ceil_div(HW, chunks);
    if (per_block < 4096) per_block = 4096;
    per_block = (per_block + 3) & ~3;
    chunks = ceil_div(HW, per_block);
    hipLaunchKernelGGL(bn_stats_kernel, dim3((unsigned)chunks, (unsigned)N, (unsigned)C), dim3(256), 0, s, z, C, HW, per_block, ws);
    MSPL_CHECK_LAUNCH("bn_batch_stats(sum)");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)ceil_div(C, 256)), dim3(256), 0, s, z, ws, C, HW, (double)N * (double)HW, eps,
                       momentum, running_mean, running_var, mean, invstd, gamma, beta, scale, shift);
    MSPL_CHECK_LAUNCH("bn_batch_stats(finalize)");
    return MSPL_OK;
}

extern "C" int mspl_sgd_step(float* p, const float* g, float* buf, int64_t n, float lr, float momentum, float weight_decay,
                             int32_t first_step, void* stream) {
    MSPL_REQUIRE(p && g && (buf || momentum == 0.f), MSPL_ERR_NULL_POINTER, "sgd_step: null pointer");
    MSPL_REQUIRE(n >= 0, MSPL_ERR_BAD_SHAPE, "sgd_step: n=%lld", (long long)n);
    if (n == 0) return MSPL_OK;
    hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, buf, n, lr, momentum,
                       weight_decay, first_step);
    MSPL_CHECK_LAUNCH("sgd_step");
    return MSPL_OK;
}
