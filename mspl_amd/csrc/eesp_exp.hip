// K2 + K3 in one launch for the stride-1 EESP blocks (levels 3 / 4 of the encoder).
//
// Reference arithmetic: nn_layers/eesp.py:68-93 -- four CDilated depthwise 3x3 convolutions of the reduced tensor with the
// hierarchical sum out_k = conv_k + out_{k-1}, torch.cat, br_after_cat (BatchNorm + PReLU), conv_1x1_exp (grouped 1x1, 4
// groups: group g reads exactly branch g of all n reduced channels) + BatchNorm, + input (residual link), module_act (PReLU).
//
// Why fused: the concatenation (4n channels) is written by K2 and read back by K3 and both launches are single rounds of
// workgroups bounded by latency (K2 12-13 us, K3 20-28 us for 7 us of matrix work at 18x30 x 512 channels x 16 images).  Here
// the branch values never leave the CU: a workgroup owns a band of TH rows of one image (TH * W <= 64 pixels) and GPW of the four
// groups, walks the reduced channels in chunks of KC and per chunk
//   (L) moves the chunk's rows (+- MAXD halo rows, zero outside the image) global -> registers -> zero-haloed LDS rows, two chunks
//       ahead of the matrix stage,
//   (D) computes the depthwise branches for the NEXT chunk on the vector unit: a wave takes one channel at a time, lane = pixel,
//       taps are ds_read_b32 with immediate offsets, the 36 weights + 12 BN/PReLU constants of the channel sit in scalar
//       registers (one packed 192-byte record per channel, s_load), same operation order as eesp_dw.hip (bit-identical values),
//       and writes them as the B operand of the matrix stage, [group][k][pixel] in LDS,
//   (M) multiplies the CURRENT chunk on the matrix cores: wave = (group, 32 output rows) x 64 pixels, v_mfma_f32_32x32x2_f32 with
//       k ascending exactly like conv1x1_pipe_kernel (bit-identical sums); the A operand (expansion weights) streams from a
//       pre-packed copy in global memory, one chunk ahead, two 16-byte loads per lane and chunk (no weight staging, no barrier
//       for it),
// with ONE barrier per chunk; the vector work of (D) runs in the shadow of the other waves' MFMAs.  Epilogue as in K3: folded BN,
// + residual, PReLU, 8-byte stores.
// Algorithmic bytes: read r (n HW) + residual (4n HW), write 4n HW: 4 * 9n * HW per image (the unfused pair: 4 * 17n * HW).
#include <stdlib.h>

#include <mutex>

#include "common.hpp"

namespace mspl {

typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int D0, int D1, int D2, int D3>
struct XDil {
    static constexpr int d(int k) { return k == 0 ? D0 : k == 1 ? D1 : k == 2 ? D2 : D3; }
    static constexpr int maxd() { return D3 > D2 ? (D3 > D1 ? (D3 > D0 ? D3 : D0) : (D1 > D0 ? D1 : D0))
                                                 : (D2 > D1 ? (D2 > D0 ? D2 : D0) : (D1 > D0 ? D1 : D0)); }
};

__device__ __forceinline__ int xe_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

constexpr int XE_REC = 48;      // floats per packed channel record: per branch k: w[9], scale, shift, alpha

// ------------------------------------------------------------------ packing (once per weight version; cached by the caller)
// dwp[c][k][12]: branch k's 3x3 weights of reduced channel c, then br_after_cat's folded scale / shift / PReLU slope of the
// concatenated channel k*n + c.  ap: the expansion weights in the lane order of the A operand, chunked by KC:
// ap[(((g*RTPG + rt)*NCHUNK + ch)*64 + lane)*(KC/2) + i] = w[g*n + rt*32 + (lane & 31)][ch*KC + 2*i + (lane >> 5)].
__global__ void eesp_exp_pack_kernel(const float* __restrict__ w4, const float* __restrict__ bscale, const float* __restrict__ bshift,
                                     const float* __restrict__ balpha, const float* __restrict__ wexp, int n, int KC,
                                     float* __restrict__ dwp, float* __restrict__ ap) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n * XE_REC) {
        const int c = i / XE_REC, f = i - c * XE_REC;
        const int k = f / 12, e = f - k * 12;
        float v;
        if (e < 9) v = w4[((size_t)k * n + c) * 9 + e];
        else if (e == 9) v = bscale[k * n + c];
        else if (e == 10) v = bshift[k * n + c];
        else v = balpha[k * n + c];
        dwp[i] = v;
    }
    const int total = 4 * n * n;
    if (i < total) {
        const int half_kc = KC / 2, nchunk = n / KC, rtpg = n / 32;
        int t = i;
        const int ii = t % half_kc;  t /= half_kc;
        const int lane = t % 64;     t /= 64;
        const int ch = t % nchunk;   t /= nchunk;
        const int rt = t % rtpg;
        const int g = t / rtpg;
        const int row = g * n + rt * 32 + (lane & 31);
        const int k = ch * KC + 2 * ii + (lane >> 5);
        ap[i] = wexp[(size_t)row * n + k];
    }
}

// The NEXT block's proj_1x1 (grouped 1x1, n outputs, n inputs per group = exactly one group's 4n/4 output rows of this block) as a
// second matrix stage on the accumulators (see the kernel's epilogue): its weights in the lane order of the A operand of k-step j,
// np[((g*RTPG + rt)*64 + lane)*16 + j] = w1[g*M1 + (lane & 31)][rt*32 + (j & 3) + 8*(j >> 2) + 4*(lane >> 5)], rows >= M1 = n/4 zero.
__global__ void eesp_exp_pack_next_kernel(const float* __restrict__ w1, int n, float* __restrict__ np) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int rtpg = n / 32, m1 = n / 4;
    if (i >= 4 * rtpg * 16 * 64) return;
    const int j = i & 15, lane = (i >> 4) & 63, wv = i >> 10;
    const int rt = wv % rtpg, g = wv / rtpg;
    const int m = lane & 31;
    const int k = rt * 32 + (j & 3) + 8 * (j >> 2) + 4 * (lane >> 5);
    np[i] = m < m1 ? w1[(size_t)(g * m1 + m) * n + k] : 0.f;
}

// ------------------------------------------------------------------ the fused kernel
// Measured on the way (in-kernel s_memrealtime timelines, tools/xe_stamp.sh): a split into four matrix waves + four depthwise
// waves per workgroup (one of each per SIMD) does NOT overlap the two kinds of work -- the depthwise waves' step got 1.2 us
// longer exactly while the matrix waves' 1.1 us of v_mfma_f32_32x32x2_f32 ran: the fp32 MFMA runs at the vector rate and holds
// the SIMD's vector issue.  So every wave does both, and a step costs (vector work) + (matrix work) per SIMD.
// W is the width of a BAND; WI the width of the image.  WI == W: a band is TH whole rows (every shape of rounds 4).  WI == 2 W (TH = 1):
// a band is half a row -- 1024-pixel-wide inputs have 128 columns at level 3, twice what the 64-pixel matrix tile holds; the staged
// rows then carry REAL halo columns on the inner side (staged per chunk, P float2 more per row), zero ones on the image border.
template <int NCH, int W, int TH, int KC, int GPW, class DS, bool NEXT, int WI = W>
__global__ __launch_bounds__(512, 4) void eesp_dw_exp_kernel(const float* __restrict__ r, const float* __restrict__ dwp,
                                                             const float* __restrict__ ap, const float* __restrict__ escale,
                                                             const float* __restrict__ eshift, const float* __restrict__ ealpha,
                                                             const float* __restrict__ res, float* __restrict__ out,
                                                             const float* __restrict__ npw, const float* __restrict__ nscale,
                                                             const float* __restrict__ nshift, const float* __restrict__ nalpha,
                                                             float* __restrict__ rnext,
                                                             int H, int bands, int nwg, unsigned long long* __restrict__ stamps) {
    constexpr int MAXD = DS::maxd();
    constexpr int ROWS = TH + 2 * MAXD;
    constexpr int P = 4;                       // zero columns on either side of a staged row (>= MAXD, even: 8-byte aligned rows)
    constexpr int RS = W + 2 * P;
    constexpr int NCHUNK = NCH / KC;
    constexpr int RTPG = NCH / 32;             // 32-row tiles per group
    constexpr int NPW = 4 / GPW;               // workgroups per band
    constexpr int PXV = TH * W;                // pixels of a band
    constexpr bool SPLIT = WI != W;            // half-row bands
    static_assert(!SPLIT || (WI == 2 * W && TH == 1), "half-row bands: TH = 1, two bands per row");
    constexpr int HW2 = SPLIT ? (W + 2 * 4) / 2 : W / 2;      // float2 items of a staged row (SPLIT: with the halo columns)
    constexpr int ITEMS = KC * ROWS * HW2;     // float2 items of a staged chunk
    constexpr int IPT = (ITEMS + 511) / 512;
    constexpr int CPW = KC / 8;                // channels per wave and chunk in the depthwise stage
    constexpr int A4 = KC / 8;                 // float4 of A per lane and chunk
    static_assert(GPW * RTPG == 8, "eight waves: (group, row tile)");
    static_assert(PXV <= 64 && (W % 2) == 0 && MAXD <= P, "band geometry");
    static_assert(KC % 8 == 0 && NCH % KC == 0 && NCHUNK >= 2, "chunking");

    constexpr int RT_F = KC * ROWS * RS, BB_F = GPW * KC * 64;          // floats of one row buffer / one B-operand buffer
    __shared__ __attribute__((aligned(16))) float lds_[2 * RT_F + 2 * BB_F];
    float* const rt_[2] = {lds_, lds_ + RT_F};
    float* const bb_[2] = {lds_ + 2 * RT_F, lds_ + 2 * RT_F + BB_F};

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, half = lane >> 5;

#ifdef MSPL_DEBUG_STAMPS
    unsigned long long st[12];
    int nst = 0;
    auto stamp = [&]() { if (stamps) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); st[nst] = __builtin_amdgcn_s_memrealtime(); } ++nst; };
#else
    auto stamp = [&]() {};
#endif
    stamp();

    int lid = xe_xcd_remap(blockIdx.x, nwg);
    int gs;
    if (NPW == 2 && (nwg & 15) == 0) {
        // Two workgroups per band: the one with groups 2-3 computes all four depthwise branches, the one with groups 0-1 only two (the
        // hierarchical sum needs every branch below a group's own): ~1.3x the time.  When the grid is more than one round of
        // workgroups (288 on 256 CUs at 18 x 30, batch 16) the leftover ones should be the SHORT ones: inside every XCD's chunk of
        // the grid (workgroups are dealt to the XCDs round-robin, a chunk starts in order) the long workgroups of the chunk's
        // bands come first, the short ones after them.  Same (band, group set) pairs, another launch order (measured alone, level 4:
        // 35.3 -> 34.8 us at batch 16, 58.0 -> 54.4 us at batch 32, the size the label lanes launch).
        const int q = nwg >> 3, half = q >> 1;
        const int xcd = lid / q, l = lid - xcd * q;
        gs = l < half ? 1 : 0;
        lid = xcd * half + (l < half ? l : l - half);
    } else {
        gs = lid % NPW;  lid /= NPW;
    }
    const int band = lid % bands;
    const int img = lid / bands;
    const int y0 = SPLIT ? band >> 1 : band * TH;
    const int x0 = SPLIT ? (band & 1) * W : 0;           // first image column of the band
    const int HW = H * WI;
    const int g_first = gs * GPW;

    const int gl = wave / RTPG, rtile = wave % RTPG;     // matrix stage: this wave's group (local) and row tile
    const int g = g_first + gl;

    // ---- staging (L): chunk c of the reduced tensor, rows y0 - MAXD .. y0 + TH - 1 + MAXD of KC channels
    const float* rimg = r + (size_t)img * NCH * HW;
    const int glin0 = (y0 - MAXD) * (W / 2);             // float2 index of the first staged row inside a plane (may be negative)
    float2 sv[IPT];
    auto load_chunk = [&](int c) {
        const float* rc = rimg + (size_t)c * KC * HW;
#pragma unroll
        for (int q = 0; q < IPT; ++q) {
            const int i = tid + 512 * q;
            const int ic = i < ITEMS ? i : 0;
            const int j = ic / (ROWS * HW2);
            const int rem = ic - j * (ROWS * HW2);
            if (!SPLIT) {
                const int gl2 = glin0 + rem;
                const bool ok = gl2 >= 0 && gl2 < H * HW2;
                const int gc = ok ? gl2 : 0;
                const float2 v = *reinterpret_cast<const float2*>(rc + (size_t)j * HW + 2 * gc);   // unconditional load from a clamped address
                sv[q] = ok ? v : make_float2(0.f, 0.f);
            } else {
                const int row = rem / HW2, c2 = rem - row * HW2;
                const int gy = y0 - MAXD + row, gx = x0 - 4 + 2 * c2;                   // (P = 4: float2-aligned columns)
                const bool ok = gy >= 0 && gy < H && gx >= 0 && gx < WI;
                const int go = ok ? gy * WI + gx : 0;
                const float2 v = *reinterpret_cast<const float2*>(rc + (size_t)j * HW + go);
                sv[q] = ok ? v : make_float2(0.f, 0.f);
            }
        }
    };
    auto store_chunk = [&](float* dst) {
#pragma unroll
        for (int q = 0; q < IPT; ++q) {
            const int i = tid + 512 * q;
            if (i < ITEMS) {
                const int j = i / (ROWS * HW2);
                const int rem = i - j * (ROWS * HW2);
                const int row = rem / HW2, c2 = rem - row * HW2;
                *reinterpret_cast<float2*>(dst + (j * ROWS + row) * RS + (SPLIT ? 0 : P) + 2 * c2) = sv[q];
            }
        }
    };

    // ---- depthwise stage (D): lane = pixel of the band; a wave takes CPW channels of the chunk, one at a time.  The channel's 48
    // constants come through scalar loads (one 192-byte record; measured against broadcast LDS reads of a staged copy: 36 vs 42 us
    // per level-4 launch), the taps are ds_read_b32 with immediate offsets into the zero-haloed rows.
    const int pxc = lane < PXV ? lane : PXV - 1;
    const int ty = (TH > 1) ? pxc / W : 0;
    const int tx = pxc - ty * W;
    const int tapbase = (ty + MAXD) * RS + P + tx;
    const int nbr = g_first + GPW;                                     // uniform: a workgroup needs branches 0 .. its last group
    auto dw_stage = [&](int c, const float* src, float* dst) {
#pragma unroll
        for (int u = 0; u < CPW; ++u) {
            const int j = wave + 8 * u;
            const float* rec = dwp + (size_t)(c * KC + j) * XE_REC;     // uniform
            float wv[XE_REC];
#pragma unroll
            for (int i = 0; i < XE_REC; ++i) wv[i] = rec[i];
            const float* tp = src + j * (ROWS * RS) + tapbase;
            float prev = 0.f;
            // taps: the centre is the same value for every dilation and two branches with the same dilation (level 4: 1, 1, 2, 3) read
            // the same nine -- each distinct tap is read from LDS once (25 reads instead of 36 per channel at level 4, 33 at level 3;
            // same values into the same multiply-adds in the same order: bit-identical)
            float tap[4][9];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k >= nbr) break;
                const int d = DS::d(k);
                int same = -1;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    if (kk < k && same < 0 && DS::d(kk) == d) same = kk;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int q = ky * 3 + kx;
                        if (same >= 0) tap[k][q] = tap[same][q];
                        else if (q == 4 && k > 0) tap[k][q] = tap[0][4];
                        else tap[k][q] = tp[(ky - 1) * d * RS + (kx - 1) * d];
                    }
                float a = 0.f;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    a = fmaf(wv[k * 12 + ky * 3 + 0], tap[k][ky * 3 + 0], a);
                    a = fmaf(wv[k * 12 + ky * 3 + 1], tap[k][ky * 3 + 1], a);
                    a = fmaf(wv[k * 12 + ky * 3 + 2], tap[k][ky * 3 + 2], a);
                }
                a += prev;                                              // hierarchical feature fusion (nn_layers/eesp.py:72-76)
                prev = a;
                if (k >= g_first) {
                    float q = fmaf(a, wv[k * 12 + 9], wv[k * 12 + 10]);
                    q = q > 0.f ? q : wv[k * 12 + 11] * q;
                    dst[((k - g_first) * KC + j) * 64 + lane] = q;
                }
            }
        }
    };

    // ---- matrix stage (M)
    floatx16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    const float4* apw = reinterpret_cast<const float4*>(ap) + ((size_t)((g * RTPG + rtile) * NCHUNK) * 64 + lane) * A4;
    float4 a_cur[A4], a_nxt[A4];
    auto load_a = [&](int c, float4 (&dst)[A4]) {
#pragma unroll
        for (int i = 0; i < A4; ++i) dst[i] = apw[(size_t)c * 64 * A4 + i];
    };
    auto mm_stage = [&](const float* src, const float4 (&av)[A4]) {
        const float* bp = src + (gl * KC + half) * 64 + 2 * li;
#pragma unroll
        for (int i = 0; i < KC / 2; ++i) {
            const float2 b2 = *reinterpret_cast<const float2*>(bp + 2 * i * 64);
            const float4 a4 = av[i >> 2];
            const float a = (i & 3) == 0 ? a4.x : (i & 3) == 1 ? a4.y : (i & 3) == 2 ? a4.z : a4.w;
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b2.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b2.y, acc1, 0, 0, 0);
        }
    };

    // ---- prologue: zero the halo columns of both row buffers (never overwritten), chunk 0 -> LDS, branches of chunk 0
    load_chunk(0);
    load_a(0, a_cur);
    if (!SPLIT) {
        for (int i = tid; i < 2 * KC * ROWS * 2 * P; i += 512) {
            const int c8 = i % (2 * P);
            const int rr = i / (2 * P);                              // (buffer, channel, row) flattened
            lds_[rr * RS + (c8 < P ? c8 : W + c8)] = 0.f;
        }
    }
    store_chunk(rt_[0]);
    load_chunk(1);
    __syncthreads();
    stamp();
    dw_stage(0, rt_[0], bb_[0]);
    store_chunk(rt_[1]);
    __syncthreads();
    stamp();

    // residual / output addressing of the epilogue (lane: pixels 2 li, 2 li + 1 of the band; rows (r & 3) + 8 (r >> 2) + 4 half)
    const int rows_here = (H - y0) < TH ? (H - y0) : TH;
    const bool pok = 2 * li < rows_here * W;
    const int ch0 = g * NCH + rtile * 32 + 4 * half;
    const size_t obase = ((size_t)img * 4 * NCH + ch0) * HW + (size_t)y0 * WI + x0 + 2 * li;
    float2 resv[16];
    constexpr int M1 = NCH / 4;                                    // rows of a group of the next projection
    constexpr int RPW = 16 / RTPG;                                 // accumulator registers of the next stage that a wave finishes
    float4 npa4[4], nsc4[RPW / 4], nsh4[RPW / 4], nal4[RPW / 4];

#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
        if (c + 2 < NCHUNK) load_chunk(c + 2);
        if (c + 1 < NCHUNK) {
            if ((c & 1) == 0) load_a(c + 1, a_nxt); else load_a(c + 1, a_cur);
            dw_stage(c + 1, rt_[(c + 1) & 1], bb_[(c + 1) & 1]);
        } else {
            // last chunk: the residual rows fly during its MFMAs
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                resv[q] = make_float2(0.f, 0.f);
                if (pok) resv[q] = *reinterpret_cast<const float2*>(res + obase + (size_t)((q & 3) + 8 * (q >> 2)) * HW);
            }
        }
        if ((c & 1) == 0) mm_stage(bb_[c & 1], a_cur); else mm_stage(bb_[c & 1], a_nxt);
        if (c + 2 < NCHUNK) store_chunk(rt_[c & 1]);
        stamp();
        if (c + 1 < NCHUNK) __syncthreads();
    }

    // ---- epilogue: folded BN, + residual, PReLU (the order of conv1x1's epilogue)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {               // two halves (register budget); the next stage's weights are requested between them
        float4 sc4[2], sh4[2], al4[2];
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
            sc4[r2] = *reinterpret_cast<const float4*>(escale + ch0 + 8 * (2 * hf + r2));
            sh4[r2] = *reinterpret_cast<const float4*>(eshift + ch0 + 8 * (2 * hf + r2));
            al4[r2] = *reinterpret_cast<const float4*>(ealpha + ch0 + 8 * (2 * hf + r2));
        }
        if (NEXT && hf == 1) {
            const float4* npl = reinterpret_cast<const float4*>(npw) + ((size_t)(g * RTPG + rtile) * 64 + lane) * 4;
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) npa4[j4] = npl[j4];
        }
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
            const int rg = 2 * hf + r2;
            const float scv[4] = {sc4[r2].x, sc4[r2].y, sc4[r2].z, sc4[r2].w}, shv[4] = {sh4[r2].x, sh4[r2].y, sh4[r2].z, sh4[r2].w};
            const float alv[4] = {al4[r2].x, al4[r2].y, al4[r2].z, al4[r2].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rr = rg * 4 + q;
                float v0 = fmaf(acc0[rr], scv[q], shv[q]) + resv[rr].x;
                float v1 = fmaf(acc1[rr], scv[q], shv[q]) + resv[rr].y;
                v0 = v0 > 0.f ? v0 : alv[q] * v0;
                v1 = v1 > 0.f ? v1 : alv[q] * v1;
                if (pok) *reinterpret_cast<float2*>(out + obase + (size_t)(q + 8 * rg) * HW) = make_float2(v0, v1);
                if (NEXT) { acc0[rr] = v0; acc1[rr] = v1; }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (NEXT) {
        // ---- the next block's proj_1x1 on what this workgroup holds: group g's n output rows are exactly the input channels of
        // its group g.  Register j of a lane is row (j & 3) + 8 (j >> 2) + 4 half of the wave's tile at the lane's pixel: as the B
        // operand of k-step j it pairs two rows of the tile (the weights are packed in that order), so the 32 x 64 tile feeds 16
        // MFMAs per pixel sub-tile without moving; the RTPG waves of a group each hold a K-slice: their partial tiles are summed
        // through LDS (the row buffers are free), every wave finishes 16 / RTPG register rows: folded BN, PReLU, 8-byte stores.
#pragma unroll
        for (int q4 = 0; q4 < RPW / 4; ++q4) {
            const int chn = g * M1 + 8 * ((rtile * RPW) / 4 + q4) + 4 * half;          // four consecutive rows per register quad
            const bool okc = 8 * ((rtile * RPW) / 4 + q4) + 4 * half < M1;
            nsc4[q4] = okc ? *reinterpret_cast<const float4*>(nscale + chn) : make_float4(0.f, 0.f, 0.f, 0.f);
            nsh4[q4] = okc ? *reinterpret_cast<const float4*>(nshift + chn) : make_float4(0.f, 0.f, 0.f, 0.f);
            nal4[q4] = okc ? *reinterpret_cast<const float4*>(nalpha + chn) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float npa[16] = {npa4[0].x, npa4[0].y, npa4[0].z, npa4[0].w, npa4[1].x, npa4[1].y, npa4[1].z, npa4[1].w,
                               npa4[2].x, npa4[2].y, npa4[2].z, npa4[2].w, npa4[3].x, npa4[3].y, npa4[3].z, npa4[3].w};
        floatx16 p0, p1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { p0[i] = 0.f; p1[i] = 0.f; }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            p0 = __builtin_amdgcn_mfma_f32_32x32x2f32(npa[j], acc0[j], p0, 0, 0, 0);
            p1 = __builtin_amdgcn_mfma_f32_32x32x2f32(npa[j], acc1[j], p1, 0, 0, 0);
        }
        // a wave keeps the RPW registers it finishes and sends the others to the waves that finish them: one round through LDS
        // (all of it is free once every wave is past the last step) for both pixel sub-tiles
        constexpr int SLOT = (RTPG - 1) * RPW;                     // registers a wave receives per sub-tile
        static_assert(2 * RT_F + 2 * BB_F >= 2 * 8 * SLOT * 64, "LDS holds the exchanged partial tiles");
        float* red = lds_;
        __syncthreads();                                           // every wave has finished the last step's LDS reads
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) {
                const int dw = r2 / RPW;                            // destination wave inside the group (compile time)
                if (dw != rtile) {                                  // uniform
                    const int si = rtile < dw ? rtile : rtile - 1;
                    red[((sub * 8 + gl * RTPG + dw) * SLOT + si * RPW + (r2 % RPW)) * 64 + lane] = sub ? p1[r2] : p0[r2];
                }
            }
        }
        __syncthreads();
        float sres[2][RPW];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int q = 0; q < RPW; ++q) {
                float own = 0.f;
#pragma unroll
                for (int r2 = 0; r2 < 16; ++r2)
                    if (r2 == rtile * RPW + q) own = sub ? p1[r2] : p0[r2];          // (rtile is uniform: a select chain, no indexing)
                float t = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < RTPG; ++w2) {                 // partials in wave order, the own one from registers
                    const int si = w2 < rtile ? w2 : w2 - 1;
                    const float v = (w2 == rtile) ? own : red[((sub * 8 + gl * RTPG + rtile) * SLOT + si * RPW + q) * 64 + lane];
                    t = (w2 == 0) ? v : t + v;
                }
                sres[sub][q] = t;
            }
        }
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const int r2 = rtile * RPW + q;
            const int row = (r2 & 3) + 8 * (r2 >> 2) + 4 * half;
            if (row < M1 && pok) {
                const int ch = g * M1 + row;
                const float4 s4 = nsc4[q >> 2], h4 = nsh4[q >> 2], a4 = nal4[q >> 2];
                const float sc = (q & 3) == 0 ? s4.x : (q & 3) == 1 ? s4.y : (q & 3) == 2 ? s4.z : s4.w;
                const float sh = (q & 3) == 0 ? h4.x : (q & 3) == 1 ? h4.y : (q & 3) == 2 ? h4.z : h4.w;
                const float al = (q & 3) == 0 ? a4.x : (q & 3) == 1 ? a4.y : (q & 3) == 2 ? a4.z : a4.w;
                float v0 = fmaf(sres[0][q], sc, sh), v1 = fmaf(sres[1][q], sc, sh);
                v0 = v0 > 0.f ? v0 : al * v0;
                v1 = v1 > 0.f ? v1 : al * v1;
                *reinterpret_cast<float2*>(rnext + ((size_t)img * NCH + ch) * HW + (size_t)y0 * WI + x0 + 2 * li) = make_float2(v0, v1);
            }
        }
    }
#ifdef MSPL_DEBUG_STAMPS
    if (stamps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp();
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 12; ++i) stamps[((size_t)blockIdx.x * 8 + wave) * 12 + i] = st[i];
        }
    }
#endif
}

// ------------------------------------------------------------------ host side
struct XePlan {
    int kind;        // 0: none; 1: n = 128, W = 30; 2: n = 64, W = 60 (288x480 / 256x480 inputs); 3: n = 128, W = 32; 4: n = 64, W = 64 (512-wide inputs);
                     // 5: n = 128, W = 64; 6: n = 64, W = 128 as half-row bands (1024-wide inputs)
    int TH, KC, GPW;
    int split;       // bands per row (1, or 2 for half-row bands)
};

static XePlan xe_plan(int n, int H, int W, const int32_t* dil) {
    XePlan p = {0, 0, 0, 0, 1};
    if (H < 1) return p;
    const bool d1123 = dil[0] == 1 && dil[1] == 1 && dil[2] == 2 && dil[3] == 3, d1234 = dil[0] == 1 && dil[1] == 2 && dil[2] == 3 && dil[3] == 4;
    if (n == 128 && d1123 && (W == 30 || W == 32)) { p.kind = W == 30 ? 1 : 3; p.TH = 2; p.KC = 16; p.GPW = 2; }
    else if (n == 64 && d1234 && (W == 60 || W == 64)) { p.kind = W == 60 ? 2 : 4; p.TH = 1; p.KC = 8; p.GPW = 4; }
    // 1024-pixel-wide inputs (Cityscapes 1024x512, train_espdnetue_city.sh:6): 64 columns at level 4 (one row per band), 128 at level 3
    // (two half-row bands per row)
    else if (n == 128 && d1123 && W == 64) { p.kind = 5; p.TH = 1; p.KC = 16; p.GPW = 2; }
    else if (n == 64 && d1234 && W == 128) { p.kind = 6; p.TH = 1; p.KC = 8; p.GPW = 4; p.split = 2; }
    return p;
}

}  // namespace mspl

using namespace mspl;

extern "C" int mspl_eesp_dw_exp_fits(int32_t N, int32_t n, int32_t H, int32_t W, const int32_t dil[4], uint32_t flags) {
    static const int enabled = MSPL_TUNE_INT("MSPL_EESP_EXP", 1);
    (void)flags;
    if (!enabled || !dil || N < 1) return 0;
    return xe_plan(n, H, W, dil).kind != 0 ? 1 : 0;
}

extern "C" int64_t mspl_eesp_dw_exp_pack_floats(int32_t n) { return (int64_t)n * XE_REC + 4ll * n * n; }

extern "C" int mspl_eesp_dw_exp_pack(const float* w4, const float* bscale, const float* bshift, const float* balpha,
                                     const float* wexp, int32_t n, int32_t H, int32_t W, const int32_t dil[4], float* packed,
                                     void* stream_) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    MSPL_REQUIRE(w4 && bscale && bshift && balpha && wexp && packed && dil, MSPL_ERR_NULL_POINTER, "eesp_dw_exp_pack: null pointer");
    const XePlan p = xe_plan(n, H, W, dil);
    MSPL_REQUIRE(p.kind != 0, MSPL_ERR_UNSUPPORTED, "eesp_dw_exp_pack: n=%d W=%d not covered", n, W);
    const int total = 4 * n * n;
    hipLaunchKernelGGL(eesp_exp_pack_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, stream, w4, bscale, bshift, balpha,
                       wexp, n, p.KC, packed, packed + (size_t)n * XE_REC);
    MSPL_CHECK_LAUNCH("eesp_dw_exp_pack");
    return MSPL_OK;
}

static int xe_launch(const float* r, const float* packed, const int32_t dil[4], int32_t N, int32_t n, int32_t H, int32_t W,
                     const mspl_epilogue_t* ep, float* out, const float* next_packed, const float* nscale, const float* nshift,
                     const float* nalpha, float* rnext, hipStream_t stream) {
    MSPL_REQUIRE(r && packed && dil && ep && out, MSPL_ERR_NULL_POINTER, "eesp_dw_exp: null pointer");
    MSPL_REQUIRE(N >= 1 && H >= 1, MSPL_ERR_BAD_SHAPE, "eesp_dw_exp: N=%d H=%d", N, H);
    if (int rc = check_epi(ep, 4 * n, "eesp_dw_exp")) return rc;
    MSPL_REQUIRE(ep->scale && ep->shift && ep->alpha && ep->residual, MSPL_ERR_NULL_POINTER,
                 "eesp_dw_exp: the stride-1 block needs scale, shift, alpha and residual");
    MSPL_REQUIRE(!ep->pre_add && !ep->reinf_r && !ep->gate && !ep->raw_out && (ep->out_ctot == 0 || (ep->out_ctot == 4 * n && ep->out_coff == 0)),
                 MSPL_ERR_UNSUPPORTED, "eesp_dw_exp: only scale/shift/alpha/residual on an un-sliced destination");
    const XePlan p = xe_plan(n, H, W, dil);
    MSPL_REQUIRE(p.kind != 0, MSPL_ERR_UNSUPPORTED, "eesp_dw_exp: n=%d H=%d W=%d dil=%d,%d,%d,%d not covered", n, H, W, dil[0], dil[1], dil[2], dil[3]);
    MSPL_REQUIRE((((uintptr_t)r | (uintptr_t)out | (uintptr_t)ep->residual | (uintptr_t)packed | (uintptr_t)ep->scale | (uintptr_t)ep->shift |
                   (uintptr_t)ep->alpha | (uintptr_t)rnext | (uintptr_t)next_packed) & 15) == 0, MSPL_ERR_BAD_SHAPE,
                 "eesp_dw_exp: operands must be 16-byte aligned");
    MSPL_REQUIRE((int64_t)N * 4 * n * H * W < (1ll << 31), MSPL_ERR_BAD_SHAPE, "eesp_dw_exp: tensor too large");
    const bool next = next_packed != nullptr;
    MSPL_REQUIRE(!next || (nscale && nshift && nalpha && rnext), MSPL_ERR_NULL_POINTER,
                 "eesp_dw_exp: the next projection needs its packed weights, scale, shift, alpha and a destination");
    const int bands = ceil_div(H, p.TH) * p.split;
    const int64_t nwg = (int64_t)N * bands * (4 / p.GPW);
    MSPL_REQUIRE(nwg < (1ll << 30), MSPL_ERR_BAD_SHAPE, "eesp_dw_exp: grid too large");
    const float* dwp = packed;
    const float* ap = packed + (size_t)n * XE_REC;
    const dim3 grid((unsigned)nwg), blk(512);
#ifdef MSPL_DEBUG_STAMPS
    static unsigned long long* stamp_buf = nullptr;
    static const int dbg_stamp = MSPL_STAMP_ENV("MSPL_XE_STAMP");
    if (dbg_stamp && !stamp_buf) (void)hipMalloc(&stamp_buf, (size_t)8192 * 8 * 12 * sizeof(unsigned long long));
    unsigned long long* stamps = nwg <= 8192 ? stamp_buf : nullptr;
#else
    unsigned long long* const stamps = nullptr;      // (the kernels' stamp argument: compiled out with the stamps)
#endif
#define XE_ARGS grid, blk, 0, stream, r, dwp, ap, ep->scale, ep->shift, ep->alpha, ep->residual, out, next_packed, nscale, nshift, nalpha, rnext, H, bands, (int)nwg, stamps
#define XE_GO(NX) do { \
    if (p.kind == 1) hipLaunchKernelGGL((eesp_dw_exp_kernel<128, 30, 2, 16, 2, XDil<1, 1, 2, 3>, NX>), XE_ARGS); \
    else if (p.kind == 2) hipLaunchKernelGGL((eesp_dw_exp_kernel<64, 60, 1, 8, 4, XDil<1, 2, 3, 4>, NX>), XE_ARGS); \
    else if (p.kind == 3) hipLaunchKernelGGL((eesp_dw_exp_kernel<128, 32, 2, 16, 2, XDil<1, 1, 2, 3>, NX>), XE_ARGS); \
    else if (p.kind == 4) hipLaunchKernelGGL((eesp_dw_exp_kernel<64, 64, 1, 8, 4, XDil<1, 2, 3, 4>, NX>), XE_ARGS); \
    else if (p.kind == 5) hipLaunchKernelGGL((eesp_dw_exp_kernel<128, 64, 1, 16, 2, XDil<1, 1, 2, 3>, NX>), XE_ARGS); \
    else hipLaunchKernelGGL((eesp_dw_exp_kernel<64, 64, 1, 8, 4, XDil<1, 2, 3, 4>, NX, 128>), XE_ARGS); } while (0)
    if (next) XE_GO(true); else XE_GO(false);
#undef XE_GO
#undef XE_ARGS
    MSPL_CHECK_LAUNCH("eesp_dw_exp");
#ifdef MSPL_DEBUG_STAMPS
    if (stamps) {   // debug only (STAMPS=1 builds): synchronous dump of the step timeline (100 MHz ticks)
        (void)hipDeviceSynchronize();
        static unsigned long long host[8192 * 8 * 12];
        (void)hipMemcpy(host, stamps, (size_t)nwg * 8 * 12 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int64_t i = 0; i < nwg * 8; ++i) { if (host[i * 12] < t0) t0 = host[i * 12]; if (host[i * 12 + 11] > t1) t1 = host[i * 12 + 11]; }
        double avg[12] = {0}, mx[12] = {0};
        for (int64_t i = 0; i < nwg * 8; ++i)
            for (int k = 0; k < 12; ++k) { const double v = (double)(host[i * 12 + k] - t0) / 100.0; avg[k] += v / (nwg * 8); if (v > mx[k]) mx[k] = v; }
        fprintf(stderr, "[xe stamp] n=%d H=%d nwg=%lld span=%.2f us; avg (max) us since first start (start, 2 prologue barriers, 8 steps' work done, end): ",
                n, H, (long long)nwg, (t1 - t0) / 100.0);
        for (int k = 0; k < 12; ++k) fprintf(stderr, "%.2f(%.2f) ", avg[k], mx[k]);
        fprintf(stderr, "\n");
    }
#endif
    return MSPL_OK;
}

extern "C" int mspl_eesp_dw_exp_fwd(const float* r, const float* packed, const int32_t dil[4], int32_t N, int32_t n, int32_t H,
                                    int32_t W, const mspl_epilogue_t* ep, float* out, void* stream_) {
    return xe_launch(r, packed, dil, N, n, H, W, ep, out, nullptr, nullptr, nullptr, nullptr, nullptr, static_cast<hipStream_t>(stream_));
}

extern "C" int64_t mspl_eesp_dw_exp_next_pack_floats(int32_t n) { return 4ll * (n / 32) * 16 * 64; }

extern "C" int mspl_eesp_dw_exp_next_pack(const float* w1, int32_t n, float* next_packed, void* stream_) {
    MSPL_REQUIRE(w1 && next_packed, MSPL_ERR_NULL_POINTER, "eesp_dw_exp_next_pack: null pointer");
    MSPL_REQUIRE(n == 64 || n == 128, MSPL_ERR_UNSUPPORTED, "eesp_dw_exp_next_pack: n=%d not covered", n);
    const int total = 4 * (n / 32) * 16 * 64;
    hipLaunchKernelGGL(eesp_exp_pack_next_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream_), w1, n,
                       next_packed);
    MSPL_CHECK_LAUNCH("eesp_dw_exp_next_pack");
    return MSPL_OK;
}

extern "C" int mspl_eesp_dw_exp_next_fwd(const float* r, const float* packed, const int32_t dil[4], int32_t N, int32_t n, int32_t H,
                                         int32_t W, const mspl_epilogue_t* ep, float* out, const float* next_packed,
                                         const float* nscale, const float* nshift, const float* nalpha, float* rnext, void* stream_) {
    MSPL_REQUIRE(next_packed && nscale && nshift && nalpha && rnext, MSPL_ERR_NULL_POINTER, "eesp_dw_exp_next: null pointer");
    return xe_launch(r, packed, dil, N, n, H, W, ep, out, next_packed, nscale, nshift, nalpha, rnext, static_cast<hipStream_t>(stream_));
}
