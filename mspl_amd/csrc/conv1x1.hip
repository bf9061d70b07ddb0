// K1 / K3 -- grouped 1x1 convolution with a fused epilogue.
//
// Reference arithmetic: nn.Conv2d(k=1, groups=g, bias=False) followed by BatchNorm (eval), optional
// DownSampler reinforcement / EESP residual, PReLU: nn_layers/eesp.py:36,55,67,77-93,117-120,142;
// nn_layers/efficient_pyramid_pool.py:22,31; nn_layers/espnet_utils.py:8-37,62-89.
//
// Per group this is C[M x P] = Wg[M x K] . X[K x P]; pixels P are contiguous in NCHW.
//
// K >= 16, HW % 4 == 0 (the path's shapes): fp32 matrix cores, 16-byte memory ops.
//   v_mfma_f32_32x32x2_f32 is exact fp32 (a k-ordered fmaf chain) at the vector-FMA rate, takes one VGPR
//   per operand and leaves the VALU free for the epilogue.  A wave owns 128 consecutive pixels of the
//   flattened (image, pixel) axis and MCW 32-row chunks of M.  Lane (half, li) loads ONE float4
//   X[k+half][4*li .. 4*li+3] per k-step and uses its four components as the B operand of four MFMAs
//   (pixel sub-tile s = pixels 4*li+s), so every HBM access is 16 bytes per lane and the accumulators of a
//   row come out as 4 consecutive pixels = one float4 store.  Weights (A operand) and per-row epilogue
//   constants are staged once per workgroup in LDS (odd row stride: conflict-free ds_read_b32).
// K >= 16 otherwise: same structure with one pixel per lane (4-byte accesses).
// K < 16: the matrix tile would be mostly padding; a VALU kernel streams 4 pixels per lane.
#include <stdlib.h>

#include <mutex>

#include "common.hpp"

namespace mspl {

int conv1x1_splitk_try(const float* x, const float* w, int N, int Cin, int Cout, int groups, int HW, const Epi& e, float* out,
                       hipStream_t s);

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct PwGeom {
    int N, Cin, Cout, G, K, M, HW;
    int KS;        // LDS row stride of A (odd, >= K rounded up to 32)
    int MB;        // rows of M handled per workgroup (multiple of 32)
    int mblocks;   // ceil(M / MB)
    int mc_total;  // 32-row chunks inside one workgroup's MB
    int WM;        // waves along M (1,2,4); WP = 4 / WM waves along pixels
    int TPW;       // pixel tiles per wave (sequential)
    int ptiles;    // pixel tiles (32 px per image, or 128 px of the flattened batch for the float4 kernel)
    int pgroups;   // ceil(ptiles / (WP * TPW))
    int vecw;      // 16-byte aligned weights: stage with float4 loads
    unsigned long long* stamps;   // tuning aid (MSPL_PW_STAMP): 4 s_memrealtime stamps per workgroup, or null
    int astage;    // register-weights kernel: 1 = weights go global -> LDS (coalesced) -> registers, 0 = global -> registers
    int AR;        // tile-pipelined kernel: weight rows actually staged (min(MB, M)); lanes of absent rows re-read row AR-1
};

// Per-row epilogue constants staged in LDS as six arrays of MB floats (scale, shift, alpha, rw0, rw1, rw2), so
// the four consecutive rows a lane owns per accumulator quad are one ds_read_b128 per constant.
constexpr int ROWC = 6;

// Stage rows [m0, m0+MB) of one group's weights into At[MB][KS] (zero padded to K32 columns) and the rows'
// epilogue constants into rowc.  Loads are independent and unrolled so that they overlap.
__device__ __forceinline__ void stage_weights(float* At, float* rowc, const float* __restrict__ wg,
                                              const PwGeom& g, const Epi& e, int m0, int cbase, int tid) {
    const int K32 = (g.K + 31) & ~31;
    if (g.vecw) {
        const int kv = K32 >> 2;
        const int total = g.MB * kv;
#pragma unroll 8
        for (int i = tid; i < total; i += 256) {
            const int m = i / kv, k = (i - m * kv) << 2;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m0 + m < g.M && k < g.K) v = *reinterpret_cast<const float4*>(wg + (size_t)m * g.K + k);
            float* d = At + m * g.KS + k;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    } else {
#pragma unroll 4
        for (int i = tid; i < g.MB * K32; i += 256) {
            const int m = i / K32, k = i - m * K32;
            At[m * g.KS + k] = (m0 + m < g.M && k < g.K) ? wg[(size_t)m * g.K + k] : 0.f;
        }
    }
    for (int m = tid; m < g.MB; m += 256) {
        EpiCh c = {1.f, 0.f, 1.f, 0.f, 0.f, 0.f};
        if (m0 + m < g.M) c = epi_channel(e, cbase + m);
        rowc[m] = c.scale; rowc[g.MB + m] = c.shift; rowc[2 * g.MB + m] = c.alpha;
        rowc[3 * g.MB + m] = c.rw0; rowc[4 * g.MB + m] = c.rw1; rowc[5 * g.MB + m] = c.rw2;
    }
}

// ------------------------------------------------------------------ MFMA kernel, NSUB pixels per lane
template <int NSUB> struct PixVec;
template <> struct PixVec<4> { typedef float4 T; };
template <> struct PixVec<2> { typedef float2 T; };
template <> struct PixVec<1> { typedef float T; };

template <int NSUB>
__device__ __forceinline__ void vec_load(const void* p, float (&v)[NSUB]) {
    if constexpr (NSUB == 4) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    else if constexpr (NSUB == 2) { const float2 t = *reinterpret_cast<const float2*>(p); v[0] = t.x; v[1] = t.y; }
    else { v[0] = *reinterpret_cast<const float*>(p); }
}
template <int NSUB>
__device__ __forceinline__ void vec_store(void* p, const float (&v)[NSUB]) {
    if constexpr (NSUB == 4) store_out4(reinterpret_cast<float*>(p), make_float4(v[0], v[1], v[2], v[3]));        // write-through (common.hpp)
    else if constexpr (NSUB == 2) store_out2(reinterpret_cast<float*>(p), make_float2(v[0], v[1]));
    else *reinterpret_cast<float*>(p) = v[0];
}

template <int NSUB, int RING>
__global__ __launch_bounds__(256, ((NSUB == 4 || RING > 4) ? 2 : 3)) void conv1x1_mfma_kernel(const float* __restrict__ x,
                                                                               const float* __restrict__ w,
                                                                               PwGeom g, Epi e,
                                                                               float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* rowc = smem;                                                    // [ROWC][MB]
    float* At = smem + (size_t)g.MB * ROWC;                                // [MB][KS]
    int bid = blockIdx.x;
    const int pg = bid % g.pgroups;  bid /= g.pgroups;
    const int mb = bid % g.mblocks;
    const int grp = bid / g.mblocks;
    const int tid = threadIdx.x;
    const int m0 = mb * g.MB;
    const int cbase = e.coff + grp * g.M + m0;   // absolute destination channel of local row 0
    const int wave = tid >> 6, lane = tid & 63;
    const int WP = 4 / g.WM;
    const int chunk = wave % g.WM, wp = wave / g.WM;         // this wave's 32-row chunk of MB and pixel slot
    const int li = lane & 31, half = lane >> 5;
    const int mrem = g.M - m0;
    const int total_px = g.N * g.HW;
    const size_t rowbytes = (size_t)g.HW * sizeof(float);
    const size_t orow = (size_t)e.hw * sizeof(float);
    const int mlh = chunk * 32 + 4 * half;                    // local row of accumulator register 0
    const int klast = g.K - 2;
    const bool active = chunk < g.mc_total;

    // Per-tile operand fetch: the residual rows (where registers allow) and the first RING groups of B.  The fetch
    // for the first tile is issued BEFORE the weights are staged, so its HBM latency hides under the staging and the
    // barrier; the fetch for tile t+1 is issued right after tile t's last MFMA.
    // B rows kb0 + 2*kk + half, kk = 0..3.  K is even on this path; a pair beyond K is clamped to the last valid
    // pair (finite data) and contributes nothing because A is zero padded to K32 columns.
    float resv[16][NSUB];
    float b[RING][4][NSUB];
    auto tile_px = [&](int t, int& gp, bool& pok, int& img, int& p) {
        const int ptile = (pg * g.TPW + t) * WP + wp;
        gp = (ptile * 32 + li) * NSUB;                        // first of this lane's NSUB pixels (flattened batch)
        pok = gp < total_px;                                  // HW % NSUB == 0: a lane's pixels share an image
        const int gpc = pok ? gp : 0;
        img = gpc / g.HW;  p = gpc - img * g.HW;
        return ptile < g.ptiles;
    };
    auto load_group = [&](const char* xb, float (&bb)[4][NSUB], int kb0) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            int kr = kb0 + 2 * kk;                            // uniform
            kr = kr < klast ? kr : klast;
            vec_load<NSUB>(xb + (size_t)kr * rowbytes, bb[kk]);
        }
    };
    auto load_residual = [&](unsigned ooff) {
        const char* rb = reinterpret_cast<const char*>(e.residual);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
#pragma unroll
            for (int s = 0; s < NSUB; ++s) resv[r][s] = 0.f;
            if (mlh + dr < mrem) vec_load<NSUB>(rb + dr * orow + ooff, resv[r]);
        }
    };
    auto fetch_res = [&](int t) {
        int gp, img, p;  bool pok;
        if (NSUB == 4 || !e.residual || !tile_px(t, gp, pok, img, p)) return;
        load_residual((unsigned)((((size_t)img * e.ctot + cbase + mlh) * (size_t)e.hw + p) * sizeof(float)));
    };
    auto fetch_b = [&](int t) {
        int gp, img, p;  bool pok;
        if (!tile_px(t, gp, pok, img, p)) return;
        const size_t voff = (((size_t)img * g.Cin + (size_t)grp * g.K + half) * g.HW + p) * sizeof(float);
        const char* xb = reinterpret_cast<const char*>(x) + voff;
#pragma unroll
        for (int i = 0; i < RING; ++i)
            if (i * 8 < g.K) load_group(xb, b[i], i * 8);
    };
    unsigned long long st0 = 0, st1 = 0, st2 = 0;
    if (g.stamps) st0 = __builtin_amdgcn_s_memrealtime();
    if (active) { fetch_res(0); fetch_b(0); }

    stage_weights(At, rowc, w + ((size_t)grp * g.M + m0) * g.K, g, e, m0, cbase, tid);
    __syncthreads();
    if (!active) return;                                      // (no barrier follows)
    if (g.stamps) st1 = __builtin_amdgcn_s_memrealtime();
    const float* arow0 = At + (size_t)(chunk * 32 + li) * g.KS + half;

    for (int t = 0; t < g.TPW; ++t) {
        int gp, img, p;  bool pok;
        if (!tile_px(t, gp, pok, img, p)) break;              // wave-uniform
        // byte offset of X[img][grp*K + half][p] from x; k advances through a uniform (SGPR) base
        const size_t voff = (((size_t)img * g.Cin + (size_t)grp * g.K + half) * g.HW + p) * sizeof(float);
        // destination-shaped tensors (out / residual / pre_add): uniform row base + one 32-bit lane offset
        const unsigned ooff = (unsigned)((((size_t)img * e.ctot + cbase + mlh) * (size_t)e.hw + p) * sizeof(float));
        const char* xb = reinterpret_cast<const char*>(x) + voff;
        floatx16 acc[NSUB];
#pragma unroll
        for (int s = 0; s < NSUB; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;

        // K loop: a ring of RING groups (8 k-values each) stays in flight; slot i is refilled right after its
        // MFMAs are issued, so ~3 groups of HBM loads overlap every group of matrix work.
#pragma unroll 1
        for (int kb0 = 0; kb0 < g.K; kb0 += 8 * RING) {
#pragma unroll
            for (int i = 0; i < RING; ++i) {
                const int kb = kb0 + 8 * i;
                if (kb < g.K) {                                   // uniform
                    const float* arow = arow0 + kb;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const float a = arow[2 * kk];
#pragma unroll
                        for (int s = 0; s < NSUB; ++s)
                            acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[i][kk][s], acc[s], 0, 0, 0);
                    }
                    if (kb + 8 * RING < g.K) load_group(xb, b[i], kb + 8 * RING);
                }
            }
        }
        if (g.stamps && t == 0) { asm volatile("s_nop 0" :: "v"(acc[0][0])); st2 = __builtin_amdgcn_s_memrealtime(); }
        if (t + 1 < g.TPW) fetch_b(t + 1);                    // next tile's first groups fly during this epilogue

        // ---- epilogue.  Sub-tile s, register r: row = (r & 3) + 8 * (r >> 2) + 4 * half, pixel gp + s.
        if (NSUB == 4 && e.residual) load_residual(ooff);
        float rr[3][NSUB];
        if (e.reinf_r) {
            const float* rp = e.reinf_r + (size_t)img * 3 * e.hw + p;
            vec_load<NSUB>(rp, rr[0]);
            vec_load<NSUB>(rp + e.hw, rr[1]);
            vec_load<NSUB>(rp + 2 * (size_t)e.hw, rr[2]);
        }
        const float* gate = e.gate ? e.gate + (size_t)img * e.ctot + cbase : nullptr;
        char* ob = reinterpret_cast<char*>(out);
        const char* pb = reinterpret_cast<const char*>(e.pre_add);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int mlb = mlh + 8 * rg;                                    // first of 4 consecutive local rows
            const float4 sc4 = *reinterpret_cast<const float4*>(rowc + mlb);
            const float4 sh4 = *reinterpret_cast<const float4*>(rowc + g.MB + mlb);
            const float4 al4 = *reinterpret_cast<const float4*>(rowc + 2 * g.MB + mlb);
            float4 w04 = make_float4(0.f, 0.f, 0.f, 0.f), w14 = w04, w24 = w04;
            if (e.reinf_r) {
                w04 = *reinterpret_cast<const float4*>(rowc + 3 * g.MB + mlb);
                w14 = *reinterpret_cast<const float4*>(rowc + 4 * g.MB + mlb);
                w24 = *reinterpret_cast<const float4*>(rowc + 5 * g.MB + mlb);
            }
            const float scv[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, shv[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
            const float alv[4] = {al4.x, al4.y, al4.z, al4.w};
            const float w0v[4] = {w04.x, w04.y, w04.z, w04.w}, w1v[4] = {w14.x, w14.y, w14.z, w14.w};
            const float w2v[4] = {w24.x, w24.y, w24.z, w24.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = rg * 4 + q, ml = mlb + q;
                const size_t rowoff = (size_t)(q + 8 * rg) * orow;           // uniform
                if (ml < mrem && pok) {
                    float v[NSUB];
#pragma unroll
                    for (int s = 0; s < NSUB; ++s) v[s] = acc[s][r];
                    if (e.raw) vec_store<NSUB>(reinterpret_cast<char*>(e.raw) + rowoff + ooff, v);      // (ctot == Cout, coff == 0: same offset)
                    if (pb) {
                        float pa[NSUB];
                        vec_load<NSUB>(pb + rowoff + ooff, pa);
#pragma unroll
                        for (int s = 0; s < NSUB; ++s) v[s] += pa[s];
                    }
                    const float gv = gate ? gate[ml] : 1.f;
#pragma unroll
                    for (int s = 0; s < NSUB; ++s) {
                        float t2 = fmaf(v[s], scv[q], shv[q]);
                        if (e.reinf_r) t2 += w0v[q] * rr[0][s] + w1v[q] * rr[1][s] + w2v[q] * rr[2][s];
                        if (e.residual) t2 += resv[r][s];
                        if (e.alpha) t2 = t2 > 0.f ? t2 : alv[q] * t2;
                        v[s] = t2 * gv;
                    }
                    vec_store<NSUB>(ob + rowoff + ooff, v);
                }
            }
            asm volatile("" ::: "memory");   // one row quad at a time: keeps hipcc from hoisting all 16 rows' work
        }
        if (t + 1 < g.TPW) fetch_res(t + 1);
    }
    if (g.stamps && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* d = g.stamps + (size_t)blockIdx.x * 4;
        d[0] = st0; d[1] = st1; d[2] = st2; d[3] = __builtin_amdgcn_s_memrealtime();
        // placement: HW_REG_HW_ID (4) and HW_REG_XCC_ID (20) packed into the top bits of d[2]'s unused range
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        g.stamps[(size_t)4 * 65536 + blockIdx.x] = ((unsigned long long)xcc << 32) | hw;
    }
}

// ------------------------------------------------------------------ MFMA kernel, weights held in registers
// For K <= 128 per group (every EESP projection / expansion of the path) a wave's 32-row chunk of the weights is only
// K/2 values per lane in the MFMA A layout, so it lives in VGPRs for the whole kernel: no LDS weight tile, no staging
// barrier before the first matrix instruction, no ds_read per MFMA, and a wave walks TPW pixel tiles with the same
// registers.  B operands stream from HBM through a ring of RING groups (4 k-steps each) exactly as above; the K loop
// is fully unrolled (static register indices).  Only the per-row epilogue constants go through LDS.
// NG = K / 8 (compile time: a uniform branch per group would force a full s_waitcnt at every join and serialise the
// ring); requires 16-byte aligned weight rows.
template <int NSUB, int NG>
__global__ __launch_bounds__(256, 2) void conv1x1_areg_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              PwGeom g, Epi e, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* rowc = smem;                                                    // [ROWC][MB]
    int bid = blockIdx.x;
    const int pg = bid % g.pgroups;  bid /= g.pgroups;
    const int mb = bid % g.mblocks;
    const int grp = bid / g.mblocks;
    const int tid = threadIdx.x;
    const int m0 = mb * g.MB;
    const int cbase = e.coff + grp * g.M + m0;   // absolute destination channel of local row 0
    for (int m = tid; m < g.MB; m += 256) {
        EpiCh c = {1.f, 0.f, 1.f, 0.f, 0.f, 0.f};
        if (m0 + m < g.M) c = epi_channel(e, cbase + m);
        rowc[m] = c.scale; rowc[g.MB + m] = c.shift; rowc[2 * g.MB + m] = c.alpha;
        rowc[3 * g.MB + m] = c.rw0; rowc[4 * g.MB + m] = c.rw1; rowc[5 * g.MB + m] = c.rw2;
    }

    const int wave = tid >> 6, lane = tid & 63;
    const int WP = 4 / g.WM;
    const int chunk = wave % g.WM, wp = wave / g.WM;         // this wave's 32-row chunk of MB and pixel slot
    const int li = lane & 31, half = lane >> 5;
    const int mrem = g.M - m0;
    const int total_px = g.N * g.HW;
    const size_t rowbytes = (size_t)g.HW * sizeof(float);

    // A operand: a[ks] = W[m0 + chunk*32 + li][2*ks + half], ks < K/2.
    constexpr int NKS = 4 * NG;
    float a[NKS];
    if (g.astage) {
        // coalesced: the workgroup stages its MB x K weight tile in LDS once (odd row stride), then every wave copies
        // its 32-row chunk into registers (conflict-free ds_read_b32); LDS is not touched again in the K loop
        float* At = smem + (size_t)g.MB * ROWC;
        const int kv = g.K >> 2, total = g.MB * kv;
        const float* wg = w + ((size_t)grp * g.M + m0) * g.K;
        constexpr int UL = 8;
        for (int base = 0; base < total; base += 256 * UL) {
            float4 v[UL];
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                const int i = base + u * 256 + tid;
                const int m = i / kv, k = (i - m * kv) << 2;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < total && m < mrem) v[u] = *reinterpret_cast<const float4*>(wg + (size_t)m * g.K + k);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                const int i = base + u * 256 + tid;
                if (i < total) {
                    const int m = i / kv, k = (i - m * kv) << 2;
                    float* d = At + m * g.KS + k;
                    d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
                }
            }
        }
        __syncthreads();
        const float* ar = At + (size_t)(chunk * 32 + li) * g.KS + half;
        if (chunk < g.mc_total) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) a[ks] = ar[2 * ks];
        }
    } else {
        // direct: one 16-byte load covers k = 4j..4j+3 = the k-steps 2j (x | y) and 2j+1 (z | w); rows beyond M are zero
        const int row = chunk * 32 + li;
        const bool rok = chunk < g.mc_total && row < mrem;
        const float* wr = w + ((size_t)grp * g.M + m0 + (rok ? row : 0)) * g.K;
#pragma unroll
        for (int j = 0; j < NKS / 2; ++j) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rok) v = *reinterpret_cast<const float4*>(wr + 4 * j);
            a[2 * j] = half ? v.y : v.x;
            a[2 * j + 1] = half ? v.w : v.z;
        }
    }
    __syncthreads();                                          // rowc visible (the only barrier)
    if (chunk >= g.mc_total) return;

    for (int t = 0; t < g.TPW; ++t) {
        const int ptile = (pg * g.TPW + t) * WP + wp;
        if (ptile >= g.ptiles) break;                         // wave-uniform
        const int gp = (ptile * 32 + li) * NSUB;              // first of this lane's NSUB pixels (flattened batch)
        const bool pok = gp < total_px;                       // HW % NSUB == 0: a lane's pixels share an image
        const int gpc = pok ? gp : 0;
        const int img = gpc / g.HW, p = gpc - img * g.HW;
        const size_t voff = (((size_t)img * g.Cin + (size_t)grp * g.K + half) * g.HW + p) * sizeof(float);
        const unsigned ooff = (unsigned)((((size_t)img * e.ctot + cbase + chunk * 32 + 4 * half) * (size_t)e.hw + p) * sizeof(float));
        const size_t orow = (size_t)e.hw * sizeof(float);
        const int mlh = chunk * 32 + 4 * half;                // local row of accumulator register 0

        float resv[16][NSUB];
        if (e.residual) {
            const char* rb = reinterpret_cast<const char*>(e.residual);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = (r & 3) + 8 * (r >> 2);
#pragma unroll
                for (int s2 = 0; s2 < NSUB; ++s2) resv[r][s2] = 0.f;
                if (mlh + dr < mrem) vec_load<NSUB>(rb + dr * orow + ooff, resv[r]);
            }
        }

        const char* xb = reinterpret_cast<const char*>(x) + voff;
        floatx16 acc[NSUB];
#pragma unroll
        for (int s2 = 0; s2 < NSUB; ++s2)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s2][r] = 0.f;

        // K loop, fully unrolled and branch-free: group gi = k-steps 4*gi .. 4*gi+3
        constexpr int RING = NG < 6 ? NG : 6;
        float b[RING][4][NSUB];
        auto load_group = [&](float (&bb)[4][NSUB], int gi) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) vec_load<NSUB>(xb + (size_t)(8 * gi + 2 * kk) * rowbytes, bb[kk]);
        };
#pragma unroll
        for (int i = 0; i < RING; ++i) load_group(b[i], i);
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int s2 = 0; s2 < NSUB; ++s2)
                    acc[s2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * gi + kk], b[gi % RING][kk][s2], acc[s2], 0, 0, 0);
            if (gi + RING < NG) load_group(b[gi % RING], gi + RING);
            __builtin_amdgcn_sched_barrier(0);          // keep the refill here (the scheduler would sink it to its use)
        }

        // ---- epilogue.  Sub-tile s, register r: row = (r & 3) + 8 * (r >> 2) + 4 * half, pixel gp + s.
        float rr[3][NSUB];
        if (e.reinf_r) {
            const float* rp = e.reinf_r + (size_t)img * 3 * e.hw + p;
            vec_load<NSUB>(rp, rr[0]);
            vec_load<NSUB>(rp + e.hw, rr[1]);
            vec_load<NSUB>(rp + 2 * (size_t)e.hw, rr[2]);
        }
        const float* gate = e.gate ? e.gate + (size_t)img * e.ctot + cbase : nullptr;
        char* ob = reinterpret_cast<char*>(out);
        const char* pb = reinterpret_cast<const char*>(e.pre_add);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int mlb = mlh + 8 * rg;                                    // first of 4 consecutive local rows
            const float4 sc4 = *reinterpret_cast<const float4*>(rowc + mlb);
            const float4 sh4 = *reinterpret_cast<const float4*>(rowc + g.MB + mlb);
            const float4 al4 = *reinterpret_cast<const float4*>(rowc + 2 * g.MB + mlb);
            float4 w04 = make_float4(0.f, 0.f, 0.f, 0.f), w14 = w04, w24 = w04;
            if (e.reinf_r) {
                w04 = *reinterpret_cast<const float4*>(rowc + 3 * g.MB + mlb);
                w14 = *reinterpret_cast<const float4*>(rowc + 4 * g.MB + mlb);
                w24 = *reinterpret_cast<const float4*>(rowc + 5 * g.MB + mlb);
            }
            const float scv[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, shv[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
            const float alv[4] = {al4.x, al4.y, al4.z, al4.w};
            const float w0v[4] = {w04.x, w04.y, w04.z, w04.w}, w1v[4] = {w14.x, w14.y, w14.z, w14.w};
            const float w2v[4] = {w24.x, w24.y, w24.z, w24.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = rg * 4 + q, ml = mlb + q;
                const size_t rowoff = (size_t)(q + 8 * rg) * orow;           // uniform
                if (ml < mrem && pok) {
                    float v[NSUB];
#pragma unroll
                    for (int s2 = 0; s2 < NSUB; ++s2) v[s2] = acc[s2][r];
                    if (e.raw) vec_store<NSUB>(reinterpret_cast<char*>(e.raw) + rowoff + ooff, v);      // (ctot == Cout, coff == 0: same offset)
                    if (pb) {
                        float pa[NSUB];
                        vec_load<NSUB>(pb + rowoff + ooff, pa);
#pragma unroll
                        for (int s2 = 0; s2 < NSUB; ++s2) v[s2] += pa[s2];
                    }
                    const float gv = gate ? gate[ml] : 1.f;
#pragma unroll
                    for (int s2 = 0; s2 < NSUB; ++s2) {
                        float t2 = fmaf(v[s2], scv[q], shv[q]);
                        if (e.reinf_r) t2 += w0v[q] * rr[0][s2] + w1v[q] * rr[1][s2] + w2v[q] * rr[2][s2];
                        if (e.residual) t2 += resv[r][s2];
                        if (e.alpha) t2 = t2 > 0.f ? t2 : alv[q] * t2;
                        v[s2] = t2 * gv;
                    }
                    vec_store<NSUB>(ob + rowoff + ooff, v);
                }
            }
            asm volatile("" ::: "memory");   // one row quad at a time: keeps hipcc from hoisting all 16 rows' work
        }
    }
}

// ------------------------------------------------------------------ MFMA kernel, tile-pipelined (small feature maps)
// At 18x30 / 36x60 a launch is a single round of workgroups that all stage, all multiply and all store at the same
// time, and a wave's time is a chain of memory round trips.  This form removes round trips instead of hiding them:
// K / 8 is a compile-time constant, so the K loop is straight-line code and the s_waitcnt counters the compiler emits
// are exact (a ring of B groups really stays in flight); the ring runs ACROSS pixel tiles -- the last groups of tile t
// are refilled with the first groups of tile t+1, which then land during tile t's epilogue -- and a workgroup walks
// several tiles with one weight staging.  Weights are read from LDS per MFMA (immediate offsets).
template <int NSUB, int NG>
__global__ __launch_bounds__(256, 3) void conv1x1_pipe_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              PwGeom g, Epi e, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* rowc = smem;                                                    // [ROWC][MB]
    float* At = smem + (size_t)g.MB * ROWC;                                // [MB][KS]
    int bid = blockIdx.x;
    const int pg = bid % g.pgroups;  bid /= g.pgroups;
    const int mb = bid % g.mblocks;
    const int grp = bid / g.mblocks;
    const int tid = threadIdx.x;
    const int m0 = mb * g.MB;
    const int cbase = e.coff + grp * g.M + m0;   // absolute destination channel of local row 0
    const int wave = tid >> 6, lane = tid & 63;
    const int WP = 4 / g.WM;
    const int chunk = wave % g.WM, wp = wave / g.WM;
    const int li = lane & 31, half = lane >> 5;
    const int mrem = g.M - m0;
    const int total_px = g.N * g.HW;
    const size_t rowbytes = (size_t)g.HW * sizeof(float);
    const size_t orow = (size_t)e.hw * sizeof(float);
    const int mlh = chunk * 32 + 4 * half;
    const bool active = chunk < g.mc_total;

    constexpr int RING = (NG % 4 == 0) ? 4 : ((NG % 3 == 0) ? 3 : (NG < 4 ? NG : 2));
    static_assert(NG % RING == 0, "ring slots must map onto the same groups in every tile");
    float b[RING][4][NSUB];
    float resv[16][NSUB];
    auto tile_px = [&](int t, int& gp, bool& pok, int& img, int& p) {
        const int ptile = (pg * g.TPW + t) * WP + wp;
        gp = (ptile * 32 + li) * NSUB;
        pok = gp < total_px;
        const int gpc = pok ? gp : 0;
        img = gpc / g.HW;  p = gpc - img * g.HW;
        return t < g.TPW && ptile < g.ptiles;
    };
    auto x_base = [&](int img, int p) {
        return reinterpret_cast<const char*>(x) + (((size_t)img * g.Cin + (size_t)grp * g.K + half) * g.HW + p) * sizeof(float);
    };
    auto load_group = [&](const char* xb, float (&bb)[4][NSUB], int gi) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) vec_load<NSUB>(xb + (size_t)(8 * gi + 2 * kk) * rowbytes, bb[kk]);
    };
    auto load_residual = [&](unsigned ooff) {
        const char* rb = reinterpret_cast<const char*>(e.residual);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
#pragma unroll
            for (int s2 = 0; s2 < NSUB; ++s2) resv[r][s2] = 0.f;
            if (mlh + dr < mrem) vec_load<NSUB>(rb + dr * orow + ooff, resv[r]);
        }
    };

    unsigned long long st0 = 0, st1 = 0, st2 = 0;
    if (g.stamps) st0 = __builtin_amdgcn_s_memrealtime();
    // Order of the first memory operations (vector loads return in order, so what a wave waits for includes everything it
    // issued earlier): the workgroup's weights first (L2 hits, needed before the barrier), then the first tile's B ring, and
    // the residual rows only AFTER the barrier -- they are not needed before the epilogue and fly during the K loop.  With the
    // residual first (round 1) the barrier waited for all of it: 7 us of the 28 us of the 512->512 expansion at 18x30
    // (s_memrealtime stamps), 2 us with this order.
    int gp = 0, img = 0, p = 0;  bool pok = false;
    bool have = active && tile_px(0, gp, pok, img, p);
    const char* xb = x_base(img, p);
    unsigned ooff = (unsigned)((((size_t)img * e.ctot + cbase + mlh) * (size_t)e.hw + p) * sizeof(float));
    {   // weights (coalesced 16-byte loads, all in flight) and per-row constants -> LDS
        const int kv = g.K >> 2, total = g.AR * kv;
        const float* wg = w + ((size_t)grp * g.M + m0) * g.K;
        constexpr int UL = 8;
        for (int base = 0; base < total; base += 256 * UL) {
            float4 v[UL];
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                const int i = base + u * 256 + tid;
                const int m = i / kv, k = (i - m * kv) << 2;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < total && m < mrem) v[u] = *reinterpret_cast<const float4*>(wg + (size_t)m * g.K + k);
            }
            if (base == 0 && have) {
#pragma unroll
                for (int i = 0; i < RING; ++i) load_group(xb, b[i], i);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                const int i = base + u * 256 + tid;
                if (i < total) {
                    const int m = i / kv, k = (i - m * kv) << 2;
                    float* d = At + m * g.KS + k;
                    d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
                }
            }
        }
        for (int m = tid; m < g.MB; m += 256) {
            EpiCh c = {1.f, 0.f, 1.f, 0.f, 0.f, 0.f};
            if (m0 + m < g.M) c = epi_channel(e, cbase + m);
            rowc[m] = c.scale; rowc[g.MB + m] = c.shift; rowc[2 * g.MB + m] = c.alpha;
            rowc[3 * g.MB + m] = c.rw0; rowc[4 * g.MB + m] = c.rw1; rowc[5 * g.MB + m] = c.rw2;
        }
    }
    __syncthreads();
    if (!have) return;                                        // (no barrier follows)
    if (e.residual) load_residual(ooff);
    if (g.stamps) st1 = __builtin_amdgcn_s_memrealtime();
    // rows beyond the staged ones (M < 32: the 16-row decoder projections) read the last staged row; their results are
    // computed and dropped (stores are masked by ml < mrem)
    const float* ar = At + (size_t)min(chunk * 32 + li, g.AR - 1) * g.KS + half;

    for (int t = 0; have; ++t) {
        // the tile after this one (its first RING groups replace this tile's last ones in the ring); when there is
        // none the same addresses are loaded again and dropped, so the loop body stays branch-free
        int gpn, imgn, pn;  bool pokn;
        const bool have_next = tile_px(t + 1, gpn, pokn, imgn, pn);
        const char* xbn = have_next ? x_base(imgn, pn) : xb;

        floatx16 acc[NSUB];
#pragma unroll
        for (int s2 = 0; s2 < NSUB; ++s2)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s2][r] = 0.f;
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const float a = ar[2 * (4 * gi + kk)];
#pragma unroll
                for (int s2 = 0; s2 < NSUB; ++s2)
                    acc[s2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[gi % RING][kk][s2], acc[s2], 0, 0, 0);
            }
            if (gi + RING < NG) load_group(xb, b[gi % RING], gi + RING);
            else load_group(xbn, b[gi % RING], gi + RING - NG);
            // without this the machine scheduler sinks every refill down to its first use (shorter live ranges) and
            // the ring degenerates into load -> s_waitcnt vmcnt(0) -> mfma
            __builtin_amdgcn_sched_barrier(0);
        }

        if (g.stamps && t == 0) { asm volatile("s_nop 0" :: "v"(acc[0][0])); st2 = __builtin_amdgcn_s_memrealtime(); }
        // ---- epilogue.  Sub-tile s, register r: row = (r & 3) + 8 * (r >> 2) + 4 * half, pixel gp + s.
        float rr[3][NSUB];
        if (e.reinf_r) {
            const float* rp = e.reinf_r + (size_t)img * 3 * e.hw + p;
            vec_load<NSUB>(rp, rr[0]);
            vec_load<NSUB>(rp + e.hw, rr[1]);
            vec_load<NSUB>(rp + 2 * (size_t)e.hw, rr[2]);
        }
        const float* gate = e.gate ? e.gate + (size_t)img * e.ctot + cbase : nullptr;
        char* ob = reinterpret_cast<char*>(out);
        const char* pb = reinterpret_cast<const char*>(e.pre_add);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int mlb = mlh + 8 * rg;
            const float4 sc4 = *reinterpret_cast<const float4*>(rowc + mlb);
            const float4 sh4 = *reinterpret_cast<const float4*>(rowc + g.MB + mlb);
            const float4 al4 = *reinterpret_cast<const float4*>(rowc + 2 * g.MB + mlb);
            float4 w04 = make_float4(0.f, 0.f, 0.f, 0.f), w14 = w04, w24 = w04;
            if (e.reinf_r) {
                w04 = *reinterpret_cast<const float4*>(rowc + 3 * g.MB + mlb);
                w14 = *reinterpret_cast<const float4*>(rowc + 4 * g.MB + mlb);
                w24 = *reinterpret_cast<const float4*>(rowc + 5 * g.MB + mlb);
            }
            const float scv[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, shv[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
            const float alv[4] = {al4.x, al4.y, al4.z, al4.w};
            const float w0v[4] = {w04.x, w04.y, w04.z, w04.w}, w1v[4] = {w14.x, w14.y, w14.z, w14.w};
            const float w2v[4] = {w24.x, w24.y, w24.z, w24.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = rg * 4 + q, ml = mlb + q;
                const size_t rowoff = (size_t)(q + 8 * rg) * orow;
                if (ml < mrem && pok) {
                    float v[NSUB];
#pragma unroll
                    for (int s2 = 0; s2 < NSUB; ++s2) v[s2] = acc[s2][r];
                    if (e.raw) vec_store<NSUB>(reinterpret_cast<char*>(e.raw) + rowoff + ooff, v);      // (ctot == Cout, coff == 0: same offset)
                    if (pb) {
                        float pa[NSUB];
                        vec_load<NSUB>(pb + rowoff + ooff, pa);
#pragma unroll
                        for (int s2 = 0; s2 < NSUB; ++s2) v[s2] += pa[s2];
                    }
                    const float gv = gate ? gate[ml] : 1.f;
#pragma unroll
                    for (int s2 = 0; s2 < NSUB; ++s2) {
                        float t2 = fmaf(v[s2], scv[q], shv[q]);
                        if (e.reinf_r) t2 += w0v[q] * rr[0][s2] + w1v[q] * rr[1][s2] + w2v[q] * rr[2][s2];
                        if (e.residual) t2 += resv[r][s2];
                        if (e.alpha) t2 = t2 > 0.f ? t2 : alv[q] * t2;
                        v[s2] = t2 * gv;
                    }
                    vec_store<NSUB>(ob + rowoff + ooff, v);
                }
            }
            asm volatile("" ::: "memory");
        }
        // advance to the next tile (its B groups are already in flight); its residual rows are requested now
        have = have_next;
        gp = gpn; pok = pokn; img = imgn; p = pn; xb = xbn;
        ooff = (unsigned)((((size_t)img * e.ctot + cbase + mlh) * (size_t)e.hw + p) * sizeof(float));
        if (have && e.residual) load_residual(ooff);
    }
    if (g.stamps && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* d = g.stamps + (size_t)blockIdx.x * 4;
        d[0] = st0; d[1] = st1; d[2] = st2; d[3] = __builtin_amdgcn_s_memrealtime();
    }
}

// ------------------------------------------------------------------ small-K VALU path
struct PwSmall {
    int N, Cin, Cout, G, K, M, HW, Q;   // Q = ceil(HW / 4) pixel quads per plane
    int mtiles;                          // ceil(M / MT)
};

template <int MT>
__global__ __launch_bounds__(256) void conv1x1_valu_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           PwSmall g, Epi e, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [G][M][K]
    const int nw = g.G * g.M * g.K;
    for (int i = threadIdx.x; i < nw; i += 256) wl[i] = w[i];
    __syncthreads();
    // blockIdx.y/z = (image, group, row tile) -- uniform splits on the scalar unit; blockIdx.x = pixel quads
    int slab = blockIdx.z * gridDim.y + blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (slab >= g.N * g.G * g.mtiles || q >= g.Q) return;
    const int mt = slab % g.mtiles;  slab /= g.mtiles;
    const int grp = slab % g.G;
    const int img = slab / g.G;
    const int p0 = q * 4;
    const bool v4 = (g.HW & 3) == 0;
    const float* xg = x + ((size_t)img * g.Cin + (size_t)grp * g.K) * (size_t)g.HW + p0;
    float acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[m][j] = 0.f;
    const float* wrow = wl + ((size_t)grp * g.M + mt * MT) * g.K;
    for (int k = 0; k < g.K; ++k) {
        float xv[4];
        if (v4) {
            const float4 t = *reinterpret_cast<const float4*>(xg + (size_t)k * g.HW);
            xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[j] = (p0 + j < g.HW) ? xg[(size_t)k * g.HW + j] : 0.f;
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float wv = (mt * MT + m < g.M) ? wrow[m * g.K + k] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[m][j] = fmaf(wv, xv[j], acc[m][j]);
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int ml = mt * MT + m;
        if (ml >= g.M) break;
        const int cabs = e.coff + grp * g.M + ml;
        const EpiCh ec = epi_channel(e, cabs);
        float* dst = out + epi_offset(e, img, cabs, p0);
        if (e.raw) {
            float* rd = e.raw + epi_offset(e, img, cabs, p0);
            if (v4) store_out4(rd, make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]));
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (p0 + j < g.HW) rd[j] = acc[m][j];
            }
        }
        if (v4) {
            store_out4(dst, epi_apply4(e, ec, acc[m], img, cabs, p0));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (p0 + j < g.HW) dst[j] = epi_apply(e, ec, acc[m][j], img, cabs, p0 + j);
        }
    }
}

// Thin projections on large maps (the decoder's 32 -> 16 and 16 -> 13 at 144x240): at most 16 output channels per group and a
// short K.  The 32-row MFMA tile is half empty there and its staging dominates (57 us for 106 MB); on the vector unit the whole
// job is 0.6 GFLOP -- nothing -- so the kernel is a pure stream: one float4 of pixels per lane and input channel, the weight
// column w[0..15][k] broadcast from LDS as four 16-byte reads, 64 FMAs.  UNR input planes are requested ahead of their use.
template <int MT, int UNR>
__global__ __launch_bounds__(256) void conv1x1_thin_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           PwSmall g, Epi e, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [G][K][MT], rows past M zero
    const int nw = g.G * g.K * MT;
    for (int i = threadIdx.x; i < nw; i += 256) {
        const int m = i % MT, k = (i / MT) % g.K, grp = i / (MT * g.K);
        wl[i] = m < g.M ? w[((size_t)grp * g.M + m) * g.K + k] : 0.f;
    }
    __syncthreads();
    int slab = blockIdx.z * gridDim.y + blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (slab >= g.N * g.G || q >= g.Q) return;
    const int grp = slab % g.G;
    const int img = slab / g.G;
    const int p0 = q * 4;
    const float* xg = x + ((size_t)img * g.Cin + (size_t)grp * g.K) * (size_t)g.HW + p0;
    float acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[m][j] = 0.f;
    const float* wk = wl + (size_t)grp * g.K * MT;
    for (int k0 = 0; k0 < g.K; k0 += UNR) {                       // K % UNR == 0 (launcher)
        float4 xv[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) xv[u] = *reinterpret_cast<const float4*>(xg + (size_t)(k0 + u) * g.HW);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
#pragma unroll
            for (int m4 = 0; m4 < MT; m4 += 4) {
                const float4 wv = *reinterpret_cast<const float4*>(wk + (k0 + u) * MT + m4);
                const float ww[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
                for (int mm = 0; mm < 4; ++mm) {
                    acc[m4 + mm][0] = fmaf(ww[mm], xv[u].x, acc[m4 + mm][0]);
                    acc[m4 + mm][1] = fmaf(ww[mm], xv[u].y, acc[m4 + mm][1]);
                    acc[m4 + mm][2] = fmaf(ww[mm], xv[u].z, acc[m4 + mm][2]);
                    acc[m4 + mm][3] = fmaf(ww[mm], xv[u].w, acc[m4 + mm][3]);
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        if (m < g.M) {                                            // uniform; no `break`, so the loop unrolls and acc stays in registers
            const int cabs = e.coff + grp * g.M + m;
            const EpiCh ec = epi_channel(e, cabs);
            if (e.raw) store_out4(e.raw + epi_offset(e, img, cabs, p0), make_float4(acc[m][0], acc[m][1], acc[m][2], acc[m][3]));
            store_out4(out + epi_offset(e, img, cabs, p0), epi_apply4(e, ec, acc[m], img, cabs, p0));
        }
    }
}

// 0 = launched.  Shapes: M <= 16 per group, K in 8..64 (multiple of 8), planes a multiple of 4 pixels, >= 400 workgroups.
static int launch_thin_try(const float* x, const float* w, int N, int Cin, int Cout, int groups, int HW, const Epi& e, float* out,
                           hipStream_t s) {
    static const int enabled = MSPL_TUNE_INT("MSPL_PW_THIN", 1);
    const int K = Cin / groups, M = Cout / groups;
    // measured (tools/bench_ops.py, bs 16): 32 -> 16 at 144x240 26.1 -> 23.0 us, 16 -> 13 18.3 -> 15.6, 128 -> 32 g4 at 72x120
    // 23.9 -> 18.9; with fewer than ~2 workgroups per CU (48 -> 16 at 72x120: 144) the MFMA kernel's finer tiles win (13 vs 20 us)
    if (!enabled || M > 16 || K < 8 || K > 64 || (K & 7) != 0 || (HW & 3) != 0 ||
        (int64_t)N * groups * ceil_div(HW / 4, 256) < 400)
        return 1;
    if (((uintptr_t)x & 15) || ((uintptr_t)out & 15) || ((uintptr_t)e.raw & 15) || ((uintptr_t)e.pre_add & 15) ||
        ((uintptr_t)e.residual & 15) || ((uintptr_t)e.reinf_r & 15) || ((uintptr_t)e.gate & 15))
        return 1;
    PwSmall g;
    g.N = N; g.Cin = Cin; g.Cout = Cout; g.G = groups; g.K = K; g.M = M; g.HW = HW;
    g.Q = HW / 4;
    g.mtiles = 1;
    const int mt = M <= 8 ? 8 : 16;
    const size_t lds = (size_t)groups * K * mt * sizeof(float);
    if (lds > 48 * 1024) return 1;
    const int64_t slabs = (int64_t)N * groups;
    if (slabs >= 65535ll * 65535ll) return 1;
    const int gy = slabs < 65535 ? (int)slabs : 65535;
    dim3 grid((unsigned)ceil_div(g.Q, 256), (unsigned)gy, (unsigned)ceil_div64(slabs, gy)), blk(256);
    if (mt == 8) hipLaunchKernelGGL((conv1x1_thin_kernel<8, 8>), grid, blk, lds, s, x, w, g, e, out);
    else hipLaunchKernelGGL((conv1x1_thin_kernel<16, 8>), grid, blk, lds, s, x, w, g, e, out);
    return 0;
}

// Launch of the tile-pipelined kernel (K % 32 == 0 or K in {16, 24}, aligned weights, small feature maps).
static int launch_pipe(const float* x, const float* w, PwGeom g, const Epi& e, const mspl_epilogue_t* ep, float* out,
                       hipStream_t s) {
    g.KS = g.K | 1;
    int mbr = ((g.M + 31) / 32) * 32;
    if (mbr > 128) mbr = 128;
    static const int dbg_plds = MSPL_TUNE_INT("MSPL_PW_PIPE_LDS", 40);     // KiB of weights per workgroup
    while (mbr > 32 && (size_t)mbr * (g.KS + ROWC) * 4 > (size_t)dbg_plds * 1024) mbr -= 32;
    if (mbr == 96) mbr = 64;
    g.MB = mbr;
    g.mblocks = ceil_div(g.M, mbr);
    g.mc_total = mbr / 32;
    g.WM = g.mc_total >= 3 ? 4 : g.mc_total;
    const int wp = 4 / g.WM;
    auto al = [](const void* p, int a) { return p == nullptr || (((uintptr_t)p) & (a - 1)) == 0; };
    auto ok = [&](int ns) {
        const int a = ns * 4;
        return g.HW % ns == 0 && al(x, a) && al(out, a) && (!ep || (al(ep->pre_add, a) && al(ep->residual, a) && al(ep->reinf_r, a)));
    };
    int nsub = ok(2) ? 2 : 1;
    static const int dbg_nsub = MSPL_TUNE_INT("MSPL_PW_NSUB", 0);
    static const int dbg_tpw = MSPL_TUNE_INT("MSPL_PW_TPW", 0);
    if ((dbg_nsub == 1 || dbg_nsub == 2) && ok(dbg_nsub)) nsub = dbg_nsub;
    g.ptiles = (int)ceil_div64((int64_t)g.N * g.HW, 32 * nsub);
    // workgroups: at most one resident round (3 per CU), each walking TPW tiles per wave
    int tpw = 1;
    static const int pipe_wgs = MSPL_TUNE_INT("MSPL_PW_PIPE_WGS", 768);
    while ((int64_t)g.G * g.mblocks * ceil_div(g.ptiles, wp * tpw) > pipe_wgs) ++tpw;
    if (dbg_tpw) tpw = dbg_tpw;
    g.TPW = tpw;
    g.pgroups = ceil_div(g.ptiles, wp * tpw);
    const int64_t blocks = (int64_t)g.G * g.mblocks * g.pgroups;
    MSPL_REQUIRE(blocks < (1ll << 31), MSPL_ERR_BAD_SHAPE, "conv1x1: grid too large");
    const dim3 grid((unsigned)blocks), blk(256);
    {
        static unsigned long long* stamp_buf = nullptr;
        static const int dbg_stamp = MSPL_STAMP_ENV("MSPL_PW_STAMP");
        if (dbg_stamp && !stamp_buf) (void)hipMalloc(&stamp_buf, (size_t)5 * 65536 * sizeof(unsigned long long));
        g.stamps = (dbg_stamp && blocks <= 65536) ? stamp_buf : nullptr;
    }
    g.AR = g.M < g.MB ? g.M : g.MB;
    const size_t lds = ((size_t)g.MB * ROWC + (size_t)g.AR * g.KS) * sizeof(float);
    MSPL_REQUIRE(lds <= 96 * 1024, MSPL_ERR_UNSUPPORTED, "conv1x1: weight tile of %zu B exceeds LDS", lds);
    if (lds > 64 * 1024) {
#define MSPL_PIPE_ATTR(NG) do { (void)hipFuncSetAttribute((const void*)conv1x1_pipe_kernel<2, NG>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); \
                                (void)hipFuncSetAttribute((const void*)conv1x1_pipe_kernel<1, NG>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); } while (0)
        static std::once_flag attr_once;
        std::call_once(attr_once, [] { MSPL_PIPE_ATTR(8); MSPL_PIPE_ATTR(12); MSPL_PIPE_ATTR(16); (void)hipGetLastError(); });
#undef MSPL_PIPE_ATTR
    }
#define MSPL_PIPE(NG) do { if (nsub == 2) hipLaunchKernelGGL((conv1x1_pipe_kernel<2, NG>), grid, blk, lds, s, x, w, g, e, out); \
                           else hipLaunchKernelGGL((conv1x1_pipe_kernel<1, NG>), grid, blk, lds, s, x, w, g, e, out); } while (0)
    switch (g.K >> 3) {
        case 2: MSPL_PIPE(2); break;
        case 3: MSPL_PIPE(3); break;
        case 4: MSPL_PIPE(4); break;
        case 6: MSPL_PIPE(6); break;
        case 8: MSPL_PIPE(8); break;
        case 12: MSPL_PIPE(12); break;
        case 16: MSPL_PIPE(16); break;
        case 32: MSPL_PIPE(32); break;
        default: MSPL_PIPE(64); break;
    }
#undef MSPL_PIPE
    MSPL_CHECK_LAUNCH("conv1x1(tile-pipelined)");
    if (g.stamps) {   // debug only: synchronous dump of the phase timeline (100 MHz ticks)
        (void)hipDeviceSynchronize();
        static unsigned long long host[4 * 65536];
        (void)hipMemcpy(host, g.stamps, (size_t)blocks * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t3 = 0; double a = 0, b = 0, c = 0, late = 0;
        for (int64_t i = 0; i < blocks; ++i) { if (host[4*i] < t0) t0 = host[4*i]; if (host[4*i+3] > t3) t3 = host[4*i+3]; a += host[4*i+1]-host[4*i]; b += host[4*i+2]-host[4*i+1]; c += host[4*i+3]-host[4*i+2]; }
        for (int64_t i = 0; i < blocks; ++i) late += host[4*i] - t0;
        fprintf(stderr, "[pw-pipe stamp] K=%d M=%d HW=%d blocks=%lld tpw=%d span=%.2fus  avg: start-delay=%.2fus fetch+stage+barrier=%.2fus kloop=%.2fus epilogue(+more tiles)=%.2fus\n",
                g.K, g.M, g.HW, (long long)blocks, g.TPW, (t3-t0)/100.0, late/blocks/100.0, a/blocks/100.0, b/blocks/100.0, c/blocks/100.0);
    }
    return MSPL_OK;
}

// Launch of the register-resident-weights kernel (K % 8 == 0, K <= 128, aligned weights).
static int launch_areg(const float* x, const float* w, PwGeom g, const Epi& e, const mspl_epilogue_t* ep, float* out,
                       hipStream_t s) {
    static const int dbg_astage = MSPL_TUNE_INT("MSPL_PW_ASTAGE", -1);
    g.astage = dbg_astage >= 0 ? dbg_astage : !(g.K <= 32 && (int64_t)g.N * g.HW >= 100000);
    g.KS = g.K | 1;
    int mbr = ((g.M + 31) / 32) * 32;
    if (mbr > 128) mbr = 128;
    while (g.astage && mbr > 32 && (size_t)mbr * (g.KS + ROWC) * 4 > 40 * 1024) mbr -= 32;
    if (mbr == 96) mbr = 64;                                   // waves along M are 1, 2 or 4
    g.MB = mbr;
    g.mblocks = ceil_div(g.M, mbr);
    g.mc_total = mbr / 32;
    g.WM = g.mc_total >= 3 ? 4 : g.mc_total;
    const int wp = 4 / g.WM;
    auto al = [](const void* p, int a) { return p == nullptr || (((uintptr_t)p) & (a - 1)) == 0; };
    auto ok = [&](int ns) {
        const int a = ns * 4;
        return g.HW % ns == 0 && al(x, a) && al(out, a) && (!ep || (al(ep->pre_add, a) && al(ep->residual, a) && al(ep->reinf_r, a)));
    };
    // wave tiles = groups x 32-row chunks x pixel tiles; aim at >= 4 per SIMD (1024 SIMDs) for balance
    auto tiles_of = [&](int ns) { return (int64_t)g.G * g.mblocks * g.mc_total * ceil_div64((int64_t)g.N * g.HW, 32 * ns); };
    int nsub = 1;
    if (ok(2) && tiles_of(2) >= 4096) nsub = 2;
    static const int dbg_nsub = MSPL_TUNE_INT("MSPL_PW_NSUB", 0);   // tuning override
    static const int dbg_tpw = MSPL_TUNE_INT("MSPL_PW_TPW", 0);
    if ((dbg_nsub == 1 || dbg_nsub == 2) && ok(dbg_nsub)) nsub = dbg_nsub;
    g.ptiles = (int)ceil_div64((int64_t)g.N * g.HW, 32 * nsub);
    int tpw = 1;
    while (tpw < 8 && tiles_of(nsub) / (tpw * 2) >= 4096) tpw *= 2;      // keep >= 4 waves per SIMD
    if (dbg_tpw) tpw = dbg_tpw;
    g.TPW = tpw;
    g.pgroups = ceil_div(g.ptiles, wp * tpw);
    const int64_t blocks = (int64_t)g.G * g.mblocks * g.pgroups;
    MSPL_REQUIRE(blocks < (1ll << 31), MSPL_ERR_BAD_SHAPE, "conv1x1: grid too large");
    const dim3 grid((unsigned)blocks), blk(256);
    const size_t lds = (size_t)g.MB * (ROWC + (g.astage ? g.KS : 0)) * sizeof(float);
#define MSPL_AREG(NG) do { if (nsub == 2) hipLaunchKernelGGL((conv1x1_areg_kernel<2, NG>), grid, blk, lds, s, x, w, g, e, out); \
                           else hipLaunchKernelGGL((conv1x1_areg_kernel<1, NG>), grid, blk, lds, s, x, w, g, e, out); } while (0)
    switch (g.K >> 3) {
        case 2: MSPL_AREG(2); break;
        case 3: MSPL_AREG(3); break;
        case 4: MSPL_AREG(4); break;
        case 6: MSPL_AREG(6); break;
        case 8: MSPL_AREG(8); break;
        case 12: MSPL_AREG(12); break;
        default: MSPL_AREG(16); break;
    }
#undef MSPL_AREG
    MSPL_CHECK_LAUNCH("conv1x1(register weights)");
    return MSPL_OK;
}

}  // namespace mspl

using namespace mspl;

static int launch_small(const float* x, const float* w, int N, int Cin, int Cout, int groups, int HW,
                        const Epi& e, float* out, hipStream_t s) {
    PwSmall g;
    g.N = N; g.Cin = Cin; g.Cout = Cout; g.G = groups; g.K = Cin / groups; g.M = Cout / groups; g.HW = HW;
    g.Q = ceil_div(HW, 4);
    const int mt = g.M <= 2 ? 2 : (g.M <= 4 ? 4 : 8);
    g.mtiles = ceil_div(g.M, mt);
    const size_t lds = (size_t)Cout * g.K * sizeof(float);
    MSPL_REQUIRE(lds <= 48 * 1024, MSPL_ERR_UNSUPPORTED, "conv1x1(small-K): weight block %zu B exceeds LDS", lds);
    const int64_t slabs = (int64_t)N * groups * g.mtiles;
    MSPL_REQUIRE(slabs < 65535ll * 65535ll, MSPL_ERR_BAD_SHAPE, "conv1x1: grid too large");
    const int gy = slabs < 65535 ? (int)slabs : 65535;
    dim3 grid((unsigned)ceil_div(g.Q, 256), (unsigned)gy, (unsigned)ceil_div64(slabs, gy)), blk(256);
    if (mt == 2) hipLaunchKernelGGL(conv1x1_valu_kernel<2>, grid, blk, lds, s, x, w, g, e, out);
    else if (mt == 4) hipLaunchKernelGGL(conv1x1_valu_kernel<4>, grid, blk, lds, s, x, w, g, e, out);
    else hipLaunchKernelGGL(conv1x1_valu_kernel<8>, grid, blk, lds, s, x, w, g, e, out);
    MSPL_CHECK_LAUNCH("conv1x1(small-K)");
    return MSPL_OK;
}

extern "C" int mspl_conv1x1_fwd(const float* x, const float* w, int32_t N, int32_t Cin, int32_t Cout,
                                int32_t groups, int32_t HW, const mspl_epilogue_t* ep, float* out,
                                void* stream) {
    MSPL_REQUIRE(x && w && out, MSPL_ERR_NULL_POINTER, "conv1x1: null pointer");
    MSPL_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && groups > 0 && HW > 0, MSPL_ERR_BAD_SHAPE,
                 "conv1x1: bad shape N=%d Cin=%d Cout=%d groups=%d HW=%d", N, Cin, Cout, groups, HW);
    MSPL_REQUIRE(Cin % groups == 0 && Cout % groups == 0, MSPL_ERR_BAD_SHAPE,
                 "conv1x1: channels (%d,%d) not divisible by groups %d", Cin, Cout, groups);
    if (int rc = check_epi(ep, Cout, "conv1x1", true)) return rc;
    const Epi e = make_epi(ep, Cout, HW);
    hipStream_t s = (hipStream_t)stream;
    PwGeom g;
    memset(&g, 0, sizeof(g));
    g.N = N; g.Cin = Cin; g.Cout = Cout; g.G = groups; g.K = Cin / groups; g.M = Cout / groups; g.HW = HW;
    if (launch_thin_try(x, w, N, Cin, Cout, groups, HW, e, out, s) == 0) {
        MSPL_CHECK_LAUNCH("conv1x1(thin)");
        return MSPL_OK;
    }
    if ((g.K < 16 || (g.K & 1)) && (size_t)Cout * g.K * 4 <= 48 * 1024) return launch_small(x, w, N, Cin, Cout, groups, HW, e, out, s);
    MSPL_REQUIRE((g.K & 1) == 0, MSPL_ERR_UNSUPPORTED, "conv1x1: odd K=%d per group with %d output channels is not supported", g.K, Cout);

    if (conv1x1_splitk_try(x, w, N, Cin, Cout, groups, HW, e, out, s) == 0) {      // few output channels per group: split-K form
        MSPL_CHECK_LAUNCH("conv1x1(split-K)");
        return MSPL_OK;
    }
    const int K32 = (g.K + 31) & ~31;
    g.KS = K32 | 1;
    g.vecw = ((g.K & 3) == 0) && ((((uintptr_t)w) & 15) == 0);
    static const int dbg_areg = MSPL_TUNE_INT("MSPL_PW_AREG", 1);
    const int ng8 = g.K >> 3;
    const bool ng_ok = ng8 == 2 || ng8 == 3 || ng8 == 4 || ng8 == 6 || ng8 == 8 || ng8 == 12 || ng8 == 16;
    // K = 256 / 512 (the decoder's first projection) only when the staged rows fit LDS (few output channels)
    const bool ng_big = (ng8 == 32 || ng8 == 64) && (size_t)((g.M < 32 ? g.M : 32) * ((g.K | 1)) + 32 * ROWC) * 4 <= 64 * 1024 && g.M <= 32;
    // measured (tools/bench_ops.py): the register-weights kernel wins for short K on large maps (no staging, no barrier);
    // for K >= 64 its 150-185 VGPRs leave 2 workgroups per CU and the 18x30 / 36x60 grids then need a second round
    const bool areg_shape = dbg_areg == 2 || (g.K <= 32 && (int64_t)N * HW >= 100000);
    static const int dbg_pipe = MSPL_TUNE_INT("MSPL_PW_PIPE", 1);
    const bool pipe_shape = true;      // measured faster than the LDS-ring and register-weights forms on every eligible shape
    if (dbg_pipe && pipe_shape && (g.K & 7) == 0 && (ng_ok || ng_big) && g.vecw &&
        (size_t)N * Cin * HW * sizeof(float) < (1ull << 32) && (size_t)N * e.ctot * HW * sizeof(float) < (1ull << 32))
        return launch_pipe(x, w, g, e, ep, out, s);
    if (dbg_areg && areg_shape && (g.K & 7) == 0 && ng_ok && g.vecw &&
        (size_t)N * Cin * HW * sizeof(float) < (1ull << 32) && (size_t)N * e.ctot * HW * sizeof(float) < (1ull << 32))
        return launch_areg(x, w, g, e, ep, out, s);
    const int rowf = ROWC;
    MSPL_REQUIRE((size_t)N * Cin * HW * sizeof(float) < (1ull << 32) && (size_t)N * e.ctot * HW * sizeof(float) < (1ull << 32),
                 MSPL_ERR_UNSUPPORTED, "conv1x1: tensor exceeds the 32-bit byte offsets of the matrix-core kernel");
    const size_t lds_cap = 96 * 1024;       // hard cap (of the 160 KiB per CU)
    static const int dbg_lds = MSPL_TUNE_INT("MSPL_PW_LDS", 0);
    const size_t lds_want = dbg_lds ? (size_t)dbg_lds * 1024 : 40 * 1024;      // preferred: several workgroups per CU
    MSPL_REQUIRE((size_t)32 * (g.KS + rowf) * 4 <= lds_cap, MSPL_ERR_UNSUPPORTED,
                 "conv1x1: K=%d per group exceeds the LDS weight tile", g.K);
    int mb = ((g.M + 31) / 32) * 32;
    if (mb > 128) mb = 128;
    while (mb > 32 && (size_t)mb * (g.KS + rowf) * 4 > lds_want) mb -= 32;
    g.MB = mb;
    g.mblocks = ceil_div(g.M, mb);
    g.mc_total = mb / 32;
    g.WM = g.mc_total >= 3 ? 4 : g.mc_total;     // one 32-row chunk per wave; chunks share B loads through L1
    const int wp = 4 / g.WM;
    const size_t lds = (size_t)g.MB * (g.KS + rowf) * sizeof(float);
    static std::once_flag attr_once;   // dynamic LDS above 64 KiB needs the opt-in (once per process, thread-safe)
    {
        constexpr int cap = 96 * 1024;            // = lds_cap
        std::call_once(attr_once, [] {
            (void)hipFuncSetAttribute((const void*)conv1x1_mfma_kernel<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            (void)hipFuncSetAttribute((const void*)conv1x1_mfma_kernel<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            (void)hipFuncSetAttribute((const void*)conv1x1_mfma_kernel<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            (void)hipFuncSetAttribute((const void*)conv1x1_mfma_kernel<1, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            (void)hipFuncSetAttribute((const void*)conv1x1_mfma_kernel<2, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            (void)hipFuncSetAttribute((const void*)conv1x1_mfma_kernel<1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            (void)hipFuncSetAttribute((const void*)conv1x1_mfma_kernel<2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            (void)hipGetLastError();
        });
    }
    // pixels per lane: the widest vector the shape/alignment allows that still yields enough waves to fill
    // 256 CUs x 4 SIMDs (small feature maps prefer narrower tiles over idle CUs)
    auto al = [](const void* p, int a) { return p == nullptr || (((uintptr_t)p) & (a - 1)) == 0; };
    auto ok = [&](int ns) {
        const int a = ns * 4;
        return HW % ns == 0 && al(x, a) && al(out, a) && (!ep || (al(ep->pre_add, a) && al(ep->residual, a) && al(ep->reinf_r, a)));
    };
    auto waves_of = [&](int ns) { return (int64_t)groups * g.mblocks * g.mc_total * ceil_div64((int64_t)N * HW, 32 * ns); };
    int nsub = 1;
    if (ok(4) && waves_of(4) >= 2048) nsub = 4;
    else if (ok(2) && waves_of(2) >= 2048) nsub = 2;
    else if (ok(2) && !ok(1)) nsub = 2;
    static const int dbg_nsub = MSPL_TUNE_INT("MSPL_PW_NSUB", 0);   // tuning override
    static const int dbg_tpw = MSPL_TUNE_INT("MSPL_PW_TPW", 0);
    if (dbg_nsub && ok(dbg_nsub)) nsub = dbg_nsub;
    g.ptiles = (int)ceil_div64((int64_t)N * HW, 32 * nsub);
    const int64_t wave_tiles = (int64_t)groups * g.mblocks * g.ptiles;
    int tpw = 1;
    while (tpw < 4 && wave_tiles / (wp * tpw * 2) >= 2048) tpw *= 2;
    if (dbg_tpw) tpw = dbg_tpw;
    g.TPW = tpw;
    g.pgroups = ceil_div(g.ptiles, wp * tpw);
    const int64_t blocks = (int64_t)groups * g.mblocks * g.pgroups;
    MSPL_REQUIRE(blocks < (1ll << 31), MSPL_ERR_BAD_SHAPE, "conv1x1: grid too large");
    static unsigned long long* stamp_buf = nullptr;
    static const int dbg_stamp = MSPL_STAMP_ENV("MSPL_PW_STAMP");
    if (dbg_stamp && !stamp_buf) (void)hipMalloc(&stamp_buf, (size_t)5 * 65536 * sizeof(unsigned long long));
    g.stamps = (dbg_stamp && blocks <= 65536) ? stamp_buf : nullptr;
    dim3 grid((unsigned)blocks), blk(256);
    // ring depth = B groups (8 k-values each) in flight per wave (deeper rings measured slower: they cost occupancy)
    int ring = 4;
    static const int dbg_ring = MSPL_TUNE_INT("MSPL_PW_RING", 0);
    if (nsub < 4 && (dbg_ring == 4 || dbg_ring == 8 || dbg_ring == 16)) ring = dbg_ring;
#define MSPL_PW(NS, RG) hipLaunchKernelGGL((conv1x1_mfma_kernel<NS, RG>), grid, blk, lds, s, x, w, g, e, out)
    if (nsub == 4) MSPL_PW(4, 4);
    else if (nsub == 2) { if (ring == 16) MSPL_PW(2, 16); else if (ring == 8) MSPL_PW(2, 8); else MSPL_PW(2, 4); }
    else { if (ring == 16) MSPL_PW(1, 16); else if (ring == 8) MSPL_PW(1, 8); else MSPL_PW(1, 4); }
#undef MSPL_PW
    if (g.stamps) {   // debug only: synchronous dump of the phase timeline (100 MHz ticks)
        (void)hipDeviceSynchronize();
        static unsigned long long host[4 * 65536];
        (void)hipMemcpy(host, g.stamps, (size_t)blocks * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t3 = 0; double a = 0, b = 0, c = 0, late = 0;
        for (int64_t i = 0; i < blocks; ++i) { if (host[4*i] < t0) t0 = host[4*i]; if (host[4*i+3] > t3) t3 = host[4*i+3]; a += host[4*i+1]-host[4*i]; b += host[4*i+2]-host[4*i+1]; c += host[4*i+3]-host[4*i+2]; }
        for (int64_t i = 0; i < blocks; ++i) late += host[4*i] - t0;
        fprintf(stderr, "[pw stamp] K=%d M=%d HW=%d blocks=%lld span=%.2fus  avg: start-delay=%.2fus fetch+stage+barrier=%.2fus kloop=%.2fus epilogue(+more tiles)=%.2fus\n",
                g.K, g.M, g.HW, (long long)blocks, (t3-t0)/100.0, late/blocks/100.0, a/blocks/100.0, b/blocks/100.0, c/blocks/100.0);
        static unsigned long long hwid[65536];
        (void)hipMemcpy(hwid, g.stamps + (size_t)4 * 65536, (size_t)blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        static int cnt[8 * 4096];
        memset(cnt, 0, sizeof(cnt));
        int used = 0, mx = 0;
        for (int64_t i = 0; i < blocks; ++i) {
            const unsigned hw = (unsigned)hwid[i], xcc = (unsigned)(hwid[i] >> 32) & 7;
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            const int key = (int)(xcc * 4096 + se * 256 + sh * 16 + cu);
            if (cnt[key]++ == 0) ++used;
            if (cnt[key] > mx) mx = cnt[key];
        }
        fprintf(stderr, "[pw stamp] placement: %d distinct CUs used, max %d workgroups on one CU (hw_id sample 0x%x xcc %u)\n", used, mx,
                (unsigned)hwid[0], (unsigned)(hwid[0] >> 32));
    }
    MSPL_CHECK_LAUNCH("conv1x1");
    return MSPL_OK;
}
