// K1 / K3 -- grouped 1x1 convolution with a fused epilogue.
//
// Reference arithmetic: nn.Conv2d(k=1, groups=g, bias=False) followed by BatchNorm (eval), optional
// DownSampler reinforcement / EESP residual, PReLU: nn_layers/eesp.py:36,55,67,77-93,117-120,142;
// nn_layers/efficient_pyramid_pool.py:22,31; nn_layers/espnet_utils.py:8-37,62-89.
//
// Per (image, group) this is C[M x P] = Wg[M x K] . X[K x P] with P = H*W contiguous in NCHW.
//
// K >= 16: fp32 matrix cores.  v_mfma_f32_32x32x2_f32 is exact fp32 (a k-ordered fmaf chain) at the
//   vector-FMA rate, takes one VGPR per operand and leaves the VALU free for the epilogue.  A wave owns a
//   32-pixel tile and MCW 32-row chunks of M.  The B operand X[k][p..p+31] is loaded straight from HBM
//   (two coalesced 128-byte rows per wave-instruction) in software-pipelined groups of 16 k-steps; the A
//   operand (weights) and the per-row epilogue constants sit in LDS (odd row stride: conflict-free
//   ds_read_b32).  Every input element is fetched from HBM once per workgroup; stores are 128-byte rows.
// K < 16: the matrix tile would be mostly padding, so a VALU kernel streams 4 pixels per lane with
//   16-byte loads/stores and LDS-broadcast weights.
#include "common.hpp"

namespace mspl {

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct PwGeom {
    int N, Cin, Cout, G, K, M, HW;
    int KS;        // LDS row stride of A (odd, >= K rounded up to 32)
    int MB;        // rows of M handled per workgroup (multiple of 32)
    int mblocks;   // ceil(M / MB)
    int mc_total;  // 32-row chunks inside one workgroup's MB
    int WM;        // waves along M (1,2,4); WP = 4 / WM waves along pixels
    int TPW;       // pixel tiles per wave (sequential)
    int ptiles;    // ceil(HW / 32)
    int pgroups;   // ceil(ptiles / (WP * TPW))
};

// Per-row epilogue constants staged in LDS (one ds_read per row instead of six global loads).
struct RowEpi { float scale, shift, alpha, rw0, rw1, rw2; };

template <int MCW>
__global__ __launch_bounds__(256, (MCW == 1 ? 4 : (MCW == 2 ? 3 : 2))) void conv1x1_mfma_kernel(const float* __restrict__ x,
                                                                               const float* __restrict__ w,
                                                                               PwGeom g, Epi e,
                                                                               float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    RowEpi* rowc = reinterpret_cast<RowEpi*>(smem);                        // [MB]
    float* Ct = smem + (size_t)g.MB * (sizeof(RowEpi) / sizeof(float));    // [4 waves][32][33]
    float* At = Ct + 4 * 32 * 33;                                          // [MB][KS], zero padded to K32
    int bid = blockIdx.x;
    const int pg = bid % g.pgroups;  bid /= g.pgroups;
    const int mb = bid % g.mblocks;  bid /= g.mblocks;
    const int grp = bid % g.G;
    const int img = bid / g.G;
    const int tid = threadIdx.x;
    const int m0 = mb * g.MB;
    const int K32 = (g.K + 31) & ~31;
    const int cbase = e.coff + grp * g.M + m0;   // absolute destination channel of local row 0

    const float* wg = w + ((size_t)grp * g.M + m0) * g.K;
    for (int i = tid; i < g.MB * K32; i += 256) {
        const int m = i / K32, k = i - m * K32;
        At[m * g.KS + k] = (m0 + m < g.M && k < g.K) ? wg[(size_t)m * g.K + k] : 0.f;
    }
    for (int m = tid; m < g.MB; m += 256) {
        RowEpi r = {1.f, 0.f, 1.f, 0.f, 0.f, 0.f};
        if (m0 + m < g.M) {
            const EpiCh c = epi_channel(e, cbase + m);
            r.scale = c.scale; r.shift = c.shift; r.alpha = c.alpha; r.rw0 = c.rw0; r.rw1 = c.rw1; r.rw2 = c.rw2;
        }
        rowc[m] = r;
    }
    __syncthreads();

    const int wave = tid >> 6, lane = tid & 63;
    const int WP = 4 / g.WM;
    const int wm = wave % g.WM, wp = wave / g.WM;
    const int li = lane & 31, half = lane >> 5;
    const float* xg = x + ((size_t)img * g.Cin + (size_t)grp * g.K) * (size_t)g.HW;
    const size_t obase = ((size_t)img * e.ctot + cbase) * (size_t)e.hw;      // uniform
    const float* gate = e.gate ? e.gate + (size_t)img * e.ctot + cbase : nullptr;
    const int mrem = g.M - m0;                                  // valid local rows
    float* ct = Ct + wave * (32 * 33);                          // this wave's 32x32 staging tile (padded)
    const size_t row2 = 2 * (size_t)g.HW;

    for (int t = 0; t < g.TPW; ++t) {
        const int ptile = (pg * g.TPW + t) * WP + wp;
        if (ptile >= g.ptiles) break;            // wave-uniform
        const int p = ptile * 32 + li;
        const bool pok = p < g.HW;
        const int pc = pok ? p : g.HW - 1;       // clamped: loads stay in bounds, results masked

        // B rows kb0 + 2*kk + half, kk = 0..15, addressed as (uniform row base) + (32-bit lane byte offset)
        // so that each load is `global_load_dword v, v_off, s[base]` with no per-load 64-bit VGPR address.
        const unsigned vb0 = (unsigned)pc * 4u;
        const unsigned vb = (unsigned)(half * g.HW + pc) * 4u;
        auto load_group = [&](float (&b)[16], int kb0) {
            if (kb0 + 32 <= g.K) {               // uniform: full group, no guards
                const char* xr = reinterpret_cast<const char*>(xg + (size_t)kb0 * g.HW);
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    b[kk] = *reinterpret_cast<const float*>(xr + vb);
                    xr += row2 * sizeof(float);
                }
            } else {                             // tail group: rows >= K contribute zeros
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    const int kr = kb0 + 2 * kk;
                    float v = 0.f;
                    if (kr < g.K) {              // uniform
                        const char* xr = reinterpret_cast<const char*>(xg + (size_t)kr * g.HW);
                        const bool both = kr + 1 < g.K;
                        v = *reinterpret_cast<const float*>(xr + (both ? vb : vb0));
                        if (!both && half) v = 0.f;
                    }
                    b[kk] = v;
                }
            }
        };
        floatx16 acc[MCW];
#pragma unroll
        for (int j = 0; j < MCW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

        float bcur[16], bnext[16];
        load_group(bcur, 0);
#pragma unroll 1
        for (int kb0 = 0; kb0 < g.K; kb0 += 32) {
            const bool more = kb0 + 32 < g.K;
            if (more) load_group(bnext, kb0 + 32);   // in flight under this group's MFMAs
#pragma unroll
            for (int j = 0; j < MCW; ++j) {
                const int chunk = wm + j * g.WM;
                if (chunk < g.mc_total) {            // wave-uniform
                    const float* arow = At + (size_t)(chunk * 32 + li) * g.KS + kb0 + half;
#pragma unroll
                    for (int kk = 0; kk < 16; ++kk)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[2 * kk], bcur[kk], acc[j], 0, 0, 0);
                }
            }
            if (more) {
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) bcur[kk] = bnext[kk];
            }
        }

        // ---- epilogue.  C/D layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
        // The tile goes through a padded LDS image so that a ROLLED loop can walk rows in order (two
        // 128-byte row segments per wave-instruction) -- an unrolled register epilogue makes hipcc hoist
        // every row's loads and masks at once and spill.
        float r0 = 0.f, r1 = 0.f, r2 = 0.f;
        if (e.reinf_r) {
            const float* rr = e.reinf_r + (size_t)img * 3 * e.hw + pc;
            r0 = rr[0]; r1 = rr[e.hw]; r2 = rr[2 * (size_t)e.hw];
        }
#pragma unroll
        for (int j = 0; j < MCW; ++j) {
            const int chunk = wm + j * g.WM;
            if (chunk >= g.mc_total) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) ct[((r & 3) + 8 * (r >> 2) + 4 * half) * 33 + li] = acc[j][r];
            // same-wave LDS write -> read: ds ops of one wave complete in order; the waitcnt is the compiler's
            const int rows = min(32, mrem - chunk * 32);
#pragma unroll 2
            for (int rr = half; rr < rows; rr += 2) {
                const int ml = chunk * 32 + rr;
                const RowEpi c = rowc[ml];
                const size_t off = obase + (size_t)ml * e.hw + pc;
                float v = ct[rr * 33 + li];
                if (e.pre_add) v += e.pre_add[off];
                v = fmaf(v, c.scale, c.shift);
                if (e.reinf_r) v += c.rw0 * r0 + c.rw1 * r1 + c.rw2 * r2;
                if (e.residual) v += e.residual[off];
                if (e.alpha) v = v > 0.f ? v : c.alpha * v;
                if (gate) v *= gate[ml];
                if (pok) out[off] = v;
            }
        }
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
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)g.N * g.G * g.mtiles * g.Q;
    if (idx >= total) return;
    const int q = (int)(idx % g.Q);  idx /= g.Q;
    const int mt = (int)(idx % g.mtiles);  idx /= g.mtiles;
    const int grp = (int)(idx % g.G);
    const int img = (int)(idx / g.G);
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
        if (v4) {
            float4 o;
            o.x = epi_apply(e, ec, acc[m][0], img, cabs, p0);
            o.y = epi_apply(e, ec, acc[m][1], img, cabs, p0 + 1);
            o.z = epi_apply(e, ec, acc[m][2], img, cabs, p0 + 2);
            o.w = epi_apply(e, ec, acc[m][3], img, cabs, p0 + 3);
            *reinterpret_cast<float4*>(dst) = o;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (p0 + j < g.HW) dst[j] = epi_apply(e, ec, acc[m][j], img, cabs, p0 + j);
        }
    }
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
    const int64_t total = (int64_t)N * groups * g.mtiles * g.Q;
    MSPL_REQUIRE(ceil_div64(total, 256) < (1ll << 31), MSPL_ERR_BAD_SHAPE, "conv1x1: grid too large");
    dim3 grid((unsigned)ceil_div64(total, 256)), blk(256);
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
    if (int rc = check_epi(ep, Cout, "conv1x1")) return rc;
    const Epi e = make_epi(ep, Cout, HW);
    hipStream_t s = (hipStream_t)stream;
    PwGeom g;
    g.N = N; g.Cin = Cin; g.Cout = Cout; g.G = groups; g.K = Cin / groups; g.M = Cout / groups; g.HW = HW;
    if (g.K < 16 && (size_t)Cout * g.K * 4 <= 48 * 1024) return launch_small(x, w, N, Cin, Cout, groups, HW, e, out, s);

    const int K32 = (g.K + 31) & ~31;
    g.KS = K32 | 1;
    const size_t lds_cap = 96 * 1024;  // of the 160 KiB per CU
    const int row_floats = g.KS + (int)(sizeof(RowEpi) / sizeof(float));
    const size_t ct_bytes = 4 * 32 * 33 * sizeof(float);
    MSPL_REQUIRE((size_t)32 * row_floats * 4 + ct_bytes <= lds_cap, MSPL_ERR_UNSUPPORTED,
                 "conv1x1: K=%d per group exceeds the LDS weight tile", g.K);
    int mb = ((g.M + 31) / 32) * 32;
    if (mb > 128) mb = 128;
    while (mb > 32 && (size_t)mb * row_floats * 4 + ct_bytes > lds_cap) mb -= 32;
    g.MB = mb;
    g.mblocks = ceil_div(g.M, mb);
    g.mc_total = mb / 32;
    g.ptiles = ceil_div(HW, 32);
    const int64_t wave_tiles = (int64_t)N * groups * g.mblocks * g.ptiles;
    // few tiles: split M over the 4 waves (more waves in flight, B re-read through L1);
    // many tiles: each wave keeps its B registers for all M chunks.
    int wm = 1;
    if (wave_tiles < 4096) wm = g.mc_total >= 4 ? 4 : (g.mc_total >= 2 ? 2 : 1);
    else if (wave_tiles < 16384 && g.mc_total >= 2) wm = 2;
    g.WM = wm;
    const int mcw = ceil_div(g.mc_total, wm);
    const int wp = 4 / wm;
    int tpw = 1;
    while (tpw < 8 && wave_tiles / (wp * tpw * 2) >= 4096) tpw *= 2;
    g.TPW = tpw;
    g.pgroups = ceil_div(g.ptiles, wp * tpw);
    const int64_t blocks = (int64_t)N * groups * g.mblocks * g.pgroups;
    MSPL_REQUIRE(blocks < (1ll << 31), MSPL_ERR_BAD_SHAPE, "conv1x1: grid too large");
    const size_t lds = (size_t)g.MB * row_floats * sizeof(float) + ct_bytes;
    dim3 grid((unsigned)blocks), blk(256);
    static bool attr_done = false;   // dynamic LDS above 64 KiB needs the opt-in (idempotent, no sync)
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv1x1_mfma_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap);
        (void)hipFuncSetAttribute((const void*)conv1x1_mfma_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap);
        (void)hipFuncSetAttribute((const void*)conv1x1_mfma_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap);
        (void)hipFuncSetAttribute((const void*)conv1x1_mfma_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap);
        (void)hipGetLastError();
        attr_done = true;
    }
    switch (mcw) {
        case 1: hipLaunchKernelGGL(conv1x1_mfma_kernel<1>, grid, blk, lds, s, x, w, g, e, out); break;
        case 2: hipLaunchKernelGGL(conv1x1_mfma_kernel<2>, grid, blk, lds, s, x, w, g, e, out); break;
        case 3: hipLaunchKernelGGL(conv1x1_mfma_kernel<3>, grid, blk, lds, s, x, w, g, e, out); break;
        default: hipLaunchKernelGGL(conv1x1_mfma_kernel<4>, grid, blk, lds, s, x, w, g, e, out); break;
    }
    MSPL_CHECK_LAUNCH("conv1x1");
    return MSPL_OK;
}
