// K6, third form -- the fused EfficientPyrPool body as a register-streaming kernel (no LDS traffic in the inner loop).
//
// Same arithmetic as pyrpool_sep.hip (nn_layers/efficient_pyramid_pool.py:36-61; the up-sampled branches collapsed into
// position-dependent separable stencils on x), same order of floating-point operations, different machine mapping.  The
// stencil form in pyrpool_sep.hip turned out to be LDS-bound: per 1x4 strip it issues ~107 LDS reads (x patch, per-row and
// per-column coefficient tables, low-resolution maps, branch tiles for the merge convolution) for ~430 FMAs, and a CU's four
// SIMDs share one LDS pipe (PMC / MSPL_PYR_STOP timings).  Here nothing in the inner loop goes through LDS:
//   * a WAVE owns one (image, plane), one block of columns and one segment of rows; lane = PXL adjacent columns; the wave
//     walks DOWN the rows.  The 5-row window of x and the 3-row window of the five branch maps live in registers and slide.
//   * row-dependent coefficients (A tables) are wave-uniform: computed once per wave into a small LDS table, read back with
//     broadcast reads (3-6 per row); column-dependent coefficients (C tables, bilinear sources of the low-resolution maps) are
//     per-lane constants computed once per wave and kept in registers; weights / BN constants of the plane are wave-uniform
//     (scalar loads).
//   * the merge convolution needs the branch values of the two neighbouring columns: v_mov_dpp wave_shr / wave_shl from the
//     neighbouring lanes; lanes 0 and 63 are halo lanes (they compute branch values but write no output), so waves never
//     exchange data and there is no barrier in the kernel.
//   * x rows and low-resolution rows are read straight from global memory (coalesced 4/8-byte accesses, L1/L2 hits for the
//     overlaps), the next row's loads are issued one row ahead.
#include <stdlib.h>

#include <algorithm>

#include "common.hpp"
#include "pyr_stencil.hpp"

namespace mspl {

constexpr int P3_MAXB = 5;
// Branch rows per trip of the row loop.  One row per trip pays ~60 of its ~312 vector instructions for moving the sliding windows at
// the back edge (hipcc does not unroll the loop by itself: `#pragma unroll 2` is refused); with two steps in the loop body the windows
// are renamed between them: rocprofv3 inside the label pass, alternating libraries, 43.4 -> 41.9 us per launch at 144x240 / 72x120,
// 24.2 -> 22.9 us for the one-pixel form.  Three and five steps (222 / 256 registers, no spill) are SLOWER (47.6 / 49.0 us).
#ifndef P3_STEPS
#define P3_STEPS 2
#endif
constexpr int P3_SEGMAX = 19;       // (SEG + 2) * 3 table entries are computed by the 64 lanes in one step

struct Pyr3Geom {
    int N, P, h, w;
    int hs[P3_MAXB], ws[P3_MAXB];
    float sh[P3_MAXB], sw[P3_MAXB];  // up branches: x -> U grid; down branches: E grid -> output grid
    int SEG, nseg, ncb, CBW;         // output rows per segment, segments per plane, column blocks, output columns per block
    unsigned total;                  // waves
};

// One up branch for the lane's PXL columns at one row.  xw[rr][j]: image row br-2+rr, column px0-2+j.
// Arow: this row's [ky][8] coefficients in LDS (same address for every lane: broadcast read).  Same evaluation order as
// up_branch_strip (pyrpool_sep.hip): per kernel row ky the vertical stencil first, then the horizontal one, accumulated over ky.
template <int T, int PXL>
__device__ __forceinline__ void p3_up_branch(const float (&xw)[5][PXL + 4], const float* __restrict__ Arow,
                                             const float (&C)[3][PXL][5], float (&b)[PXL]) {
    constexpr int R0 = (5 - T) / 2;
    constexpr int NCOL = PXL + T - 1;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        float a[5];
        {
            const float4 t4 = *reinterpret_cast<const float4*>(Arow + ky * 8);
            a[0] = t4.x; a[1] = t4.y; a[2] = t4.z; a[3] = t4.w;
            a[4] = T > 4 ? Arow[ky * 8 + 4] : 0.f;
        }
        float cs[NCOL];
#pragma unroll
        for (int s = 0; s < NCOL; ++s) {
            float v = a[0] * xw[R0][R0 + s];
#pragma unroll
            for (int rr = 1; rr < T; ++rr) v = fmaf(a[rr], xw[R0 + rr][R0 + s], v);
            cs[s] = v;
        }
#pragma unroll
        for (int j = 0; j < PXL; ++j) {
            float v = b[j];
#pragma unroll
            for (int s = 0; s < T; ++s) v = fmaf(C[ky][j][s], cs[j + s], v);
            b[j] = v;
        }
    }
}

template <int PXL> struct P3Vec;
template <> struct P3Vec<1> { typedef float T; };
template <> struct P3Vec<2> { typedef float2 T; };

// Branch layout handled here (the reference's scales sorted descending: 2.0, 1.5, 1.0, 0.5, 0.1):
//   branch 0: up (T0 taps), 1: up (T1 taps), 2: same, 3: down, 4: down.
// Every read-only operand is a `const float* __restrict__` kernel argument of its own: only then does hipcc turn the
// wave-uniform reads of the plane's weights into scalar loads (pointers inside the geometry struct are not known to be
// unaliased with `out`, and the per-row constants came back as 13 global_load_dwordx4 per lane and row).
// TRAIN: the training forward also keeps the branch values before merge_layer.0 (zcat, torch.cat order) and the bare merge convolution
// (mraw): what the backward kernels of pyrpool_train.hip read.
template <int T0, int T1, int PXL, bool TRAIN>
__global__ __launch_bounds__(256, 2) void pyrpool_stream_kernel(const float* __restrict__ x, const float* __restrict__ sw0,
                                                                const float* __restrict__ sw1, const float* __restrict__ sw2,
                                                                const float* __restrict__ de3, const float* __restrict__ de4,
                                                                const float* __restrict__ br_scale, const float* __restrict__ br_shift,
                                                                const float* __restrict__ br_alpha, const float* __restrict__ merge_w,
                                                                const float* __restrict__ ep_scale,
                                                                const float* __restrict__ ep_shift,
                                                                const float* __restrict__ ep_alpha, int ep_ctot, int ep_coff,
                                                                Pyr3Geom g, float* __restrict__ out, float* __restrict__ zcat,
                                                                float* __restrict__ mraw) {
    // The four waves of a workgroup take four consecutive planes of the SAME column block and row segment: the row tables (A) and
    // the channel-independent column tables (G) are identical for them, so they are computed once per workgroup (a quarter of
    // the column tasks per wave) into LDS; after the one barrier every wave folds its own plane's 3x3 weights into its
    // per-lane C tables and never touches LDS tables of columns again.
    __shared__ __attribute__((aligned(16))) float At[2][(P3_SEGMAX + 2) * 24];        // [up branch][row][ky][8]
    __shared__ float Gt[2][3][PXL][5][64];                                             // [up branch][kx][column][tap][lane]
    // merge_layer.2's 45 weights of the wave's plane, [branch][12] (9 used): read back with three broadcast 16-byte LDS reads per
    // branch and row.  As 45 scalar registers (rounds 1-2) they did not fit next to the other plane constants: 44 spilled SGPRs,
    // 32 v_readlane per row step (412 -> 382 vector instructions per row; same box, three launches in flight: 16 400 -> 16 700 images/s).
    __shared__ __attribute__((aligned(16))) float Wm[4][5][12];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned bid = blockIdx.x;
    const int sgi = bid % g.nseg;  bid /= g.nseg;
    const int cb = bid % g.ncb;  bid /= g.ncb;
    const int plane = (int)bid * 4 + wave;                                 // (N * P) % 4 == 0 (checked by the launcher)
    const int n = plane / g.P, c = plane - n * g.P;
    const int h = g.h, w = g.w;
    const int ys = sgi * g.SEG, ye = min(ys + g.SEG, h);
    const int px0 = cb * g.CBW + (lane - 1) * PXL;                      // lane 0 / 63: halo columns of the block
    const bool writer = lane >= 1 && lane <= 62 && px0 < w;

    // ---- per-workgroup setup
    // (1) row tables of the two up branches (wave 0: branch 0, wave 1: branch 1): entry = lane -> (row index, ky), rows ys-1 .. ye
    if (wave < 2) {
        const int ri = lane / 3, ky = lane - 3 * ri;
        if (ri < g.SEG + 2) {
            float acc[5];
            if (wave == 0) p3_coeffs<T0>(ys - 1 + ri, ky, h, g.hs[0], g.sh[0], acc);
            else p3_coeffs<T1>(ys - 1 + ri, ky, h, g.hs[1], g.sh[1], acc);
            float* d = &At[wave][ri * 24 + ky * 8];
            d[0] = acc[0]; d[1] = acc[1]; d[2] = acc[2]; d[3] = acc[3]; d[4] = acc[4];
        }
    }
    // (2) column tables G_kx[column][tap]: 6 * PXL tasks (branch, kx, column) per lane, task t goes to wave t % 4
#pragma unroll
    for (int t = 0; t < 6 * PXL; ++t) {
        if ((t & 3) == wave) {                                          // uniform
            const int ub = t / (3 * PXL), r = t - ub * 3 * PXL, kx = r / PXL, j = r - kx * PXL;     // compile-time
            float acc[5];
            if (ub == 0) p3_coeffs<T0>(px0 + j, kx, w, g.ws[0], g.sw[0], acc);
            else p3_coeffs<T1>(px0 + j, kx, w, g.ws[1], g.sw[1], acc);
#pragma unroll
            for (int q = 0; q < 5; ++q) Gt[ub][kx][j][q][lane] = acc[q];
        }
    }
    __syncthreads();
    // (2b) per-lane C_ky[j][s] = sum_kx w[ky][kx] * G_kx[j][s] with this wave's plane's weights
    float C0[3][PXL][5], C1[3][PXL][5];
    {
        const float* w0p = sw0 + (size_t)c * 9;
        const float* w1p = sw1 + (size_t)c * 9;
#pragma unroll
        for (int j = 0; j < PXL; ++j)
#pragma unroll
            for (int s2 = 0; s2 < 5; ++s2) {
                const float a0 = Gt[0][0][j][s2][lane], a1 = Gt[0][1][j][s2][lane], a2 = Gt[0][2][j][s2][lane];
                const float b0 = Gt[1][0][j][s2][lane], b1 = Gt[1][1][j][s2][lane], b2 = Gt[1][2][j][s2][lane];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    C0[ky][j][s2] = fmaf(w0p[ky * 3 + 2], a2, fmaf(w0p[ky * 3 + 1], a1, w0p[ky * 3] * a0));
                    C1[ky][j][s2] = fmaf(w1p[ky * 3 + 2], b2, fmaf(w1p[ky * 3 + 1], b1, w1p[ky * 3] * b0));
                }
            }
    }
    // (3) per-lane bilinear column sources of the two low-resolution maps
    unsigned dxa[2][PXL], dxb[2][PXL];  float dw0[2][PXL], dw1[2][PXL];        // byte offsets inside a low-resolution row
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < PXL; ++j) {
            const int px = min(max(px0 + j, 0), w - 1);
            int xa, xb;
            bilinear_src(g.sw[3 + i], px, g.ws[3 + i], xa, xb, dw0[i][j], dw1[i][j]);
            dxa[i][j] = (unsigned)xa * 4u;  dxb[i][j] = (unsigned)xb * 4u;
        }
    // (4) wave-uniform constants of this plane (scalar loads)
    const float* wsame = sw2 + (size_t)c * 9;
    if (lane < 45) Wm[wave][lane / 9][lane % 9] = merge_w[(size_t)c * 45 + lane];      // (a wave's LDS operations execute in order: no barrier)
    const float4* wmq = reinterpret_cast<const float4*>(&Wm[wave][0][0]);
    float bsc[5], bsh[5], bal[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) { bsc[i] = br_scale[i * g.P + c]; bsh[i] = br_shift[i * g.P + c]; bal[i] = br_alpha[i * g.P + c]; }
    const int cabs = ep_coff + c;
    const float esc = ep_scale ? ep_scale[cabs] : 1.f, esh = ep_shift ? ep_shift[cabs] : 0.f, eal = ep_alpha ? ep_alpha[cabs] : 1.f;
    // The plane index is wave-uniform (n, c follow from the wave index); said so explicitly, the plane bases are kernel argument +
    // scalar offset and the row bases of the loop are computed on the scalar unit (412 -> 392 vector instructions per row step).
    const unsigned plane_id = __builtin_amdgcn_readfirstlane((unsigned)(n * g.P + c));
    const float* xpl = x + (size_t)plane_id * (size_t)(h * w);
    const float* e3 = de3 + (size_t)plane_id * (size_t)(g.hs[3] * g.ws[3]);
    const float* e4 = de4 + (size_t)plane_id * (size_t)(g.hs[4] * g.ws[4]);
    float* opl = out + ((size_t)__builtin_amdgcn_readfirstlane((unsigned)(n * ep_ctot + cabs))) * (size_t)(h * w);
    bool colin[PXL];
#pragma unroll
    for (int j = 0; j < PXL; ++j) colin[j] = px0 + j >= 0 && px0 + j < w;

    // x row loader: columns px0-2 .. px0+PXL+1 of row r (zero outside the image).  Buffer loads through a per-ROW descriptor built on
    // the scalar unit (base = the row, records = the row's bytes, or 0 for a row outside the image) + a per-lane byte offset that
    // never changes (columns outside the image: an offset past the records): the range check of the load returns the zeros, so
    // no vector instruction goes into addressing or masking (rounds 2-4: one 64-bit address add per load and one multiply by a
    // 0/1 mask per value -- 12 of the ~500 vector instructions of a row step; 28 with the low-resolution rows below).
    // PXL == 2: px0 is even and so is w (launcher), so the six columns are three 8-byte loads, each inside or outside as a whole.
    constexpr unsigned P3_OOB = 0x7ffffff0u;
    unsigned xoff[PXL + 4];
#pragma unroll
    for (int j = 0; j < PXL + 4; ++j) {
        const int cx = px0 - 2 + j;
        xoff[j] = (cx >= 0 && cx < w) ? (unsigned)cx * 4u : P3_OOB;
    }
    auto load_row = [&](int r, float (&v)[PXL + 4]) {
        const bool rin = r >= 0 && r < h;                                                   // uniform
        const float* row = xpl + (size_t)(rin ? r : 0) * w;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(row), 0, rin ? w * 4 : 0, 0x00020000);
        if (PXL == 2) {
#pragma unroll
            for (int j = 0; j < PXL + 4; j += 2) {
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, xoff[j], 0, 0);
                v[j] = __uint_as_float(t.x);  v[j + 1] = __uint_as_float(t.y);
            }
        } else {
#pragma unroll
            for (int j = 0; j < PXL + 4; ++j) v[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, xoff[j], 0, 0));
        }
    };

    float xw[5][PXL + 4];
    // rows br-2 .. br+2 for the first branch row br = ys - 1: rows ys-3 .. ys+1
#pragma unroll
    for (int rr = 0; rr < 5; ++rr) load_row(ys - 3 + rr, xw[rr]);
    // low-resolution values of the NEXT branch row, requested one row ahead: [map][column][ya/xa, ya/xb, yb/xa, yb/xb]
    float en[2][PXL][4];
    float enwy0[2], enwy1[2];
    auto load_e = [&](int r) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int ya, yb;
            bilinear_src(g.sh[3 + i], min(max(r, 0), h - 1), g.hs[3 + i], ya, yb, enwy0[i], enwy1[i]);       // uniform
            const int rowb = g.ws[3 + i] * 4;
            const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((i == 0 ? e3 : e4) + ya * g.ws[3 + i]), 0, rowb, 0x00020000);
            const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((i == 0 ? e3 : e4) + yb * g.ws[3 + i]), 0, rowb, 0x00020000);
#pragma unroll
            for (int j = 0; j < PXL; ++j) {
                en[i][j][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, dxa[i][j], 0, 0));
                en[i][j][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, dxb[i][j], 0, 0));
                en[i][j][2] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb, dxa[i][j], 0, 0));
                en[i][j][3] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb, dxb[i][j], 0, 0));
            }
        }
    };
    load_e(ys - 1);
    // Merge convolution, accumulated as the branch rows arrive: branch row br contributes kernel row 2 of output row br-1, kernel
    // row 1 of output row br and kernel row 0 of output row br+1.  acc[0]: output row br-1 (complete after this row), acc[1]: row
    // br, acc[2]: row br+1.  (Summation order over (branch, kernel row) differs from pyrpool_sep's branch-major order: fp32
    // rounding-level differences, covered by the tests' tolerance against the oracle.)
    float acc[3][PXL];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int j = 0; j < PXL; ++j) acc[r][j] = 0.f;

    // Two branch rows per trip of the loop: the sliding windows (x rows, low-resolution samples, merge accumulators) are renamed
    // between the two steps instead of moved, the moves are paid once per trip at the back edge.
    auto row_step = [&](const int br) __attribute__((always_inline)) {
        // ---- branch maps of row br
        float bv[5][PXL];
        const bool rowin = br >= 0 && br < h;
        float ecur[2][PXL][4], wy0c[2], wy1c[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            wy0c[i] = enwy0[i]; wy1c[i] = enwy1[i];
#pragma unroll
            for (int j = 0; j < PXL; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) ecur[i][j][q] = en[i][j][q];
        }
        if (br < ye) load_e(br + 1);                       // next row's low-resolution values fly during this row's arithmetic
        float xn[PXL + 4];
        load_row(br + 3, xn);                              // the row that enters the window at the end of this iteration
        __builtin_amdgcn_sched_barrier(0);
        if (rowin) {
            const float* A0 = &At[0][(br - (ys - 1)) * 24];
            const float* A1 = &At[1][(br - (ys - 1)) * 24];
#pragma unroll
            for (int j = 0; j < PXL; ++j) { bv[0][j] = 0.f; bv[1][j] = 0.f; bv[2][j] = 0.f; }
            p3_up_branch<T0, PXL>(xw, A0, C0, bv[0]);
            p3_up_branch<T1, PXL>(xw, A1, C1, bv[1]);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float w0 = wsame[ky * 3], w1 = wsame[ky * 3 + 1], w2 = wsame[ky * 3 + 2];
#pragma unroll
                for (int j = 0; j < PXL; ++j) {
                    bv[2][j] = fmaf(w0, xw[1 + ky][j + 1], bv[2][j]);
                    bv[2][j] = fmaf(w1, xw[1 + ky][j + 2], bv[2][j]);
                    bv[2][j] = fmaf(w2, xw[1 + ky][j + 3], bv[2][j]);
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < PXL; ++j) {
                    const float top = dw0[i][j] * ecur[i][j][0] + dw1[i][j] * ecur[i][j][1];
                    const float bot = dw0[i][j] * ecur[i][j][2] + dw1[i][j] * ecur[i][j][3];
                    bv[3 + i][j] = wy0c[i] * top + wy1c[i] * bot;
                }
            if (TRAIN && writer && br >= ys && br < ye) {          // rows ys-1 and ye belong to the neighbouring segments
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    float* zp = zcat + (((size_t)n * 5 + i) * g.P + c) * (size_t)h * w + (size_t)br * w + px0;
                    if (PXL == 2) *reinterpret_cast<float2*>(zp) = make_float2(bv[i][0], bv[i][PXL - 1]);
                    else zp[0] = bv[i][0];
                }
            }
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < PXL; ++j) {
                    float v = fmaf(bv[i][j], bsc[i], bsh[i]);
                    v = v > 0.f ? v : bal[i] * v;
                    bv[i][j] = colin[j] ? v : 0.f;           // zero outside the image: the merge convolution's padding
                }
        } else {
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < PXL; ++j) bv[i][j] = 0.f;
        }
        // ---- this row's contributions to the three output rows it touches; neighbouring columns come from the neighbouring lanes
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            float bx[PXL + 2];
            bx[0] = p3_from_left(bv[i][PXL - 1]);
#pragma unroll
            for (int j = 0; j < PXL; ++j) bx[1 + j] = bv[i][j];
            bx[PXL + 1] = p3_from_right(bv[i][0]);
            const float4 q0 = wmq[i * 3], q1 = wmq[i * 3 + 1], q2 = wmq[i * 3 + 2];
            const float wv[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
#pragma unroll
            for (int r = 0; r < 3; ++r) {                   // acc[r] <-> output row br-1+r <-> kernel row ky = 2 - r
                const int ky = 2 - r;
                const float w0 = wv[ky * 3], w1 = wv[ky * 3 + 1], w2 = wv[ky * 3 + 2];
#pragma unroll
                for (int j = 0; j < PXL; ++j) {
                    acc[r][j] = fmaf(w0, bx[j], acc[r][j]);
                    acc[r][j] = fmaf(w1, bx[j + 1], acc[r][j]);
                    acc[r][j] = fmaf(w2, bx[j + 2], acc[r][j]);
                }
            }
        }
        // ---- slide the x window (row br+3, requested at the top of this iteration, enters it)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int j = 0; j < PXL + 4; ++j) xw[rr][j] = xw[rr + 1][j];
#pragma unroll
        for (int j = 0; j < PXL + 4; ++j) xw[4][j] = xn[j];
        // ---- output row y = br - 1 is complete
        const int y = br - 1;
        if (y >= ys) {
            float v[PXL];
#pragma unroll
            for (int j = 0; j < PXL; ++j) {
                v[j] = fmaf(acc[0][j], esc, esh);
                v[j] = (ep_alpha && v[j] <= 0.f) ? eal * v[j] : v[j];
            }
            if (writer) {
                float* dst = opl + (size_t)y * w + px0;
                if (PXL == 2) store_out2(dst, make_float2(v[0], v[PXL - 1]));
                else dst[0] = v[0];
                if (TRAIN) {
                    float* rp = mraw + ((size_t)n * g.P + c) * (size_t)h * w + (size_t)y * w + px0;
                    if (PXL == 2) *reinterpret_cast<float2*>(rp) = make_float2(acc[0][0], acc[0][PXL - 1]);
                    else rp[0] = acc[0][0];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < PXL; ++j) { acc[0][j] = acc[1][j]; acc[1][j] = acc[2][j]; acc[2][j] = 0.f; }
    };
    constexpr int STEPS = TRAIN ? 2 : P3_STEPS;            // (the training forward holds more per row: two steps fit its registers)
#pragma unroll 1
    for (int br = ys - 1; br <= ye; br += STEPS) {
        row_step(br);
#pragma unroll
        for (int u = 1; u < STEPS; ++u)
            if (br + u <= ye) row_step(br + u);
    }
}

// Returns MSPL_OK when launched, 1 when the shape is left to the LDS-tiled kernels, < 0 on error.
int pyrpool_stream_try(const float* x, int N, int P, int h, int w, int nb, const int32_t* hs, const int32_t* ws,
                       const float* const* stage_w, const float* const* down_e, const float* br_scale,
                       const float* br_shift, const float* br_alpha, const float* merge_w, const Epi& e, float* out,
                       hipStream_t stream, float* zcat) {
    static const int off = (MSPL_TUNE_INT("MSPL_PYR_STREAM", 1) == 0);
    // maps narrower than ~half a wave leave most lanes idle: the LDS-tiled form is faster there (18x30: 12 vs 17 us)
    static const int min_w = MSPL_TUNE_INT("MSPL_PYR_STREAM_MINW", 40);
    if (off || nb != 5 || w < min_w || ((int64_t)N * P) % 4 != 0) return 1;
    if (e.pre_add || e.residual || e.reinf_r || e.gate) return 1;    // only the scale/shift/PReLU epilogue
    // branch pattern: up, up, same, down, down (strictly)
    for (int i = 0; i < 5; ++i) if (hs[i] <= 0 || ws[i] <= 0) return 1;
    if (!(hs[0] > h && ws[0] > w && hs[1] > h && ws[1] > w)) return 1;
    if (!(hs[2] == h && ws[2] == w)) return 1;
    if (!(hs[3] <= h && ws[3] <= w && hs[4] <= h && ws[4] <= w && (hs[3] < h || ws[3] < w) && (hs[4] < h || ws[4] < w))) return 1;
    if (!stage_w[0] || !stage_w[1] || !stage_w[2] || !down_e[3] || !down_e[4]) return 1;
    int taps[2];
    for (int i = 0; i < 2; ++i) {
        const int R = std::max(p3_stencil_radius(h, hs[i]), p3_stencil_radius(w, ws[i]));
        if (R > 2) return 1;
        taps[i] = R <= 1 ? 3 : 5;
    }
    const int PXL = w <= 62 ? 1 : 2;
    if (PXL == 2 && ((w & 1) || (((uintptr_t)out) & 7) || (((uintptr_t)x) & 7))) return 1;     // 8-byte stores and x row loads
    if (zcat && (!e.raw || e.ctot != P || e.coff != 0)) return 1;      // training forward: un-sliced destination + raw output
    if (zcat && PXL == 2 && ((((uintptr_t)zcat) | ((uintptr_t)e.raw)) & 7)) return 1;
    Pyr3Geom g;
    memset(&g, 0, sizeof(g));
    g.N = N; g.P = P; g.h = h; g.w = w;
    for (int i = 0; i < 5; ++i) {
        g.hs[i] = hs[i]; g.ws[i] = ws[i];
        if (i < 2) { g.sh[i] = bilinear_scale(h, hs[i]); g.sw[i] = bilinear_scale(w, ws[i]); }
        if (i > 2) { g.sh[i] = bilinear_scale(hs[i], h); g.sw[i] = bilinear_scale(ws[i], w); }
    }
    g.CBW = 62 * PXL;
    g.ncb = ceil_div(w, g.CBW);
    // rows per segment: enough waves to put >= 2 on every SIMD when the maps allow, halo rows (2 of SEG + 2 branch rows, 6 of
    // SEG + 6 x rows) kept small otherwise
    static const int dbg_seg = MSPL_TUNE_INT("MSPL_PYR_SEG", 0);
    int seg = std::min(h, P3_SEGMAX);
    while (seg > 6 && (int64_t)N * P * g.ncb * ceil_div(h, seg) < 2048) --seg;
    seg = ceil_div(h, ceil_div(h, seg));
    if (dbg_seg > 0 && dbg_seg <= P3_SEGMAX) seg = std::min(dbg_seg, h);
    g.SEG = seg;
    g.nseg = ceil_div(h, seg);
    const int64_t waves = (int64_t)N * P * g.ncb * g.nseg;
    if (waves >= (1ll << 31)) return 1;
    g.total = (unsigned)waves;
    const dim3 grid((unsigned)(waves / 4)), blk(256);          // workgroup = 4 consecutive planes of one (column block, segment)
#define MSPL_P3_LAUNCH(A, B, L) do { \
        if (zcat) hipLaunchKernelGGL((pyrpool_stream_kernel<A, B, L, true>), grid, blk, 0, stream, x, stage_w[0], stage_w[1], stage_w[2], down_e[3], down_e[4], br_scale, br_shift, br_alpha, merge_w, e.scale, e.shift, e.alpha, e.ctot, e.coff, g, out, zcat, e.raw); \
        else hipLaunchKernelGGL((pyrpool_stream_kernel<A, B, L, false>), grid, blk, 0, stream, x, stage_w[0], stage_w[1], stage_w[2], down_e[3], down_e[4], br_scale, br_shift, br_alpha, merge_w, e.scale, e.shift, e.alpha, e.ctot, e.coff, g, out, (float*)nullptr, (float*)nullptr); } while (0)
    if (PXL == 1) {
        if (taps[0] == 3 && taps[1] == 3) MSPL_P3_LAUNCH(3, 3, 1);
        else if (taps[0] == 3) MSPL_P3_LAUNCH(3, 5, 1);
        else if (taps[1] == 3) MSPL_P3_LAUNCH(5, 3, 1);
        else MSPL_P3_LAUNCH(5, 5, 1);
    } else {
        if (taps[0] == 3 && taps[1] == 3) MSPL_P3_LAUNCH(3, 3, 2);
        else if (taps[0] == 3) MSPL_P3_LAUNCH(3, 5, 2);
        else if (taps[1] == 3) MSPL_P3_LAUNCH(5, 3, 2);
        else MSPL_P3_LAUNCH(5, 5, 2);
    }
#undef MSPL_P3_LAUNCH
    MSPL_CHECK_LAUNCH("pyrpool_fused(streaming form)");
    return MSPL_OK;
}

}  // namespace mspl
