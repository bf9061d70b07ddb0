// Backward of the fused EfficientPyrPool body (the training step's counterpart of pyrpool.hip).
//
// Forward (mspl_pyrpool_fused_train_fwd, pyrpool.hip; nn_layers/efficient_pyramid_pool.py:39-58): per projected plane c
//     t_i = branch_i(x_c)                       i = 0..nb-1        (kept: zcat, torch.cat order)
//     y_i = PReLU(t_i * bs_i + bh_i)            merge_layer.0 (frozen BatchNorm folded + PReLU)
//     m   = sum_i conv3x3(y_i, wm[c, i])        merge_layer.1 (Shuffle) + merge_layer.2 conv    (kept: mraw)
//     out = PReLU(m * ms + mh)                  merge_layer.2 BatchNorm + PReLU
// The unfused backward moved the 5x-wide concatenation through HBM nine times (data gradient of the grouped 3x3, the
// un-shuffle copy, its weight gradient, the BatchNorm/PReLU backward, five slice copies) and the 4x / 2.25x up-sampled
// intermediates of the scale > 1 branches another ~26 times.  Two kernels replace all of that:
//   pyr_merge_bwd_kernel   gy, mraw, zcat -> gt (nb,N,P,h,w) = dL/dt_i, plus every parameter gradient of merge_layer.0 / .2
//   pyr_branch_bwd_kernel  gt_i of the up-sampled / same-size branches + x -> dL/dx (one write, the low-resolution branches'
//                          contributions added in), plus the stage weights' gradients; nothing at up-sampled resolution
//                          ever exists in memory.
#include <stdlib.h>

#include "common.hpp"
#include "pyr_stencil.hpp"

#include <type_traits>

namespace mspl {

// ------------------------------------------------------------------------------------------------ merge backward
constexpr int MB_TH = 16, MB_TW = 64;      // tile: 16 rows x 64 columns; thread = one 1x4 strip
constexpr int MB_GS = 68;                  // LDS row stride of the g_m halo tile (66 columns used)

struct MbGeom {
    int N, P, h, w;
    int tiles_x, tiles_y;
    const float *br_scale, *br_shift, *br_alpha, *br_mean, *br_inv;      // nb*P (merge_layer.0; mean/inv null: plain scale/shift gradients)
    const float* merge_w;                                                // (P, nb, 3, 3)
    const float *m_scale, *m_shift, *m_alpha, *m_mean, *m_inv;           // P (merge_layer.2's BatchNorm + PReLU)
    float *g_br_scale, *g_br_shift, *g_br_alpha;                         // nb*P, accumulated (atomics)
    float* g_merge_w;                                                    // P*nb*9
    float *g_m_scale, *g_m_shift, *g_m_alpha;                            // P
};

// One WAVE per branch: a workgroup (NB waves) owns a 16-row band of one (image, channel) plane and walks its 64-column tiles.  All
// waves build the g_m halo tile in LDS together; then wave i evaluates branch i for the tile's 256 1x4 strips (four per lane).  A lane
// so carries ONE branch's nine weight-gradient accumulators and three BatchNorm / PReLU sums (12 registers instead of 60), its
// branch's constants are wave-uniform scalars, and the final reduction is 12 values per wave.  (One thread per strip doing all five
// branches: 234 VGPRs, 204 spilled SGPRs, two waves per SIMD, 378 ds_bpermute in the tail -- 248 us for 16 x 16 x 144x240 whose
// traffic is ~85 us.)
// (the plane's constants come through `const float* __restrict__` kernel arguments of their own: only then does hipcc read the
// wave-uniform values with scalar loads into SGPRs)
template <int NB>
__global__ __launch_bounds__(64 * NB) void pyr_merge_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ mraw,
                                                               const float* __restrict__ zcat, const float* __restrict__ k_merge_w,
                                                               const float* __restrict__ k_br_scale, const float* __restrict__ k_br_shift,
                                                               const float* __restrict__ k_br_alpha, const float* __restrict__ k_m_scale,
                                                               const float* __restrict__ k_m_shift, const float* __restrict__ k_m_alpha,
                                                               MbGeom g, float* __restrict__ gt) {
    constexpr int NT = 64 * NB;
    constexpr int NEL = (MB_TH + 2) * (MB_TW + 2);
    constexpr int NST = (NEL + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) float G[(MB_TH + 2) * MB_GS];
    __shared__ float redq[NB][3];
    int b = blockIdx.x;
    const int ty = b % g.tiles_y;  b /= g.tiles_y;
    const int c = b % g.P;
    const int n = b / g.P;
    const int y0 = ty * MB_TH;
    const int tid = threadIdx.x, lane = tid & 63;
    const int bi = __builtin_amdgcn_readfirstlane(tid >> 6);             // this wave's branch
    const int h = g.h, w = g.w;
    const size_t plane = (size_t)h * w;
    const float* gyp = gy + ((size_t)n * g.P + c) * plane;
    const float* mp = mraw + ((size_t)n * g.P + c) * plane;
    const float msc = k_m_scale[c], msh = k_m_shift[c];
    const bool mact = k_m_alpha != nullptr;
    const float mal = mact ? k_m_alpha[c] : 1.f;
    const bool vec = (w & 3) == 0;

    float wmi[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wmi[k] = k_merge_w[((size_t)c * NB + bi) * 9 + k];
    const int ch = bi * g.P + c;
    const float bs = k_br_scale[ch], bh = k_br_shift[ch], ba = k_br_alpha[ch];
    const float* zpl = zcat + (((size_t)n * NB + bi) * g.P + c) * plane;
    float* opl = gt + (((size_t)bi * g.N + n) * g.P + c) * plane;
    float dw[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) dw[k] = 0.f;
    float a_sc = 0.f, a_sh = 0.f, a_al = 0.f;
    float q_sc = 0.f, q_sh = 0.f, q_al = 0.f;

    for (int tx = 0; tx < g.tiles_x; ++tx) {
        const int x0 = tx * MB_TW;
        // this wave's branch values of its four strips requested FIRST: they do not depend on the g_m tile, so their round trip runs
        // under the tile's own loads and arithmetic
        float zall[4][4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int sidx = lane + 64 * it, r = sidx >> 4, s4 = (sidx & 15) * 4;
            const int y = y0 + r, xb = x0 + s4;
            if (y < h && xb < w) {
                const float* zp = zpl + (size_t)y * w + xb;
                if (vec) {
                    const float4 t4 = *reinterpret_cast<const float4*>(zp);
                    zall[it][0] = t4.x; zall[it][1] = t4.y; zall[it][2] = t4.z; zall[it][3] = t4.w;
                } else {
                    const int nv = min(4, w - xb);
#pragma unroll
                    for (int j = 0; j < 4; ++j) zall[it][j] = j < nv ? zp[j] : 0.f;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) zall[it][j] = 0.f;
            }
        }
        // ---- g_m = dL/dm on the tile + a one-pixel halo (zero outside the image): BatchNorm/PReLU backward of merge_layer.2
        {
            float gvv[NST], mvv[NST];
            // all loads of the tile first, then the arithmetic (interleaved, hipcc waits for each pair before the next is issued)
#pragma unroll
            for (int k = 0; k < NST; ++k) {
                const int t = tid + NT * k;
                const int R = t / (MB_TW + 2), Cq = t - R * (MB_TW + 2);
                const int py = y0 - 1 + R, px = x0 - 1 + Cq;
                const bool in = t < NEL && py >= 0 && py < h && px >= 0 && px < w;
                const size_t o = (size_t)min(max(py, 0), h - 1) * w + min(max(px, 0), w - 1);
                const float a = gyp[o], b2 = mp[o];
                gvv[k] = in ? a : 0.f;
                mvv[k] = b2;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < NST; ++k) {
                const int t = tid + NT * k;
                if (t < NEL) {
                    const int R = t / (MB_TW + 2), Cq = t - R * (MB_TW + 2);
                    const int py = y0 - 1 + R, px = x0 - 1 + Cq;
                    const bool in = py >= 0 && py < h && px >= 0 && px < w;
                    const float gv = gvv[k], mv = mvv[k];
                    const float u = mv * msc + msh;
                    const bool pos = !mact || u > 0.f;
                    const float gz = pos ? gv : mal * gv;                       // gv = 0 outside the image
                    if (in && R >= 1 && R <= MB_TH && Cq >= 1 && Cq <= MB_TW) {  // the tile's own pixels: every pixel exactly once
                        q_sc += gz * mv;
                        q_sh += gz;
                        if (!pos) q_al += gv * u;
                    }
                    G[R * MB_GS + Cq] = gz * msc;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int sidx = lane + 64 * it, r = sidx >> 4, s4 = (sidx & 15) * 4;
            const int y = y0 + r, xb = x0 + s4;
            if (y < h && xb < w) {
                float win[3][6];
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) {
                    const float4 a = *reinterpret_cast<const float4*>(&G[(r + rr) * MB_GS + s4]);
                    const float2 b2 = *reinterpret_cast<const float2*>(&G[(r + rr) * MB_GS + s4 + 4]);
                    win[rr][0] = a.x; win[rr][1] = a.y; win[rr][2] = a.z; win[rr][3] = a.w; win[rr][4] = b2.x; win[rr][5] = b2.y;
                }
                const int nv = min(4, w - xb);
                float* op = opl + (size_t)y * w + xb;
                float ov[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool live = j < nv;
                    const float zvj = zall[it][j];
                    const float u = zvj * bs + bh;
                    const bool pos = u > 0.f;
                    const float yv = live ? (pos ? u : ba * u) : 0.f;
                    float gyi = 0.f;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const float gm = win[2 - ky][j + 2 - kx];
                            gyi = fmaf(wmi[ky * 3 + kx], gm, gyi);
                            dw[ky * 3 + kx] = fmaf(yv, gm, dw[ky * 3 + kx]);
                        }
                    if (!live) gyi = 0.f;
                    const float gz = pos ? gyi : ba * gyi;
                    a_sc += gz * zvj;
                    a_sh += gz;
                    if (!pos) a_al += gyi * u;
                    ov[j] = gz * bs;
                }
                if (vec) {
                    *reinterpret_cast<float4*>(op) = make_float4(ov[0], ov[1], ov[2], ov[3]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (j < nv) op[j] = ov[j];
                }
            }
        }
        __syncthreads();      // the next tile overwrites G
    }

    // ---- parameter gradients: 15 wave sums; the branch's twelve go out from this wave, merge_layer.2's three through LDS
    float v[15];
#pragma unroll
    for (int k = 0; k < 9; ++k) v[k] = dw[k];
    v[9] = a_sc;  v[10] = a_sh;  v[11] = a_al;  v[12] = q_sc;  v[13] = q_sh;  v[14] = q_al;
#pragma unroll
    for (int k = 0; k < 15; ++k)             // every lane ends with the wave's total (seven DPP adds + one readlane: the xor ladder was 90 ds_bpermute)
        v[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_sum_dpp(v[k])), 63));
    if (lane < 9) {
        float t = v[0];
#pragma unroll
        for (int k = 1; k < 9; ++k) t = lane == k ? v[k] : t;
        atomicAdd(&g.g_merge_w[((size_t)c * NB + bi) * 9 + lane], t);
    } else if (lane == 9) {
        // frozen BatchNorm folded into (scale, shift) = (gamma*inv, beta - mean*gamma*inv): (d scale, d shift) -> (d gamma, d beta)
        atomicAdd(&g.g_br_scale[ch], g.br_inv ? (v[9] - g.br_mean[ch] * v[10]) * g.br_inv[ch] : v[9]);
        atomicAdd(&g.g_br_shift[ch], v[10]);
        atomicAdd(&g.g_br_alpha[ch], v[11]);
    } else if (lane == 10) {
        redq[bi][0] = v[12];  redq[bi][1] = v[13];  redq[bi][2] = v[14];
    }
    __syncthreads();
    if (tid == 0) {
        float t_sc = 0.f, t_sh = 0.f, t_al = 0.f;
#pragma unroll
        for (int i = 0; i < NB; ++i) { t_sc += redq[i][0];  t_sh += redq[i][1];  t_al += redq[i][2]; }
        atomicAdd(&g.g_m_scale[c], g.m_inv ? (t_sc - g.m_mean[c] * t_sh) * g.m_inv[c] : t_sc);
        atomicAdd(&g.g_m_shift[c], t_sh);
        if (mact && g.g_m_alpha) atomicAdd(&g.g_m_alpha[c], t_al);
    }
}

// ------------------------------------------------------------------------------------------------ branch backward
// Branches with hs >= h (bilinear up -> depthwise 3x3 -> adaptive average pool; hs == h is the plain depthwise 3x3).
//   forward   up[u,v]  = sum of 4 bilinear taps of x            (u,v on the hs x ws grid; align_corners)
//             cv[u,v]  = sum_k w[k] * up[(u,v) + k - 1]          (zero padding)
//             t[o,q]   = mean of cv over the adaptive window of (o,q)
//   backward  gp[u,v]  = sum over the outputs whose window holds (u,v) of gt / area
//             gu[u,v]  = sum_k w[k] * gp[(u,v) - (k - 1)]
//             gx[y,x]  = sum over the grid points that interpolate from (y,x) of weight * gu
//             gw[k]    = sum_{u,v} gp[u,v] * up[(u,v) + k - 1]
// A workgroup owns a band of UB_TH rows of one plane and walks its UB_TW-column tiles; per tile everything at grid resolution
// lives in LDS.  Grid points are OWNED (for gw) by the tile that holds their first bilinear source (ya, xa): a partition.
constexpr int UB_TH = 16, UB_TW = 32, UB_HX = 2;   // x-space tile, halo of the staged x tile
constexpr int UB_XW = UB_TW + 2 * UB_HX;           // 36
constexpr int UB_KG = 8;                           // grid rows (columns) that can interpolate from one x row (column)
constexpr int UB_MAXB = 3;

struct UbBranch {
    int hs, ws;
    float sh, sw;          // bilinear scales x grid -> up grid
    int UHT, UWT;          // grid rows / columns staged per tile (one-point halo included)
    int OH, OW;            // gt rows / columns staged per tile
    const float* w;        // (P, 9) stage weights
    float* gw;             // (P, 9), accumulated
    const float* gt;       // (N, P, h, w) gradient of this branch's output
};

struct UbGeom {
    int N, P, h, w, nb;
    int tiles_x, tiles_y;
    UbBranch b[UB_MAXB];
    int off_up, off_gp, off_gu, off_gt, off_tab;   // LDS offsets (floats), sized for the largest branch
    const float* add0;     // optional (N,P,h,w) tensors added into gx (the low-resolution branches' contributions)
    const float* add1;
};

// table layout per branch (float words, ints bit-cast), rebuilt per tile:
//   UR[UHT][4] = {x-tile offset of source row a, of source row b, w0, w1}    (zero weights: outside the grid / the x tile)
//   UC[UWT][4] = {x-tile column a, column b, w0, w1}
//   PR[UHT][4] = {first gt-tile row of the pooling windows that hold this grid row, their count, owned flag, 0};  PC[UWT][4] likewise
//   GY[UB_TH][2 + UB_KG] = {first grid-tile row that interpolates from this x row, count, weights...};  GX[UB_TW][2 + UB_KG]
__device__ __forceinline__ int ub_base(int p0, float s) {      // a grid index safely below every grid point that touches x index p0
    if (s <= 0.f) return 0;
    const int e = (int)floorf((float)(p0 - 1) / s) - 1;
    return e < 0 ? 0 : e;
}

__global__ __launch_bounds__(256) void pyr_branch_bwd_kernel(const float* __restrict__ x, UbGeom g, float* __restrict__ gx) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xs = smem;                                   // (UB_TH + 4) x UB_XW
    float* UP = smem + g.off_up;
    float* GP = smem + g.off_gp;
    float* GU = smem + g.off_gu;
    float* GT = smem + g.off_gt;
    float* TAB = smem + g.off_tab;
    __shared__ float red[4][9];
    int b = blockIdx.x;
    const int ty = b % g.tiles_y;  b /= g.tiles_y;
    const int c = b % g.P;
    const int n = b / g.P;
    const int y0 = ty * UB_TH;
    const int tid = threadIdx.x;
    const int h = g.h, w = g.w;
    const size_t plane = (size_t)h * w;
    const float* xp = x + ((size_t)n * g.P + c) * plane;
    float dwacc[UB_MAXB][9];
#pragma unroll
    for (int i = 0; i < UB_MAXB; ++i)
#pragma unroll
        for (int k = 0; k < 9; ++k) dwacc[i][k] = 0.f;

    for (int tx = 0; tx < g.tiles_x; ++tx) {
        const int x0 = tx * UB_TW;
        float acc[2] = {0.f, 0.f};                      // the thread's two x pixels: p = tid, tid + 256 -> (p / UB_TW, p % UB_TW)
        // ---- x tile (+2 halo), zero outside the image
        for (int t = tid; t < (UB_TH + 2 * UB_HX) * UB_XW; t += 256) {
            const int R = t / UB_XW, Cq = t - R * UB_XW;
            const int py = y0 - UB_HX + R, px = x0 - UB_HX + Cq;
            xs[t] = (py >= 0 && py < h && px >= 0 && px < w) ? xp[(size_t)py * w + px] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < UB_MAXB; ++i) {
            if (i >= g.nb) break;
            const UbBranch& B = g.b[i];
            const int hs = B.hs, ws = B.ws, UHT = B.UHT, UWT = B.UWT, OH = B.OH, OW = B.OW;
            float* UR = TAB;  float* UC = UR + 4 * UHT;  float* PR = UC + 4 * UWT;  float* PC = PR + 4 * UHT;
            float* GY = PC + 4 * UWT;  float* GX = GY + UB_TH * (2 + UB_KG);
            const int ub = ub_base(y0, B.sh) - 1, vb = ub_base(x0, B.sw) - 1;       // grid coordinates of tile row / column 0
            const int ob = max(ub, 0) * h / hs, qb = max(vb, 0) * w / ws;         // first gt row / column staged
            __syncthreads();                            // previous branch / tile done with TAB and the tiles
            // ---- tables
            for (int t = tid; t < UHT + UWT; t += 256) {
                const bool row = t < UHT;
                const int R = row ? t : t - UHT;
                const int u = (row ? ub : vb) + R, gs = row ? hs : ws, isz = row ? h : w, p0 = row ? y0 : x0, tl = row ? UB_TH : UB_TW;
                int a = p0, bb = p0, first = 0, cnt = 0, own = 0;
                float w0 = 0.f, w1 = 0.f;
                if (u >= 0 && u < gs) {
                    bilinear_src(row ? B.sh : B.sw, u, isz, a, bb, w0, w1);
                    own = (a >= p0 && a < p0 + tl) ? 1 : 0;
                    if (a < p0 - UB_HX || bb > p0 + tl - 1 + UB_HX) { a = p0; bb = p0; w0 = 0.f; w1 = 0.f; }   // sources outside the staged x tile: never needed there
                    const int lo = (int)(((unsigned)u * (unsigned)isz) / (unsigned)gs);
                    const int hi = (int)((((unsigned)(u + 1)) * (unsigned)isz + gs - 1) / (unsigned)gs) - 1;
                    first = lo - (row ? ob : qb);
                    cnt = hi - lo + 1;
                    if (first < 0 || first + cnt > (row ? OH : OW)) cnt = 0;       // cannot happen for rows that are used (host sizes OH / OW)
                }
                float* U4 = (row ? UR : UC) + 4 * R;
                U4[0] = __int_as_float(row ? (a - p0 + UB_HX) * UB_XW : (a - p0 + UB_HX));
                U4[1] = __int_as_float(row ? (bb - p0 + UB_HX) * UB_XW : (bb - p0 + UB_HX));
                U4[2] = w0;  U4[3] = w1;
                float* P4 = (row ? PR : PC) + 4 * R;
                P4[0] = __int_as_float(row ? first * OW : first);  P4[1] = __int_as_float(cnt);  P4[2] = __int_as_float(own);  P4[3] = 0.f;
            }
            for (int t = tid; t < UB_TH + UB_TW; t += 256) {      // gather tables: which grid rows / columns interpolate from x row / column p
                const bool row = t < UB_TH;
                const int l = row ? t : t - UB_TH;
                const int p = (row ? y0 : x0) + l, gs = row ? hs : ws, isz = row ? h : w, T = row ? UHT : UWT, base = row ? ub : vb;
                float* Gt = (row ? GY : GX) + l * (2 + UB_KG);
                int first = 0, cnt = 0;
                if (p < isz) {
                    bool open = false;
                    for (int R = 1; R < T - 1; ++R) {
                        const int u = base + R;
                        int a = -1, bb = -1;  float w0 = 0.f, w1 = 0.f;
                        if (u >= 0 && u < gs) bilinear_src(row ? B.sh : B.sw, u, isz, a, bb, w0, w1);
                        const bool hit = a == p || bb == p;
                        if (hit && !open) { open = true; first = R; }
                        if (open && R - first < UB_KG) {          // every slot between the first and the last contributing row is written
                            Gt[2 + R - first] = hit ? (a == p ? w0 : 0.f) + (bb == p ? w1 : 0.f) : 0.f;
                            if (hit) cnt = R - first + 1;
                        }
                    }
                }
                Gt[0] = __int_as_float(row ? first * UWT : first);  Gt[1] = __int_as_float(cnt);
            }
            // ---- gt tile, pre-scaled by 1 / window area
            {
                const float* gtp = B.gt + ((size_t)n * g.P + c) * plane;
                for (int t = tid; t < OH * OW; t += 256) {
                    const int R = t / OW, Cq = t - R * OW;
                    const int o = ob + R, q = qb + Cq;
                    float v = 0.f;
                    if (o < h && q < w) {
                        const int rc = (int)((((unsigned)(o + 1)) * (unsigned)hs + h - 1) / (unsigned)h) - (int)(((unsigned)o * (unsigned)hs) / (unsigned)h);
                        const int cc = (int)((((unsigned)(q + 1)) * (unsigned)ws + w - 1) / (unsigned)w) - (int)(((unsigned)q * (unsigned)ws) / (unsigned)w);
                        v = gtp[(size_t)o * w + q] / (float)(rc * cc);
                    }
                    GT[t] = v;
                }
            }
            __syncthreads();
            // ---- up-sampled tile and pooled-gradient tile on the grid
            for (int t = tid; t < UHT * UWT; t += 256) {
                const int R = t / UWT, Cq = t - R * UWT;
                const float4 rr = *reinterpret_cast<const float4*>(UR + 4 * R);
                const float4 cc = *reinterpret_cast<const float4*>(UC + 4 * Cq);
                const float* ra = xs + __float_as_int(rr.x);
                const float* rb = xs + __float_as_int(rr.y);
                const int xa = __float_as_int(cc.x), xb = __float_as_int(cc.y);
                const float top = cc.z * ra[xa] + cc.w * ra[xb];
                const float bot = cc.z * rb[xa] + cc.w * rb[xb];
                UP[t] = rr.z * top + rr.w * bot;
                const float4 pr = *reinterpret_cast<const float4*>(PR + 4 * R);
                const float4 pc = *reinterpret_cast<const float4*>(PC + 4 * Cq);
                const int rcnt = __float_as_int(pr.y), ccnt = __float_as_int(pc.y);
                const float* p0 = GT + __float_as_int(pr.x) + __float_as_int(pc.x);
                float s = 0.f;
                for (int a2 = 0; a2 < rcnt; ++a2)
                    for (int b2 = 0; b2 < ccnt; ++b2) s += p0[a2 * OW + b2];
                GP[t] = s;
            }
            __syncthreads();
            // ---- gradient on the grid (transposed 3x3) and the stage weights' gradient over the owned grid points
            {
                float wk[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) wk[k] = B.w[(size_t)c * 9 + k];
                for (int t = tid; t < (UHT - 2) * (UWT - 2); t += 256) {
                    const int R = 1 + t / (UWT - 2), Cq = 1 + t - (R - 1) * (UWT - 2);
                    const float* gpc = GP + R * UWT + Cq;
                    float gu = 0.f;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) gu = fmaf(wk[ky * 3 + kx], gpc[(1 - ky) * UWT + (1 - kx)], gu);
                    GU[R * UWT + Cq] = gu;
                    if (__float_as_int(PR[4 * R + 2]) && __float_as_int(PC[4 * Cq + 2])) {
                        const float gp = gpc[0];
                        const float* upc = UP + R * UWT + Cq;
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx)
                                dwacc[i][ky * 3 + kx] = fmaf(gp, upc[(ky - 1) * UWT + (kx - 1)], dwacc[i][ky * 3 + kx]);
                    }
                }
            }
            __syncthreads();
            // ---- gather to the x grid
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int p = tid + 256 * j;
                const int yl = p / UB_TW, xl = p - yl * UB_TW;
                const float* gyt = GY + yl * (2 + UB_KG);
                const float* gxt = GX + xl * (2 + UB_KG);
                const int rcnt = __float_as_int(gyt[1]), ccnt = __float_as_int(gxt[1]);
                const float* p0 = GU + __float_as_int(gyt[0]) + __float_as_int(gxt[0]);
                float s = 0.f;
                for (int a2 = 0; a2 < rcnt; ++a2) {
                    float rs = 0.f;
                    for (int b2 = 0; b2 < ccnt; ++b2) rs = fmaf(gxt[2 + b2], p0[a2 * UWT + b2], rs);
                    s = fmaf(gyt[2 + a2], rs, s);
                }
                acc[j] += s;
            }
        }
        // ---- one write of dL/dx
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int p = tid + 256 * j;
            const int yl = p / UB_TW, xl = p - yl * UB_TW;
            const int y = y0 + yl, xx = x0 + xl;
            if (y < h && xx < w) {
                const size_t o = ((size_t)n * g.P + c) * plane + (size_t)y * w + xx;
                float v = acc[j];
                if (g.add0) v += g.add0[o];
                if (g.add1) v += g.add1[o];
                gx[o] = v;
            }
        }
    }
    // ---- stage weight gradients
#pragma unroll
    for (int i = 0; i < UB_MAXB; ++i) {
        if (i >= g.nb) break;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            float t = dwacc[i][k];
            t = wave_sum_dpp(t);                              // total in lane 63
            if ((tid & 63) == 63) red[tid >> 6][k] = t;
        }
        __syncthreads();
        if (tid < 9) atomicAdd(&g.b[i].gw[(size_t)c * 9 + tid], (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]));
    }
}


// ------------------------------------------------------------------------------------------------ branch backward, streaming form
// The same gradients from the separable-stencil view (pyr_stencil.hpp): with A_ky (rows) and G_kx (columns) the banded matrices of
// a branch,   t = sum_ky sum_kx w[ky][kx] A_ky x G_kx^T   so
//     gx          = sum_ky A_ky^T (gt C_ky),    C_ky = sum_kx w[ky][kx] G_kx          (T taps per row / column, T = 3 or 5)
//     gw[ky][kx]  = < gt, A_ky x G_kx^T >
// everything at the map's own resolution.  Machine mapping = pyrpool_stream.hip's: a wave owns one plane, a block of 124 columns
// (lane = 2 adjacent columns, lanes 0 / 63 are halo lanes) and a segment of rows, and walks DOWN the rows of gt: row p is
// multiplied by the lane's column coefficients, the +-R column spill goes to the neighbouring lanes by DPP, the T gx rows that p
// touches accumulate in registers and the finished one is written; the T-row window of x for the weight gradient slides in
// registers too.  No LDS in the row loop except broadcast reads of the wave-uniform row coefficients.  One branch per launch
// (scale 2.0: T = 3, scale 1.5: T = 5, the same-size branch: T = 3 with unit coefficients); later launches add to gx.
constexpr int SB_SEGMAX = 19;

struct SbGeom {
    int N, P, h, w;
    int hs, ws;
    float sh, sw;
    int SEG, nseg, ncb, CBW;
    int accumulate;          // gx += (a later branch's launch)
};

template <int T>
__global__ __launch_bounds__(256) void pyr_branch_bwd_stream_kernel(const float* __restrict__ x, const float* __restrict__ gt,
                                                                    const float* __restrict__ wst, const float* __restrict__ add0,
                                                                    const float* __restrict__ add1, SbGeom g,
                                                                    float* __restrict__ gx, float* __restrict__ gw) {
    constexpr int R = (T - 1) / 2, PXL = 2, NC = PXL + 2 * R;
    __shared__ __attribute__((aligned(16))) float At[(SB_SEGMAX + 4) * 24];          // [row ys-R ..][ky][8]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned bid = blockIdx.x;
    const int sgi = bid % g.nseg;  bid /= g.nseg;
    const int cb = bid % g.ncb;  bid /= g.ncb;
    const int plane = (int)bid * 4 + wave;                                 // (N * P) % 4 == 0 (checked by the launcher)
    const int n = plane / g.P, c = plane - n * g.P;
    const int h = g.h, w = g.w;
    const int ys = sgi * g.SEG, ye = min(ys + g.SEG, h);
    const int px0 = cb * g.CBW + (lane - 1) * PXL;
    const bool writer = lane >= 1 && lane <= 62 && px0 < w;

    // row coefficients of the segment (shared by the four planes of the workgroup)
    for (int t = threadIdx.x; t < (g.SEG + 2 * R) * 3; t += 256) {
        const int ri = t / 3, ky = t - 3 * ri;
        float acc[5];
        p3_coeffs<T>(ys - R + ri, ky, h, g.hs, g.sh, acc);
        float* d = &At[ri * 24 + ky * 8];
        d[0] = acc[0]; d[1] = acc[1]; d[2] = acc[2]; d[3] = acc[3]; d[4] = acc[4];
    }
    // the lane's column coefficients: G[kx][i][s] multiplies x[., px0 + i - R + s]; C[ky] = sum_kx w[ky][kx] G[kx]
    float G[3][PXL][T], C[3][PXL][T];
    {
        const float* wp = wst + (size_t)c * 9;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int i = 0; i < PXL; ++i) {
                float acc[5];
                p3_coeffs<T>(px0 + i, kx, w, g.ws, g.sw, acc);
#pragma unroll
                for (int s2 = 0; s2 < T; ++s2) G[kx][i][s2] = acc[s2];
            }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int i = 0; i < PXL; ++i)
#pragma unroll
                for (int s2 = 0; s2 < T; ++s2)
                    C[ky][i][s2] = fmaf(wp[ky * 3 + 2], G[2][i][s2], fmaf(wp[ky * 3 + 1], G[1][i][s2], wp[ky * 3] * G[0][i][s2]));
    }
    __syncthreads();

    const size_t pl = (size_t)h * w;
    const float* xpl = x + ((size_t)n * g.P + c) * pl;
    const float* gpl = gt + ((size_t)n * g.P + c) * pl;
    float* opl = gx + ((size_t)n * g.P + c) * pl;
    const float* a0p = add0 ? add0 + ((size_t)n * g.P + c) * pl : nullptr;
    const float* a1p = add1 ? add1 + ((size_t)n * g.P + c) * pl : nullptr;
    unsigned xoff[NC];  bool xin[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int cx = px0 - R + j;
        xin[j] = cx >= 0 && cx < w;
        xoff[j] = (unsigned)min(max(cx, 0), w - 1) * 4u;
    }
    const bool gin0 = px0 >= 0 && px0 < w, gin1 = px0 + 1 >= 0 && px0 + 1 < w;
    const unsigned goff0 = (unsigned)min(max(px0, 0), w - 1) * 4u, goff1 = (unsigned)min(max(px0 + 1, 0), w - 1) * 4u;
    auto load_row = [&](int r, float (&v)[NC]) {
        const bool rin = r >= 0 && r < h;                                                   // uniform
        const char* row = reinterpret_cast<const char*>(xpl + (size_t)min(max(r, 0), h - 1) * w);
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const float t = *reinterpret_cast<const float*>(row + xoff[j]);
            v[j] = (rin && xin[j]) ? t : 0.f;
        }
    };
    auto load_g = [&](int r, float (&v)[PXL]) {
        const bool rin = r >= 0 && r < h;
        const char* row = reinterpret_cast<const char*>(gpl + (size_t)min(max(r, 0), h - 1) * w);
        const float t0 = *reinterpret_cast<const float*>(row + goff0), t1 = *reinterpret_cast<const float*>(row + goff1);
        v[0] = (rin && gin0) ? t0 : 0.f;
        v[1] = (rin && gin1) ? t1 : 0.f;
    };

    float xw[T][NC];                        // x rows p-R .. p+R
#pragma unroll
    for (int q = 0; q < T; ++q) load_row(ys - 2 * R + q, xw[q]);
    float xn[NC];
    load_row(ys - R + R + 1, xn);           // row (p + 1) + R of the next iteration
    float gv[PXL], gn[PXL];
    load_g(ys - R, gv);
    load_g(ys - R + 1, gn);
    float gacc[T][PXL];
#pragma unroll
    for (int q = 0; q < T; ++q) { gacc[q][0] = 0.f; gacc[q][1] = 0.f; }
    float dw[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) dw[k] = 0.f;

#pragma unroll 1
    for (int p = ys - R; p < ye + R; ++p) {
        const float* Ap = &At[(p - (ys - R)) * 24];
        float A[3][T];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const float4 t4 = *reinterpret_cast<const float4*>(Ap + ky * 8);
            A[ky][0] = t4.x; A[ky][1] = t4.y; A[ky][2] = t4.z;
            if constexpr (T > 3) { A[ky][3] = t4.w; A[ky][T - 1] = Ap[ky * 8 + 4]; }
        }
        // ---- gx: row p of gt through the transposed column stencils, then scattered over the T rows it touches
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            float loc[NC];
#pragma unroll
            for (int j = 0; j < NC; ++j) loc[j] = 0.f;
#pragma unroll
            for (int i = 0; i < PXL; ++i)
#pragma unroll
                for (int s2 = 0; s2 < T; ++s2) loc[i + s2] = fmaf(C[ky][i][s2], gv[i], loc[i + s2]);
            float hg[PXL];
            hg[0] = loc[R];  hg[1] = loc[R + 1];
#pragma unroll
            for (int i2 = 0; i2 < R; ++i2) {
                hg[i2] += p3_from_left(loc[R + PXL + i2]);               // the left neighbour's spill over its right edge
                hg[PXL - R + i2] += p3_from_right(loc[i2]);              // the right neighbour's spill over its left edge
            }
#pragma unroll
            for (int q = 0; q < T; ++q) {
                gacc[q][0] = fmaf(A[ky][q], hg[0], gacc[q][0]);
                gacc[q][1] = fmaf(A[ky][q], hg[1], gacc[q][1]);
            }
        }
        // ---- gw over the rows this segment owns
        if (p >= ys && p < ye && writer) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                float V[NC];
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    float v = A[ky][0] * xw[0][j];
#pragma unroll
                    for (int q = 1; q < T; ++q) v = fmaf(A[ky][q], xw[q][j], v);
                    V[j] = v;
                }
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    float acc = 0.f;
#pragma unroll
                    for (int i = 0; i < PXL; ++i) {
                        float qv = G[kx][i][0] * V[i];
#pragma unroll
                        for (int s2 = 1; s2 < T; ++s2) qv = fmaf(G[kx][i][s2], V[i + s2], qv);
                        acc = fmaf(gv[i], qv, acc);
                    }
                    dw[ky * 3 + kx] += acc;
                }
            }
        }
        // ---- gx row p - R is complete
        const int r = p - R;
        if (r >= ys && writer) {
            const size_t o = (size_t)r * w + px0;
            float v0 = gacc[0][0], v1 = gacc[0][1];
            if (a0p) { const float2 t = *reinterpret_cast<const float2*>(a0p + o); v0 += t.x; v1 += t.y; }
            if (a1p) { const float2 t = *reinterpret_cast<const float2*>(a1p + o); v0 += t.x; v1 += t.y; }
            if (g.accumulate) { const float2 t = *reinterpret_cast<const float2*>(opl + o); v0 += t.x; v1 += t.y; }
            *reinterpret_cast<float2*>(opl + o) = make_float2(v0, v1);
        }
#pragma unroll
        for (int q = 0; q < T - 1; ++q) { gacc[q][0] = gacc[q + 1][0]; gacc[q][1] = gacc[q + 1][1]; }
        gacc[T - 1][0] = 0.f;  gacc[T - 1][1] = 0.f;
        // ---- slide
#pragma unroll
        for (int q = 0; q < T - 1; ++q)
#pragma unroll
            for (int j = 0; j < NC; ++j) xw[q][j] = xw[q + 1][j];
#pragma unroll
        for (int j = 0; j < NC; ++j) xw[T - 1][j] = xn[j];
        gv[0] = gn[0];  gv[1] = gn[1];
        if (p + 1 < ye + R) {
            load_row(p + 2 + R, xn);
            load_g(p + 2, gn);
        }
    }
    // ---- stage weight gradient: wave sum, one atomic per tap and wave
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const float t = wave_sum_dpp(dw[k]);                      // total in lane 63
        if (lane == 63) atomicAdd(&gw[(size_t)c * 9 + k], t);
    }
}

// 0 = launched, 1 = shape not covered by the streaming form, < 0 error
static int pyr_branch_bwd_stream_try(const float* x, const float* gt, const float* wst, const float* add0, const float* add1, int N,
                                     int P, int h, int w, int hs, int ws, int accumulate, float* gx, float* gw, hipStream_t stream,
                                     bool dry) {
    static const int off = (MSPL_TUNE_INT("MSPL_PYR_BWD_STREAM", 1) == 0);
    if (off || (w & 1) || ((int64_t)N * P) % 4 != 0 || hs < h || ws < w || h < 2 || w < 2) return 1;
    if (!dry && ((((uintptr_t)gx) | ((uintptr_t)add0) | ((uintptr_t)add1)) & 7)) return 1;
    // (a branch of the map's own size is the plain 3x3: its coefficient tables have one unit entry inside the 3-tap band)
    const int Rr = (hs == h && ws == w) ? 1 : std::max(p3_stencil_radius(h, hs), p3_stencil_radius(w, ws));
    if (Rr > 2) return 1;
    if (dry) return 0;
    SbGeom g;
    memset(&g, 0, sizeof(g));
    g.N = N; g.P = P; g.h = h; g.w = w; g.hs = hs; g.ws = ws;
    g.sh = bilinear_scale(h, hs); g.sw = bilinear_scale(w, ws);
    g.CBW = 124;
    g.ncb = ceil_div(w, g.CBW);
    int seg = std::min(h, SB_SEGMAX);
    while (seg > 8 && (int64_t)N * P * g.ncb * ceil_div(h, seg) < 2048) --seg;
    seg = ceil_div(h, ceil_div(h, seg));
    g.SEG = seg;
    g.nseg = ceil_div(h, seg);
    g.accumulate = accumulate;
    const int64_t waves = (int64_t)N * P * g.ncb * g.nseg;
    if (waves >= (1ll << 31)) return 1;
    const dim3 grid((unsigned)(waves / 4)), blk(256);
    if (Rr <= 1) hipLaunchKernelGGL(pyr_branch_bwd_stream_kernel<3>, grid, blk, 0, stream, x, gt, wst, add0, add1, g, gx, gw);
    else hipLaunchKernelGGL(pyr_branch_bwd_stream_kernel<5>, grid, blk, 0, stream, x, gt, wst, add0, add1, g, gx, gw);
    MSPL_CHECK_LAUNCH("pyrpool_branch_bwd(streaming form)");
    return 0;
}


// ---- all branches with hs >= h in ONE walk: up to two up-sampled branches (T0 / T1 taps) and the same-size branch.
// Cheaper algebra than the one-branch kernel above: with K_kx = gt G_kx (the row of gt through the transposed COLUMN stencil of
// kernel column kx; the +-R spill exchanged with the neighbouring lanes) both gradients come from the same three vectors:
//     gx rows  += A_ky^T ( sum_kx w[ky][kx] K_kx )          gw[ky][kx] += < K_kx, A_ky x >  (own columns only)
// i.e. 3T + 9 multiply-adds per column for the weight gradient instead of 12T + 9, and only the lane's OWN columns of x are needed
// (one 8-byte load per row).  The same-size branch is the plain transposed 3x3 (neighbour columns by DPP).
template <int T0, int T1, int NUP, bool SAME>
__global__ __launch_bounds__(256) void pyr_branch_bwd_stream3_kernel(const float* __restrict__ x, const float* __restrict__ gt0,
                                                                     const float* __restrict__ gt1, const float* __restrict__ gt2,
                                                                     const float* __restrict__ w0, const float* __restrict__ w1,
                                                                     const float* __restrict__ w2, const float* __restrict__ add0,
                                                                     const float* __restrict__ add1, SbGeom g, int hs1, int ws1, float sh1,
                                                                     float sw1, float* __restrict__ gx, float* __restrict__ gw0,
                                                                     float* __restrict__ gw1, float* __restrict__ gw2) {
    constexpr int R0 = (T0 - 1) / 2, R1 = NUP > 1 ? (T1 - 1) / 2 : 0, RS = SAME ? 1 : 0;
    constexpr int RM = R0 > R1 ? (R0 > RS ? R0 : RS) : (R1 > RS ? R1 : RS);      // rows / columns of reach
    constexpr int TW = 2 * RM + 1;
    __shared__ __attribute__((aligned(16))) float At[2][(SB_SEGMAX + 4) * 24];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned bid = blockIdx.x;
    const int sgi = bid % g.nseg;  bid /= g.nseg;
    const int cb = bid % g.ncb;  bid /= g.ncb;
    const int plane = (int)bid * 4 + wave;
    const int n = plane / g.P, c = plane - n * g.P;
    const int h = g.h, w = g.w;
    const int ys = sgi * g.SEG, ye = min(ys + g.SEG, h);
    const int px0 = cb * g.CBW + (lane - 1) * 2;
    const bool writer = lane >= 1 && lane <= 62 && px0 < w;
    const float wmask = writer ? 1.f : 0.f;

    for (int t = threadIdx.x; t < (g.SEG + 2 * RM) * 3 * NUP; t += 256) {
        const int ub = t / ((g.SEG + 2 * RM) * 3), t2 = t - ub * (g.SEG + 2 * RM) * 3;
        const int ri = t2 / 3, ky = t2 - 3 * ri;
        float acc[5];
        if (ub == 0) p3_coeffs<T0>(ys - RM + ri, ky, h, g.hs, g.sh, acc);
        else p3_coeffs<T1>(ys - RM + ri, ky, h, hs1, sh1, acc);
        float* d = &At[ub][ri * 24 + ky * 8];
        d[0] = acc[0]; d[1] = acc[1]; d[2] = acc[2]; d[3] = acc[3]; d[4] = acc[4];
    }
    float G0[3][2][T0], G1[3][2][T1];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float acc[5];
            p3_coeffs<T0>(px0 + i, kx, w, g.ws, g.sw, acc);
#pragma unroll
            for (int s2 = 0; s2 < T0; ++s2) G0[kx][i][s2] = acc[s2];
            if (NUP > 1) {
                p3_coeffs<T1>(px0 + i, kx, w, ws1, sw1, acc);
#pragma unroll
                for (int s2 = 0; s2 < T1; ++s2) G1[kx][i][s2] = acc[s2];
            }
        }
    __syncthreads();
    const float* wp0 = w0 + (size_t)c * 9;
    const float* wp1 = (NUP > 1 ? w1 : w0) + (size_t)c * 9;
    const float* wp2 = (SAME ? w2 : w0) + (size_t)c * 9;

    const size_t pl = (size_t)h * w;
    const size_t pbase = ((size_t)n * g.P + c) * pl;
    const bool cin = px0 >= 0 && px0 < w;                // both of the lane's columns are inside or outside together (w, px0 even)
    const unsigned coff = (unsigned)min(max(px0, 0), w - 2) * 4u;
    // Rows / columns outside the image: the address is clamped and the value multiplied by 0 WHEN IT IS CONSUMED.  (With a select
    // hipcc puts the load behind a branch and waits for it on the spot; with the multiply next to the load it waits there too: either
    // way one exposed memory round trip per row and operand.  Raw values requested at the top of an iteration, masked at its end.)
    const float cmask = cin ? 1.f : 0.f;
    auto rmask = [&](int r) { return (r >= 0 && r < h) ? cmask : 0.f; };
    auto load_raw = [&](const float* base, int r) {
        const char* row = reinterpret_cast<const char*>(base + pbase + (size_t)min(max(r, 0), h - 1) * w);
        return *reinterpret_cast<const float2*>(row + coff);
    };
    auto load2 = [&](const float* base, int r, float (&v)[2]) {      // prologue only
        const float2 t = load_raw(base, r);
        const float m = rmask(r);
        v[0] = t.x * m;  v[1] = t.y * m;
    };

    float xw[TW][2];                        // x rows p-RM .. p+RM, own columns
#pragma unroll
    for (int q = 0; q < TW; ++q) load2(x, ys - 2 * RM + q, xw[q]);
    float ga[2], gb[2], gc[2];
    load2(gt0, ys - RM, ga);
    if (NUP > 1) load2(gt1, ys - RM, gb);
    if (SAME) load2(gt2, ys - RM, gc);
    float gacc[TW][2];
#pragma unroll
    for (int q = 0; q < TW; ++q) { gacc[q][0] = 0.f; gacc[q][1] = 0.f; }
    float dwa[9], dwb[9], dwc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) { dwa[k] = 0.f; dwb[k] = 0.f; dwc[k] = 0.f; }

#pragma unroll 1
    for (int p = ys - RM; p < ye + RM; ++p) {
        const bool own_row = p >= ys && p < ye;                  // uniform
        // next row's operands: requested now, consumed (masked) at the end of this iteration
        const float2 xr = load_raw(x, p + 1 + RM), gar = load_raw(gt0, p + 1);
        const float2 gbr = NUP > 1 ? load_raw(gt1, p + 1) : make_float2(0.f, 0.f);
        const float2 gcr = SAME ? load_raw(gt2, p + 1) : make_float2(0.f, 0.f);
        // ---- one up-sampled branch: K_kx, then gx rows and (owned rows) the weight gradient
        auto up_branch = [&](auto tt, const float (&G)[3][2][decltype(tt)::value], const float* Ab, const float* wp, const float (&gv)[2],
                             float (&dw)[9]) {
            constexpr int T = decltype(tt)::value, R = (T - 1) / 2, NC = 2 + 2 * R, Q0 = RM - R;
            float A[3][T];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float4 t4 = *reinterpret_cast<const float4*>(Ab + ky * 8);
                A[ky][0] = t4.x; A[ky][1] = t4.y; A[ky][2] = t4.z;
                if constexpr (T > 3) { A[ky][3] = t4.w; A[ky][T - 1] = Ab[ky * 8 + 4]; }
            }
            float hk[3][2];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                float loc[NC];
#pragma unroll
                for (int j = 0; j < NC; ++j) loc[j] = 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int s2 = 0; s2 < T; ++s2) loc[i + s2] = fmaf(G[kx][i][s2], gv[i], loc[i + s2]);
                hk[kx][0] = loc[R];  hk[kx][1] = loc[R + 1];
#pragma unroll
                for (int i2 = 0; i2 < R; ++i2) {
                    hk[kx][i2] += p3_from_left(loc[R + 2 + i2]);
                    hk[kx][2 - R + i2] += p3_from_right(loc[i2]);
                }
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float wa = wp[ky * 3], wb = wp[ky * 3 + 1], wc = wp[ky * 3 + 2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float hg = fmaf(wc, hk[2][i], fmaf(wb, hk[1][i], wa * hk[0][i]));
#pragma unroll
                    for (int q = 0; q < T; ++q) gacc[Q0 + q][i] = fmaf(A[ky][q], hg, gacc[Q0 + q][i]);
                }
            }
            if (own_row) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        float v = A[ky][0] * xw[Q0][i];
#pragma unroll
                        for (int q = 1; q < T; ++q) v = fmaf(A[ky][q], xw[Q0 + q][i], v);
                        v *= wmask;
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) dw[ky * 3 + kx] = fmaf(hk[kx][i], v, dw[ky * 3 + kx]);
                    }
            }
        };
        up_branch(std::integral_constant<int, T0>(), G0, &At[0][(p - (ys - RM)) * 24], wp0, ga, dwa);
        if (NUP > 1) up_branch(std::integral_constant<int, T1>(), G1, &At[1][(p - (ys - RM)) * 24], wp1, gb, dwb);
        if (SAME) {
            // gx[r][c] += w[ky][kx] * gt[r - ky + 1][c - kx + 1]: row p of gt goes to rows p + ky - 1
            const float e[4] = {p3_from_left(gc[1]), gc[0], gc[1], p3_from_right(gc[0])};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float wa = wp2[ky * 3], wb = wp2[ky * 3 + 1], wc = wp2[ky * 3 + 2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    gacc[RM - 1 + ky][i] = fmaf(wc, e[i], fmaf(wb, e[i + 1], fmaf(wa, e[i + 2], gacc[RM - 1 + ky][i])));
            }
            if (own_row) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const float xe[4] = {p3_from_left(xw[RM - 1 + ky][1]), xw[RM - 1 + ky][0], xw[RM - 1 + ky][1],
                                         p3_from_right(xw[RM - 1 + ky][0])};
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
                        dwc[ky * 3 + kx] = fmaf(gc[1] * wmask, xe[1 + kx], fmaf(gc[0] * wmask, xe[kx], dwc[ky * 3 + kx]));
                }
            }
        }
        // ---- gx row p - RM is complete
        const int r = p - RM;
        if (r >= ys && writer) {
            const size_t o = pbase + (size_t)r * w + px0;
            float v0 = gacc[0][0], v1 = gacc[0][1];
            if (add0 && add1) {
                const float2 t = *reinterpret_cast<const float2*>(add0 + o), t2 = *reinterpret_cast<const float2*>(add1 + o);
                v0 += t.x + t2.x;  v1 += t.y + t2.y;
            } else if (add0) {
                const float2 t = *reinterpret_cast<const float2*>(add0 + o);
                v0 += t.x;  v1 += t.y;
            }
            *reinterpret_cast<float2*>(gx + o) = make_float2(v0, v1);
        }
#pragma unroll
        for (int q = 0; q < TW - 1; ++q) { gacc[q][0] = gacc[q + 1][0]; gacc[q][1] = gacc[q + 1][1]; xw[q][0] = xw[q + 1][0]; xw[q][1] = xw[q + 1][1]; }
        gacc[TW - 1][0] = 0.f;  gacc[TW - 1][1] = 0.f;
        {
            const float mx = rmask(p + 1 + RM), mg = rmask(p + 1);
            xw[TW - 1][0] = xr.x * mx;  xw[TW - 1][1] = xr.y * mx;
            ga[0] = gar.x * mg;  ga[1] = gar.y * mg;
            if (NUP > 1) { gb[0] = gbr.x * mg;  gb[1] = gbr.y * mg; }
            if (SAME) { gc[0] = gcr.x * mg;  gc[1] = gcr.y * mg; }
        }
    }
    auto flush = [&](float (&dw)[9], float* gw) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const float t = wave_sum_dpp(dw[k]);                  // total in lane 63
            if (lane == 63) atomicAdd(&gw[(size_t)c * 9 + k], t);
        }
    };
    flush(dwa, gw0);
    if (NUP > 1) flush(dwb, gw1);
    if (SAME) flush(dwc, gw2);
}

// the standard pattern [up, up, same] (or [up, same] / [up, up] / [up]) in one launch; 0 = launched, 1 = not covered
static int pyr_branch_bwd_stream3_try(const float* x, int N, int P, int h, int w, int nb, const int32_t* hs, const int32_t* ws,
                                      const float* const* stage_w, const float* const* gt, float* const* gw, const float* add0,
                                      const float* add1, float* gx, hipStream_t stream, bool dry) {
    static const int off = (MSPL_TUNE_INT("MSPL_PYR_BWD_STREAM3", 1) == 0);
    if (off || (w & 1) || w < 2 || h < 2 || ((int64_t)N * P) % 4 != 0 || nb < 1 || nb > 3) return 1;
    const bool same = hs[nb - 1] == h && ws[nb - 1] == w;
    const int nup = nb - (same ? 1 : 0);
    if (nup < 1 || nup > 2) return 1;
    int taps[2] = {3, 3};
    for (int i = 0; i < nup; ++i) {
        if (hs[i] < h || ws[i] < w || (hs[i] == h && ws[i] == w)) return 1;
        const int Rr = std::max(p3_stencil_radius(h, hs[i]), p3_stencil_radius(w, ws[i]));
        if (Rr > 2) return 1;
        taps[i] = Rr <= 1 ? 3 : 5;
    }
    if (!dry && ((((uintptr_t)gx) | ((uintptr_t)add0) | ((uintptr_t)add1) | ((uintptr_t)x)) & 7)) return 1;
    if (dry) return 0;
    for (int i = 0; i < nb; ++i)
        if ((((uintptr_t)gt[i]) & 7)) return 1;
    SbGeom g;
    memset(&g, 0, sizeof(g));
    g.N = N; g.P = P; g.h = h; g.w = w; g.hs = hs[0]; g.ws = ws[0];
    g.sh = bilinear_scale(h, hs[0]); g.sw = bilinear_scale(w, ws[0]);
    const int hs1 = nup > 1 ? hs[1] : hs[0], ws1 = nup > 1 ? ws[1] : ws[0];
    const float sh1 = bilinear_scale(h, hs1), sw1 = bilinear_scale(w, ws1);
    g.CBW = 124;
    g.ncb = ceil_div(w, g.CBW);
    int seg = std::min(h, SB_SEGMAX);
    while (seg > 8 && (int64_t)N * P * g.ncb * ceil_div(h, seg) < 2048) --seg;
    seg = ceil_div(h, ceil_div(h, seg));
    g.SEG = seg;
    g.nseg = ceil_div(h, seg);
    const int64_t waves = (int64_t)N * P * g.ncb * g.nseg;
    if (waves >= (1ll << 31)) return 1;
    const dim3 grid((unsigned)(waves / 4)), blk(256);
    const float* g1 = nup > 1 ? gt[1] : gt[0];
    const float* g2 = same ? gt[nb - 1] : gt[0];
    const float* s1 = nup > 1 ? stage_w[1] : stage_w[0];
    const float* s2 = same ? stage_w[nb - 1] : stage_w[0];
    float* o1 = nup > 1 ? gw[1] : gw[0];
    float* o2 = same ? gw[nb - 1] : gw[0];
#define MSPL_SB3(A, B, U, S) hipLaunchKernelGGL((pyr_branch_bwd_stream3_kernel<A, B, U, S>), grid, blk, 0, stream, x, gt[0], g1, g2, stage_w[0], s1, s2, add0, add1, g, hs1, ws1, sh1, sw1, gx, gw[0], o1, o2)
    if (nup == 2 && same) {
        if (taps[0] == 3 && taps[1] == 5) MSPL_SB3(3, 5, 2, true);
        else if (taps[0] == 3 && taps[1] == 3) MSPL_SB3(3, 3, 2, true);
        else if (taps[0] == 5 && taps[1] == 5) MSPL_SB3(5, 5, 2, true);
        else MSPL_SB3(5, 3, 2, true);
    } else if (nup == 2) {
        if (taps[0] == 3 && taps[1] == 5) MSPL_SB3(3, 5, 2, false);
        else if (taps[0] == 3 && taps[1] == 3) MSPL_SB3(3, 3, 2, false);
        else if (taps[0] == 5 && taps[1] == 5) MSPL_SB3(5, 5, 2, false);
        else MSPL_SB3(5, 3, 2, false);
    } else if (same) {
        if (taps[0] == 3) MSPL_SB3(3, 3, 1, true); else MSPL_SB3(5, 3, 1, true);
    } else {
        if (taps[0] == 3) MSPL_SB3(3, 3, 1, false); else MSPL_SB3(5, 3, 1, false);
    }
#undef MSPL_SB3
    MSPL_CHECK_LAUNCH("pyrpool_branch_bwd(streaming, all branches)");
    return 0;
}

}  // namespace mspl

using namespace mspl;

extern "C" int mspl_pyrpool_merge_bwd(const float* gy, const float* mraw, const float* zcat, int32_t N, int32_t P, int32_t h,
                                      int32_t w, int32_t nb, const float* br_scale, const float* br_shift, const float* br_alpha,
                                      const float* br_mean, const float* br_inv, const float* merge_w, const float* m_scale,
                                      const float* m_shift, const float* m_alpha, const float* m_mean, const float* m_inv,
                                      float* gt, float* g_br_scale, float* g_br_shift, float* g_br_alpha, float* g_merge_w,
                                      float* g_m_scale, float* g_m_shift, float* g_m_alpha, void* stream) {
    MSPL_REQUIRE(gy && mraw && zcat && br_scale && br_shift && br_alpha && merge_w && m_scale && m_shift && gt && g_br_scale &&
                 g_br_shift && g_br_alpha && g_merge_w && g_m_scale && g_m_shift, MSPL_ERR_NULL_POINTER, "pyrpool_merge_bwd: null pointer");
    MSPL_REQUIRE((br_mean == nullptr) == (br_inv == nullptr) && (m_mean == nullptr) == (m_inv == nullptr), MSPL_ERR_NULL_POINTER,
                 "pyrpool_merge_bwd: BatchNorm mean and inverse deviation must be given together");
    MSPL_REQUIRE((m_alpha == nullptr) == (g_m_alpha == nullptr), MSPL_ERR_NULL_POINTER, "pyrpool_merge_bwd: m_alpha / g_m_alpha mismatch");
    MSPL_REQUIRE(N > 0 && P > 0 && h > 0 && w > 0, MSPL_ERR_BAD_SHAPE, "pyrpool_merge_bwd: bad shape N=%d P=%d %dx%d", N, P, h, w);
    MSPL_REQUIRE(nb >= 1 && nb <= 5, MSPL_ERR_UNSUPPORTED, "pyrpool_merge_bwd: %d branches (1..5)", nb);
    MSPL_REQUIRE((int64_t)N * nb * P * h * w < (1ll << 40), MSPL_ERR_BAD_SHAPE, "pyrpool_merge_bwd: tensor too large");
    MbGeom g;
    memset(&g, 0, sizeof(g));
    g.N = N; g.P = P; g.h = h; g.w = w;
    g.tiles_x = ceil_div(w, MB_TW); g.tiles_y = ceil_div(h, MB_TH);
    g.br_scale = br_scale; g.br_shift = br_shift; g.br_alpha = br_alpha; g.br_mean = br_mean; g.br_inv = br_inv;
    g.merge_w = merge_w;
    g.m_scale = m_scale; g.m_shift = m_shift; g.m_alpha = m_alpha; g.m_mean = m_mean; g.m_inv = m_inv;
    g.g_br_scale = g_br_scale; g.g_br_shift = g_br_shift; g.g_br_alpha = g_br_alpha; g.g_merge_w = g_merge_w;
    g.g_m_scale = g_m_scale; g.g_m_shift = g_m_shift; g.g_m_alpha = g_m_alpha;
    const int64_t blocks = (int64_t)N * P * g.tiles_y;
    MSPL_REQUIRE(blocks < (1ll << 31), MSPL_ERR_BAD_SHAPE, "pyrpool_merge_bwd: grid too large");
    const dim3 grid((unsigned)blocks);
    hipStream_t s = (hipStream_t)stream;
    switch (nb) {
        case 1: hipLaunchKernelGGL(pyr_merge_bwd_kernel<1>, grid, dim3(64), 0, s, gy, mraw, zcat, merge_w, br_scale, br_shift, br_alpha, m_scale, m_shift, m_alpha, g, gt); break;
        case 2: hipLaunchKernelGGL(pyr_merge_bwd_kernel<2>, grid, dim3(128), 0, s, gy, mraw, zcat, merge_w, br_scale, br_shift, br_alpha, m_scale, m_shift, m_alpha, g, gt); break;
        case 3: hipLaunchKernelGGL(pyr_merge_bwd_kernel<3>, grid, dim3(192), 0, s, gy, mraw, zcat, merge_w, br_scale, br_shift, br_alpha, m_scale, m_shift, m_alpha, g, gt); break;
        case 4: hipLaunchKernelGGL(pyr_merge_bwd_kernel<4>, grid, dim3(256), 0, s, gy, mraw, zcat, merge_w, br_scale, br_shift, br_alpha, m_scale, m_shift, m_alpha, g, gt); break;
        default: hipLaunchKernelGGL(pyr_merge_bwd_kernel<5>, grid, dim3(320), 0, s, gy, mraw, zcat, merge_w, br_scale, br_shift, br_alpha, m_scale, m_shift, m_alpha, g, gt); break;
    }
    MSPL_CHECK_LAUNCH("pyrpool_merge_bwd");
    return MSPL_OK;
}

// host-side plan of the branch-backward tiles; returns the LDS bytes, 0 when a branch is not covered
static size_t ub_plan(int N, int P, int h, int w, int nb, const int32_t* hs, const int32_t* ws, UbGeom& g) {
    if (nb < 1 || nb > UB_MAXB || h < 2 || w < 2) return 0;
    memset(&g, 0, sizeof(g));
    g.N = N; g.P = P; g.h = h; g.w = w; g.nb = nb;
    g.tiles_x = ceil_div(w, UB_TW); g.tiles_y = ceil_div(h, UB_TH);
    int max_u = 0, max_g = 0, max_tab = 0;
    for (int i = 0; i < nb; ++i) {
        UbBranch& B = g.b[i];
        if (hs[i] < h || ws[i] < w || hs[i] < 2 || ws[i] < 2) return 0;
        if ((int64_t)(hs[i] - 1) > 3ll * (h - 1) || (int64_t)(ws[i] - 1) > 3ll * (w - 1)) return 0;   // <= 8 grid rows per x row (UB_KG)
        B.hs = hs[i]; B.ws = ws[i];
        B.sh = bilinear_scale(h, hs[i]); B.sw = bilinear_scale(w, ws[i]);
        // grid rows that can touch the x rows [y0 - 1, y0 + TH]: (TH + 2) / s + slack, plus the one-point halo on both sides
        B.UHT = (int)((double)(UB_TH + 2) / (double)B.sh) + 6;
        B.UWT = (int)((double)(UB_TW + 2) / (double)B.sw) + 6;
        B.OH = (int)(((int64_t)(B.UHT + 1) * h + hs[i] - 1) / hs[i]) + 2;
        B.OW = (int)(((int64_t)(B.UWT + 1) * w + ws[i] - 1) / ws[i]) + 2;
        max_u = max_u > B.UHT * B.UWT ? max_u : B.UHT * B.UWT;
        max_g = max_g > B.OH * B.OW ? max_g : B.OH * B.OW;
        const int tab = 8 * (B.UHT + B.UWT) + (UB_TH + UB_TW) * (2 + UB_KG);
        max_tab = max_tab > tab ? max_tab : tab;
    }
    auto al = [](int v) { return (v + 3) & ~3; };
    int off = al((UB_TH + 2 * UB_HX) * UB_XW);
    g.off_up = off;  off += al(max_u);
    g.off_gp = off;  off += al(max_u);
    g.off_gu = off;  off += al(max_u);
    g.off_gt = off;  off += al(max_g);
    g.off_tab = off; off += al(max_tab);
    return (size_t)off * sizeof(float);
}

extern "C" int mspl_pyrpool_branch_bwd_fits(int32_t N, int32_t P, int32_t h, int32_t w, int32_t nb, const int32_t* hs,
                                            const int32_t* ws) {
    if (!hs || !ws || N <= 0 || P <= 0) return 0;
    if (pyr_branch_bwd_stream3_try(nullptr, N, P, h, w, nb, hs, ws, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, true) == 0) return 1;
    bool all_stream = nb >= 1;
    for (int i = 0; i < nb; ++i)
        all_stream = all_stream && pyr_branch_bwd_stream_try(nullptr, nullptr, nullptr, nullptr, nullptr, N, P, h, w, hs[i], ws[i], 0, nullptr, nullptr, nullptr, true) == 0;
    if (all_stream) return 1;
    UbGeom g;
    const size_t lds = ub_plan(N, P, h, w, nb, hs, ws, g);
    return lds > 0 && lds <= 64 * 1024 && (int64_t)N * P * g.tiles_y < (1ll << 31);
}

extern "C" int mspl_pyrpool_branch_bwd(const float* x, int32_t N, int32_t P, int32_t h, int32_t w, int32_t nb, const int32_t* hs,
                                       const int32_t* ws, const float* const* stage_w, const float* const* gt, float* const* gw,
                                       const float* add0, const float* add1, float* gx, void* stream) {
    MSPL_REQUIRE(x && hs && ws && stage_w && gt && gw && gx, MSPL_ERR_NULL_POINTER, "pyrpool_branch_bwd: null pointer");
    MSPL_REQUIRE(N > 0 && P > 0 && h > 0 && w > 0, MSPL_ERR_BAD_SHAPE, "pyrpool_branch_bwd: bad shape N=%d P=%d %dx%d", N, P, h, w);
    MSPL_REQUIRE(nb >= 1 && nb <= UB_MAXB, MSPL_ERR_UNSUPPORTED, "pyrpool_branch_bwd: %d branches (1..%d)", nb, UB_MAXB);
    for (int i = 0; i < nb; ++i) MSPL_REQUIRE(stage_w[i] && gt[i] && gw[i], MSPL_ERR_NULL_POINTER, "pyrpool_branch_bwd: branch %d pointer", i);
    {   // every branch in one walk (the standard [up, up, same] pattern)
        const int rc = pyr_branch_bwd_stream3_try(x, N, P, h, w, nb, hs, ws, stage_w, gt, gw, add0, add1, gx, (hipStream_t)stream, false);
        if (rc <= 0) return rc;
    }
    {   // streaming form: one launch per branch, the first one overwrites gx (and adds add0 / add1), the others accumulate
        bool all_stream = true;
        for (int i = 0; i < nb; ++i) {
            MSPL_REQUIRE(stage_w[i] && gt[i] && gw[i], MSPL_ERR_NULL_POINTER, "pyrpool_branch_bwd: branch %d pointer", i);
            all_stream = all_stream && pyr_branch_bwd_stream_try(x, gt[i], stage_w[i], add0, add1, N, P, h, w, hs[i], ws[i], 0, gx, gw[i], nullptr, true) == 0;
        }
        all_stream = all_stream && !((((uintptr_t)gx) | ((uintptr_t)add0) | ((uintptr_t)add1)) & 7);
        if (all_stream) {
            for (int i = 0; i < nb; ++i) {
                const int rc = pyr_branch_bwd_stream_try(x, gt[i], stage_w[i], i == 0 ? add0 : nullptr, i == 0 ? add1 : nullptr, N, P, h, w,
                                                         hs[i], ws[i], i > 0, gx, gw[i], (hipStream_t)stream, false);
                MSPL_REQUIRE(rc <= 0, MSPL_ERR_UNSUPPORTED, "pyrpool_branch_bwd: streaming form refused branch %d after accepting it", i);
                if (rc < 0) return rc;
            }
            return MSPL_OK;
        }
    }
    UbGeom g;
    const size_t lds = ub_plan(N, P, h, w, nb, hs, ws, g);
    MSPL_REQUIRE(lds > 0 && lds <= 64 * 1024, MSPL_ERR_UNSUPPORTED,
                 "pyrpool_branch_bwd: %d branches on a %dx%d map are not covered (sizes must be in [1x, 3x], tile LDS %zu B)", nb, h, w, lds);
    for (int i = 0; i < nb; ++i) {
        MSPL_REQUIRE(stage_w[i] && gt[i] && gw[i], MSPL_ERR_NULL_POINTER, "pyrpool_branch_bwd: branch %d pointer", i);
        g.b[i].w = stage_w[i]; g.b[i].gt = gt[i]; g.b[i].gw = gw[i];
    }
    g.add0 = add0; g.add1 = add1;
    const int64_t blocks = (int64_t)N * P * g.tiles_y;
    MSPL_REQUIRE(blocks < (1ll << 31), MSPL_ERR_BAD_SHAPE, "pyrpool_branch_bwd: grid too large");
    hipLaunchKernelGGL(pyr_branch_bwd_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, x, g, gx);
    MSPL_CHECK_LAUNCH("pyrpool_branch_bwd");
    return MSPL_OK;
}
