// Backward of the LOW-RESOLUTION pyramid branches (scales < 1) of EfficientPyrPool, the part between the transposed bilinear
// interpolation and the full-resolution gradient, for every such branch of a pyramid in ONE launch.
//
// Reference arithmetic (nn_layers/efficient_pyramid_pool.py:44-47, autograd of it): per branch i and projected channel c
//     p = adaptive_avg_pool2d(x_c, (hs, ws));   e = dw3x3_c(p);   branch value = bilinear_up(e)
// so, given g_e = dL/de (mspl_bilinear_bwd of the branch-major gradient the merge backward wrote):
//     g_p[u, v]   = sum_{ky,kx} w[ky][kx] * g_e[u - ky + 1, v - kx + 1]                 (zero outside the map)
//     g_w[ky][kx] += sum_{u,v} g_e[u, v] * p[u + ky - 1, v + kx - 1]
//     g_x[y, x]   = sum over the pooling windows (i, j) that contain (y, x) of g_p[i, j] / area(i, j)
// Rounds 3-4 ran this as three launches per branch (conv3x3 with flipped weights, the generic 3x3 weight gradient, the adaptive
// pool's gather backward) on maps of 5x5 .. 72x120 values: six launches of pure latency per pyramid and step, thirty per train
// step.  Here a workgroup owns (plane, branch, band of full-resolution rows): g_e of the plane sits zero-haloed in LDS (<= 36 KB),
// g_p of the rows a band touches is evaluated once into LDS, the band's share of the low-resolution pixels feeds the nine tap sums
// (wave + LDS reduction, nine atomics per workgroup into the parameter's gradient buffer).
#include <stdlib.h>

#include <mutex>

#include "common.hpp"

namespace mspl {

constexpr int PDB_MAXB = 2;

struct PdbGeom {
    int N, P, h, w, nb, bands;
    int hs[PDB_MAXB], ws[PDB_MAXB];
    int ger[PDB_MAXB], gpr[PDB_MAXB];  // LDS rows of g_e (haloed columns) / g_p a band needs at most
    const float* ge[PDB_MAXB];        // (N, P, hs, ws)
    const float* pooled[PDB_MAXB];    // (N, P, hs, ws)
    const float* wts[PDB_MAXB];       // (P, 1, 3, 3)
    float* gw[PDB_MAXB];              // (P, 1, 3, 3), accumulated
    float* gx[PDB_MAXB];              // (N, P, h, w)
};

// ATen's adaptive pooling window of output o: [floor(o * I / O), ceil((o + 1) * I / O))
__device__ __forceinline__ int pdb_s(int o, int I, int O) { return (int)(((unsigned)o * (unsigned)I) / (unsigned)O); }
__device__ __forceinline__ int pdb_e(int o, int I, int O) { return (int)((((unsigned)(o + 1)) * (unsigned)I + O - 1) / (unsigned)O); }

__global__ __launch_bounds__(256) void pyr_down_mid_bwd_kernel(PdbGeom g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int plane = blockIdx.x, bi = blockIdx.y, band = blockIdx.z;
    const int c = plane % g.P;
    const int hs = g.hs[bi], ws = g.ws[bi], h = g.h, w = g.w;
    const int tid = threadIdx.x;
    const int WS2 = ws + 2;
    __shared__ float red[9][4];
    // The band's low-resolution rows: its share [r0, r1) of the weight gradient, and the rows [ia, ib] whose g_p its full-resolution rows
    // can touch, which read g_e one row further on either side.  Only those rows of g_e / g_p live in LDS (a whole 64x120 map + its g_p
    // were 62 KB: two workgroups per CU; a quarter of it is nine).
    const int r0 = band * hs / g.bands, r1 = (band + 1) * hs / g.bands;
    const int y0 = band * h / g.bands, y1 = (band + 1) * h / g.bands;
    const int ia = (int)(((unsigned)y0 * (unsigned)hs) / (unsigned)h);
    const int ib = y1 > y0 ? min(hs - 1, (int)(((unsigned)(y1 - 1) * (unsigned)hs) / (unsigned)h) + 2) : ia - 1;
    const int lo = min(r0, ia - 1), hi = max(r1 - 1, ib + 1);            // actual rows of g_e held (-1 and hs are the zero halo)
    const int nge = min(hi - lo + 1, g.ger[bi]);                          // (the host sized ger / gpr from the same expressions)
    float* GE = smem;                                  // [nge][(ws + 2)], zero halo; local row = actual row - lo
    float* GP = smem + (size_t)g.ger[bi] * WS2;        // [ib - ia + 1][ws]; local row = i - ia
    // ---- g_e rows of the band -> LDS
    const float* gep = g.ge[bi] + (size_t)plane * hs * ws;
    for (int i = tid; i < nge * WS2; i += 256) {
        const int lr = i / WS2, q = i - lr * WS2;
        const int r = lo + lr;
        const bool in = r >= 0 && r < hs && q >= 1 && q <= ws;
        GE[i] = in ? gep[r * ws + (q - 1)] : 0.f;
    }
    const float* w9 = g.wts[bi] + (size_t)c * 9;
    const float w00 = w9[0], w01 = w9[1], w02 = w9[2], w10 = w9[3], w11 = w9[4], w12 = w9[5], w20 = w9[6], w21 = w9[7], w22 = w9[8];
    __syncthreads();
    // ---- weight gradient: this band's share of the low-resolution rows
    {
        const float* pp = g.pooled[bi] + (size_t)plane * hs * ws;
        float s[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) s[k] = 0.f;
        const int cnt = (r1 - r0) * ws;
        for (int i = tid; i < cnt; i += 256) {
            const int u = r0 + i / ws, v = i - (i / ws) * ws;
            const float gv = GE[(u - lo) * WS2 + v + 1];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int y = u + ky - 1, x = v + kx - 1;
                    const float pv = (y >= 0 && y < hs && x >= 0 && x < ws) ? pp[y * ws + x] : 0.f;
                    s[ky * 3 + kx] = fmaf(gv, pv, s[ky * 3 + kx]);
                }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            s[k] = wave_sum_dpp(s[k]);                        // total in lane 63
            if ((tid & 63) == 63) red[k][tid >> 6] = s[k];
        }
        __syncthreads();
        if (tid < 9 && cnt > 0) atomicAdd(g.gw[bi] + (size_t)c * 9 + tid, (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]));
    }
    // ---- g_p of the low-resolution rows this band's full-resolution rows can touch -> LDS (evaluated once per cell, not once per
    // full-resolution pixel: a 2x2 window has four pixels, the 0.1-scale windows ~100):
    // g_p[i, j] = sum w[ky][kx] * g_e[i - ky + 1, j - kx + 1]: rows i - 1 .. i + 1 of g_e = local rows i - 1 - lo ..
    {
        const int ngp = min(ib - ia + 1, g.gpr[bi]);
        for (int t = tid; t < ngp * ws; t += 256) {
            const int li = t / ws, j = t - li * ws;
            const float* q = GE + (ia + li - 1 - lo) * WS2 + j;      // q[dy * WS2 + dx] = g_e[i - 1 + dy][j - 1 + dx]
            float v = w00 * q[2 * WS2 + 2];
            v = fmaf(w01, q[2 * WS2 + 1], v);  v = fmaf(w02, q[2 * WS2], v);
            v = fmaf(w10, q[WS2 + 2], v);      v = fmaf(w11, q[WS2 + 1], v);  v = fmaf(w12, q[WS2], v);
            v = fmaf(w20, q[2], v);            v = fmaf(w21, q[1], v);        v = fmaf(w22, q[0], v);
            GP[li * ws + j] = v;
        }
    }
    // Which windows contain a row / a column: ATen's windows overlap when the sizes do not divide, so an index sits in up to two
    // (three allowed for) of them.  Rows: a small LDS table built once per workgroup (one thread per row: the divisions are not
    // repeated per pixel); columns: per-thread registers (a thread keeps its column while it walks the band's rows).
    int* RT = reinterpret_cast<int*>(GP + (size_t)g.gpr[bi] * ws);   // [y1 - y0][4]: first window, then 1 / height of up to three windows (0 = not inside)
    for (int r = tid; r < y1 - y0; r += 256) {
        const int y = y0 + r;
        const int i0 = (int)(((unsigned)y * (unsigned)hs) / (unsigned)h);
        RT[4 * r] = i0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i = i0 + k;
            float rh = 0.f;
            if (i < hs) { const int ys = pdb_s(i, h, hs), ye = pdb_e(i, h, hs); if (y >= ys && y < ye) rh = 1.0f / (float)(ye - ys); }
            RT[4 * r + 1 + k] = __float_as_int(rh);
        }
    }
    __syncthreads();
    // (1 / height per row in the table, 1 / width per thread: a pixel costs nine LDS reads and twelve multiply-adds; dividing every term by
    // (height * width) behind two nested conditions was ~10 instructions per term: 75 -> 65 us on the 128x240 map at batch 16; with the
    // band-local LDS footprint and eight bands 39 us)
    float* gxp = g.gx[bi] + (size_t)plane * h * w;
    for (int x = tid; x < w; x += 256) {
        const int j0 = (int)(((unsigned)x * (unsigned)ws) / (unsigned)w);
        float rw[3];
        int jc[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int j = j0 + k;
            rw[k] = 0.f;
            jc[k] = min(j, ws - 1);                      // (a window that does not exist reads a finite neighbour with weight 0)
            if (j < ws) { const int xs = pdb_s(j, w, ws), xe = pdb_e(j, w, ws); if (x >= xs && x < xe) rw[k] = 1.0f / (float)(xe - xs); }
        }
        for (int r = 0; r < y1 - y0; ++r) {             // uniform
            const int i0 = RT[4 * r];
            float acc = 0.f;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float rh = __int_as_float(RT[4 * r + 1 + a]);       // uniform
                if (rh == 0.f) continue;                // (rows beyond the evaluated ones are never read)
                const float* gp = GP + (i0 + a - ia) * ws;
                const float rowv = fmaf(rw[2], gp[jc[2]], fmaf(rw[1], gp[jc[1]], rw[0] * gp[jc[0]]));
                acc = fmaf(rh, rowv, acc);
            }
            gxp[(size_t)(y0 + r) * w + x] = acc;
        }
    }
}

}  // namespace mspl

using namespace mspl;

extern "C" int mspl_pyr_down_mid_bwd(const float* const* g_e, const float* const* pooled, const float* const* stage_w, int32_t N,
                                     int32_t P, int32_t h, int32_t w, int32_t nb, const int32_t* hs, const int32_t* ws,
                                     float* const* gw, float* const* gx, void* stream) {
    MSPL_REQUIRE(g_e && pooled && stage_w && hs && ws && gw && gx, MSPL_ERR_NULL_POINTER, "pyr_down_mid_bwd: null pointer");
    MSPL_REQUIRE(N > 0 && P > 0 && h > 0 && w > 0, MSPL_ERR_BAD_SHAPE, "pyr_down_mid_bwd: bad shape N=%d P=%d %dx%d", N, P, h, w);
    MSPL_REQUIRE(nb >= 1 && nb <= PDB_MAXB, MSPL_ERR_UNSUPPORTED, "pyr_down_mid_bwd: %d branches (1..%d)", nb, PDB_MAXB);
    MSPL_REQUIRE((int64_t)N * P < 65536ll * 32768ll && (int64_t)h * w < (1ll << 30), MSPL_ERR_BAD_SHAPE, "pyr_down_mid_bwd: map too large");
    PdbGeom g;
    memset(&g, 0, sizeof(g));
    g.N = N; g.P = P; g.h = h; g.w = w; g.nb = nb;
    size_t lds = 0;
    for (int i = 0; i < nb; ++i) {
        MSPL_REQUIRE(hs[i] > 0 && ws[i] > 0 && hs[i] <= h && ws[i] <= w, MSPL_ERR_BAD_SHAPE,
                     "pyr_down_mid_bwd: branch %d size %dx%d for a %dx%d map", i, hs[i], ws[i], h, w);
        MSPL_REQUIRE(g_e[i] && pooled[i] && stage_w[i] && gw[i] && gx[i], MSPL_ERR_NULL_POINTER, "pyr_down_mid_bwd: branch %d has a null pointer", i);
        // (the window arithmetic multiplies sizes in 32 bits)
        MSPL_REQUIRE((int64_t)(h + 1) * hs[i] < (1ll << 31) && (int64_t)(w + 1) * ws[i] < (1ll << 31), MSPL_ERR_BAD_SHAPE,
                     "pyr_down_mid_bwd: map too large for the 32-bit window arithmetic");
        g.hs[i] = hs[i]; g.ws[i] = ws[i]; g.ge[i] = g_e[i]; g.pooled[i] = pooled[i]; g.wts[i] = stage_w[i]; g.gw[i] = gw[i]; g.gx[i] = gx[i];
    }
    // bands of full-resolution rows per plane: enough workgroups to fill the chip several times over (a band stages only the
    // low-resolution rows it needs)
    const int64_t planes = (int64_t)N * P;
    int bands = 1;
    // (>= 16 full-resolution rows per band: on the 16- and 32-row maps of levels 3-4 more bands only multiply the staging and the atomics:
    // 10.8 -> 15.3 us, 7.6 -> 14.1 us with 2-row bands)
    while (bands < 16 && planes * nb * bands < 4096 && 32 * bands <= h) bands *= 2;
    int rows_max = 0;
    for (int i = 0; i < nb; ++i) {
        int ger = 1, gpr = 1;
        for (int b = 0; b < bands; ++b) {               // the kernel's own expressions
            const int r0 = b * hs[i] / bands, r1 = (b + 1) * hs[i] / bands;
            const int y0 = b * h / bands, y1 = (b + 1) * h / bands;
            const int ia = (int)(((unsigned)y0 * (unsigned)hs[i]) / (unsigned)h);
            const int ib = y1 > y0 ? std::min(hs[i] - 1, (int)(((unsigned)(y1 - 1) * (unsigned)hs[i]) / (unsigned)h) + 2) : ia - 1;
            ger = std::max(ger, std::max(r1 - 1, ib + 1) - std::min(r0, ia - 1) + 1);
            gpr = std::max(gpr, ib - ia + 1);
            rows_max = std::max(rows_max, y1 - y0);
        }
        g.ger[i] = ger; g.gpr[i] = gpr;
        const size_t bts = ((size_t)ger * (ws[i] + 2) + (size_t)gpr * ws[i]) * sizeof(float);      // g_e rows (haloed columns) + g_p rows
        if (bts > lds) lds = bts;
    }
    lds += (size_t)rows_max * 4 * sizeof(int);       // the row table of a band
    static std::once_flag once;
    static bool attr_ok = false;
    std::call_once(once, [] { attr_ok = hipFuncSetAttribute((const void*)pyr_down_mid_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess; });
    MSPL_REQUIRE(lds <= (attr_ok ? 128 : 64) * 1024, MSPL_ERR_UNSUPPORTED, "pyr_down_mid_bwd: a %zu-byte low-resolution map does not fit the workgroup's LDS", lds);
    g.bands = bands;
    MSPL_REQUIRE(planes < (1ll << 31), MSPL_ERR_BAD_SHAPE, "pyr_down_mid_bwd: too many planes");
    hipLaunchKernelGGL(pyr_down_mid_bwd_kernel, dim3((unsigned)planes, (unsigned)nb, (unsigned)bands), dim3(256), lds, (hipStream_t)stream, g);
    MSPL_CHECK_LAUNCH("pyr_down_mid_bwd");
    return MSPL_OK;
}
