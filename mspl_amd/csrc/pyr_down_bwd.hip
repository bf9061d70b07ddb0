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
    float* GE = smem;                                  // [(hs + 2)][(ws + 2)], zero halo
    __shared__ float red[9][4];
    // ---- g_e of the plane -> LDS
    const float* gep = g.ge[bi] + (size_t)plane * hs * ws;
    for (int i = tid; i < (hs + 2) * WS2; i += 256) {
        const int r = i / WS2, q = i - r * WS2;
        const bool in = r >= 1 && r <= hs && q >= 1 && q <= ws;
        GE[i] = in ? gep[(r - 1) * ws + (q - 1)] : 0.f;
    }
    const float* w9 = g.wts[bi] + (size_t)c * 9;
    const float w00 = w9[0], w01 = w9[1], w02 = w9[2], w10 = w9[3], w11 = w9[4], w12 = w9[5], w20 = w9[6], w21 = w9[7], w22 = w9[8];
    __syncthreads();
    // ---- weight gradient: this band's share of the low-resolution rows
    {
        const int r0 = band * hs / g.bands, r1 = (band + 1) * hs / g.bands;
        const float* pp = g.pooled[bi] + (size_t)plane * hs * ws;
        float s[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) s[k] = 0.f;
        const int cnt = (r1 - r0) * ws;
        for (int i = tid; i < cnt; i += 256) {
            const int u = r0 + i / ws, v = i - (i / ws) * ws;
            const float gv = GE[(u + 1) * WS2 + v + 1];
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
    // g_p[i, j] = sum w[ky][kx] * g_e[i - ky + 1, j - kx + 1] = sum w[ky][kx] * GE[i + 2 - ky][j + 2 - kx]
    float* GP = smem + (size_t)(hs + 2) * WS2;         // [hs][ws]
    const int y0 = band * h / g.bands, y1 = (band + 1) * h / g.bands;
    {
        const int ia = (int)(((unsigned)y0 * (unsigned)hs) / (unsigned)h);
        const int ib = y1 > y0 ? min(hs - 1, (int)(((unsigned)(y1 - 1) * (unsigned)hs) / (unsigned)h) + 2) : ia - 1;
        for (int t = tid; t < (ib - ia + 1) * ws; t += 256) {
            const int i = ia + t / ws, j = t - (t / ws) * ws;
            const float* q = GE + i * WS2 + j;         // GE[i + dy][j + dx], dy, dx in 0..2
            float v = w00 * q[2 * WS2 + 2];
            v = fmaf(w01, q[2 * WS2 + 1], v);  v = fmaf(w02, q[2 * WS2], v);
            v = fmaf(w10, q[WS2 + 2], v);      v = fmaf(w11, q[WS2 + 1], v);  v = fmaf(w12, q[WS2], v);
            v = fmaf(w20, q[2], v);            v = fmaf(w21, q[1], v);        v = fmaf(w22, q[0], v);
            GP[i * ws + j] = v;
        }
    }
    // Which windows contain a row / a column: ATen's windows overlap when the sizes do not divide, so an index sits in up to two
    // (three allowed for) of them.  Rows: a small LDS table built once per workgroup (one thread per row: the divisions are not
    // repeated per pixel); columns: per-thread registers (a thread keeps its column while it walks the band's rows).
    int* RT = reinterpret_cast<int*>(GP + (size_t)hs * ws);         // [y1 - y0][4]: first window, then up to three heights (0 = not inside)
    for (int r = tid; r < y1 - y0; r += 256) {
        const int y = y0 + r;
        const int i0 = (int)(((unsigned)y * (unsigned)hs) / (unsigned)h);
        RT[4 * r] = i0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int i = i0 + k;
            int hh = 0;
            if (i < hs) { const int ys = pdb_s(i, h, hs), ye = pdb_e(i, h, hs); if (y >= ys && y < ye) hh = ye - ys; }
            RT[4 * r + 1 + k] = hh;
        }
    }
    __syncthreads();
    float* gxp = g.gx[bi] + (size_t)plane * h * w;
    for (int x = tid; x < w; x += 256) {
        const int j0 = (int)(((unsigned)x * (unsigned)ws) / (unsigned)w);
        int jw[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int j = j0 + k;
            jw[k] = 0;
            if (j < ws) { const int xs = pdb_s(j, w, ws), xe = pdb_e(j, w, ws); if (x >= xs && x < xe) jw[k] = xe - xs; }
        }
        for (int r = 0; r < y1 - y0; ++r) {             // uniform
            const int i0 = RT[4 * r];
            float acc = 0.f;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const int hh = RT[4 * r + 1 + a];       // uniform
                if (hh == 0) continue;
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    if (jw[k]) acc += GP[(i0 + a) * ws + j0 + k] / (float)(hh * jw[k]);
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
        const size_t b = ((size_t)(hs[i] + 2) * (ws[i] + 2) + (size_t)hs[i] * ws[i]) * sizeof(float);      // g_e (haloed) + g_p
        if (b > lds) lds = b;
    }
    lds += (size_t)h * 4 * sizeof(int);              // the row table of a band (<= h rows)
    static std::once_flag once;
    static bool attr_ok = false;
    std::call_once(once, [] { attr_ok = hipFuncSetAttribute((const void*)pyr_down_mid_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess; });
    MSPL_REQUIRE(lds <= (attr_ok ? 128 : 64) * 1024, MSPL_ERR_UNSUPPORTED, "pyr_down_mid_bwd: a %zu-byte low-resolution map does not fit the workgroup's LDS", lds);
    // bands of full-resolution rows per plane: enough workgroups to fill the chip (a plane's g_e is re-read per band: <= 36 KB from L2)
    const int64_t planes = (int64_t)N * P;
    int bands = 1;
    while (bands < 16 && planes * nb * bands < 1024 && 2 * bands <= h) bands *= 2;
    g.bands = bands;
    MSPL_REQUIRE(planes < (1ll << 31), MSPL_ERR_BAD_SHAPE, "pyr_down_mid_bwd: too many planes");
    hipLaunchKernelGGL(pyr_down_mid_bwd_kernel, dim3((unsigned)planes, (unsigned)nb, (unsigned)bands), dim3(256), lds, (hipStream_t)stream, g);
    MSPL_CHECK_LAUNCH("pyr_down_mid_bwd");
    return MSPL_OK;
}
