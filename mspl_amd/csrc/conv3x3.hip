// Generic grouped 3x3 convolution (padding 1, dilation 1, stride 1|2), direct and LDS-tiled, with the
// shared fused epilogue.  Covers the small-K spatial convolutions of the path:
//   level1 stem 3->32 s2            model/classification/espnetv2.py:61
//   inp_reinf.0 3->3                nn_layers/eesp.py:118
//   pyramid depthwise stages        nn_layers/efficient_pyramid_pool.py:24
//   merge CBR (groups = proj)       nn_layers/efficient_pyramid_pool.py:30  (reads through Shuffle, :29)
//   EfficientPWConv expansion       nn_layers/efficient_pt.py:21
// A workgroup stages the cin_g input planes of one (image, group, spatial tile) in LDS (coalesced rows,
// zero-filled halo); a thread produces a 1x4 output strip for COB output channels, so every LDS row
// window is reused COB*3 times from registers.  Weight reads are LDS broadcasts.
#include <stdlib.h>

#include <algorithm>

#include "common.hpp"

namespace mspl {

struct C3Geom {
    int N, Cin, Cout, G, cin_g, cout_g, H, W, Ho, Wo, sg;
    int TH, TW;        // output tile
    int tiles_y, tiles_x;
    int IH, IWS;       // staged input rows / LDS row stride (floats, multiple of 4)
    int XS;            // TW / 4 strips per tile row
    int coblks;        // cout_g / COB
    int nvec;          // 16-byte chunks per staged row
    unsigned mag_nvec, mag_ih;   // ceil(2^32 / d): exact division of small counts by mul-hi
    unsigned xcd_per, total;     // XCD-contiguous tile order (common.hpp): neighbouring tiles share halo rows
    int fast;                    // W % 4 == 0, no Shuffle, 16-byte-aligned x: lean staging (2: stride 1 and one tile per row)
    unsigned mag_nv;             // ceil(2^32 / (W / 4))
};

// UNIW: one output-channel block per group (cout_g == COB): the weights a thread needs are the same for the whole workgroup, so
// they are read through a uniform address (scalar loads, SGPR operands of the FMAs) instead of 27 LDS broadcasts per input
// channel and strip -- for the 8 -> 3 grouped expansion that was 216 ds_read_b32 per 864 FMAs.
template <int STRIDE, int COB, bool UNIW>
__global__ __launch_bounds__(256) void conv3x3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      C3Geom g, Epi e, float* __restrict__ out) {
    constexpr int NR = (STRIDE == 1) ? 6 : 9;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tile = smem;                                         // cin_g * IH * IWS
    float* wl = smem + (size_t)g.cin_g * g.IH * g.IWS;          // cout_g * cin_g * 9

    int bid = (int)xcd_contiguous(blockIdx.x, g.xcd_per);
    if ((unsigned)bid >= g.total) return;
    const int txi = bid % g.tiles_x;  bid /= g.tiles_x;
    const int tyi = bid % g.tiles_y;  bid /= g.tiles_y;
    const int grp = bid % g.G;
    const int img = bid / g.G;
    const int oy0 = tyi * g.TH, ox0 = txi * g.TW;
    const int iy0 = oy0 * STRIDE - 1, ix0 = ox0 * STRIDE - 1;   // input coords of LDS (0,0)
    const int tid = threadIdx.x;

    // the group's weights: the first 1024 are requested into registers NOW and written to LDS after the tile (their
    // latency then hides under the tile's HBM round trip instead of preceding it); any rest goes the plain way
    const int nw = g.cout_g * g.cin_g * 9;
    const float* wg = w + (size_t)grp * nw;
    float wreg[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) wreg[u] = (tid + 256 * u < nw) ? wg[tid + 256 * u] : 0.f;

    // stage input planes; LDS (r, j) <-> input (iy0 + r, ix0 + j).  The (plane, row, 16-byte chunk) space is walked
    // flat by all 256 threads with UL independent global loads in flight per thread before the first LDS write
    // (aligned 16-byte loads when W % 4 == 0, scalar LDS writes since ix0 = -1 mod 4, zero fill outside the image).
    const int per_plane = g.IH * g.IWS;
    if (STRIDE == 1 && g.fast == 2) {
        // Lean staging, whole-row tiles (stride 1, TW == W): LDS column j <-> input column j - 1, W / 4 chunks per row, four
        // unguarded LDS stores per chunk, the four padding columns of a row written separately.  128 -> 48 in 16 groups at 16 x 72x120:
        // 41.3 us (general walk) -> 40.6 (the guarded lean form below) -> 35.6 us (this form with 8-row tiles).
        constexpr int UL = 5;
        const int nv = g.W >> 2;
        const int rows = g.cin_g * g.IH;
        const int total = rows * nv;
        const float* xg = x + ((size_t)img * g.Cin + (size_t)grp * g.cin_g) * (size_t)g.H * g.W;
        for (int base = 0; base < total; base += 256 * UL) {
            float4 t4[UL];
            int dst[UL];
            bool in[UL];
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                const int i = min(base + u * 256 + tid, total - 1);      // the last chunk is re-written by the spare threads
                const int rr = (int)__umulhi((unsigned)i, g.mag_nv), v = i - rr * nv;
                const int ci = (int)__umulhi((unsigned)rr, g.mag_ih), r = rr - ci * g.IH;
                const int iy = iy0 + r;
                in[u] = iy >= 0 && iy < g.H;
                const int iyc = min(max(iy, 0), g.H - 1);
                t4[u] = *reinterpret_cast<const float4*>(xg + (unsigned)((ci * g.H + iyc) * g.W + 4 * v));   // < 2^29 (host check)
                dst[u] = rr * g.IWS + 4 * v + 1;
            }
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                float* d = tile + dst[u];
                d[0] = in[u] ? t4[u].x : 0.f;  d[1] = in[u] ? t4[u].y : 0.f;
                d[2] = in[u] ? t4[u].z : 0.f;  d[3] = in[u] ? t4[u].w : 0.f;
            }
        }
        // padding columns -1 and W .. W + 2 (LDS 0, W + 1 .. W + 3: the strips read 8 floats from 4 * xs)
        for (int i = tid; i < rows * 4; i += 256) {
            const int rr = i >> 2, q = i & 3;
            tile[rr * g.IWS + (q == 0 ? 0 : g.W + q)] = 0.f;
        }
    } else if (g.fast) {
        // Lean staging (uniform choice; W % 4 == 0, no Shuffle, 16-byte-aligned planes): a staged row is a whole image row or padding,
        // a 16-byte chunk lies wholly inside or outside the row (ix0 = -1 mod 4), so a chunk is one aligned load or zeros.  The
        // general walk below costs ~100 vector instructions per chunk (four guarded scalar loads, 64-bit addresses, the Shuffle index
        // division).  Stem 3 -> 32 stride 2 at 16 x 288x480: 35.8 -> 33.5 us.
        constexpr int UL = 5;
        const int c_lo = ix0 - 3;
        const int total = g.cin_g * g.IH * g.nvec;
        const float* xg = x + ((size_t)img * g.Cin + (size_t)grp * g.cin_g) * (size_t)g.H * g.W;
        for (int base = 0; base < total; base += 256 * UL) {
            float4 t4[UL];
            int dst[UL], jcol[UL];
            bool in[UL];
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                const int i = min(base + u * 256 + tid, total - 1);      // the last chunk is re-written by the spare threads
                const int rr = (int)__umulhi((unsigned)i, g.mag_nvec), v = i - rr * g.nvec;
                const int ci = (int)__umulhi((unsigned)rr, g.mag_ih), r = rr - ci * g.IH;
                const int iy = iy0 + r, c0 = c_lo + 4 * v;
                in[u] = iy >= 0 && iy < g.H && c0 >= 0 && c0 < g.W;
                const int iyc = min(max(iy, 0), g.H - 1), c0c = min(max(c0, 0), g.W - 4);
                t4[u] = *reinterpret_cast<const float4*>(xg + (unsigned)((ci * g.H + iyc) * g.W + c0c));   // < 2^29 (host check)
                jcol[u] = 4 * v - 3;
                dst[u] = rr * g.IWS + jcol[u];
            }
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                const float e0 = in[u] ? t4[u].x : 0.f, e1 = in[u] ? t4[u].y : 0.f, e2 = in[u] ? t4[u].z : 0.f, e3 = in[u] ? t4[u].w : 0.f;
                float* d = tile + dst[u];
                const int jq = jcol[u];                                   // LDS column of element 0: 4 v - 3
                if (jq >= 0 && jq < g.IWS) d[0] = e0;
                if (jq + 1 >= 0 && jq + 1 < g.IWS) d[1] = e1;
                if (jq + 2 >= 0 && jq + 2 < g.IWS) d[2] = e2;
                if (jq + 3 < g.IWS) d[3] = e3;
            }
        }
    } else {
        constexpr int UL = 4;
        const int c_lo = ix0 - 3;                    // ix0 = 4k - 1  ->  floor4(ix0) = ix0 - 3
        const int total = g.cin_g * g.IH * g.nvec;
        const bool w4 = (g.W & 3) == 0;
        for (int base = 0; base < total; base += 256 * UL) {
            float e4[UL][4];
            int dbase[UL], c0s[UL];
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                const int i = base + u * 256 + tid;
                e4[u][0] = e4[u][1] = e4[u][2] = e4[u][3] = 0.f;
                dbase[u] = -1;  c0s[u] = 0;
                if (i < total) {
                    const int rr = (int)__umulhi((unsigned)i, g.mag_nvec), v = i - rr * g.nvec;
                    const int ci = (int)__umulhi((unsigned)rr, g.mag_ih), r = rr - ci * g.IH;
                    const int iy = iy0 + r;
                    const int c0 = c_lo + 4 * v;
                    dbase[u] = rr * g.IWS;  c0s[u] = c0;
                    if (iy >= 0 && iy < g.H) {
                        int lc = grp * g.cin_g + ci;
                        if (g.sg > 0) lc = (lc % g.sg) * (g.Cin / g.sg) + lc / g.sg;
                        const float* src = x + (((size_t)img * g.Cin + lc) * g.H + iy) * (size_t)g.W;
                        if (w4 && c0 >= 0 && c0 + 3 < g.W) {
                            const float4 t4 = *reinterpret_cast<const float4*>(src + c0);
                            e4[u][0] = t4.x; e4[u][1] = t4.y; e4[u][2] = t4.z; e4[u][3] = t4.w;
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                if (c0 + q >= 0 && c0 + q < g.W) e4[u][q] = src[c0 + q];
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                if (dbase[u] < 0) continue;
                float* dst = tile + dbase[u];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int j = c0s[u] + q - ix0;
                    if (j >= 0 && j < g.IWS) dst[j] = e4[u][q];
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (tid + 256 * u < nw) wl[tid + 256 * u] = wreg[u];
    for (int i = tid + 1024; i < nw; i += 256) wl[i] = wg[i];
    // per-channel epilogue constants of the group -> LDS (read from global inside the item loop they were one dependent
    // round trip per output channel and item)
    float* cl = wl + nw;                                         // [cout_g][6] = scale, shift, alpha, rw0, rw1, rw2
    for (int c = tid; c < g.cout_g; c += 256) {
        const EpiCh ec = epi_channel(e, e.coff + grp * g.cout_g + c);
        float* d = cl + c * 6;
        d[0] = ec.scale; d[1] = ec.shift; d[2] = ec.alpha; d[3] = ec.rw0; d[4] = ec.rw1; d[5] = ec.rw2;
    }
    __syncthreads();

    const int rows_here = min(g.TH, g.Ho - oy0);
    const int items = g.coblks * rows_here * g.XS;
    const int hw = g.Ho * g.Wo;
    for (int it = tid; it < items; it += 256) {
        const int xs = it % g.XS;
        const int t2 = it / g.XS;
        const int ty = t2 % rows_here;
        const int cb = t2 / rows_here;
        const int xb = ox0 + xs * 4;
        if (xb >= g.Wo) continue;
        float acc[COB][4];
#pragma unroll
        for (int c = 0; c < COB; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[c][j] = 0.f;
        for (int ci = 0; ci < g.cin_g; ++ci) {
            const float* lp = tile + (size_t)ci * per_plane + (size_t)(ty * STRIDE) * g.IWS + xs * 4 * STRIDE;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                float rv[NR + 3];
                const float* row = lp + ky * g.IWS;
                const float4 a = *reinterpret_cast<const float4*>(row);
                const float4 b = *reinterpret_cast<const float4*>(row + 4);
                rv[0] = a.x; rv[1] = a.y; rv[2] = a.z; rv[3] = a.w;
                rv[4] = b.x; rv[5] = b.y; rv[6] = b.z; rv[7] = b.w;
                if (STRIDE == 2) rv[8] = row[8];
#pragma unroll
                for (int c = 0; c < COB; ++c) {
                    const float* wp = UNIW ? wg + ((size_t)c * g.cin_g + ci) * 9 + ky * 3
                                           : wl + ((size_t)(cb * COB + c) * g.cin_g + ci) * 9 + ky * 3;
                    const float w0 = wp[0], w1 = wp[1], w2 = wp[2];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[c][j] = fmaf(w0, rv[j * STRIDE], acc[c][j]);
                        acc[c][j] = fmaf(w1, rv[j * STRIDE + 1], acc[c][j]);
                        acc[c][j] = fmaf(w2, rv[j * STRIDE + 2], acc[c][j]);
                    }
                }
            }
        }
        const int y = oy0 + ty;
        const int pix = y * g.Wo + xb;
#pragma unroll
        for (int c = 0; c < COB; ++c) {
            const int cabs = e.coff + grp * g.cout_g + cb * COB + c;
            const float* cc = cl + (cb * COB + c) * 6;
            const EpiCh ec = {cc[0], cc[1], cc[2], cc[3], cc[4], cc[5]};
            float* dst = out + ((size_t)img * e.ctot + cabs) * (size_t)hw + pix;
            if (e.raw) {                                      // training forward: the bare convolution result as well
                float* rd = e.raw + ((size_t)img * e.ctot + cabs) * (size_t)hw + pix;
                if (((g.Wo & 3) == 0)) store_out4(rd, make_float4(acc[c][0], acc[c][1], acc[c][2], acc[c][3]));
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (xb + j < g.Wo) rd[j] = acc[c][j];
                }
            }
            if (((g.Wo & 3) == 0)) {
                store_out4(dst, epi_apply4(e, ec, acc[c], img, cabs, pix));
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (xb + j < g.Wo) dst[j] = epi_apply(e, ec, acc[c][j], img, cabs, pix + j);
            }
        }
    }
}


// ------------------------------------------------------------------ depthwise 3x3, stride 1: register-streaming form
// (EfficientPWConv's expansion over the 32-channel level-1 map, nn_layers/efficient_pt.py:21 with groups = gcd(nin, nout) =
// channels.)  The LDS-tiled kernel above spends ~85 vector instructions per output pixel on a depthwise plane (staging and
// index arithmetic for 9 FMAs); the whole label pass is bound by vector-instruction issue, not by HBM (344 M wave-instructions
// per pass).  Here a wave owns a column block and a row segment of one plane and walks down the rows: lane
// = 4 adjacent columns (one 16-byte load per row), the left / right neighbour columns come from the neighbouring lanes
// (v_mov_dpp wave_shr / wave_shl; lanes 0 and 63 are halo lanes), the 3-row window lives in registers, the next row is
// requested one row ahead.  ~20 vector instructions per pixel, no LDS, no barrier.
struct DwsGeom {
    int N, C, H, W;
    int SEG, nseg, ncb;
    unsigned total;      // waves
};

__device__ __forceinline__ float dws_from_left(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dws_from_right(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}

__global__ __launch_bounds__(256) void dwconv3x3_stream_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               DwsGeom g, Epi e, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    unsigned wid = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (wid >= g.total) return;                              // wave-uniform; no barrier in this kernel
    const int sgi = wid % g.nseg;  wid /= g.nseg;
    const int cb = wid % g.ncb;  wid /= g.ncb;
    const int c = wid % g.C;
    const int n = wid / g.C;
    const int H = g.H, W = g.W;
    const int ys = sgi * g.SEG, ye = min(ys + g.SEG, H);
    const int c0 = cb * 248 + (lane - 1) * 4;
    const bool colin = c0 >= 0 && c0 < W;                    // W % 4 == 0: all four columns in or out
    const bool writer = lane >= 1 && lane <= 62 && colin;
    const float* xp = x + ((size_t)n * g.C + c) * (size_t)H * W;
    const float* wp = w + (size_t)c * 9;
    const float w00 = wp[0], w01 = wp[1], w02 = wp[2], w10 = wp[3], w11 = wp[4], w12 = wp[5], w20 = wp[6], w21 = wp[7], w22 = wp[8];
    const int cabs = e.coff + c;
    const EpiCh ec = epi_channel(e, cabs);
    auto load_row = [&](int r) -> float4 {
        if (!colin || r < 0 || r >= H) return make_float4(0.f, 0.f, 0.f, 0.f);
        return *reinterpret_cast<const float4*>(xp + (size_t)r * W + c0);
    };
    // rows y-1, y, y+1 as six-column windows (columns c0-1 .. c0+4)
    float r0[6], r1[6], r2[6];
    auto widen = [&](const float4& v, float (&r)[6]) {
        r[1] = v.x; r[2] = v.y; r[3] = v.z; r[4] = v.w;
        r[0] = dws_from_left(v.w);
        r[5] = dws_from_right(v.x);
    };
    widen(load_row(ys - 1), r0);
    widen(load_row(ys), r1);
    float4 nxt = load_row(ys + 1);
    float* op = out + ((size_t)n * e.ctot + cabs) * (size_t)e.hw;
#pragma unroll 1
    for (int y = ys; y < ye; ++y) {
        widen(nxt, r2);
        nxt = load_row(y + 2);
        float acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = 0.f;
            a = fmaf(w00, r0[j], a); a = fmaf(w01, r0[j + 1], a); a = fmaf(w02, r0[j + 2], a);
            a = fmaf(w10, r1[j], a); a = fmaf(w11, r1[j + 1], a); a = fmaf(w12, r1[j + 2], a);
            a = fmaf(w20, r2[j], a); a = fmaf(w21, r2[j + 1], a); a = fmaf(w22, r2[j + 2], a);
            acc[j] = a;
        }
        if (writer) {
            const int pix = y * W + c0;
            if (e.raw) store_out4(e.raw + (op - out) + pix, make_float4(acc[0], acc[1], acc[2], acc[3]));
            store_out4(op + pix, epi_apply4(e, ec, acc, n, cabs, pix));
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) { r0[j] = r1[j]; r1[j] = r2[j]; }
    }
}

static bool dwconv3x3_stream_ok(const C3Geom& g, const float* x, const float* out, const Epi& e) {
    auto al16 = [](const void* p) { return p == nullptr || (((uintptr_t)p) & 15) == 0; };
    return g.cin_g == 1 && g.cout_g == 1 && g.sg == 0 && (g.W & 3) == 0 && g.W >= 64 && g.H >= 8 && al16(x) && al16(out) &&
           al16(e.pre_add) && al16(e.residual) && al16(e.reinf_r) && (int64_t)g.H * g.W < (1ll << 29);
}

static int launch_dws(const float* x, const float* w, const C3Geom& g3, const Epi& e, float* out, hipStream_t s) {
    DwsGeom g;
    g.N = g3.N; g.C = g3.Cin; g.H = g3.H; g.W = g3.W;
    g.ncb = ceil_div(g.W, 248);
    int seg = g.H < 32 ? g.H : 32;
    while (seg > 8 && (int64_t)g.N * g.C * g.ncb * ceil_div(g.H, seg) < 4096) --seg;
    seg = ceil_div(g.H, ceil_div(g.H, seg));
    g.SEG = seg;
    g.nseg = ceil_div(g.H, seg);
    const int64_t waves = (int64_t)g.N * g.C * g.ncb * g.nseg;
    MSPL_REQUIRE(waves < (1ll << 31), MSPL_ERR_BAD_SHAPE, "conv3x3(depthwise): grid too large");
    g.total = (unsigned)waves;
    hipLaunchKernelGGL(dwconv3x3_stream_kernel, dim3((unsigned)ceil_div64(waves, 4)), dim3(256), 0, s, x, w, g, e, out);
    MSPL_CHECK_LAUNCH("conv3x3(depthwise, streaming)");
    return MSPL_OK;
}

// ------------------------------------------------------------------ grouped 3x3, stride 1, few channels per group: streaming form
// The PWConv expansion 256 -> 64 in 64 groups at 36x60 (30 us in the LDS-tiled kernel for 7 us of bytes) and its data gradient.  Same
// machine mapping as the depthwise kernel above, generalised: a lane owns 4 columns of one row, W / 4 lanes cover a row, and
// 64 / (W / 4) units (image, group, row segment) share a wave, all of the SAME group, so the group's CG * COB * 9 weights and
// the epilogue constants are wave-uniform (scalar loads).  An input row is read once (CG 16-byte loads per lane, halo columns from
// the neighbouring lanes by DPP) and contributes to the three output rows it touches, which are carried in three accumulator
// sets: no LDS, no barrier, no 3-row window in registers.  Summation order per output: kernel row, input channel, kernel column.
struct GcsGeom {
    int N, Cin, Cout, G, H, W, sg;
    int LPR, SUB;            // lanes per row (W / 4), units per wave (64 / LPR)
    int SEG, nseg;           // output rows per unit, units per (image, group)
    int upg, wpg;            // units and waves per group
    unsigned total;          // waves
};

template <int CG, int COB>
__global__ __launch_bounds__(256) void gconv3x3_stream_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              GcsGeom g, Epi e, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const unsigned wid = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (wid >= g.total) return;                              // wave-uniform; no barrier in this kernel
    const int grp = wid / g.wpg, wig = wid - grp * g.wpg;    // uniform
    const int sub = lane / g.LPR, cl = lane - sub * g.LPR;
    const int unit = wig * g.SUB + sub;                      // (image, segment) of this lane's unit
    const bool live = sub < g.SUB && unit < g.upg;
    const int uc = live ? unit : 0;
    const int img = uc / g.nseg, sgi = uc - img * g.nseg;
    const int H = g.H, W = g.W;
    const int ys = sgi * g.SEG, ye = min(ys + g.SEG, H);
    const bool lok = cl > 0, rok = cl < g.LPR - 1;           // neighbours inside the same row
    // input planes of the group (Shuffle-aware channel index, nn_layers/cnn_utils.py:109-125), uniform
    const float* xch[CG];
#pragma unroll
    for (int ci = 0; ci < CG; ++ci) {
        int lc = grp * CG + ci;
        if (g.sg > 0) lc = (lc % g.sg) * (g.Cin / g.sg) + lc / g.sg;
        xch[ci] = x + (size_t)lc * H * W;
    }
    const size_t img_off = (size_t)img * g.Cin * H * W + (size_t)cl * 4;        // per lane
    const float* wg = w + (size_t)grp * COB * CG * 9;
    auto load_row = [&](int r, float4 (&v)[CG]) {
        const bool in = live && r >= 0 && r < H;
        const size_t off = img_off + (size_t)min(max(r, 0), H - 1) * W;
#pragma unroll
        for (int ci = 0; ci < CG; ++ci) {
            const float4 t = *reinterpret_cast<const float4*>(xch[ci] + off);
            v[ci] = in ? t : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    float acc[3][COB][4];                                    // [0]: output row r-1, [1]: row r, [2]: row r+1 at input row r
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < COB; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[a][c][j] = 0.f;
    float4 cur[CG], nxt[CG];
    load_row(ys - 1, cur);
    const int hw = H * W;
#pragma unroll 1
    for (int t = 0; t < g.SEG + 2; ++t) {                    // uniform trip count; r = ys - 1 + t
        const int r = ys - 1 + t;
        load_row(r + 1, nxt);                                 // next row's loads fly during this row's arithmetic
#pragma unroll
        for (int ci = 0; ci < CG; ++ci) {
            float win[6];
            win[1] = cur[ci].x; win[2] = cur[ci].y; win[3] = cur[ci].z; win[4] = cur[ci].w;
            const float fl = dws_from_left(cur[ci].w), fr = dws_from_right(cur[ci].x);
            win[0] = lok ? fl : 0.f;
            win[5] = rok ? fr : 0.f;
#pragma unroll
            for (int a = 0; a < 3; ++a) {                     // accumulator a <-> kernel row 2 - a
                const int ky = 2 - a;
#pragma unroll
                for (int c = 0; c < COB; ++c) {
                    const float* wk = wg + ((size_t)c * CG + ci) * 9 + ky * 3;
                    const float w0 = wk[0], w1 = wk[1], w2 = wk[2];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[a][c][j] = fmaf(w0, win[j], acc[a][c][j]);
                        acc[a][c][j] = fmaf(w1, win[j + 1], acc[a][c][j]);
                        acc[a][c][j] = fmaf(w2, win[j + 2], acc[a][c][j]);
                    }
                }
            }
        }
        // output row y = r - 1 is complete
        const int y = r - 1;
        if (live && y >= ys && y < ye) {
            const int pix = y * W + cl * 4;
#pragma unroll
            for (int c = 0; c < COB; ++c) {
                const int cabs = e.coff + grp * COB + c;
                const EpiCh ec = epi_channel(e, cabs);
                const size_t o = ((size_t)img * e.ctot + cabs) * (size_t)hw + pix;
                if (e.raw) store_out4(e.raw + o, make_float4(acc[0][c][0], acc[0][c][1], acc[0][c][2], acc[0][c][3]));
                store_out4(out + o, epi_apply4(e, ec, acc[0][c], img, cabs, pix));
            }
        }
#pragma unroll
        for (int c = 0; c < COB; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[0][c][j] = acc[1][c][j]; acc[1][c][j] = acc[2][c][j]; acc[2][c][j] = 0.f; }
#pragma unroll
        for (int ci = 0; ci < CG; ++ci) cur[ci] = nxt[ci];
    }
}

// Returns MSPL_OK when launched, 1 when the shape is left to the LDS-tiled kernel.
static int gconv3x3_stream_try(const float* x, const float* w, const C3Geom& g3, const Epi& e, float* out, hipStream_t s) {
    static const int off = (MSPL_TUNE_INT("MSPL_GC3S", 1) == 0);
    auto al16 = [](const void* p) { return p == nullptr || (((uintptr_t)p) & 15) == 0; };
    if (off || (g3.W & 3) != 0 || g3.W < 16 || g3.W > 256 || g3.H < 4) return 1;
    if (!(al16(x) && al16(out) && al16(e.pre_add) && al16(e.residual) && al16(e.reinf_r) && al16(e.raw))) return 1;
    if ((int64_t)g3.N * g3.Cin * g3.H * g3.W >= (1ll << 31) || (int64_t)g3.N * e.ctot * g3.H * g3.W >= (1ll << 31)) return 1;
    GcsGeom g;
    g.N = g3.N; g.Cin = g3.Cin; g.Cout = g3.Cout; g.G = g3.G; g.H = g3.H; g.W = g3.W; g.sg = g3.sg;
    g.LPR = g.W / 4;
    g.SUB = 64 / g.LPR;
    // rows per unit: enough waves for >= 2 per SIMD when the map allows; each unit reads 2 halo rows
    int seg = std::min(g.H, 24);
    while (seg > 6 && (int64_t)g.G * ceil_div64((int64_t)g.N * ceil_div(g.H, seg), g.SUB) < 2048) --seg;
    seg = ceil_div(g.H, ceil_div(g.H, seg));
    g.SEG = seg;
    g.nseg = ceil_div(g.H, seg);
    g.upg = g.N * g.nseg;
    g.wpg = ceil_div(g.upg, g.SUB);
    const int64_t waves = (int64_t)g.G * g.wpg;
    if (waves >= (1ll << 31)) return 1;
    g.total = (unsigned)waves;
    const dim3 grid((unsigned)ceil_div64(waves, 4)), blk(256);
#define MSPL_GCS(A, B) hipLaunchKernelGGL((gconv3x3_stream_kernel<A, B>), grid, blk, 0, s, x, w, g, e, out)
    // measured (tools/c3_probe.py, batch 16): 256 -> 64 in 64 groups at 36x60: 24.5 -> 16.0 us.  NOT used for 8 -> 3 per group (128 -> 48
    // at 72x120: 63 us against 41 us tiled: 237 VGPRs, two waves per SIMD, the 864 FMAs per row and lane are not hidden) nor for
    // ungrouped 3 -> 3 (one group = too few waves: 18.8 against 6.0 us); those stay with the LDS-tiled kernel.
    if (g3.G < 16) return 1;
    if (g3.cin_g == 4 && g3.cout_g == 1) MSPL_GCS(4, 1);
    else if (g3.cin_g == 1 && g3.cout_g == 4) MSPL_GCS(1, 4);
    else return 1;
#undef MSPL_GCS
    MSPL_CHECK_LAUNCH("conv3x3(grouped, streaming)");
    return MSPL_OK;
}

template <int STRIDE>
static int launch3(const float* x, const float* w, C3Geom g, const Epi& e, float* out, hipStream_t s) {
    int cob = 1;
    if (g.cout_g % 8 == 0) cob = 8;
    else if (g.cout_g % 4 == 0) cob = 4;
    else if (g.cout_g % 3 == 0) cob = 3;
    else if (g.cout_g % 2 == 0) cob = 2;
    g.coblks = g.cout_g / cob;
    const int wo4 = (g.Wo + 3) & ~3;
    g.TW = wo4;
    if (wo4 > 128) {          // widest tile <= 128 columns that wastes the fewest columns in the last tile of a row
        int best = 64, best_waste = 1 << 30;
        for (int tw = 128; tw >= 64; tw -= 4) {
            const int waste = ceil_div(g.Wo, tw) * tw - g.Wo;
            if (waste < best_waste) { best_waste = waste; best = tw; }
        }
        g.TW = best;
    }
    g.XS = g.TW / 4;
    g.tiles_x = ceil_div(g.Wo, g.TW);
    g.IWS = (((g.TW - 1) * STRIDE + 3 + 3) & ~3) + 4;
    auto lds_of = [&](int th) {
        return ((size_t)g.cin_g * ((th - 1) * STRIDE + 3) * g.IWS + (size_t)g.cout_g * g.cin_g * 9 + (size_t)g.cout_g * 6) * sizeof(float);
    };
    g.fast = (g.W & 3) == 0 && g.W >= 8 && g.sg == 0 && (((uintptr_t)x) & 15) == 0 && (int64_t)g.cin_g * g.H * g.W < (1ll << 29);
    if (g.fast && STRIDE == 1 && g.tiles_x == 1 && g.TW == g.W && g.IWS >= g.W + 4) g.fast = 2;
    g.mag_nv = (unsigned)((0x100000000ull + (g.W >> 2) - 1) / std::max(g.W >> 2, 1));
    int th = g.Ho < 16 ? g.Ho : 16;
    // lean staging: a tile of 8 rows of the 8-plane groups (41.9 KB) keeps 240 of 256 threads on strips; at 4 rows it was 120
    static const int fast_cap_kb = MSPL_TUNE_INT("MSPL_C3_CAPKB", 48);
    const size_t lds_cap = g.fast == 2 ? (size_t)fast_cap_kb * 1024 : 40 * 1024;
    while (th > 1 && lds_of(th) > lds_cap) th = (th + 1) / 2;
    // keep 256 threads busy: shrink the tile only while it still holds >= 256 strips
    while (th > 2 && (int64_t)g.coblks * (th / 2) * g.XS >= 512) th = th / 2;
    // ... and shrink it further while the grid would not even give every CU a few workgroups (a lone workgroup per CU
    // is one wave per SIMD: nothing hides its LDS latency -- the 3->32 stem ran at a third of its speed that way)
    while (th > 2 && (int64_t)g.N * g.G * ceil_div(g.Ho, th) * ceil_div(g.Wo, g.TW) < 1536) th = th / 2;
    MSPL_REQUIRE(lds_of(th) <= 64 * 1024, MSPL_ERR_UNSUPPORTED,
                 "conv3x3: tile (cin_g=%d, cout_g=%d) does not fit LDS", g.cin_g, g.cout_g);
    g.TH = th;
    g.IH = (th - 1) * STRIDE + 3;
    g.nvec = (g.IWS + 3 + 3) >> 2;
    g.mag_nvec = (unsigned)((0x100000000ull + g.nvec - 1) / g.nvec);
    g.mag_ih = (unsigned)((0x100000000ull + g.IH - 1) / g.IH);
    g.tiles_y = ceil_div(g.Ho, th);
    const int64_t blocks = (int64_t)g.N * g.G * g.tiles_y * g.tiles_x;
    MSPL_REQUIRE(blocks < (1ll << 31), MSPL_ERR_BAD_SHAPE, "conv3x3: grid too large");
    g.total = (unsigned)blocks; g.xcd_per = xcd_per(blocks);
    dim3 grid(8u * g.xcd_per), blk(256);
    const size_t lds = lds_of(th);
    static const int no_uniw = (MSPL_TUNE_INT("MSPL_C3_UNIW", 1) == 0);
    const bool uniw = g.coblks == 1 && !no_uniw;
#define MSPL_C3(CB) do { if (uniw) hipLaunchKernelGGL((conv3x3_kernel<STRIDE, CB, true>), grid, blk, lds, s, x, w, g, e, out); \
                         else hipLaunchKernelGGL((conv3x3_kernel<STRIDE, CB, false>), grid, blk, lds, s, x, w, g, e, out); } while (0)
    switch (cob) {
        case 8: MSPL_C3(8); break;
        case 4: MSPL_C3(4); break;
        case 3: MSPL_C3(3); break;
        case 2: MSPL_C3(2); break;
        default: MSPL_C3(1); break;
    }
#undef MSPL_C3
    MSPL_CHECK_LAUNCH("conv3x3");
    return MSPL_OK;
}

}  // namespace mspl

using namespace mspl;

extern "C" int mspl_conv3x3_fwd(const float* x, const float* w, int32_t N, int32_t Cin, int32_t Cout,
                                int32_t groups, int32_t H, int32_t W, int32_t stride, int32_t shuffle_groups,
                                const mspl_epilogue_t* ep, float* out, void* stream) {
    MSPL_REQUIRE(x && w && out, MSPL_ERR_NULL_POINTER, "conv3x3: null pointer");
    MSPL_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && groups > 0 && H > 0 && W > 0, MSPL_ERR_BAD_SHAPE,
                 "conv3x3: bad shape N=%d Cin=%d Cout=%d groups=%d H=%d W=%d", N, Cin, Cout, groups, H, W);
    MSPL_REQUIRE(Cin % groups == 0 && Cout % groups == 0, MSPL_ERR_BAD_SHAPE,
                 "conv3x3: channels (%d,%d) not divisible by groups %d", Cin, Cout, groups);
    MSPL_REQUIRE(stride == 1 || stride == 2, MSPL_ERR_UNSUPPORTED, "conv3x3: stride %d (1 or 2)", stride);
    MSPL_REQUIRE(shuffle_groups >= 0 && (shuffle_groups == 0 || Cin % shuffle_groups == 0), MSPL_ERR_BAD_SHAPE,
                 "conv3x3: shuffle groups %d do not divide Cin=%d", shuffle_groups, Cin);
    if (int rc = check_epi(ep, Cout, "conv3x3", true)) return rc;
    C3Geom g;
    memset(&g, 0, sizeof(g));
    g.N = N; g.Cin = Cin; g.Cout = Cout; g.G = groups; g.cin_g = Cin / groups; g.cout_g = Cout / groups;
    g.H = H; g.W = W; g.sg = shuffle_groups;
    g.Ho = (H - 1) / stride + 1;
    g.Wo = (W - 1) / stride + 1;
    const Epi e = make_epi(ep, Cout, g.Ho * g.Wo);
    hipStream_t s = (hipStream_t)stream;
    static const int no_dws = (MSPL_TUNE_INT("MSPL_DWS", 1) == 0);       // tuning aid: 0 = LDS-tiled form only
    if (stride == 1 && !no_dws && dwconv3x3_stream_ok(g, x, out, e)) return launch_dws(x, w, g, e, out, s);
    if (stride == 1) {
        const int rc = gconv3x3_stream_try(x, w, g, e, out, s);
        if (rc <= 0) return rc;
    }
    return stride == 1 ? launch3<1>(x, w, g, e, out, s) : launch3<2>(x, w, g, e, out, s);
}
