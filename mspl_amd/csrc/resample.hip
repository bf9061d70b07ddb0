// Pooling / resampling / pointwise kernels of the decoder and the DownSampler, all with the shared epilogue.
//   avgpool3x3s2      nn_layers/eesp.py:115,128,136-140  (AvgPool2d(3, stride 2, pad 1), count_include_pad)
//   bilinear          F.interpolate(mode='bilinear', align_corners=True): efficient_pyramid_pool.py:48,50;
//                     model/segmentation/espdnet_ue.py:110,301-302
//   adaptive_avgpool  F.adaptive_avg_pool2d: efficient_pyramid_pool.py:46,52
//   pointwise         BatchNorm+PReLU blocks: espdnet_ue.py:89-97, cnn_utils.py:85-105
//   gap_gate          EfficientPWConv.wt_layer: efficient_pt.py:13-17
// All are streaming kernels: one thread per 4 consecutive output pixels of a row (16-byte stores when the
// row length allows), source reads served by L1/L2 (every source line is touched by neighbouring lanes).
#include <stdlib.h>

#include "common.hpp"

namespace mspl {

struct RsGeom {
    int N, C, Hi, Wi, Ho, Wo;
    int XS;           // ceil(Wo / 4)
    unsigned mag_xs;  // ceil(2^32 / XS): s / XS == __umulhi(s, mag_xs) for s * XS < 2^32
    float sh, sw;     // bilinear scales
    int half_pixel;   // bilinear: 0 = align_corners=True, 1 = ATen's align_corners=False rule (src = scale*(dst+0.5)-0.5, >= 0)
};

// Grid layout of the strip kernels: blockIdx.x walks the Ho x XS strips of one plane, (blockIdx.y, blockIdx.z) the
// N*C planes.  The plane split is uniform (scalar unit); the strip split is one mul-hi -- the 64-bit div/mod chains of
// a flat index were a large share of these kernels' instructions.
__device__ __forceinline__ bool strip_decode(const RsGeom& g, int& n, int& c, int& y, int& x0) {
    const int plane = blockIdx.z * gridDim.y + blockIdx.y;
    const unsigned s = blockIdx.x * 256u + threadIdx.x;
    if (plane >= g.N * g.C || s >= (unsigned)(g.Ho * g.XS)) return false;
    n = plane / g.C;  c = plane - n * g.C;
    y = g.XS == 1 ? (int)s : (int)__umulhi(s, g.mag_xs);       // (ceil(2^32 / 1) does not fit the 32-bit magic)
    x0 = ((int)s - y * g.XS) * 4;
    return true;
}

static inline dim3 strip_grid(const RsGeom& g) {
    const int planes = g.N * g.C;
    const int gy = planes < 65535 ? planes : 65535;
    return dim3((unsigned)ceil_div(g.Ho * g.XS, 256), (unsigned)gy, (unsigned)ceil_div(planes, gy));
}

__device__ __forceinline__ void strip_store(const Epi& e, const RsGeom& g, int n, int c, int y, int x0,
                                            const float (&acc)[4], float* out) {
    const int cabs = e.coff + c;
    const EpiCh ec = epi_channel(e, cabs);
    const int pix = y * g.Wo + x0;
    float* dst = out + epi_offset(e, n, cabs, pix);
    if ((g.Wo & 3) == 0) {
        store_out4(dst, epi_apply4(e, ec, acc, n, cabs, pix));
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (x0 + j < g.Wo) dst[j] = epi_apply(e, ec, acc[j], n, cabs, pix + j);
    }
}

// PSUM: also accumulate the plane sums of the INPUT (the EfficientPWConv gates need mean_hw of exactly the tensors the three
// DownSamplers pool: l1, l2, l3 -- reading them again for a plane-mean kernel was 4 % of a pass's HBM traffic).  Every thread owns
// the 2x8 input block (rows 2y, 2y+1; columns 2*x0 .. 2*x0+7) it loads anyway; one partial per workgroup (one plane per workgroup).
template <bool PSUM>
__global__ __launch_bounds__(256) void avgpool3x3s2_kernel(const float* __restrict__ x, RsGeom g, Epi e,
                                                           float* __restrict__ out, float* __restrict__ psum) {
    int n, c, y, x0;
    const bool live = strip_decode(g, n, c, y, x0);
    float own = 0.f;
    if (live) {
    const float* src = x + ((size_t)n * g.C + c) * (size_t)g.Hi * g.Wi;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // input columns 2*x0-1 .. 2*x0+7; 2*x0 is a multiple of 8, so with Wi % 4 == 0 the 8 interior columns are
    // two aligned 16-byte loads and only the left neighbour is a scalar load.  The three rows are read branch-free
    // (clamped row index, zero weight outside the image) so that all nine loads are in flight together.
    const bool vec = ((g.Wi & 3) == 0) && (2 * x0 + 7 < g.Wi);
    if (vec) {
        float4 a[3], b[3];
        float l[3], m[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * y - 1 + ky;
            m[ky] = (iy >= 0 && iy < g.Hi) ? 1.f : 0.f;
            const float* row = src + (size_t)min(max(iy, 0), g.Hi - 1) * g.Wi + 2 * x0;
            a[ky] = *reinterpret_cast<const float4*>(row);
            b[ky] = *reinterpret_cast<const float4*>(row + 4);
            l[ky] = row[x0 > 0 ? -1 : 0];
        }
        const float lm = x0 > 0 ? 1.f : 0.f;
        if (PSUM) {
            own = ((a[1].x + a[1].y) + (a[1].z + a[1].w)) + ((b[1].x + b[1].y) + (b[1].z + b[1].w));
            own += m[2] * (((a[2].x + a[2].y) + (a[2].z + a[2].w)) + ((b[2].x + b[2].y) + (b[2].z + b[2].w)));
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const float w = m[ky];
            acc[0] += w * ((lm * l[ky] + a[ky].x) + a[ky].y);
            acc[1] += w * ((a[ky].y + a[ky].z) + a[ky].w);
            acc[2] += w * ((a[ky].w + b[ky].x) + b[ky].y);
            acc[3] += w * ((b[ky].y + b[ky].z) + b[ky].w);
        }
    } else {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * y - 1 + ky;
            if (iy < 0 || iy >= g.Hi) continue;
            const float* row = src + (size_t)iy * g.Wi;
            float rv[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const int ix = 2 * x0 - 1 + i;
                rv[i] = (ix >= 0 && ix < g.Wi) ? row[ix] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += rv[2 * j] + rv[2 * j + 1] + rv[2 * j + 2];
            if (PSUM && ky >= 1) {
#pragma unroll
                for (int i = 1; i < 9; ++i) own += rv[i];          // columns 2*x0 .. 2*x0+7 (zero outside the image)
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] *= (1.0f / 9.0f);   // count_include_pad=True: divisor is always 9
    strip_store(e, g, n, c, y, x0, acc, out);
    }
    if (PSUM) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) own += __shfl_down(own, o, 64);
        __shared__ float part[4];
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = own;
        __syncthreads();
        const int plane = blockIdx.z * gridDim.y + blockIdx.y;
        // one slot per (plane, workgroup): the gate sums the slots in order -- deterministic, and nothing to zero
        if (threadIdx.x == 0 && plane < g.N * g.C) psum[(size_t)plane * gridDim.x + blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
    }
}

// The DownSampler's pool (nn_layers/eesp.py:131-144: avg_pool(input) into channels [0, nin) of the block's output, + the image
// reinforcement, module PReLU) in its own lean form.  The generic strip kernel above spent ~220 vector instructions per strip on
// this launch (64-bit addressing of ~13 loads, a plane / C division per thread, the epilogue's optional operands behind uniform
// branches): 13 M of a label pass's 262 M.  Here: grid = (strips, channel, image) -- no division; 32-bit offsets on uniform bases;
// the epilogue is exactly scale / shift / three reinforcement planes / PReLU; the plane-sum partial goes through DPP adds.
// Same arithmetic in the same order as avgpool3x3s2_kernel<true> with that epilogue (bit-identical outputs and plane sums).
struct DpGeom {
    int C, Hi, Wi, Ho, Wo, XS;        // input channels (= pooled channels), sizes, strips per output row
    unsigned mag_xs;
    int ctot;                         // channels of the destination tensor
};

// RAG: rows of Wo = 4k + 2 outputs (the level-4 DownSampler of a 480-wide input: 60 -> 30 columns).  The last strip of a row
// owns two outputs and its second 16-byte piece lies past the row (taken as zeros, as the padding is); pixel offsets are
// 8-byte aligned only, so the epilogue moves float2 halves.  That launch used to take the generic strip kernel above, where
// the ragged strip sent every wave through both its load paths and the guarded element stores: 5.4 M of a label pass's vector
// instructions for one launch.  Outputs are bit-identical to that kernel's; the plane-sum partials of a ragged strip are
// added in the tree order of the other strips (the generic kernel's element path added them left to right).
template <bool RAG>
__global__ __launch_bounds__(256) void avgpool3x3s2_down_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, const float* __restrict__ alpha,
                                                                const float* __restrict__ reinf_r, const float* __restrict__ reinf_w,
                                                                DpGeom g, float* __restrict__ out, float* __restrict__ psum) {
    const int c = blockIdx.y, n = blockIdx.z;
    const unsigned s = blockIdx.x * 256u + threadIdx.x;
    const bool live = s < (unsigned)(g.Ho * g.XS);
    float own = 0.f;
    if (live) {
        const int y = g.XS == 1 ? (int)s : (int)__umulhi(s, g.mag_xs);
        const int x0 = ((int)s - y * g.XS) * 4;
        const float* src = x + ((size_t)n * g.C + c) * (size_t)g.Hi * g.Wi;          // uniform
        float4 a[3], b[3];
        float l[3], m[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * y - 1 + ky;
            m[ky] = (iy >= 0 && iy < g.Hi) ? 1.f : 0.f;
            const unsigned o = (unsigned)(min(max(iy, 0), g.Hi - 1) * g.Wi + 2 * x0);
            a[ky] = *reinterpret_cast<const float4*>(src + o);
            if (!RAG) {
                b[ky] = *reinterpret_cast<const float4*>(src + o + 4);
            } else {
                const bool bok = 2 * x0 + 4 < g.Wi;                                  // whole piece inside the row or whole piece outside
                const float4 t = *reinterpret_cast<const float4*>(src + (bok ? o + 4 : o));
                b[ky] = bok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            l[ky] = src[x0 > 0 ? o - 1 : o];
        }
        const float lm = x0 > 0 ? 1.f : 0.f;
        own = ((a[1].x + a[1].y) + (a[1].z + a[1].w)) + ((b[1].x + b[1].y) + (b[1].z + b[1].w));
        own += m[2] * (((a[2].x + a[2].y) + (a[2].z + a[2].w)) + ((b[2].x + b[2].y) + (b[2].z + b[2].w)));
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const float w = m[ky];
            acc[0] += w * ((lm * l[ky] + a[ky].x) + a[ky].y);
            acc[1] += w * ((a[ky].y + a[ky].z) + a[ky].w);
            acc[2] += w * ((a[ky].w + b[ky].x) + b[ky].y);
            acc[3] += w * ((b[ky].y + b[ky].z) + b[ky].w);
        }
        const unsigned pix = (unsigned)(y * g.Wo + x0), hw = (unsigned)(g.Ho * g.Wo);
        const float sc = scale[c], sh = shift[c], al = alpha[c];                     // uniform
        const float rw0 = reinf_w[c * 3], rw1 = reinf_w[c * 3 + 1], rw2 = reinf_w[c * 3 + 2];
        const float* r = reinf_r + (size_t)n * 3 * hw;                               // uniform
        float4 r0, r1, r2;
        const bool hi = !RAG || x0 + 2 < g.Wo;                                       // RAG: outputs x0 + 2, x0 + 3 exist
        if (!RAG) {
            r0 = *reinterpret_cast<const float4*>(r + pix);
            r1 = *reinterpret_cast<const float4*>(r + hw + pix);
            r2 = *reinterpret_cast<const float4*>(r + 2 * hw + pix);
        } else {
            const unsigned p2 = hi ? pix + 2 : pix;
            const float2 a0 = *reinterpret_cast<const float2*>(r + pix), b0 = *reinterpret_cast<const float2*>(r + p2);
            const float2 a1 = *reinterpret_cast<const float2*>(r + hw + pix), b1 = *reinterpret_cast<const float2*>(r + hw + p2);
            const float2 a2 = *reinterpret_cast<const float2*>(r + 2 * hw + pix), b2 = *reinterpret_cast<const float2*>(r + 2 * hw + p2);
            r0 = make_float4(a0.x, a0.y, b0.x, b0.y);
            r1 = make_float4(a1.x, a1.y, b1.x, b1.y);
            r2 = make_float4(a2.x, a2.y, b2.x, b2.y);
        }
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaf(acc[j] * (1.0f / 9.0f), sc, sh);
        v[0] += rw0 * r0.x + rw1 * r1.x + rw2 * r2.x;
        v[1] += rw0 * r0.y + rw1 * r1.y + rw2 * r2.y;
        v[2] += rw0 * r0.z + rw1 * r1.z + rw2 * r2.z;
        v[3] += rw0 * r0.w + rw1 * r1.w + rw2 * r2.w;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.0f ? v[j] : al * v[j];
        float* dst = out + ((size_t)n * g.ctot + c) * (size_t)hw + pix;
        if (!RAG) {
            store_out4(dst, make_float4(v[0], v[1], v[2], v[3]));
        } else {
            store_out2(dst, make_float2(v[0], v[1]));
            if (hi) store_out2(dst + 2, make_float2(v[2], v[3]));
        }
    }
    own = wave_sum_dpp(own);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 63) part[threadIdx.x >> 6] = own;
    __syncthreads();
    // one slot per (plane, workgroup), planes in (n, c) order: the gate sums the slots in order
    if (threadIdx.x == 0) psum[((size_t)n * g.C + c) * gridDim.x + blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

// The column sources / weights depend on the output column only: a workgroup computes the table of the columns it
// touches once into LDS (4 floats per column) instead of four bilinear_src evaluations per thread; source reads use
// 32-bit offsets from the plane base.
__global__ __launch_bounds__(256) void bilinear_kernel(const float* __restrict__ x, RsGeom g, Epi e,
                                                       float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float xtab[];      // [4 * XS][4] = {xa, xb, wx0, wx1}
    const int ncol = 4 * g.XS;
    for (int c = threadIdx.x; c < ncol; c += 256) {
        int xa, xb;  float wx0, wx1;
        if (g.half_pixel) bilinear_src_hp(g.sw, min(c, g.Wo - 1), g.Wi, xa, xb, wx0, wx1);
        else bilinear_src(g.sw, min(c, g.Wo - 1), g.Wi, xa, xb, wx0, wx1);
        xtab[4 * c] = __int_as_float(xa); xtab[4 * c + 1] = __int_as_float(xb); xtab[4 * c + 2] = wx0; xtab[4 * c + 3] = wx1;
    }
    __syncthreads();
    int n, c, y, x0;
    if (!strip_decode(g, n, c, y, x0)) return;
    const float* src = x + ((size_t)n * g.C + c) * (size_t)g.Hi * g.Wi;
    int y0i, y1i;  float wy0, wy1;
    if (g.half_pixel) bilinear_src_hp(g.sh, y, g.Hi, y0i, y1i, wy0, wy1);
    else bilinear_src(g.sh, y, g.Hi, y0i, y1i, wy0, wy1);
    const int r0 = y0i * g.Wi, r1 = y1i * g.Wi;
    float acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float4 t = *reinterpret_cast<const float4*>(xtab + 4 * (x0 + j));
        const int xa = __float_as_int(t.x), xb = __float_as_int(t.y);
        const float top = t.z * src[r0 + xa] + t.w * src[r0 + xb];
        const float bot = t.z * src[r1 + xa] + t.w * src[r1 + xb];
        acc[j] = wy0 * top + wy1 * bot;
    }
    strip_store(e, g, n, c, y, x0, acc, out);
}


// Bilinear, register-streaming form (the decoder's up-merge: bu_br(pw + upsample2(bu)), model/segmentation/espdnet_ue.py:276-280).
// The strip kernel above issues 16 scattered 4-byte loads and ~36 vector instructions per output pixel; the label pass is bound
// by vector-instruction issue.  Here a wave owns PW planes x one segment of output rows and walks DOWN the rows: lane = (plane,
// 4 adjacent output columns); the column sources / weights of its 4 pixels are per-lane constants; the row sources are
// wave-uniform.  A source row is fetched and interpolated horizontally ONCE (top = wx0*p[xa] + wx1*p[xb], the reference's own
// expression) and serves every output row that uses it (two for the x2 merge); per output row only the vertical blend, the
// epilogue and one 16-byte store remain.  ~12 vector instructions per pixel, no LDS, no barrier.
struct BsGeom {
    int N, C, Hi, Wi, Ho, Wo, XS, PW;   // PW: planes side by side in one wave (64 / XS)
    int SEG, nseg, pgroups;             // output rows per segment, segments per plane, ceil(N*C / PW)
    unsigned total;                     // waves
    float sh, sw;
    int half_pixel;
};

__global__ __launch_bounds__(256) void bilinear_stream_kernel(const float* __restrict__ x, BsGeom g, Epi e, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    unsigned wid = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (wid >= g.total) return;                              // wave-uniform; no barrier in this kernel
    const int sgi = wid % g.nseg;
    const int pg = wid / g.nseg;
    const int pl = lane / g.XS, xs = lane - pl * g.XS;       // plane slot and strip of this lane
    const int plane = pg * g.PW + pl;
    const bool live = pl < g.PW && plane < g.N * g.C;
    const int n = live ? plane / g.C : 0, c = live ? plane - n * g.C : 0;
    const int x0 = xs * 4;
    const float* src = x + (size_t)(live ? plane : 0) * (size_t)g.Hi * g.Wi;
    int xa[4], xb[4];  float wx0[4], wx1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int xx = min(x0 + j, g.Wo - 1);
        if (g.half_pixel) bilinear_src_hp(g.sw, xx, g.Wi, xa[j], xb[j], wx0[j], wx1[j]);
        else bilinear_src(g.sw, xx, g.Wi, xa[j], xb[j], wx0[j], wx1[j]);
    }
    const int cabs = e.coff + c;
    const EpiCh ec = epi_channel(e, cabs);
    const int ys = sgi * g.SEG, ye = min(ys + g.SEG, g.Ho);
    float ha[4] = {0.f, 0.f, 0.f, 0.f}, hb[4] = {0.f, 0.f, 0.f, 0.f};      // horizontally interpolated source rows ia / ib
    int ia = -1, ib = -1;
    auto hrow = [&](int r, float (&h)[4]) {
        const float* row = src + (size_t)r * g.Wi;
#pragma unroll
        for (int j = 0; j < 4; ++j) h[j] = wx0[j] * row[xa[j]] + wx1[j] * row[xb[j]];
    };
#pragma unroll 1
    for (int y = ys; y < ye; ++y) {
        int y0i, y1i;  float wy0, wy1;
        if (g.half_pixel) bilinear_src_hp(g.sh, y, g.Hi, y0i, y1i, wy0, wy1);
        else bilinear_src(g.sh, y, g.Hi, y0i, y1i, wy0, wy1);            // uniform
        if (y0i != ia) {                                                   // uniform branches
            if (y0i == ib) {
#pragma unroll
                for (int j = 0; j < 4; ++j) ha[j] = hb[j];
            } else {
                hrow(y0i, ha);
            }
            ia = y0i;
        }
        if (y1i != ib) {
            if (y1i == ia) {
#pragma unroll
                for (int j = 0; j < 4; ++j) hb[j] = ha[j];
            } else {
                hrow(y1i, hb);
            }
            ib = y1i;
        }
        if (live) {
            float acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = wy0 * ha[j] + wy1 * hb[j];
            const int pix = y * g.Wo + x0;
            store_out4(out + epi_offset(e, n, cabs, pix), epi_apply4(e, ec, acc, n, cabs, pix));
        }
    }
}

__device__ __forceinline__ int ada_start(int o, int I, int O) { return (int)(((int64_t)o * I) / O); }
__device__ __forceinline__ int ada_end(int o, int I, int O) { return (int)((((int64_t)(o + 1)) * I + O - 1) / O); }

__global__ __launch_bounds__(256) void adaptive_avgpool_kernel(const float* __restrict__ x, RsGeom g, Epi e,
                                                               float* __restrict__ out) {
    int n, c, y, x0;
    if (!strip_decode(g, n, c, y, x0)) return;
    const float* src = x + ((size_t)n * g.C + c) * (size_t)g.Hi * g.Wi;
    const int ys = ada_start(y, g.Hi, g.Ho), ye = ada_end(y, g.Hi, g.Ho);
    float acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int xx = min(x0 + j, g.Wo - 1);
        const int xs = ada_start(xx, g.Wi, g.Wo), xe = ada_end(xx, g.Wi, g.Wo);
        float s = 0.f;
        for (int iy = ys; iy < ye; ++iy) {
            const float* row = src + (size_t)iy * g.Wi;
            for (int ix = xs; ix < xe; ++ix) s += row[ix];
        }
        acc[j] = s / (float)((ye - ys) * (xe - xs));
    }
    strip_store(e, g, n, c, y, x0, acc, out);
}

// Large windows (the 0.1-scale pyramid branch pools ~10x10..20x20 inputs per output): one wave per output
// pixel, lanes stride over the window so that each window row is read coalesced; shuffle reduction.
__global__ __launch_bounds__(256) void adaptive_avgpool_wave_kernel(const float* __restrict__ x, RsGeom g, Epi e,
                                                                    float* __restrict__ out, int64_t total_out) {
    const int64_t o = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= total_out) return;                       // wave-uniform
    const int lane = threadIdx.x & 63;
    int64_t t = o;
    const int ox = (int)(t % g.Wo);  t /= g.Wo;
    const int oy = (int)(t % g.Ho);  t /= g.Ho;
    const int c = (int)(t % g.C);
    const int n = (int)(t / g.C);
    const float* src = x + ((size_t)n * g.C + c) * (size_t)g.Hi * g.Wi;
    const int ys = ada_start(oy, g.Hi, g.Ho), ye = ada_end(oy, g.Hi, g.Ho);
    const int xs = ada_start(ox, g.Wi, g.Wo), xe = ada_end(ox, g.Wi, g.Wo);
    const int ww = xe - xs, cnt = (ye - ys) * ww;
    float s = 0.f;
    for (int i = lane; i < cnt; i += 64) {
        const int r = i / ww, q = i - r * ww;
        s += src[(size_t)(ys + r) * g.Wi + xs + q];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
    if (lane == 0) {
        const int cabs = e.coff + c;
        const EpiCh ec = epi_channel(e, cabs);
        const int pix = oy * g.Wo + ox;
        out[epi_offset(e, n, cabs, pix)] = epi_apply(e, ec, s / (float)cnt, n, cabs, pix);
    }
}

// x and out share the destination geometry (N, ctot, HW); 4 pixels per thread; blockIdx.y/z = plane (n, c).
__global__ __launch_bounds__(256) void pointwise_kernel(const float* __restrict__ x, int N, int C, Epi e,
                                                        float* __restrict__ out) {
    const int plane = blockIdx.z * gridDim.y + blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (plane >= N * C || q >= ((e.hw + 3) >> 2)) return;
    const int n = plane / C, c = plane - n * C;
    const int cabs = e.coff + c;
    const EpiCh ec = epi_channel(e, cabs);
    const int p0 = q * 4;
    const size_t off = epi_offset(e, n, cabs, p0);
    if ((e.hw & 3) == 0) {
        const float4 v = *reinterpret_cast<const float4*>(x + off);
        const float a4[4] = {v.x, v.y, v.z, v.w};
        store_out4(out + off, epi_apply4(e, ec, a4, n, cabs, p0));
    } else {
        for (int j = 0; j < 4 && p0 + j < e.hw; ++j) out[off + j] = epi_apply(e, ec, x[off + j], n, cabs, p0 + j);
    }
}

// One workgroup per (n, c) plane: mean over HW with 16-byte loads, wave shuffles, then LDS across waves.
__global__ __launch_bounds__(256) void plane_mean_kernel(const float* __restrict__ x, int HW, float* __restrict__ mean) {
    const float* src = x + (size_t)blockIdx.x * HW;
    float s = 0.f;
    if ((HW & 3) == 0) {
        const float4* s4 = reinterpret_cast<const float4*>(src);
        // (unrolled: four 16-byte loads in flight per thread, same order of additions)
#pragma unroll 4
        for (int i = threadIdx.x; i < (HW >> 2); i += 256) { const float4 v = s4[i]; s += (v.x + v.y) + (v.z + v.w); }
    } else {
        for (int i = threadIdx.x; i < HW; i += 256) s += src[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) mean[blockIdx.x] = ((part[0] + part[1]) + (part[2] + part[3])) / (float)HW;
}

// One wave per (n, co): lanes stride over Cin (coalesced weight row), shuffle reduction, sigmoid.
__global__ __launch_bounds__(256) void gate_kernel(const float* __restrict__ mean, const float* __restrict__ w,
                                                   int N, int Cin, int Cout, int nblk, float mscale, float* __restrict__ gate) {
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= N * Cout) return;                      // wave-uniform
    const int lane = threadIdx.x & 63;
    const int n = idx / Cout, co = idx - n * Cout;
    const float* m = mean + (size_t)n * Cin * nblk;            // nblk partial sums per channel (1: a finished mean)
    const float* wr = w + (size_t)co * Cin;
    float s = 0.f;
    for (int k = lane; k < Cin; k += 64) {
        float mk = m[(size_t)k * nblk];
        for (int j = 1; j < nblk; ++j) mk += m[(size_t)k * nblk + j];
        s = fmaf(wr[k], mk, s);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
    if (lane == 0) gate[idx] = 1.0f / (1.0f + expf(-(s * mscale)));
}

static int resample_common(const char* who, const float* x, float* out, int N, int C, int Hi, int Wi, int Ho, int Wo,
                           const mspl_epilogue_t* ep, RsGeom& g, Epi& e, int64_t& total) {
    MSPL_REQUIRE(x && out, MSPL_ERR_NULL_POINTER, "%s: null pointer", who);
    MSPL_REQUIRE(N > 0 && C > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0, MSPL_ERR_BAD_SHAPE,
                 "%s: bad shape N=%d C=%d in=%dx%d out=%dx%d", who, N, C, Hi, Wi, Ho, Wo);
    if (int rc = check_epi(ep, C, who)) return rc;
    g.N = N; g.C = C; g.Hi = Hi; g.Wi = Wi; g.Ho = Ho; g.Wo = Wo;
    g.XS = ceil_div(Wo, 4);
    g.mag_xs = (unsigned)((0x100000000ull + g.XS - 1) / g.XS);
    MSPL_REQUIRE((int64_t)Ho * g.XS * g.XS < (1ll << 32), MSPL_ERR_BAD_SHAPE, "%s: plane too large", who);
    g.sh = bilinear_scale(Hi, Ho);
    g.sw = bilinear_scale(Wi, Wo);
    g.half_pixel = 0;
    e = make_epi(ep, C, Ho * Wo);
    total = (int64_t)N * C * Ho * g.XS;
    MSPL_REQUIRE(ceil_div64(total, 256) < (1ll << 31), MSPL_ERR_BAD_SHAPE, "%s: grid too large", who);
    return MSPL_OK;
}

}  // namespace mspl

using namespace mspl;

extern "C" int mspl_avgpool3x3s2_fwd(const float* x, int32_t N, int32_t C, int32_t H, int32_t W,
                                     const mspl_epilogue_t* ep, float* out, void* stream) {
    RsGeom g; Epi e; int64_t total;
    const int Ho = H > 0 ? (H - 1) / 2 + 1 : 0, Wo = W > 0 ? (W - 1) / 2 + 1 : 0;
    if (int rc = resample_common("avgpool3x3s2", x, out, N, C, H, W, Ho, Wo, ep, g, e, total)) return rc;
    hipLaunchKernelGGL(avgpool3x3s2_kernel<false>, strip_grid(g), dim3(256), 0, (hipStream_t)stream, x, g, e, out, nullptr);
    MSPL_CHECK_LAUNCH("avgpool3x3s2");
    return MSPL_OK;
}

// Partial plane sums per plane written by mspl_avgpool3x3s2_psum_fwd for an (H, W) input (the psum buffer holds N*C times this).
extern "C" int mspl_avgpool3x3s2_psum_blocks(int32_t H, int32_t W) {
    if (H <= 0 || W <= 0) return MSPL_ERR_BAD_SHAPE;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    return ceil_div(Ho * ceil_div(Wo, 4), 256);
}

extern "C" int mspl_avgpool3x3s2_psum_fwd(const float* x, int32_t N, int32_t C, int32_t H, int32_t W,
                                          const mspl_epilogue_t* ep, float* out, float* psum, void* stream) {
    MSPL_REQUIRE(psum, MSPL_ERR_NULL_POINTER, "avgpool3x3s2_psum: null pointer");
    RsGeom g; Epi e; int64_t total;
    const int Ho = H > 0 ? (H - 1) / 2 + 1 : 0, Wo = W > 0 ? (W - 1) / 2 + 1 : 0;
    if (int rc = resample_common("avgpool3x3s2_psum", x, out, N, C, H, W, Ho, Wo, ep, g, e, total)) return rc;
    // the DownSampler's call (scale, shift, PReLU and the image reinforcement on an un-gated destination with whole 16-byte rows): lean form
    if (e.scale && e.shift && e.alpha && e.reinf_r && e.reinf_w && !e.pre_add && !e.residual && !e.gate && !e.raw && e.coff == 0 &&
        (W & 3) == 0 && (H & 1) == 0 && C <= 65535 && N <= 65535 && (int64_t)H * W < (1ll << 30) &&
        ((((uintptr_t)x) | ((uintptr_t)out) | ((uintptr_t)e.reinf_r)) & 15) == 0) {
        // W = 4k: Wo = 2k is a whole number of 16-byte strips or (RAG) ends in one 8-byte half; plane sizes Ho * Wo are even
        DpGeom d;
        d.C = C; d.Hi = H; d.Wi = W; d.Ho = Ho; d.Wo = Wo; d.XS = g.XS; d.mag_xs = g.mag_xs; d.ctot = e.ctot;
        const dim3 grid((unsigned)ceil_div(Ho * g.XS, 256), (unsigned)C, (unsigned)N);
        if ((Wo & 3) == 0)
            hipLaunchKernelGGL(avgpool3x3s2_down_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, e.scale, e.shift, e.alpha,
                               e.reinf_r, e.reinf_w, d, out, psum);
        else
            hipLaunchKernelGGL(avgpool3x3s2_down_kernel<true>, grid, dim3(256), 0,
                           (hipStream_t)stream, x, e.scale, e.shift, e.alpha, e.reinf_r, e.reinf_w, d, out, psum);
        MSPL_CHECK_LAUNCH("avgpool3x3s2_psum(DownSampler form)");
        return MSPL_OK;
    }
    hipLaunchKernelGGL(avgpool3x3s2_kernel<true>, strip_grid(g), dim3(256), 0, (hipStream_t)stream, x, g, e, out, psum);
    MSPL_CHECK_LAUNCH("avgpool3x3s2_psum");
    return MSPL_OK;
}

// EfficientPWConv gate from plane SUMS (mspl_avgpool3x3s2_psum_fwd): gate = sigmoid(W . sums / HW).
extern "C" int mspl_gate_from_sums_fwd(const float* psum, const float* w, int32_t N, int32_t Cin, int32_t Cout, int32_t nblk,
                                       int32_t HW, float* gate, void* stream) {
    MSPL_REQUIRE(psum && w && gate, MSPL_ERR_NULL_POINTER, "gate_from_sums: null pointer");
    MSPL_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && HW > 0 && nblk > 0, MSPL_ERR_BAD_SHAPE,
                 "gate_from_sums: bad shape N=%d Cin=%d Cout=%d nblk=%d HW=%d", N, Cin, Cout, nblk, HW);
    hipLaunchKernelGGL(gate_kernel, dim3((unsigned)ceil_div(N * Cout, 4)), dim3(256), 0, (hipStream_t)stream, psum, w, N, Cin, Cout,
                       nblk, 1.0f / (float)HW, gate);
    MSPL_CHECK_LAUNCH("gate_from_sums");
    return MSPL_OK;
}

extern "C" int mspl_bilinear_fwd(const float* x, int32_t N, int32_t C, int32_t Hi, int32_t Wi, int32_t Ho,
                                 int32_t Wo, int32_t align_corners, const mspl_epilogue_t* ep, float* out, void* stream) {
    RsGeom g; Epi e; int64_t total;
    if (int rc = resample_common("bilinear", x, out, N, C, Hi, Wi, Ho, Wo, ep, g, e, total)) return rc;
    g.half_pixel = align_corners ? 0 : 1;
    if (g.half_pixel) {        // ATen area_pixel_compute_scale(align_corners=False, no scale factor): in / out
        g.sh = (float)Hi / (float)Ho;
        g.sw = (float)Wi / (float)Wo;
    }
    {   // register-streaming form: rows of whole 16-byte strips, at most one wave wide, enough rows to walk
        auto al16 = [](const void* p) { return p == nullptr || (((uintptr_t)p) & 15) == 0; };
        static const int no_stream = (MSPL_TUNE_INT("MSPL_BILINEAR_STREAM", 1) == 0);
        if (!no_stream && (Wo & 3) == 0 && g.XS <= 64 && Ho >= 8 && al16(out) && al16(e.pre_add) && al16(e.residual) && al16(e.reinf_r) &&
            (int64_t)Hi * Wi < (1ll << 29) && (int64_t)Ho * Wo < (1ll << 29)) {
            BsGeom b;
            b.N = N; b.C = C; b.Hi = Hi; b.Wi = Wi; b.Ho = Ho; b.Wo = Wo; b.XS = g.XS; b.PW = 64 / g.XS;
            b.sh = g.sh; b.sw = g.sw; b.half_pixel = g.half_pixel;
            b.pgroups = ceil_div(N * C, b.PW);
            int seg = Ho < 32 ? Ho : 32;
            while (seg > 6 && (int64_t)b.pgroups * ceil_div(Ho, seg) < 4096) --seg;
            seg = ceil_div(Ho, ceil_div(Ho, seg));
            b.SEG = seg; b.nseg = ceil_div(Ho, seg);
            const int64_t waves = (int64_t)b.pgroups * b.nseg;
            if (waves < (1ll << 31)) {
                b.total = (unsigned)waves;
                hipLaunchKernelGGL(bilinear_stream_kernel, dim3((unsigned)ceil_div64(waves, 4)), dim3(256), 0, (hipStream_t)stream, x, b, e, out);
                MSPL_CHECK_LAUNCH("bilinear(streaming)");
                return MSPL_OK;
            }
        }
    }
    const size_t lds = (size_t)16 * g.XS * sizeof(float);
    MSPL_REQUIRE(lds <= 64 * 1024 && (int64_t)Hi * Wi < (1ll << 31), MSPL_ERR_UNSUPPORTED, "bilinear: output rows of %d pixels do not fit the column table", Wo);
    hipLaunchKernelGGL(bilinear_kernel, strip_grid(g), dim3(256), lds, (hipStream_t)stream, x, g, e, out);
    MSPL_CHECK_LAUNCH("bilinear");
    return MSPL_OK;
}

extern "C" int mspl_adaptive_avgpool_fwd(const float* x, int32_t N, int32_t C, int32_t Hi, int32_t Wi,
                                         int32_t Ho, int32_t Wo, const mspl_epilogue_t* ep, float* out,
                                         void* stream) {
    RsGeom g; Epi e; int64_t total;
    if (int rc = resample_common("adaptive_avgpool", x, out, N, C, Hi, Wi, Ho, Wo, ep, g, e, total)) return rc;
    if ((int64_t)Hi * Wi >= 32ll * Ho * Wo) {         // mean window >= 32 inputs: wave per output
        const int64_t total_out = (int64_t)N * C * Ho * Wo;
        MSPL_REQUIRE(ceil_div64(total_out, 4) < (1ll << 31), MSPL_ERR_BAD_SHAPE, "adaptive_avgpool: grid too large");
        hipLaunchKernelGGL(adaptive_avgpool_wave_kernel, dim3((unsigned)ceil_div64(total_out, 4)), dim3(256), 0,
                           (hipStream_t)stream, x, g, e, out, total_out);
        MSPL_CHECK_LAUNCH("adaptive_avgpool(wave)");
        return MSPL_OK;
    }
    hipLaunchKernelGGL(adaptive_avgpool_kernel, strip_grid(g), dim3(256), 0, (hipStream_t)stream, x, g, e, out);
    MSPL_CHECK_LAUNCH("adaptive_avgpool");
    return MSPL_OK;
}

extern "C" int mspl_pointwise_fwd(const float* x, int32_t N, int32_t C, int32_t HW, const mspl_epilogue_t* ep,
                                  float* out, void* stream) {
    MSPL_REQUIRE(x && out, MSPL_ERR_NULL_POINTER, "pointwise: null pointer");
    MSPL_REQUIRE(N > 0 && C > 0 && HW > 0, MSPL_ERR_BAD_SHAPE, "pointwise: bad shape N=%d C=%d HW=%d", N, C, HW);
    if (int rc = check_epi(ep, C, "pointwise")) return rc;
    const Epi e = make_epi(ep, C, HW);
    const int planes = N * C, gy = planes < 65535 ? planes : 65535;
    const dim3 grid((unsigned)ceil_div((HW + 3) / 4, 256), (unsigned)gy, (unsigned)ceil_div(planes, gy));
    hipLaunchKernelGGL(pointwise_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, N, C, e, out);
    MSPL_CHECK_LAUNCH("pointwise");
    return MSPL_OK;
}

extern "C" int mspl_gap_gate_fwd(const float* x, const float* w, int32_t N, int32_t Cin, int32_t Cout,
                                 int32_t HW, float* mean_ws, float* gate, void* stream) {
    MSPL_REQUIRE(x && w && mean_ws && gate, MSPL_ERR_NULL_POINTER, "gap_gate: null pointer");
    MSPL_REQUIRE(N > 0 && Cin > 0 && Cout > 0 && HW > 0, MSPL_ERR_BAD_SHAPE,
                 "gap_gate: bad shape N=%d Cin=%d Cout=%d HW=%d", N, Cin, Cout, HW);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(plane_mean_kernel, dim3((unsigned)(N * Cin)), dim3(256), 0, s, x, HW, mean_ws);
    MSPL_CHECK_LAUNCH("gap_gate(mean)");
    hipLaunchKernelGGL(gate_kernel, dim3((unsigned)ceil_div(N * Cout, 4)), dim3(256), 0, s, mean_ws, w, N, Cin, Cout, 1, 1.0f, gate);
    MSPL_CHECK_LAUNCH("gap_gate(gate)");
    return MSPL_OK;
}
