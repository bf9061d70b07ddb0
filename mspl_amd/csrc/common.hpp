// Shared device/host helpers for libmspl_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>

#include "../../include/mspl_hip.h"

namespace mspl {

constexpr int WAVE = 64;

// ------------------------------------------------------------------ host-side error plumbing
void set_error(const char* fmt, ...);

#define MSPL_REQUIRE(cond, code, ...)            \
    do {                                         \
        if (!(cond)) {                           \
            ::mspl::set_error(__VA_ARGS__);      \
            return (code);                       \
        }                                        \
    } while (0)

#define MSPL_CHECK_LAUNCH(name)                                                      \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) {                                                     \
            ::mspl::set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return MSPL_ERR_HIP;                                                     \
        }                                                                            \
    } while (0)

// In-kernel phase stamps (MSPL_PW_STAMP / MSPL_DW_STAMP) are a tuning aid with a static hipMalloc and a blocking hipMemcpy in
// the launcher: they would break a stream capture, so release builds compile the switch out (make STAMPS=1 brings it back).
#ifdef MSPL_DEBUG_STAMPS
#define MSPL_STAMP_ENV(name) (getenv(name) ? atoi(getenv(name)) : 0)
#else
#define MSPL_STAMP_ENV(name) 0
#endif

// Tuning switches (tile shapes, forcing an alternative kernel form, stopping a kernel after a phase -- DESIGN.md "Tuning aids"): the
// default build compiles every one of them to its default, so no environment variable can change which kernel a C caller gets
// (include/mspl_hip.h: "no global mutable state").  `make TUNING=1` (-DMSPL_TUNING) brings the environment reads back for the
// probes under tools/.  Launch-shape choices a CALLER may legitimately make travel in mspl_epilogue_t.flags (MSPL_LAUNCH_*).
#ifdef MSPL_TUNING
#define MSPL_TUNE_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#else
#define MSPL_TUNE_INT(name, dflt) (dflt)
#endif

// Write-through (sc1) 16- and 8-byte stores for kernel OUTPUTS.  A plain store leaves its line dirty in the XCD's L2 and the
// whole of a kernel's output is written back at the kernel boundary (+ bytes / 6 TB/s before the successor starts: 3 us behind
// 17.7 MB, MI355X_MICROARCH.md "boundary" row; measured here: K2 at 36x60 15.6 -> 12.3 us).  With sc1 the bytes leave L2 while the
// kernel still computes.  Only for data nobody in the same launch reads back.  The asm string ends in s_nop 1 (hipcc does not
// know the store still reads its data registers: cdna_hip_programming.md section 5.7).
// Measured on the whole label pass (bench.py, round 2): write-through in K2 alone is neutral for the pass (14 767 vs 14 755
// images/s) and lifts K2 itself from 0.285 to 0.332 of the HBM roof; write-through in EVERY kernel costs 3 % (14 330): the 1x1
// kernels' 8-byte sc1 stores are slower than plain ones and their consumers lose the same-XCD L2 hits.  So these helpers are
// plain stores unless the library is built with WT=1 (-DMSPL_WT_STORES); K2 has its own switch (MSPL_DW_WT, default on).
typedef float mspl_f32x4 __attribute__((ext_vector_type(4)));
typedef float mspl_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_out4(float* p, const float4& v) {
#ifndef MSPL_WT_STORES
    *reinterpret_cast<float4*>(p) = v;
#else
    const mspl_f32x4 d = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(d) : "memory");
#endif
}
__device__ __forceinline__ void store_out2(float* p, const float2& v) {
#ifndef MSPL_WT_STORES
    *reinterpret_cast<float2*>(p) = v;
#else
    const mspl_f32x2 d = {v.x, v.y};
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(d) : "memory");
#endif
}


// XCD-contiguous workgroup order.  The hardware deals workgroups to the 8 XCDs round-robin by linear id, and each XCD has its own
// L2: tiles that share halo rows / low-resolution source maps should sit on ONE XCD.  With the grid padded to 8 * per workgroups,
// physical id b works on logical tile (b % 8) * per + b / 8, so XCD k walks the contiguous logical range [k*per, (k+1)*per).
// Logical ids >= total do nothing (the whole workgroup leaves: safe before any barrier).
__device__ __forceinline__ unsigned xcd_contiguous(unsigned b, unsigned per) { return (b & 7u) * per + (b >> 3); }
static inline unsigned xcd_per(int64_t total) { return (unsigned)((total + 7) / 8); }

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------ device epilogue
// Device copy of mspl_epilogue_t with the destination geometry resolved.
struct Epi {
    const float* scale;
    const float* shift;
    const float* alpha;
    const float* pre_add;
    const float* residual;
    const float* reinf_r;
    const float* reinf_w;
    const float* gate;
    int ctot;   // destination tensor channels
    int coff;   // first destination channel of this op
    int hw;     // destination pixels per plane
    float* raw; // convolutions: also store the bare accumulator here (same shape as the destination, ctot == C), or null
    unsigned flags;   // MSPL_LAUNCH_* of the call (host side only)
};

static inline Epi make_epi(const mspl_epilogue_t* ep, int C_default, int hw) {
    Epi e;
    memset(&e, 0, sizeof(e));
    e.ctot = C_default;
    e.coff = 0;
    e.hw = hw;
    if (ep) {
        e.scale = ep->scale; e.shift = ep->shift; e.alpha = ep->alpha;
        e.pre_add = ep->pre_add; e.residual = ep->residual;
        e.reinf_r = ep->reinf_r; e.reinf_w = ep->reinf_w; e.gate = ep->gate;
        if (ep->out_ctot > 0) { e.ctot = ep->out_ctot; e.coff = ep->out_coff; }
        e.raw = ep->raw_out;
        e.flags = ep->flags;
    }
    return e;
}

static inline unsigned epi_flags(const mspl_epilogue_t* ep) { return ep ? ep->flags : 0u; }

static inline int check_epi(const mspl_epilogue_t* ep, int C, const char* who, bool raw_ok = false) {
    if (!ep) return MSPL_OK;
    MSPL_REQUIRE(ep->struct_size == sizeof(mspl_epilogue_t), MSPL_ERR_BAD_SHAPE,
                 "%s: epilogue struct size %u, this library expects %zu (caller built against another mspl_hip.h: ABI %d)", who,
                 ep->struct_size, sizeof(mspl_epilogue_t), 3);
    MSPL_REQUIRE(!ep->raw_out || (raw_ok && (ep->out_ctot == 0 || (ep->out_ctot == C && ep->out_coff == 0))), MSPL_ERR_UNSUPPORTED,
                 "%s: raw_out is for un-sliced convolution outputs only", who);
    if (ep->out_ctot > 0) {
        MSPL_REQUIRE(ep->out_coff >= 0 && ep->out_coff + C <= ep->out_ctot, MSPL_ERR_BAD_SHAPE,
                     "%s: channel slice [%d,%d) outside destination with %d channels", who,
                     ep->out_coff, ep->out_coff + C, ep->out_ctot);
    } else {
        MSPL_REQUIRE(ep->out_coff == 0, MSPL_ERR_BAD_SHAPE, "%s: out_coff without out_ctot", who);
    }
    MSPL_REQUIRE((ep->reinf_r == nullptr) == (ep->reinf_w == nullptr), MSPL_ERR_NULL_POINTER,
                 "%s: reinf_r and reinf_w must be given together", who);
    return MSPL_OK;
}

// Per-channel constants of the epilogue, fetched once per (thread, channel).
struct EpiCh {
    float scale, shift, alpha, rw0, rw1, rw2;
};

__device__ __forceinline__ EpiCh epi_channel(const Epi& e, int cabs) {
    EpiCh c;
    c.scale = e.scale ? e.scale[cabs] : 1.0f;
    c.shift = e.shift ? e.shift[cabs] : 0.0f;
    c.alpha = e.alpha ? e.alpha[cabs] : 1.0f;
    if (e.reinf_w) {
        c.rw0 = e.reinf_w[cabs * 3 + 0];
        c.rw1 = e.reinf_w[cabs * 3 + 1];
        c.rw2 = e.reinf_w[cabs * 3 + 2];
    } else {
        c.rw0 = c.rw1 = c.rw2 = 0.0f;
    }
    return c;
}

// v: accumulator; n: image; cabs: absolute destination channel; p: pixel in plane.
__device__ __forceinline__ float epi_apply(const Epi& e, const EpiCh& c, float v, int n, int cabs, int p) {
    const size_t off = ((size_t)n * e.ctot + cabs) * (size_t)e.hw + p;
    if (e.pre_add) v += e.pre_add[off];
    v = fmaf(v, c.scale, c.shift);
    if (e.reinf_r) {
        const float* r = e.reinf_r + (size_t)n * 3 * e.hw + p;
        v += c.rw0 * r[0] + c.rw1 * r[e.hw] + c.rw2 * r[2 * (size_t)e.hw];
    }
    if (e.residual) v += e.residual[off];
    if (e.alpha) v = v > 0.0f ? v : c.alpha * v;
    if (e.gate) v *= e.gate[(size_t)n * e.ctot + cabs];
    return v;
}

// Four consecutive pixels p .. p+3 of one row (hw % 4 == 0 and p % 4 == 0: every per-pixel operand is one 16-byte
// load instead of four 4-byte loads with their own 64-bit address arithmetic).  Same operation order as epi_apply.
__device__ __forceinline__ float4 epi_apply4(const Epi& e, const EpiCh& c, const float (&a)[4], int n, int cabs, int p) {
    const size_t off = ((size_t)n * e.ctot + cabs) * (size_t)e.hw + p;
    float v[4] = {a[0], a[1], a[2], a[3]};
    if (e.pre_add) {
        const float4 t = *reinterpret_cast<const float4*>(e.pre_add + off);
        v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = fmaf(v[j], c.scale, c.shift);
    if (e.reinf_r) {
        const float* r = e.reinf_r + (size_t)n * 3 * e.hw + p;
        const float4 r0 = *reinterpret_cast<const float4*>(r);
        const float4 r1 = *reinterpret_cast<const float4*>(r + e.hw);
        const float4 r2 = *reinterpret_cast<const float4*>(r + 2 * (size_t)e.hw);
        v[0] += c.rw0 * r0.x + c.rw1 * r1.x + c.rw2 * r2.x;
        v[1] += c.rw0 * r0.y + c.rw1 * r1.y + c.rw2 * r2.y;
        v[2] += c.rw0 * r0.z + c.rw1 * r1.z + c.rw2 * r2.z;
        v[3] += c.rw0 * r0.w + c.rw1 * r1.w + c.rw2 * r2.w;
    }
    if (e.residual) {
        const float4 t = *reinterpret_cast<const float4*>(e.residual + off);
        v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
    }
    if (e.alpha) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.0f ? v[j] : c.alpha * v[j];
    }
    if (e.gate) {
        const float gv = e.gate[(size_t)n * e.ctot + cabs];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= gv;
    }
    return make_float4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ size_t epi_offset(const Epi& e, int n, int cabs, int p) {
    return ((size_t)n * e.ctot + cabs) * (size_t)e.hw + p;
}

// ATen bilinear source index / weight, align_corners=True (UpSampleKernel: area_pixel_compute_source_index
// + guard_index_and_lambda).  scale = (in-1)/(out-1) in fp32, 0 when out == 1.
__device__ __forceinline__ void bilinear_src(float scale, int dst, int in_size, int& i0, int& i1, float& w0, float& w1) {
    float real = scale * (float)dst;
    int idx = (int)floorf(real);
    if (idx > in_size - 1) idx = in_size - 1;
    float lam = real - (float)idx;
    lam = fminf(fmaxf(lam, 0.0f), 1.0f);
    i0 = idx;
    i1 = idx + ((idx < in_size - 1) ? 1 : 0);
    w1 = lam;
    w0 = 1.0f - lam;
}

// ATen's align_corners=False source rule (area_pixel_compute_source_index, not cubic): src = scale*(dst+0.5)-0.5, clamped at 0;
// scale = in/out.  Used by the DeepLab-style heads (nn_layers/aspp.py:93, model/segmentation/deeplabv3.py:40).
__device__ __forceinline__ void bilinear_src_hp(float scale, int dst, int in_size, int& i0, int& i1, float& w0, float& w1) {
    float real = scale * ((float)dst + 0.5f) - 0.5f;
    real = real < 0.f ? 0.f : real;
    int idx = (int)floorf(real);
    if (idx > in_size - 1) idx = in_size - 1;
    float lam = real - (float)idx;
    lam = fminf(fmaxf(lam, 0.0f), 1.0f);
    i0 = idx;
    i1 = idx + ((idx < in_size - 1) ? 1 : 0);
    w1 = lam;
    w0 = 1.0f - lam;
}

// Sum over the 64 lanes of a wave, total in lane 63: seven DPP adds (a __shfl_down ladder is six ds_bpermute round trips).
__device__ __forceinline__ float wave_sum_dpp(float v) {
    // total in lane 63 (row_shr 1, 2, 3 within rows of 16, then across bank groups and rows: the canonical 7-step DPP reduction)
#define MSPL_DPP(x, ctrl, rm, bm) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, rm, bm, true))
    float t = v + MSPL_DPP(v, 0x111, 0xf, 0xf);
    t += MSPL_DPP(v, 0x112, 0xf, 0xf);
    t += MSPL_DPP(v, 0x113, 0xf, 0xf);
    t += MSPL_DPP(t, 0x114, 0xf, 0xe);
    t += MSPL_DPP(t, 0x118, 0xf, 0xc);
    t += MSPL_DPP(t, 0x142, 0xa, 0xf);          // row_bcast:15
    t += MSPL_DPP(t, 0x143, 0xc, 0xf);          // row_bcast:31
#undef MSPL_DPP
    return t;
}

static inline float bilinear_scale(int in_size, int out_size) {
    return out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.0f;
}

}  // namespace mspl
