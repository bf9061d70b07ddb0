/*
 * mspl_hip.h -- C ABI of libmspl_hip.so: hand-written HIP kernels (gfx950 / CDNA4) for the MSPL
 * segmentation + multi-source pseudo-label hot path.
 *
 * The reference (ShigemichiMatsuzaki/MSPL) is pure Python on stock PyTorch: it has no FFI, so these
 * entry points replace *ATen op sequences* inside the reference's Python modules.  Each entry cites
 * the reference lines whose arithmetic it performs (paths relative to the reference root).
 *
 * Conventions
 *  - plain C types only; every tensor is a raw device pointer, dense NCHW fp32 unless stated.
 *  - the caller owns every buffer (inputs, outputs, workspaces); the library never allocates,
 *    frees or retains device memory.
 *  - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream),
 *    re-entrant, capturable into a hipGraph (no host synchronisation inside).
 *  - return value: MSPL_OK (0) or a negative mspl_status; mspl_last_error() gives the text for the
 *    calling thread.  No C++ exception crosses the boundary.
 *  - an output may be a channel slice of a larger tensor (free torch.cat): the epilogue carries the
 *    destination tensor's total channel count and the channel offset of this op's first output.
 */
#ifndef MSPL_HIP_H
#define MSPL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    MSPL_OK = 0,
    MSPL_ERR_BAD_SHAPE = -1,      /* non-positive / inconsistent dimensions */
    MSPL_ERR_UNSUPPORTED = -2,    /* e.g. dilation set or class count outside the compiled variants */
    MSPL_ERR_NULL_POINTER = -3,
    MSPL_ERR_HIP = -4             /* launch failed; text carries hipGetErrorString */
} mspl_status;

/*
 * Fused epilogue shared by the producer kernels.  For an accumulator value v of output channel c
 * (absolute channel index in the destination tensor = out_coff + c), image n, pixel p:
 *      v += pre_add[n, c, p]                                   (if pre_add)
 *      v  = v * scale[c] + shift[c]                            (folded eval-mode BatchNorm / conv bias)
 *      v += sum_j reinf_w[c*3+j] * reinf_r[n, j, p]            (DownSampler input reinforcement)
 *      v += residual[n, c, p]                                  (EESP identity link)
 *      v  = v > 0 ? v : alpha[c] * v                           (per-channel PReLU)
 *      v *= gate[n, c]                                         (EfficientPWConv sigmoid gate)
 * Every pointer may be NULL (step skipped).  scale/shift/alpha/reinf_w/gate are indexed by the
 * ABSOLUTE destination channel; pre_add/residual have the destination tensor's shape.
 * raw_out (convolutions only: mspl_conv1x1_fwd, mspl_conv3x3_fwd; the output must not be a channel slice): the accumulator value v BEFORE any of the
 * steps above is also stored there, same shape as the destination -- the training forward keeps it for the backward of the
 * BatchNorm / PReLU (d gamma and the PReLU sign need the un-transformed convolution result), which saves the separate
 * BatchNorm + PReLU launch of the unfused form.
 */
typedef struct {
    const float* scale;
    const float* shift;
    const float* alpha;
    const float* pre_add;
    const float* residual;
    const float* reinf_r;   /* (N, 3, Ho, Wo) */
    const float* reinf_w;   /* (out_ctot, 3) */
    const float* gate;      /* (N, out_ctot) */
    int32_t out_ctot;       /* channels of the destination tensor */
    int32_t out_coff;       /* first destination channel written by this call */
    float* raw_out;         /* see above; NULL = not wanted */
    uint32_t struct_size;   /* sizeof(mspl_epilogue_t) of the CALLER's header: a library built against another layout rejects the
                               call (MSPL_ERR_BAD_SHAPE, "epilogue struct size") instead of reading fields that are not there */
    uint32_t flags;         /* MSPL_LAUNCH_* launch-shape preferences of THIS call (never change results) */
} mspl_epilogue_t;

/* Launch-shape preference of a call: the caller keeps several independent passes in flight (mspl_amd.uest.PipelinedLabelPass), so
 * fewer, longer workgroups are preferred (another pass fills the ramp and tail) and the big-LDS fused K1+K2 launch is not used.
 * Per call, not per process: the library holds no mutable global state and is re-entrant (rounds 1-2 had a process-wide switch). */
#define MSPL_LAUNCH_THROUGHPUT 1u
/* Form of the stride-2 depthwise launch (mspl_eesp_dw_hff_fwd): by default the library picks between the register-streaming and
 * the direct / LDS-tiled forms from the shape; OFF never takes the streaming form, FORCE takes it whenever the shape allows.  The
 * forms are bit-identical (tests compare them through these flags; rounds 2-4 read an environment variable per call instead). */
#define MSPL_LAUNCH_K2_STREAM_OFF 2u
#define MSPL_LAUNCH_K2_STREAM_FORCE 4u

const char* mspl_version(void);
/* ABI revision of this library; it changes whenever a struct layout or a signature in this header does.  3 = this header. */
int mspl_abi_version(void);
/* Copies the calling thread's last error text (NUL-terminated) into buf; returns its length. */
size_t mspl_last_error(char* buf, size_t cap);

/* ---------------------------------------------------------------------------------------------
 * K2  EESP split/transform/HFF: 4 parallel dilated depthwise 3x3 convs of one (N,n,H,W) tensor,
 *     hierarchical add out_k += out_{k-1}, channel concat, then the epilogue (br_after_cat).
 *     Replaces nn_layers/eesp.py:68-80 (+ espnet_utils.py:118-142 CDilated, :39-60 BR).
 *     w: (4, n, 3, 3) = spp_dw[0..3].conv.weight stacked; dil[4] in {1,2,3,4}, ascending sets
 *     {1,2,3,4} {1,1,2,3} {1,1,1,2} are compiled; stride in {1,2}; padding = dilation.
 *     out: (N, 4n, Ho, Wo) slice, Ho = (H-1)/stride + 1.
 */
int mspl_eesp_dw_hff_fwd(const float* x, const float* w, const int32_t dil[4], int32_t stride,
                         int32_t N, int32_t n, int32_t H, int32_t W,
                         const mspl_epilogue_t* ep, float* out, void* stream);

/* K1/K3  grouped 1x1 convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32) + epilogue.
 *     Replaces nn_layers/eesp.py:67 (proj_1x1), :77-93 (conv_1x1_exp + residual + module_act),
 *     nn_layers/eesp.py:117-120,142 (inp_reinf), efficient_pyramid_pool.py:22,31 (projection / final
 *     1x1), espnet_utils.py:8-37,62-89.   x: (N,Cin,HW)  w: (Cout, Cin/groups)  out: (N,.,HW) slice.
 */
int mspl_conv1x1_fwd(const float* x, const float* w, int32_t N, int32_t Cin, int32_t Cout,
                     int32_t groups, int32_t HW, const mspl_epilogue_t* ep, float* out, void* stream);

/* Generic grouped 3x3 convolution, padding 1, dilation 1, stride 1 or 2 (direct, LDS-tiled) + epilogue.
 *     Replaces model/classification/espnetv2.py:61 (level1 stem), nn_layers/eesp.py:118 (inp_reinf.0),
 *     efficient_pyramid_pool.py:24 (depthwise stages), :30 (merge CBR), efficient_pt.py:21 (expansion).
 *     shuffle_groups > 0 reads the input through the channel shuffle of cnn_utils.py:119-125
 *     (logical channel j = physical channel (j % sg) * (Cin / sg) + j / sg), so Shuffle is never
 *     materialised.  w: (Cout, Cin/groups, 3, 3).
 */
int mspl_conv3x3_fwd(const float* x, const float* w, int32_t N, int32_t Cin, int32_t Cout,
                     int32_t groups, int32_t H, int32_t W, int32_t stride, int32_t shuffle_groups,
                     const mspl_epilogue_t* ep, float* out, void* stream);

/* K1 + K2 in one launch for stride-1 EESP blocks whose planes fit LDS (nn_layers/eesp.py:60-80: proj_1x1 = grouped 1x1 + BN + PReLU,
 * then the four dilated depthwise 3x3 + HFF + cat + br_after_cat).  x (N,Cin,H,W); wp (n, Cin/groups): the projection's weights;
 * pscale/pshift/palpha (n): its folded BatchNorm and PReLU (NULL = identity); w (4,n,3,3), dil, ep, out (N,4n,H,W): as
 * mspl_eesp_dw_hff_fwd with stride 1.  Covered: Cin/groups in {64,128}, n/groups a multiple of 16, H*W % 4 == 0, W even, W,H <= 64,
 * dilations {1,1,2,3} or {1,2,3,4}, M = n/groups in {16,32}; mspl_eesp_proj_dw_hff_fits() returns 1 for covered shapes unless
 * launch_flags has MSPL_LAUNCH_THROUGHPUT (with several launches in flight the two-launch form measured faster) (callers run
 * mspl_conv1x1_fwd + mspl_eesp_dw_hff_fwd otherwise; _fwd returns MSPL_ERR_UNSUPPORTED). */
int mspl_eesp_proj_dw_hff_fits(int32_t N, int32_t Cin, int32_t n, int32_t groups, int32_t H, int32_t W, const int32_t dil[4],
                               uint32_t launch_flags);
int mspl_eesp_proj_dw_hff_fwd(const float* x, const float* wp, const float* pscale, const float* pshift, const float* palpha,
                              const float* w, const int32_t dil[4], int32_t N, int32_t Cin, int32_t n, int32_t groups,
                              int32_t H, int32_t W, const mspl_epilogue_t* ep, float* out, void* stream);

/* K2 + K3 in one launch for stride-1 EESP blocks (nn_layers/eesp.py:68-93: the four dilated depthwise 3x3 of the reduced tensor +
 * HFF + cat + br_after_cat, then conv_1x1_exp (grouped 1x1, 4 groups) + BatchNorm + residual link + module_act).  The 4n-channel
 * concatenation never exists in memory: a workgroup computes the branch values of a band of rows chunk by chunk of reduced channels
 * into LDS and feeds them to the matrix cores as the B operand of the expansion.  Results are bit-identical to
 * mspl_eesp_dw_hff_fwd followed by mspl_conv1x1_fwd.
 *   r (N,n,H,W): proj_1x1's output.  packed: mspl_eesp_dw_exp_pack_floats(n) floats written by mspl_eesp_dw_exp_pack from
 *   w4 (4,n,3,3), br_after_cat's folded scale/shift and PReLU slope (4n each) and conv_1x1_exp's weight (4n, n) -- the caller
 *   caches it per weight version.  ep: scale/shift (conv_1x1_exp's folded BN), alpha (module_act) and residual (the block's
 *   input, (N,4n,H,W)) are all required; nothing else may be set.  out (N,4n,H,W).
 * Covered (mspl_eesp_dw_exp_fits returns 1): (n, W, dil) = (128, 30 | 32, {1,1,2,3}) or (64, 60 | 64, {1,2,3,4}) -- levels 4 / 3 of
 * ESPDNet(-UE) s=2.0 for 480- and 512-pixel-wide inputs --, any H; callers run the two-launch form otherwise (_fwd / _pack return
 * MSPL_ERR_UNSUPPORTED). */
int mspl_eesp_dw_exp_fits(int32_t N, int32_t n, int32_t H, int32_t W, const int32_t dil[4], uint32_t launch_flags);
int64_t mspl_eesp_dw_exp_pack_floats(int32_t n);
int mspl_eesp_dw_exp_pack(const float* w4, const float* bscale, const float* bshift, const float* balpha, const float* wexp,
                          int32_t n, int32_t H, int32_t W, const int32_t dil[4], float* packed, void* stream);
int mspl_eesp_dw_exp_fwd(const float* r, const float* packed, const int32_t dil[4], int32_t N, int32_t n, int32_t H, int32_t W,
                         const mspl_epilogue_t* ep, float* out, void* stream);
/* The same launch + the NEXT block's proj_1x1 (nn_layers/eesp.py:67 of the following EESP block: grouped 1x1 over the 4n channels
 * this launch has just produced, + BatchNorm + PReLU) as a second matrix stage on the accumulators: group g's n output rows are
 * exactly the input channels of the next projection's group g.  next_packed: mspl_eesp_dw_exp_next_pack_floats(n) floats written
 * by mspl_eesp_dw_exp_next_pack from the next block's proj_1x1 weight (n, n); nscale / nshift / nalpha (n): its folded BatchNorm
 * and PReLU; rnext (N,n,H,W): the reduced tensor the next block's K2 reads (what mspl_conv1x1_fwd would have produced from `out`;
 * equal to it up to the summation order of the K = n products: the four K-slices of a group are summed through LDS). */
int64_t mspl_eesp_dw_exp_next_pack_floats(int32_t n);
int mspl_eesp_dw_exp_next_pack(const float* w1, int32_t n, float* next_packed, void* stream);
int mspl_eesp_dw_exp_next_fwd(const float* r, const float* packed, const int32_t dil[4], int32_t N, int32_t n, int32_t H, int32_t W,
                              const mspl_epilogue_t* ep, float* out, const float* next_packed, const float* nscale,
                              const float* nshift, const float* nalpha, float* rnext, void* stream);

/* AvgPool2d(kernel 3, stride 2, padding 1, count_include_pad) + epilogue.
 *     Replaces nn_layers/eesp.py:115,128 and the image pyramid of :136-140.
 */
int mspl_avgpool3x3s2_fwd(const float* x, int32_t N, int32_t C, int32_t H, int32_t W,
                          const mspl_epilogue_t* ep, float* out, void* stream);

/* Bilinear resize (ATen index rules) + epilogue.  align_corners != 0: F.interpolate(align_corners=True) at
 *     efficient_pyramid_pool.py:48,50 and espdnet_ue.py:110,301-302; align_corners == 0: the default rule the DeepLab-style
 *     heads use (nn_layers/aspp.py:93, model/segmentation/deeplabv3.py:40).
 */
int mspl_bilinear_fwd(const float* x, int32_t N, int32_t C, int32_t Hi, int32_t Wi, int32_t Ho,
                      int32_t Wo, int32_t align_corners, const mspl_epilogue_t* ep, float* out, void* stream);

/* adaptive_avg_pool2d (window [floor(i*I/O), ceil((i+1)*I/O)) ) + epilogue.
 *     Replaces efficient_pyramid_pool.py:46,52.
 */
int mspl_adaptive_avgpool_fwd(const float* x, int32_t N, int32_t C, int32_t Hi, int32_t Wi,
                              int32_t Ho, int32_t Wo, const mspl_epilogue_t* ep, float* out,
                              void* stream);

/* Elementwise epilogue only (BatchNorm+PReLU "BR" blocks, espdnet_ue.py:89-97, cnn_utils.py:85-105).
 * x has the destination tensor's shape (N, out_ctot, HW); channels [out_coff, out_coff+C) are processed. */
int mspl_pointwise_fwd(const float* x, int32_t N, int32_t C, int32_t HW, const mspl_epilogue_t* ep,
                       float* out, void* stream);

/* EfficientPWConv gate: sigmoid(W . global_avg_pool(x)).  Replaces efficient_pt.py:13-17,26.
 *     x: (N,Cin,HW)  w: (Cout,Cin)  mean_ws: (N,Cin) workspace  gate: (N,Cout).
 */
int mspl_gap_gate_fwd(const float* x, const float* w, int32_t N, int32_t Cin, int32_t Cout,
                      int32_t HW, float* mean_ws, float* gate, void* stream);

/* RGB-D fusion gate blend.  Replaces nn_layers/fusion_gate.py:26-47 after its 1x1 convolution (which runs through
 * mspl_conv1x1_fwd on the two halves of the weight, no concatenated copy):
 *     z != NULL:  w = sigmoid(z); out = rgb*w + depth*(1-w)          (is_trainable=True,  :27-41)
 *     z == NULL:  out = rgb + depth                                  (is_trainable=False, :43-44)
 * All operands `count` fp32 elements, 16-byte aligned.  The reference's `torch.ones(size).to('cuda')` temporary
 * (:38) is not materialised. */
int mspl_fusion_gate_fwd(const float* z, const float* rgb, const float* depth, int64_t count, float* out, void* stream);
/* Its backward: grgb = gy*w, gdepth = gy*(1-w), gz = gy*(rgb-depth)*w*(1-w)   (z == NULL: grgb = gdepth = gy). */
int mspl_fusion_gate_bwd(const float* z, const float* rgb, const float* depth, const float* gy, int64_t count,
                         float* gz, float* grgb, float* gdepth, void* stream);

/* mspl_avgpool3x3s2_fwd that also leaves partial plane sums of its input: psum (N*C, nblk) with nblk =
 * mspl_avgpool3x3s2_psum_blocks(H, W) (one partial per workgroup of a plane, every slot written, summed in order by the
 * consumer: deterministic), and the EfficientPWConv gate computed from them: gate (N,Cout) = sigmoid(W (Cout,Cin) . sum_j psum / HW).
 * Together they replace mspl_gap_gate_fwd's extra read of the encoder outputs (efficient_pt.py:13-17,26) where a DownSampler
 * pools the same tensor. */
int mspl_avgpool3x3s2_psum_blocks(int32_t H, int32_t W);
int mspl_avgpool3x3s2_psum_fwd(const float* x, int32_t N, int32_t C, int32_t H, int32_t W, const mspl_epilogue_t* ep, float* out,
                               float* psum, void* stream);
int mspl_gate_from_sums_fwd(const float* psum, const float* w, int32_t N, int32_t Cin, int32_t Cout, int32_t nblk, int32_t HW,
                            float* gate, void* stream);

/* K6 prologue: the low-resolution branches' maps for mspl_pyrpool_fused_fwd in ONE launch (one workgroup per (image,
 *     channel) plane): out[i] (N,P,hs[i],ws[i]) = dw3x3(adaptive_avg_pool2d(x, (hs[i],ws[i]))) with stage_w[i] (P,1,3,3);
 *     nn_layers/efficient_pyramid_pool.py:44-50 for the scales < 1.  A workgroup stages a band of input rows and the
 *     pooled rows it yields in LDS; mspl_pyr_down_prep_lds_bytes() returns the bytes that takes for a shape, or 0 when no
 *     band fits (very wide maps: use mspl_adaptive_avgpool_fwd + mspl_conv3x3_fwd per branch instead). */
int64_t mspl_pyr_down_prep_lds_bytes(int32_t N, int32_t P, int32_t h, int32_t w, int32_t nb, const int32_t* hs,
                                     const int32_t* ws);
int mspl_pyr_down_prep_fwd(const float* x, int32_t N, int32_t P, int32_t h, int32_t w, int32_t nb, const int32_t* hs,
                           const int32_t* ws, const float* const* stage_w, float* const* out, void* stream);
/* The training forward of the same step: pooled[i] (N,P,hs[i],ws[i]), when given, also receives the pooled map itself
 *     (adaptive_avg_pool2d(x, (hs[i],ws[i]))): the depthwise convolution's weight gradient reads it.  pooled == NULL: as above. */
int mspl_pyr_down_prep_train_fwd(const float* x, int32_t N, int32_t P, int32_t h, int32_t w, int32_t nb, const int32_t* hs,
                                 const int32_t* ws, const float* const* stage_w, float* const* out, float* const* pooled,
                                 void* stream);
/* Autograd of those branches between the transposed bilinear interpolation and the full-resolution gradient, every branch of a
 *     pyramid in ONE launch (nn_layers/efficient_pyramid_pool.py:44-47): given g_e[i] = dL/d(dw3x3 output) (mspl_bilinear_bwd),
 *     gw[i] (P,1,3,3) += the depthwise weight gradient (atomically: gw may be the parameter's gradient buffer), gx[i] (N,P,h,w) =
 *     adaptive_avg_pool2d^T(dw3x3^T(g_e[i])).  nb <= 2.  Replaces conv3x3 with flipped weights + mspl_conv_bwd_weight +
 *     mspl_adaptive_avgpool_bwd per branch. */
int mspl_pyr_down_mid_bwd(const float* const* g_e, const float* const* pooled, const float* const* stage_w, int32_t N, int32_t P,
                          int32_t h, int32_t w, int32_t nb, const int32_t* hs, const int32_t* ws, float* const* gw,
                          float* const* gx, void* stream);

/* K6  fused EfficientPyrPool body: all branches + merge_layer.0 (BN+PReLU) + Shuffle + merge_layer.2 (grouped
 *     3x3 + BN + PReLU) in one pass over the projected tensor.  Replaces nn_layers/efficient_pyramid_pool.py:39-58
 *     (everything between projection_layer and the final 1x1) and cnn_utils.py:119-125.
 *     x: (N,P,h,w) projection output.  Branch i works on an hs[i] x ws[i] grid:
 *       hs>h : bilinear_up -> depthwise 3x3 (stage_w[i], (P,1,3,3)) -> adaptive_avg_pool, all inside the kernel;
 *       hs==h: depthwise 3x3 (stage_w[i]);
 *       hs<h : down_e[i] = (N,P,hs,ws) = dw3x3(adaptive_avg_pool(x)) computed beforehand; up-sampled here.
 *     br_scale/shift/alpha: nb*P folded merge_layer.0 constants (branch-major, like torch.cat);
 *     merge_w: (P, nb, 3, 3) = merge_layer.2 conv weight; ep: merge_layer.2's BN+PReLU; out: (N,P,h,w) slice.
 */
int mspl_pyrpool_fused_fwd(const float* x, int32_t N, int32_t P, int32_t h, int32_t w, int32_t nb,
                           const int32_t* hs, const int32_t* ws, const float* const* stage_w,
                           const float* const* down_e, const float* br_scale, const float* br_shift,
                           const float* br_alpha, const float* merge_w, const mspl_epilogue_t* ep,
                           float* out, void* stream);

/* ---- fused EfficientPyrPool body on the training path (nn_layers/efficient_pyramid_pool.py:39-58 + its autograd backward) ----
 * Forward: mspl_pyrpool_fused_fwd that also keeps what the backward needs: zcat (N, nb*P, h, w) = the branch values BEFORE
 * merge_layer.0's BatchNorm + PReLU, in torch.cat order (channel i*P + c), and -- through ep->raw_out -- the bare result of
 * merge_layer.2's convolution.  Always the LDS-tiled table form (MSPL_ERR_UNSUPPORTED when a tile does not fit LDS: callers run
 * the branch-by-branch form then). */
int mspl_pyrpool_fused_train_fwd(const float* x, int32_t N, int32_t P, int32_t h, int32_t w, int32_t nb,
                                 const int32_t* hs, const int32_t* ws, const float* const* stage_w,
                                 const float* const* down_e, const float* br_scale, const float* br_shift,
                                 const float* br_alpha, const float* merge_w, const mspl_epilogue_t* ep,
                                 float* out, float* zcat, void* stream);
/* 1 when mspl_pyrpool_fused_train_fwd covers these shapes (same planning code, nothing is launched). */
int mspl_pyrpool_fused_train_fits(int32_t N, int32_t P, int32_t h, int32_t w, int32_t nb, const int32_t* hs, const int32_t* ws);
/* Backward of merge_layer.2 (grouped 3x3 over the shuffled concatenation + BatchNorm + PReLU) and merge_layer.0 (BatchNorm +
 * PReLU over the concatenation) in one pass: gy (N,P,h,w) = dL/d(body output), mraw = the kept convolution result, zcat as above.
 * gt (nb, N, P, h, w), BRANCH-major: dL/d(branch value), i.e. through both BatchNorm/PReLU pairs and the transposed convolution.
 * Parameter gradients are ACCUMULATED (atomics; the caller zeroes them or passes the parameters' own gradient buffers):
 * g_br_scale/g_br_shift/g_br_alpha (nb*P), g_merge_w (P,nb,3,3), g_m_scale/g_m_shift/g_m_alpha (P).  br_mean/br_inv (nb*P) and
 * m_mean/m_inv (P): the frozen BatchNorms' running mean and rsqrt(var + eps) -- then (scale, shift) are the folded
 * (gamma*inv, beta - mean*gamma*inv) and the g_*_scale / g_*_shift outputs receive d gamma / d beta; NULL: plain d scale / d shift. */
int mspl_pyrpool_merge_bwd(const float* gy, const float* mraw, const float* zcat, int32_t N, int32_t P, int32_t h,
                           int32_t w, int32_t nb, const float* br_scale, const float* br_shift, const float* br_alpha,
                           const float* br_mean, const float* br_inv, const float* merge_w, const float* m_scale,
                           const float* m_shift, const float* m_alpha, const float* m_mean, const float* m_inv,
                           float* gt, float* g_br_scale, float* g_br_shift, float* g_br_alpha, float* g_merge_w,
                           float* g_m_scale, float* g_m_shift, float* g_m_alpha, void* stream);
/* Backward of the branches with hs >= h (adaptive_avg_pool2d(dw3x3(bilinear_up(x))); hs == h: the plain depthwise 3x3), up to
 * three of them in one launch: gt[i] (N,P,h,w) = dL/d(branch i's value), stage_w[i] (P,1,3,3).  gx (N,P,h,w) is OVERWRITTEN with
 * the sum of the branches' input gradients + add0 + add1 (optional (N,P,h,w) tensors: the low-resolution branches'
 * contributions); gw[i] (P,1,3,3) are ACCUMULATED.  Nothing at up-sampled resolution is written to memory.
 * _fits: 1 when the shapes are covered (branch sizes within [1x, 3x] of the map, tile fits LDS). */
int mspl_pyrpool_branch_bwd_fits(int32_t N, int32_t P, int32_t h, int32_t w, int32_t nb, const int32_t* hs, const int32_t* ws);
int mspl_pyrpool_branch_bwd(const float* x, int32_t N, int32_t P, int32_t h, int32_t w, int32_t nb, const int32_t* hs,
                            const int32_t* ws, const float* const* stage_w, const float* const* gt, float* const* gw,
                            const float* add0, const float* add1, float* gx, void* stream);

/* K8+K9  label epilogue: bilinear(align_corners) upsample of both heads to (H,W), o = main + 0.5*aux,
 *     class = first-max argmax_c o (== np.argmax of softmax2d(o) up to exp() rounding ties), optional
 *     id LUT, optional softmax probabilities and KL(main||aux) map.
 *     Replaces espdnet_ue.py:301-302 + uest_seg_multi_os.py:685-691 (get_output), :903-912 (argmax+LUT),
 *     loss_fns/segmentation_loss.py:181-189 (PixelwiseKLD).
 *     main: (N,C,Hm,Wm)  aux: (N,C,Ha,Wa) or NULL (single-head nets: o = main, kld = 0)
 *     lut: C bytes or NULL;  labels: (N,H,W) uint8;  prob: (N,C,H,W) or NULL;  kld: (N,H,W) or NULL;
 *     main_up/aux_up: (N,C,H,W) or NULL -- the upsampled logits themselves (what model(x) returns).
 *     C <= 255 (labels are uint8).
 */
int mspl_label_epilogue_fwd(const float* main, const float* aux, int32_t N, int32_t C,
                            int32_t Hm, int32_t Wm, int32_t Ha, int32_t Wa, int32_t H, int32_t W,
                            const uint8_t* lut, uint8_t* labels, float* prob, float* kld,
                            float* main_up, float* aux_up, void* stream);

/* K8+K9 with the class histogram of the labels it writes: the single-source label pass (uest_seg_multi_os.py:785-815,
 *     one model relabelling its own domain) without a separate merge launch -- what label_epilogue + merge_labels(S=1,
 *     thresh=1) computed.  hist: num_classes (<= 32) uint64 bins, accumulated into (caller zeroes); labels >= num_classes
 *     are not counted.  C <= 24; labels required; kld optional.  workspace: caller-owned scratch of at least
 *     mspl_label_epilogue_hist_workspace_bytes(N, H, W) bytes (per-workgroup partial counts, summed by a second tiny launch:
 *     same-address device atomics from ~10^4 workgroups serialise).
 */
int64_t mspl_label_epilogue_hist_workspace_bytes(int32_t N, int32_t H, int32_t W);
/* 1 when mspl_label_epilogue_hist_fwd covers the shape (the LDS-staged tile of both heads fits; Ha = Wa = 0: no second head);
 * callers run mspl_label_epilogue_fwd + mspl_merge_labels_fwd(S = 1, thresh = 1) otherwise. */
int mspl_label_epilogue_hist_fits(int32_t N, int32_t C, int32_t Hm, int32_t Wm, int32_t Ha, int32_t Wa, int32_t H, int32_t W);
int mspl_label_epilogue_hist_fwd(const float* main, const float* aux, int32_t N, int32_t C,
                                 int32_t Hm, int32_t Wm, int32_t Ha, int32_t Wa, int32_t H, int32_t W,
                                 const uint8_t* lut, uint8_t* labels, float* kld, unsigned long long* hist,
                                 int32_t num_classes, void* workspace, int64_t workspace_bytes, void* stream);

/* K10  cross-source label merge + class histogram.  Replaces uest_seg_multi_os.py:695-718
 *     (merge_outputs) and :919-921.  src[s]: npix uint8 class maps (already in target ids),
 *     S <= 8, num_classes <= 32.  out[p] = first-max argmax_c count_c(p), or `fill` when the
 *     winning count < thresh.  hist (num_classes x uint64, device) is ACCUMULATED into (caller zeroes).
 */
int mspl_merge_labels_fwd(const uint8_t* const* src, int32_t S, int64_t npix, int32_t num_classes,
                          int32_t thresh, int32_t fill, uint8_t* out, unsigned long long* hist,
                          void* stream);

/* =============================================================================================
 * Training step (uest_seg_multi_os.py:958-1089, BatchNorm frozen = eval mode).
 * Backward of the forward ops above + K11 (loss) + Adam.  All gradients fp32, same layouts as the forward.
 * ============================================================================================= */

/* Data gradient of a bias-free grouped convolution (K in {1,3}, padding = dilation*(K-1)/2; the autograd backward
 * of nn.Conv2d in espnet_utils.py / cnn_utils.py / efficient_pyramid_pool.py).  gy: (N,Cout,Ho,Wo)  w: (Cout,Cin/g,K,K)
 * gx: (N,Cin,H,W), overwritten or accumulated into. */
int mspl_conv_bwd_data(const float* gy, const float* w, int32_t N, int32_t Cin, int32_t Cout, int32_t groups,
                       int32_t H, int32_t W, int32_t K, int32_t stride, int32_t dilation, int32_t accumulate,
                       float* gx, void* stream);

/* Weight gradient of the same convolution.  x: (N,Cin,H,W)  gw: (Cout,Cin/g,K,K). */
int mspl_conv_bwd_weight(const float* gy, const float* x, int32_t N, int32_t Cin, int32_t Cout, int32_t groups,
                         int32_t H, int32_t W, int32_t K, int32_t stride, int32_t dilation, int32_t accumulate,
                         float* gw, void* stream);
/* Weight gradients of nprob grouped 1x1 convolutions in as few launches as possible, each ACCUMULATED (atomically) into gw[i] --
 *     parameter gradient buffers that were zeroed at the start of the step.  Problem i: gy[i] (N,Cout,HW), x[i] (N,Cin,HW),
 *     gw[i] (Cout, Cin/groups).  Nothing in a backward chain waits for a weight gradient, so the training steps queue them
 *     (mspl_amd.autograd.WgradQueue) and flush the queue here: ~33 launches of 8-25 us per uest step become a handful. */
int mspl_conv1x1_wgrad_batch(const float* const* gy, const float* const* x, float* const* gw, const float* const* rowscale,
                             const int32_t* N, const int32_t* Cin, const int32_t* Cout, const int32_t* groups, const int32_t* HW, int32_t nprob,
                             void* stream);
/* rowscale: NULL, or per problem NULL / a (Cout) vector s: gw[i][co, :] += s[co] * sum -- gy[i] is then the gradient BEFORE a
 *     per-output-channel scale (a caller that keeps only the unscaled gradient in memory). */

/* DownSampler tail (nn_layers/eesp.py:131-144) without materialising torch.cat: y = PReLU(cat[a, b] + reinf).  a (N,nin,HW): avg-pooled
 * input; b (N,C-nin,HW): the strided EESP branch; reinf (N,C,HW) or NULL; alpha (C).  Backward: ga / gb (shapes of a / b, contiguous),
 * greinf (N,C,HW; NULL iff reinf is), galpha (C, ACCUMULATED with atomics: caller zeroes).  HW % 4 == 0, 16-byte aligned operands
 * (MSPL_ERR_UNSUPPORTED otherwise: callers fall back to cat + mspl_pointwise_fwd / mspl_affine_prelu_bwd). */
int mspl_down_tail_fwd(const float* a, const float* b, const float* reinf, const float* alpha, int32_t N, int32_t nin, int32_t C,
                       int32_t HW, float* y, void* stream);
int mspl_down_tail_bwd(const float* a, const float* b, const float* reinf, const float* gy, const float* alpha, int32_t N, int32_t nin,
                       int32_t C, int32_t HW, float* ga, float* gb, float* greinf, float* galpha, void* stream);

/* Backward of y = PReLU((c + pre_add) * scale + shift + residual) (folded eval BatchNorm + PReLU, mspl_pointwise_fwd).
 * Any of pre_add/residual/scale/shift/alpha may be NULL.  Outputs: gz = dL/d(pre-activation) (also the residual's
 * gradient; may be NULL), gc = gz*scale (gradient of c and pre_add; may be NULL); gscale/gshift/galpha (C floats each,
 * ACCUMULATED with atomics: caller zeroes; may be NULL). */
int mspl_affine_prelu_bwd(const float* c, const float* pre_add, const float* residual, const float* gy,
                          const float* scale, const float* shift, const float* alpha, int32_t N, int32_t C,
                          int32_t HW, float* gz, float* gc, float* gscale, float* gshift, float* galpha, void* stream);

/* Backward of mspl_avgpool3x3s2_fwd (gather form, gx overwritten). */
int mspl_avgpool3x3s2_bwd(const float* gy, int32_t N, int32_t C, int32_t H, int32_t W, float* gx, void* stream);
/* Backward of mspl_bilinear_fwd and of mspl_adaptive_avgpool_fwd (both gather form: one thread per input pixel,
 * deterministic, gx overwritten). */
int mspl_bilinear_bwd(const float* gy, int32_t N, int32_t C, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo,
                      float* gx, void* stream);
int mspl_adaptive_avgpool_bwd(const float* gy, int32_t N, int32_t C, int32_t Hi, int32_t Wi, int32_t Ho, int32_t Wo,
                              float* gx, void* stream);

/* out[i] = sum_p a[i,p] * b[i,p] over `planes` planes of HW elements (b NULL: plain sum).  Gate / GAP gradients. */
int mspl_plane_dot(const float* a, const float* b, int32_t planes, int32_t HW, float* out, void* stream);
/* gx[i,p] (+)= v[i] * mul. */
int mspl_plane_broadcast(const float* v, int32_t planes, int32_t HW, float mul, int32_t accumulate, float* gx,
                         void* stream);
/* Same backward when (scale, shift) is a folded FROZEN BatchNorm, scale = gamma*inv, shift = beta - mean*gamma*inv
 * (inv = rsqrt(running_var + eps)): the per-channel sums are transformed on the fly and ACCUMULATED into ggamma / gbeta /
 * galpha -- which may be the parameters' own gradient buffers (the caller's optimizer zeroes them once per step). */
int mspl_bn_prelu_bwd(const float* c, const float* pre_add, const float* residual, const float* gy, const float* scale,
                      const float* shift, const float* alpha, const float* bn_mean, const float* bn_inv, int32_t N, int32_t C,
                      int32_t HW, float* gz, float* gc, float* ggamma, float* gbeta, float* galpha, void* stream);
/* Backward of mspl_gap_gate_fwd's gate = sigmoid(W . mean): gw (Cout,Cin), gmean (N,Cin). */
int mspl_gap_gate_bwd(const float* ggate, const float* gate, const float* mean, const float* w, int32_t N,
                      int32_t Cin, int32_t Cout, float* gw, float* gmean, void* stream);
/* Same, but gw (Cout,Cin) is ACCUMULATED into with atomic adds (the parameter's own gradient buffer, zeroed by the caller). */
int mspl_gap_gate_bwd_accum(const float* ggate, const float* gate, const float* mean, const float* w, int32_t N,
                      int32_t Cin, int32_t Cout, float* gw, float* gmean, void* stream);
/* Backward of the hierarchical feature fusion of K2: out_k = sum_{j>=k} g_j over the 4 branch blocks of g (N,4n,HW);
 * out is branch-major (4,N,n,HW). */
int mspl_hff_suffix_sum(const float* g, int32_t N, int32_t n, int32_t HW, float* out, void* stream);

/* br_after_cat's BatchNorm + PReLU backward fused with mspl_hff_suffix_sum (the EESP block between conv_1x1_exp and the four
 * depthwise branches, nn_layers/eesp.py:76-80): z (N,4n,HW) = K2's raw concatenation, gy = dL/d(PReLU(BN(z))), scale/shift/alpha
 * (4n; NULL = identity / no activation); out (4,N,n,HW) branch-major = suffix sums of gy * (u > 0 ? 1 : alpha) * scale;
 * gscale/gshift/galpha (4n) ACCUMULATED, with bn_mean/bn_inv (4n) they receive d gamma / d beta of the frozen BatchNorm. */
int mspl_hff_bn_prelu_suffix_bwd(const float* z, const float* gy, const float* scale, const float* shift, const float* alpha,
                                 const float* bn_mean, const float* bn_inv, int32_t N, int32_t n, int32_t HW, float* out,
                                 float* gscale, float* gshift, float* galpha, void* stream);

/* K11  fused PixelwiseKLD + UncertaintyWeightedSegmentationLoss (loss_fns/segmentation_loss.py:146-189) as used at
 *      uest_seg_multi_os.py:1020-1023:  loss = ce_scale * mean_pix(w[t] * -log_softmax(pred+0.5aux)[t] * exp(-kld))
 *      + mean_pix(kld), kld NOT detached.  class_weights: C floats with the ignore class already zeroed; the mean runs
 *      over ALL pixels.  loss_acc (1 float) is accumulated into (caller zeroes); gpred/gaux (N,C,HW) = d loss / d logits,
 *      kld_out (N,HW) optional. */
int mspl_uw_loss_fwd_bwd(const float* pred, const float* aux, const int64_t* target, const float* class_weights,
                         int32_t N, int32_t C, int32_t HW, float ce_scale, float* loss_acc, float* gpred,
                         float* gaux, float* kld_out, void* stream);
/* The same with the loss AND both gradients multiplied by out_scale (a micro-batch lane of a step back-propagates loss / lanes:
 * the factor rides on the kernel's 1/npix instead of two full-size multiplies in the backward). */
int mspl_uw_loss_scaled_fwd_bwd(const float* pred, const float* aux, const int64_t* target, const float* class_weights,
                                int32_t N, int32_t C, int32_t HW, float ce_scale, float out_scale, float* loss_acc,
                                float* gpred, float* gaux, float* kld_out, void* stream);
/* K11 at head resolution: the same loss taken from the decoder's two outputs BEFORE their bilinear up-sampling to the label map
 * (model/segmentation/espdnet_ue.py:301-302, F.interpolate(..., mode='bilinear', align_corners=True) of main (N,C,Hm,Wm) and
 * aux (N,C,Ha,Wa) to H x W; target (N,H,W) int64): the up-sampling happens inside the loss kernel, the two full-size logit tensors are
 * never written.  gpred / gaux (N,C,H,W) = d loss / d (up-sampled logits), as mspl_uw_loss_scaled_fwd_bwd writes them: hand each to
 * mspl_bilinear_bwd for the gradient of its low-resolution map.  loss_acc is ACCUMULATED into (caller zeroes).  Class counts
 * mspl_uw_loss_heads_supported() says 1 for (1..8, 13, 20) and heads no larger than the label map; MSPL_ERR_UNSUPPORTED otherwise
 * (callers then take the three-step form). */
int mspl_uw_loss_heads_supported(int32_t C);
int mspl_uw_loss_heads_fwd_bwd(const float* main_lo, const float* aux_lo, const int64_t* target, const float* class_weights,
                               int32_t N, int32_t C, int32_t Hm, int32_t Wm, int32_t Ha, int32_t Wa, int32_t H, int32_t W,
                               float ce_scale, float out_scale, float* loss_acc, float* gpred, float* gaux, void* stream);

/* K13  dense (groups = 1) 1x1 / dilated 3x3 convolution on the fp32 matrix cores: the ASPP heads of nn_layers/aspp.py:7-99
 *      (Conv2d(Cin, Cout, k, padding = dilation, dilation) + BatchNorm + ReLU).  x: (N,Cin,H,W); w_packed: the conv weight
 *      re-laid as (k*k, Cout, Cin) (tap-major, 16-byte aligned; Cin % 32 == 0); output (N,Cout,H,W) slice with the
 *      scale / shift / alpha epilogue (fold the conv bias into shift; alpha = 0 is ReLU). */
int mspl_dense_conv_fwd(const float* x, const float* w_packed, int32_t N, int32_t Cin, int32_t Cout, int32_t H, int32_t W,
                        int32_t ksize, int32_t dilation, const mspl_epilogue_t* ep, float* out, void* stream);

/* MIOU.get_iou (utilities/metrics/segmentation_miou.py:13-44) on the device: argmax (first maximum) of logits (N,C,HW), or
 * ready labels (N,HW) uint8 (pass exactly one of the two), against int64 targets, in the reference's uint8 arithmetic (+1,
 * 255 wraps to 0 = ignored).  hist: 3*num_classes counters [area_inter | area_pred | area_mask], accumulated into (caller
 * zeroes); area_union = pred + mask - inter + 1e-6 is the caller's. */
int mspl_miou_areas_fwd(const float* logits, const uint8_t* labels, const int64_t* target, int32_t N, int32_t C, int32_t HW,
                        int32_t num_classes, unsigned long long* hist, void* stream);

/* Evaluation step epilogue: val_seg_ue (utilities/train_eval_seg.py:279-302) / the body of test() (uest_seg_multi_os.py:1196-1198)
 * after the network, in one pass over the low-resolution heads -- full-resolution logits are never written:
 *     o = upsample(main) + aux_weight * upsample(aux)   (bilinear, align_corners; aux NULL or aux_weight 0: main alone)
 *     loss_sums[0] += sum_valid w[t] * (lse(o) - o[t]),  loss_sums[1] += sum_valid w[t]     (nn.CrossEntropyLoss(weight, ignore_index)
 *         = loss_sums[0] / loss_sums[1]; valid: t != ignore_index and 0 <= t < C; class_weights NULL = 1; doubles, ACCUMULATED)
 *     areas[0..3K) += [area_inter | area_pred | area_mask] of MIOU(K).get_iou(o, target) in the reference's uint8 arithmetic
 *         (as mspl_miou_areas_fwd; K = miou_classes, normally C - 1)
 *     labels (N,H,W) uint8: optional argmax map (first maximum).
 * mspl_eval_batch_finalize: AverageMeter.update(loss.item(), n) on the device -- acc[0] += (loss_sums[0]/loss_sums[1]) * n,
 * acc[1] += n, acc[2] = this batch's loss; loss_sums is cleared for the next batch. */
int mspl_eval_epilogue_fwd(const float* main, const float* aux, const int64_t* target, const float* class_weights,
                           int32_t N, int32_t C, int32_t Hm, int32_t Wm, int32_t Ha, int32_t Wa, int32_t H, int32_t W,
                           float aux_weight, int32_t ignore_index, int32_t miou_classes, double* loss_sums,
                           unsigned long long* areas, uint8_t* labels, void* stream);
int mspl_eval_batch_finalize(double* loss_sums, double* acc, int32_t batch_images, void* stream);

/* Stand-alone loss modules (callers that compose them themselves instead of the fused K11 form):
 *  PixelwiseKLD.forward (loss_fns/segmentation_loss.py:181-189): kld (N,HW) = sum_c softmax(d1)*(log_softmax(d1)-log_softmax(d2));
 *  its backward: gd1/gd2 (N,C,HW) from gkld (N,HW); either output may be NULL. */
int mspl_pixelwise_kld_fwd(const float* d1, const float* d2, int32_t N, int32_t C, int32_t HW, float* kld, void* stream);
int mspl_pixelwise_kld_bwd(const float* d1, const float* d2, const float* gkld, int32_t N, int32_t C, int32_t HW,
                           float* gd1, float* gd2, void* stream);
/* Weighted cross entropy sums: sums[0] += sum_pix w[t] * -log_softmax(pred)[t] * exp(-u),  sums[1] += sum_{valid pix} w[t]
 * (caller zeroes sums; pixels with t == ignore_index or t outside [0,C) contribute nothing; u_weight / class_weights may be
 * NULL = 1).  UncertaintyWeightedSegmentationLoss.forward (segmentation_loss.py:155-175) = sums[0] / (N*HW);
 * SegmentationLoss 'ce' = nn.CrossEntropyLoss(weight, ignore_index) (segmentation_loss.py:24) = sums[0] / sums[1].
 * Backward: g = upstream gradient (1 device float); den = device pointer to sums[1] for the CrossEntropyLoss normalisation
 * or NULL for the mean over all N*HW pixels; gpred (N,C,HW) and gu (N,HW) are overwritten (either may be NULL). */
int mspl_weighted_ce_fwd(const float* pred, const int64_t* target, const float* u_weight, const float* class_weights,
                         int32_t ignore_index, int32_t N, int32_t C, int32_t HW, float* sums, void* stream);
int mspl_weighted_ce_bwd(const float* pred, const int64_t* target, const float* u_weight, const float* class_weights,
                         int32_t ignore_index, int32_t N, int32_t C, int32_t HW, const float* g, const float* den,
                         float* gpred, float* gu, void* stream);

/* ---- loader-side transforms: the step in front of the hot path -------------------------------------
 * Replace, for a whole batch of decoded uint8 images, data_loader/segmentation/greenhouse.py:216-222 (val_transforms =
 * Resize(size) -> Normalize()|Tensorize()) i.e. transforms/segmentation/data_transforms.py:191-212 (PIL BILINEAR for
 * rgb/depth, PIL NEAREST for labels) and :15-46 (to_tensor /255, normalize (x-MEAN)/STD).  Pillow's arithmetic (22-bit
 * fixed-point triangle filter, horizontal pass first, uint8 between the passes) is reproduced bit for bit.
 *
 * Host-side table builders (no GPU work; plain host arrays):
 *   mspl_resample_ksize(in, out)            taps per output sample (>0) or a negative status
 *   mspl_resample_coeffs(in, out, bounds, kk)   bounds (out,2) = (first source index, count); kk (out, ksize) weights
 *   mspl_nearest_index(in, out, idx)        source index per destination index for PIL NEAREST
 */
int mspl_resample_ksize(int32_t in_size, int32_t out_size);
int mspl_resample_coeffs(int32_t in_size, int32_t out_size, int32_t* bounds, int32_t* kk);
int mspl_nearest_index(int32_t in_size, int32_t out_size, int32_t* idx);
/* src (N,Hs,Ws,C) uint8 HWC (C = 3 RGB or 1 depth) -> out (N,C,H,W) fp32.  xb/xk/kx and yb/yk/ky: DEVICE copies of the
 * tables above for Ws->W and Hs->H (xb/xk/tmp may be NULL when Ws == W: Pillow skips that pass).  mean/std: C device
 * floats or both NULL (Tensorize).  flip: N device bytes (RandomFlip's mirror, data_transforms.py:49-66) or NULL.
 * tmp: (N,Hs,W,C) uint8 workspace. */
int mspl_preprocess_u8_fwd(const uint8_t* src, int32_t N, int32_t Hs, int32_t Ws, int32_t C, int32_t H, int32_t W,
                           const int32_t* xb, const int32_t* xk, int32_t kx, const int32_t* yb, const int32_t* yk,
                           int32_t ky, const float* mean, const float* stdv, const uint8_t* flip, uint8_t* tmp,
                           float* out, void* stream);
/* Label maps: src (N,Hs,Ws) uint8 -> out (N,H,W) int64 (torch.LongTensor(np.array(label)), data_transforms.py:38), NEAREST. */
int mspl_resize_label_fwd(const uint8_t* src, int32_t N, int32_t Hs, int32_t Ws, int32_t H, int32_t W,
                          const int32_t* yi, const int32_t* xi, const uint8_t* flip, int64_t* out, void* stream);

/* Native label-map writer (host threads + zlib; no device work): replaces the per-image `Image.fromarray(label).save(png)`
 * of the label loop (uest_seg_multi_os.py:929-931).  create(workers, deflate level 0..9) -> handle or NULL.
 * submit: `host` = n maps (n,H,W) uint8 in host memory that must stay valid and unchanged until the ticket is done; paths: n
 * file names (copied); event: a hipEvent_t recorded after the device->host copy that fills `host`, or NULL when the data is
 * already there.  Returns a ticket >= 0 or a negative status.  poll(ticket, block): 1 = all n files written (ticket is then
 * forgotten), 0 = pending, negative = failed.  destroy drains the queue and joins the threads. */
void* mspl_png_writer_create(int32_t workers, int32_t level);
int64_t mspl_png_writer_submit(void* writer, const uint8_t* host, int32_t n, int32_t H, int32_t W, const char* const* paths,
                               void* event);
int mspl_png_writer_poll(void* writer, int64_t ticket, int32_t block);
int mspl_png_writer_destroy(void* writer);

/* ---- supervised-loop pieces -------------------------------------------------------------------------
 * nn.BatchNorm2d in train() (model.train(), utilities/train_eval_seg.py:174): per-channel batch mean and 1/sqrt(biased
 * var + eps) over (N,HW) of z (N,C,HW); when running_mean/running_var are given they are updated in place,
 * r = (1-momentum)*r + momentum*stat (unbiased variance M/(M-1)).  ws: 2*C doubles of device workspace. */
int mspl_bn_batch_stats_fwd(const float* z, int32_t N, int32_t C, int32_t HW, float eps, float momentum,
                            float* running_mean, float* running_var, double* ws, float* mean, float* invstd, void* stream);
/* Same, and also the fold the affine / PReLU kernel applies: scale = gamma * invstd, shift = beta - mean * scale (C floats each). */
int mspl_bn_batch_stats_fold_fwd(const float* z, int32_t N, int32_t C, int32_t HW, float eps, float momentum,
                                 float* running_mean, float* running_var, const float* gamma, const float* beta,
                                 double* ws, float* mean, float* invstd, float* scale, float* shift, void* stream);
/* The training-mode BatchNorm + PReLU node of the supervised loop with HALF the launches: a per-BatchNorm persistent workspace
 * (mspl_bn_fused_workspace_bytes(C) bytes, zeroed ONCE by the caller, handed back zeroed by every call) replaces the memset + finalize
 * launches of mspl_bn_batch_stats_fold_fwd and the separate mspl_bn_batch_stats_bwd_coeffs launch: the workgroup that adds a
 * channel's last partial finishes the channel.  _fused_fwd: same outputs as _fold_fwd; num_batches_tracked (may be NULL): the module's
 * int64 counter, incremented by one (nn.BatchNorm2d does it per forward: an ATen launch per BatchNorm otherwise).  mspl_bn_train_prelu_bwd: the backward of
 * y = PReLU(z * scale + shift + residual) with (scale, shift) the batch-statistics fold of z: gres (may be NULL), gc (may be NULL: see mspl_bn_train_prelu_bwd_apply) = direct gradient
 * of z through the affine map, ggamma / gbeta (accumulate != 0: added to), galpha (ACCUMULATED, caller zeroes or passes the parameter's
 * gradient), and the coefficients p, q (C floats each) of the statistics' path: dL/dz = p * z + q + gc (one mspl_pointwise_fwd). */
/* merge_layer.0 (BatchNorm fold + PReLU over the concatenation), Shuffle and merge_layer.2's grouped 3x3 from KEPT branch values
 * (nn_layers/efficient_pyramid_pool.py:51-58): out (N,P,h,w) = the bare convolution result, zcat (N, nb*P, h, w) in torch.cat order,
 * merge_w (P, nb, 3, 3).  The batch-statistics pyramid node runs it between the two statistics passes. */
int mspl_pyrpool_merge_fwd(const float* zcat, int32_t N, int32_t P, int32_t h, int32_t w, int32_t nb, const float* br_scale,
                           const float* br_shift, const float* br_alpha, const float* merge_w, float* out, void* stream);
/* Statistics path of a batch-statistics BatchNorm over a concatenation whose gradient is branch-major (the pyramid body's merge_layer.0):
 * g (nb,N,P,HW) += p[i*P+c] * z[n, i*P+c] + q[i*P+c] with z (N, nb*P, HW); HW % 4 == 0. */
int mspl_bn_stats_path_add(float* g, const float* z, const float* p, const float* q, int32_t N, int32_t P, int32_t nb, int32_t HW,
                           void* stream);
int64_t mspl_bn_fused_workspace_bytes(int32_t C);
int mspl_bn_batch_stats_fused_fwd(const float* z, int32_t N, int32_t C, int32_t HW, float eps, float momentum,
                                  float* running_mean, float* running_var, const float* gamma, const float* beta,
                                  void* ws_zeroed, float* mean, float* invstd, float* scale, float* shift,
                                  int64_t* num_batches_tracked, void* stream);
int mspl_bn_train_prelu_bwd(const float* z, const float* residual, const float* gy, const float* scale, const float* shift,
                            const float* alpha, const float* gamma, const float* mean, const float* invstd, int32_t N, int32_t C,
                            int32_t HW, float* gres, float* gc, void* ws_zeroed, int32_t accumulate, float* ggamma, float* gbeta,
                            float* galpha, float* p, float* q, void* stream);
/* Small planes (mspl_bn_train_small_fits: N * HW <= 40 960 values per channel -- levels 3-5 of the path): the whole node per channel in
 * one workgroup and ONE launch each way, no workspace.  _fwd: y = PReLU(BatchNorm_train(z) + residual) + the statistics outputs of
 * mspl_bn_batch_stats_fused_fwd (residual / alpha / running_* / num_batches_tracked may be NULL).  _bwd: gz (dL/dz through both the
 * affine map and the statistics), gres (NULL without a residual), d gamma / d beta (accumulate != 0: added to), galpha (ACCUMULATED). */
int mspl_bn_train_small_fits(int32_t N, int32_t C, int32_t HW);
int mspl_bn_train_small_fwd(const float* z, const float* residual, const float* gamma, const float* beta, const float* alpha,
                            int32_t N, int32_t C, int32_t HW, float eps, float momentum, float* running_mean, float* running_var,
                            int64_t* num_batches_tracked, float* mean, float* invstd, float* scale, float* shift, float* y, void* stream);
int mspl_bn_train_small_bwd(const float* z, const float* residual, const float* gy, const float* scale, const float* shift,
                            const float* alpha, const float* gamma, const float* mean, const float* invstd, int32_t N, int32_t C,
                            int32_t HW, int32_t accumulate, float* gz, float* gres, float* ggamma, float* gbeta, float* galpha,
                            void* stream);
/* The residual-free node's second pass with the direct gradient recomputed instead of read back (mspl_bn_train_prelu_bwd then takes
 * gc = NULL and only reads): gz = p * z + q + (z * scale + shift > 0 ? gy : alpha * gy) * scale; alpha may be NULL (no PReLU). */
int mspl_bn_train_prelu_bwd_apply(const float* z, const float* gy, const float* scale, const float* shift, const float* alpha,
                                  const float* p, const float* q, int32_t N, int32_t C, int32_t HW, float* gz, void* stream);
/* Backward of the statistics' dependence on z, per channel: t = gscale - mean * gshift; ggamma = t * invstd; gbeta = gshift (NULL:
 * not wanted); p = -gamma * t * invstd^3 / M; q = -gshift * scale / M - p * mean  (then gz = p * z + q, one mspl_pointwise_fwd).
 * accumulate != 0: ggamma / gbeta are added to (the parameters' gradient buffers). */
int mspl_bn_batch_stats_bwd_coeffs(const float* gscale, const float* gshift, const float* gamma, const float* mean,
                                   const float* invstd, const float* scale, int32_t C, double M, int32_t accumulate,
                                   float* ggamma, float* gbeta, float* p, float* q, void* stream);
/* torch.optim.SGD (train_segmentation.py:253) on a flat fp32 buffer: g += wd*p; buf = first_step ? g : momentum*buf + g;
 * p -= lr*buf  (dampening 0, no Nesterov; buf may be NULL when momentum == 0).  One call per learning-rate group. */
int mspl_sgd_step(float* p, const float* g, float* buf, int64_t n, float lr, float momentum, float weight_decay,
                  int32_t first_step, void* stream);

/* NIDLoss (loss_fns/segmentation_loss.py:54-144), the two heavy steps: soft-arg-max of the label logits (:124-141) and the
 * soft joint histogram of get_probabilities (:76-101).  camera (B,3,H,W) network input, label (B,C,H,W) logits, K image bins.
 * out: K*Cl + K + Cl floats = joint (K,Cl) | p_c (K) | p_l (Cl), Cl = min(C,K) (the reference fills label bins inside its
 * loop over image bins), all divided by norm = B*H*W.  ws: mspl_nid_workspace_floats(C,H,W,K) floats.
 * Backward w.r.t. the label logits: gjoint (K,Cl) and gpl (Cl) = dL/d joint, dL/d p_l ALREADY divided by norm; glabel
 * (B,C,H,W) is overwritten.  K, C <= 32. */
int64_t mspl_nid_workspace_floats(int32_t C, int32_t H, int32_t W, int32_t K);
int mspl_nid_hist_fwd(const float* camera, const float* label, int32_t B, int32_t C, int32_t H, int32_t W, int32_t K,
                      float bw_camera, float bw_label, float* ws, float* out, void* stream);
int mspl_nid_hist_bwd(const float* camera, const float* label, int32_t B, int32_t C, int32_t H, int32_t W, int32_t K,
                      float bw_camera, float bw_label, const float* gjoint, const float* gpl, float* glabel, void* stream);

/* Backward of mspl_eesp_dw_hff_fwd (K2) in two launches.  gs: the suffix-summed output gradient from mspl_hff_suffix_sum,
 * branch-major (4,N,n,Ho,Wo); x: the forward input (N,n,H,W); w4: (4,n,3,3); dil: 4 dilations; stride 1|2.
 * gx (N,n,H,W): overwritten, NULL = skip.  gw: 4 device pointers to (n,3,3) buffers that are ACCUMULATED into (zeroed by
 * the caller, or the parameters' own gradient buffers), NULL = skip. */
int mspl_eesp_dw_bwd(const float* gs, const float* x, const float* w4, const int32_t* dil, int32_t stride, int32_t N,
                     int32_t n, int32_t H, int32_t W, float* gx, float* const* gw, void* stream);

/* Stride-1 EESP block, everything between conv_1x1_exp's data gradient and proj_1x1's BatchNorm in ONE launch (nn_layers/eesp.py:68-80
 * backward): br_after_cat's BatchNorm + PReLU backward, the HFF suffix sum, the data gradient and the weight gradients of the four
 * dilated depthwise branches.  z (N,4n,H,W) = K2's raw concatenation, gy = dL/d(PReLU(BN(z))), x (N,n,H,W) = K2's input, w4 (4,n,3,3),
 * dil[4] in 1..4; scale/shift/alpha/bn_mean/bn_inv (4n) as mspl_hff_bn_prelu_suffix_bwd.  gx (N,n,H,W) overwritten; gw[4] (n,3,3),
 * gscale/gshift/galpha (4n) ACCUMULATED.  With proj_c (N,n,H,W) = proj_1x1's bare convolution result (x = PReLU(proj_c * proj_scale +
 * proj_shift)), proj_1x1's BatchNorm + PReLU backward is applied before the store: gx is then dL/d(proj_c) and g_proj_scale /
 * g_proj_shift / g_proj_alpha (n, ACCUMULATED; d gamma / d beta with proj_mean / proj_inv) its parameter gradients; proj_c NULL:
 * gx = dL/dx.  _fits: 1 when the haloed row band fits LDS. */
int mspl_eesp_bwd_fused_fits(int32_t N, int32_t n, int32_t H, int32_t W, const int32_t* dil);
int mspl_eesp_bwd_fused(const float* z, const float* gy, const float* x, const float* w4, const int32_t* dil,
                        const float* scale, const float* shift, const float* alpha, const float* bn_mean,
                        const float* bn_inv, int32_t N, int32_t n, int32_t H, int32_t W, float* gx, float* const* gw,
                        float* gscale, float* gshift, float* galpha, const float* proj_c, const float* proj_scale,
                        const float* proj_shift, const float* proj_alpha, const float* proj_mean, const float* proj_inv,
                        float* g_proj_scale, float* g_proj_shift, float* g_proj_alpha, void* stream);

/* mspl_hff_bn_prelu_suffix_bwd for br_after_cat in train(): out_k = sum_{m >= k} (p z + q + direct gradient)_m, branch-major (4,N,n,HW);
 * (scale, shift) = the batch fold, stat_p / stat_q from mspl_bn_train_prelu_bwd (which also produced the parameter gradients).  For the
 * strided blocks and for shapes mspl_eesp_bwd_fused_bnstat does not cover; mspl_eesp_dw_bwd follows. */
int mspl_hff_bn_stat_suffix_bwd(const float* z, const float* gy, const float* scale, const float* shift, const float* alpha,
                                const float* stat_p, const float* stat_q, int32_t N, int32_t n, int32_t HW, float* out, void* stream);
/* The same launch for br_after_cat in train() (batch statistics, the supervised loop): (scale, shift) = the batch fold, stat_p / stat_q
 * (4n each) = the statistics-path coefficients of mspl_bn_train_prelu_bwd (which has already produced d gamma / d beta / d alpha): the
 * suffix sum runs over gz = p * z + q + direct gradient; gx = dL/dx of K2's input, gw accumulated as above. */
int mspl_eesp_bwd_fused_bnstat(const float* z, const float* gy, const float* x, const float* w4, const int32_t* dil,
                               const float* scale, const float* shift, const float* alpha, const float* stat_p, const float* stat_q,
                               int32_t N, int32_t n, int32_t H, int32_t W, float* gx, float* const* gw, void* stream);

/* out = srcs[0] + ... + srcs[n-1] (1 <= n <= 8 equally shaped fp32 tensors of `count` elements, count % 4 == 0, 16-byte aligned; srcs
 * is a HOST array of device pointers): the gradient of a tensor with several consumers in one launch (autograd.FanOutFn). */
int mspl_sum_n(const float* const* srcs, int32_t n, int64_t count, float* out, void* stream);
/* The same sum + a constant per (N*C) plane: out[plane, :] = sum_k srcs[k][plane, :] + mul * plane_const[plane] -- the gradient of a
 *     global average pool (EfficientPWConv's gate, nn_layers/efficient_pt.py:25-29) joins the sum as N*C values instead of a
 *     broadcast full-size tensor.  srcs: n tensors of planes * HW floats. */
int mspl_sum_n_planes(const float* const* srcs, int32_t n, const float* plane_const, float mul, int32_t planes, int32_t HW, float* out,
                      void* stream);

/* Transposed copies of many convolution weights in one launch (the weights of the data-gradient convolutions of a training
 * step; replaces one ATen permute copy + flip per convolution, autograd.ConvFn.backward).  seg_table: device array of
 * { const float* src; float* dst; int32 groups, cin_g, cout_g, k, numel, pad } (40 bytes each); src is (G*cout_g, cin_g, k, k),
 * dst is (G*cin_g, cout_g, k, k) with both spatial axes reversed.  block_table: device array of int32 pairs {segment, first
 * element} -- one workgroup of 256 destination elements each. */
int mspl_transpose_weights(const void* seg_table, const void* block_table, int32_t nblocks, void* stream);

/* torch.optim.Adam step on a flat fp32 buffer (L2 weight decay folded into the gradient; bias correction by `step`). */
int mspl_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int32_t step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MSPL_HIP_H */
