"""ctypes binding of libmspl_hip.so (the C ABI declared in include/mspl_hip.h).

There is no CPU or eager-PyTorch fallback: if the library is missing the import fails loudly, and
every entry point raises RuntimeError with the library's own error text on a non-zero status --
mirroring the reference's failure mode (a RuntimeError from ATen on a shape mismatch).
"""
import ctypes
import os

# torch first, always: torch carries its own copy of the HIP runtime and this library binds /opt/rocm's.  The two coexist in
# one process only when torch's is loaded first (the order every test and bench.py use); with this library first, the first
# launch from it fails with hipErrorNoDevice (seen with `python __graft_entry__.py smoke`, where build() imports the package
# before anything imported torch).
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
# (the probes under tools/ point MSPL_HIP_LIB at a `make TUNING=1` / `STAMPS=1` build, libmspl_hip_tuning.so; the product library
# itself reads no environment variable)
LIB_PATH = os.environ.get('MSPL_HIP_LIB') or os.path.join(_HERE, 'lib', 'libmspl_hip.so')

c_f32p = ctypes.c_void_p
c_i32 = ctypes.c_int32
c_i64 = ctypes.c_int64


class Epilogue(ctypes.Structure):
    """mspl_epilogue_t."""
    _fields_ = [('scale', ctypes.c_void_p), ('shift', ctypes.c_void_p), ('alpha', ctypes.c_void_p),
                ('pre_add', ctypes.c_void_p), ('residual', ctypes.c_void_p), ('reinf_r', ctypes.c_void_p),
                ('reinf_w', ctypes.c_void_p), ('gate', ctypes.c_void_p),
                ('out_ctot', c_i32), ('out_coff', c_i32), ('raw_out', ctypes.c_void_p),
                ('struct_size', ctypes.c_uint32), ('flags', ctypes.c_uint32)]


ABI_VERSION = 3                 # include/mspl_hip.h: mspl_abi_version()
LAUNCH_THROUGHPUT = 1           # MSPL_LAUNCH_THROUGHPUT
LAUNCH_K2_STREAM_OFF = 2        # MSPL_LAUNCH_K2_STREAM_OFF
LAUNCH_K2_STREAM_FORCE = 4      # MSPL_LAUNCH_K2_STREAM_FORCE


_EP = ctypes.POINTER(Epilogue)

# name -> argument types (all return int status); the single source for the symbol-export test
SIGNATURES = {
    'mspl_eesp_dw_hff_fwd': [c_f32p, c_f32p, ctypes.POINTER(c_i32), c_i32, c_i32, c_i32, c_i32, c_i32, _EP,
                             c_f32p, ctypes.c_void_p],
    'mspl_eesp_proj_dw_hff_fits': [c_i32] * 6 + [ctypes.POINTER(c_i32), ctypes.c_uint32],
    'mspl_eesp_proj_dw_hff_fwd': [c_f32p] * 6 + [ctypes.POINTER(c_i32)] + [c_i32] * 6 + [_EP, c_f32p, ctypes.c_void_p],
    'mspl_eesp_dw_exp_fits': [c_i32] * 4 + [ctypes.POINTER(c_i32), ctypes.c_uint32],
    'mspl_eesp_dw_exp_pack_floats': [c_i32],
    'mspl_eesp_dw_exp_pack': [c_f32p] * 5 + [c_i32] * 3 + [ctypes.POINTER(c_i32), c_f32p, ctypes.c_void_p],
    'mspl_eesp_dw_exp_fwd': [c_f32p, c_f32p, ctypes.POINTER(c_i32)] + [c_i32] * 4 + [_EP, c_f32p, ctypes.c_void_p],
    'mspl_eesp_dw_exp_next_pack_floats': [c_i32],
    'mspl_eesp_dw_exp_next_pack': [c_f32p, c_i32, c_f32p, ctypes.c_void_p],
    'mspl_eesp_dw_exp_next_fwd': [c_f32p, c_f32p, ctypes.POINTER(c_i32)] + [c_i32] * 4 + [_EP, c_f32p] + [c_f32p] * 5 + [ctypes.c_void_p],
    'mspl_conv1x1_fwd': [c_f32p, c_f32p, c_i32, c_i32, c_i32, c_i32, c_i32, _EP, c_f32p, ctypes.c_void_p],
    'mspl_conv3x3_fwd': [c_f32p, c_f32p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, _EP, c_f32p,
                         ctypes.c_void_p],
    'mspl_avgpool3x3s2_fwd': [c_f32p, c_i32, c_i32, c_i32, c_i32, _EP, c_f32p, ctypes.c_void_p],
    'mspl_bilinear_fwd': [c_f32p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, _EP, c_f32p, ctypes.c_void_p],
    'mspl_adaptive_avgpool_fwd': [c_f32p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, _EP, c_f32p, ctypes.c_void_p],
    'mspl_avgpool3x3s2_psum_fwd': [c_f32p, c_i32, c_i32, c_i32, c_i32, _EP, c_f32p, c_f32p, ctypes.c_void_p],
    'mspl_avgpool3x3s2_psum_blocks': [c_i32, c_i32],
    'mspl_gate_from_sums_fwd': [c_f32p, c_f32p, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32p, ctypes.c_void_p],
    'mspl_pointwise_fwd': [c_f32p, c_i32, c_i32, c_i32, _EP, c_f32p, ctypes.c_void_p],
    'mspl_gap_gate_fwd': [c_f32p, c_f32p, c_i32, c_i32, c_i32, c_i32, c_f32p, c_f32p, ctypes.c_void_p],
    'mspl_fusion_gate_fwd': [c_f32p, c_f32p, c_f32p, c_i64, c_f32p, ctypes.c_void_p],
    'mspl_fusion_gate_bwd': [c_f32p, c_f32p, c_f32p, c_f32p, c_i64, c_f32p, c_f32p, c_f32p, ctypes.c_void_p],
    'mspl_pyrpool_fused_fwd': [c_f32p, c_i32, c_i32, c_i32, c_i32, c_i32, ctypes.POINTER(c_i32), ctypes.POINTER(c_i32),
                               ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), c_f32p, c_f32p, c_f32p,
                               c_f32p, _EP, c_f32p, ctypes.c_void_p],
    'mspl_pyrpool_fused_train_fwd': [c_f32p, c_i32, c_i32, c_i32, c_i32, c_i32, ctypes.POINTER(c_i32), ctypes.POINTER(c_i32),
                                     ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), c_f32p, c_f32p, c_f32p,
                                     c_f32p, _EP, c_f32p, c_f32p, ctypes.c_void_p],
    'mspl_pyrpool_fused_train_fits': [c_i32] * 5 + [ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)],
    'mspl_pyrpool_merge_bwd': [c_f32p] * 3 + [c_i32] * 5 + [c_f32p] * 19 + [ctypes.c_void_p],
    'mspl_pyrpool_branch_bwd_fits': [c_i32] * 5 + [ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)],
    'mspl_pyrpool_branch_bwd': [c_f32p] + [c_i32] * 5 + [ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)] +
                               [ctypes.POINTER(ctypes.c_void_p)] * 3 + [c_f32p, c_f32p, c_f32p, ctypes.c_void_p],
    'mspl_label_epilogue_fwd': [c_f32p, c_f32p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                ctypes.c_void_p, ctypes.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_void_p],
    'mspl_label_epilogue_hist_fwd': [c_f32p, c_f32p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                     ctypes.c_void_p, ctypes.c_void_p, c_f32p, ctypes.c_void_p, c_i32, ctypes.c_void_p, c_i64,
                                     ctypes.c_void_p],
    'mspl_label_epilogue_hist_workspace_bytes': [c_i32, c_i32, c_i32],
    'mspl_label_epilogue_hist_fits': [c_i32] * 8,
    'mspl_conv_bwd_data': [c_f32p, c_f32p] + [c_i32] * 10 + [c_f32p, ctypes.c_void_p],
    'mspl_conv_bwd_weight': [c_f32p, c_f32p] + [c_i32] * 10 + [c_f32p, ctypes.c_void_p],
    'mspl_affine_prelu_bwd': [c_f32p] * 7 + [c_i32] * 3 + [c_f32p] * 5 + [ctypes.c_void_p],
    'mspl_bn_train_small_fits': [c_i32] * 3,
    'mspl_bn_train_small_fwd': [c_f32p] * 5 + [c_i32] * 3 + [ctypes.c_float] * 2 + [c_f32p] * 2 + [ctypes.c_void_p] + [c_f32p] * 5 + [ctypes.c_void_p],
    'mspl_bn_train_small_bwd': [c_f32p] * 9 + [c_i32] * 4 + [c_f32p] * 5 + [ctypes.c_void_p],
    'mspl_bn_train_prelu_bwd_apply': [c_f32p] * 7 + [c_i32] * 3 + [c_f32p, ctypes.c_void_p],
    'mspl_pyrpool_merge_fwd': [c_f32p] + [c_i32] * 5 + [c_f32p] * 5 + [ctypes.c_void_p],
    'mspl_bn_stats_path_add': [c_f32p] * 4 + [c_i32] * 4 + [ctypes.c_void_p],
    'mspl_bn_fused_workspace_bytes': [c_i32],
    'mspl_bn_batch_stats_fused_fwd': [c_f32p, c_i32, c_i32, c_i32, ctypes.c_float, ctypes.c_float] + [c_f32p] * 4 + [ctypes.c_void_p] + [c_f32p] * 4
                                     + [ctypes.c_void_p, ctypes.c_void_p],
    'mspl_bn_train_prelu_bwd': [c_f32p] * 9 + [c_i32] * 3 + [c_f32p, c_f32p, ctypes.c_void_p, c_i32] + [c_f32p] * 5 + [ctypes.c_void_p],
    'mspl_down_tail_fwd': [c_f32p] * 4 + [c_i32] * 4 + [c_f32p, ctypes.c_void_p],
    'mspl_down_tail_bwd': [c_f32p] * 5 + [c_i32] * 4 + [c_f32p] * 4 + [ctypes.c_void_p],
    'mspl_bn_prelu_bwd': [c_f32p] * 9 + [c_i32] * 3 + [c_f32p] * 5 + [ctypes.c_void_p],
    'mspl_avgpool3x3s2_bwd': [c_f32p, c_i32, c_i32, c_i32, c_i32, c_f32p, ctypes.c_void_p],
    'mspl_bilinear_bwd': [c_f32p] + [c_i32] * 6 + [c_f32p, ctypes.c_void_p],
    'mspl_adaptive_avgpool_bwd': [c_f32p] + [c_i32] * 6 + [c_f32p, ctypes.c_void_p],
    'mspl_plane_dot': [c_f32p, c_f32p, c_i32, c_i32, c_f32p, ctypes.c_void_p],
    'mspl_plane_broadcast': [c_f32p, c_i32, c_i32, ctypes.c_float, c_i32, c_f32p, ctypes.c_void_p],
    'mspl_gap_gate_bwd': [c_f32p] * 4 + [c_i32] * 3 + [c_f32p, c_f32p, ctypes.c_void_p],
    'mspl_gap_gate_bwd_accum': [c_f32p] * 4 + [c_i32] * 3 + [c_f32p, c_f32p, ctypes.c_void_p],
    'mspl_hff_bn_prelu_suffix_bwd': [c_f32p] * 7 + [c_i32] * 3 + [c_f32p] * 4 + [ctypes.c_void_p],
    'mspl_hff_suffix_sum': [c_f32p, c_i32, c_i32, c_i32, c_f32p, ctypes.c_void_p],
    'mspl_uw_loss_fwd_bwd': [c_f32p, c_f32p, ctypes.c_void_p, c_f32p, c_i32, c_i32, c_i32, ctypes.c_float] + [c_f32p] * 4
                            + [ctypes.c_void_p],
    'mspl_uw_loss_scaled_fwd_bwd': [c_f32p, c_f32p, ctypes.c_void_p, c_f32p, c_i32, c_i32, c_i32, ctypes.c_float, ctypes.c_float] + [c_f32p] * 4
                                   + [ctypes.c_void_p],
    'mspl_uw_loss_heads_supported': [c_i32],
    'mspl_uw_loss_heads_fwd_bwd': [c_f32p, c_f32p, ctypes.c_void_p, c_f32p] + [c_i32] * 8 + [ctypes.c_float] * 2 + [c_f32p] * 3 + [ctypes.c_void_p],
    'mspl_pixelwise_kld_fwd': [c_f32p, c_f32p, c_i32, c_i32, c_i32, c_f32p, ctypes.c_void_p],
    'mspl_pixelwise_kld_bwd': [c_f32p, c_f32p, c_f32p, c_i32, c_i32, c_i32, c_f32p, c_f32p, ctypes.c_void_p],
    'mspl_weighted_ce_fwd': [c_f32p, ctypes.c_void_p, c_f32p, c_f32p, c_i32, c_i32, c_i32, c_i32, c_f32p, ctypes.c_void_p],
    'mspl_weighted_ce_bwd': [c_f32p, ctypes.c_void_p, c_f32p, c_f32p, c_i32, c_i32, c_i32, c_i32, c_f32p, c_f32p, c_f32p, c_f32p,
                             ctypes.c_void_p],
    'mspl_pyr_down_prep_lds_bytes': [c_i32, c_i32, c_i32, c_i32, c_i32, ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)],
    'mspl_pyr_down_prep_fwd': [c_f32p, c_i32, c_i32, c_i32, c_i32, c_i32, ctypes.POINTER(c_i32), ctypes.POINTER(c_i32),
                               ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p],
    'mspl_pyr_down_prep_train_fwd': [c_f32p, c_i32, c_i32, c_i32, c_i32, c_i32, ctypes.POINTER(c_i32), ctypes.POINTER(c_i32),
                                     ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                     ctypes.c_void_p],
    'mspl_conv1x1_wgrad_batch': [ctypes.POINTER(ctypes.c_void_p)] * 4 + [ctypes.POINTER(c_i32)] * 5 + [c_i32, ctypes.c_void_p],
    'mspl_pyr_down_mid_bwd': [ctypes.POINTER(ctypes.c_void_p)] * 3 + [c_i32] * 5 + [ctypes.POINTER(c_i32)] * 2 +
                             [ctypes.POINTER(ctypes.c_void_p)] * 2 + [ctypes.c_void_p],
    'mspl_dense_conv_fwd': [c_f32p, c_f32p, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, _EP, c_f32p,
                            ctypes.c_void_p],
    'mspl_eval_epilogue_fwd': [c_f32p, c_f32p, ctypes.c_void_p, c_f32p] + [c_i32] * 8 + [ctypes.c_float, c_i32, c_i32, ctypes.c_void_p,
                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p],
    'mspl_eval_batch_finalize': [ctypes.c_void_p, ctypes.c_void_p, c_i32, ctypes.c_void_p],
    'mspl_miou_areas_fwd': [c_f32p, ctypes.c_void_p, ctypes.c_void_p, c_i32, c_i32, c_i32, c_i32, ctypes.c_void_p, ctypes.c_void_p],
    'mspl_resample_ksize': [c_i32, c_i32],
    'mspl_resample_coeffs': [c_i32, c_i32, ctypes.c_void_p, ctypes.c_void_p],
    'mspl_nearest_index': [c_i32, c_i32, ctypes.c_void_p],
    'mspl_preprocess_u8_fwd': [ctypes.c_void_p] + [c_i32] * 6 + [ctypes.c_void_p, ctypes.c_void_p, c_i32, ctypes.c_void_p,
                               ctypes.c_void_p, c_i32] + [ctypes.c_void_p] * 6,
    'mspl_resize_label_fwd': [ctypes.c_void_p] + [c_i32] * 5 + [ctypes.c_void_p] * 5,
    'mspl_bn_batch_stats_fwd': [c_f32p, c_i32, c_i32, c_i32, ctypes.c_float, ctypes.c_float] + [ctypes.c_void_p] * 6,
    'mspl_sgd_step': [c_f32p, c_f32p, c_f32p, c_i64, ctypes.c_float, ctypes.c_float, ctypes.c_float, c_i32, ctypes.c_void_p],
    'mspl_bn_batch_stats_fold_fwd': [c_f32p, c_i32, c_i32, c_i32, ctypes.c_float, ctypes.c_float] + [c_f32p] * 4 + [ctypes.c_void_p] + [c_f32p] * 4 + [ctypes.c_void_p],
    'mspl_bn_batch_stats_bwd_coeffs': [c_f32p] * 6 + [c_i32, ctypes.c_double, c_i32] + [c_f32p] * 4 + [ctypes.c_void_p],
    'mspl_nid_workspace_floats': [c_i32, c_i32, c_i32, c_i32],
    'mspl_nid_hist_fwd': [c_f32p, c_f32p] + [c_i32] * 5 + [ctypes.c_float, ctypes.c_float, c_f32p, c_f32p, ctypes.c_void_p],
    'mspl_nid_hist_bwd': [c_f32p, c_f32p] + [c_i32] * 5 + [ctypes.c_float, ctypes.c_float, c_f32p, c_f32p, c_f32p, ctypes.c_void_p],
    'mspl_eesp_bwd_fused_fits': [c_i32] * 4 + [ctypes.POINTER(c_i32)],
    'mspl_hff_bn_stat_suffix_bwd': [c_f32p] * 7 + [c_i32] * 3 + [c_f32p, ctypes.c_void_p],
    'mspl_eesp_bwd_fused_bnstat': [c_f32p] * 4 + [ctypes.POINTER(c_i32)] + [c_f32p] * 5 + [c_i32] * 4 + [c_f32p, ctypes.POINTER(ctypes.c_void_p)] + [ctypes.c_void_p],
    'mspl_eesp_bwd_fused': [c_f32p] * 4 + [ctypes.POINTER(c_i32)] + [c_f32p] * 5 + [c_i32] * 4 + [c_f32p, ctypes.POINTER(ctypes.c_void_p)] +
                           [c_f32p] * 12 + [ctypes.c_void_p],
    'mspl_eesp_dw_bwd': [c_f32p, c_f32p, c_f32p, ctypes.POINTER(c_i32), c_i32, c_i32, c_i32, c_i32, c_i32, c_f32p,
                         ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p],
    'mspl_png_writer_create': [c_i32, c_i32],
    'mspl_png_writer_submit': [ctypes.c_void_p, ctypes.c_void_p, c_i32, c_i32, c_i32, ctypes.POINTER(ctypes.c_char_p), ctypes.c_void_p],
    'mspl_png_writer_poll': [ctypes.c_void_p, c_i64, c_i32],
    'mspl_png_writer_destroy': [ctypes.c_void_p],
    'mspl_abi_version': [],
    'mspl_sum_n': [ctypes.POINTER(ctypes.c_void_p), c_i32, c_i64, c_f32p, ctypes.c_void_p],
    'mspl_sum_n_planes': [ctypes.POINTER(ctypes.c_void_p), c_i32, c_f32p, ctypes.c_float, c_i32, c_i32, c_f32p, ctypes.c_void_p],
    'mspl_transpose_weights': [ctypes.c_void_p, ctypes.c_void_p, c_i32, ctypes.c_void_p],
    'mspl_adam_step': [c_f32p, c_f32p, c_f32p, c_f32p, c_i64] + [ctypes.c_float] * 5 + [c_i32, ctypes.c_void_p],
    'mspl_merge_labels_fwd': [ctypes.POINTER(ctypes.c_void_p), c_i32, c_i64, c_i32, c_i32, c_i32, ctypes.c_void_p,
                              ctypes.c_void_p, ctypes.c_void_p],
}


def _load():
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            'mspl_amd: %s not found. Build it with `python -c "import __graft_entry__ as g; g.build()"` or '
            '`make -C mspl_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    lib.mspl_pyr_down_prep_lds_bytes.restype = ctypes.c_int64      # a size query, not a status
    lib.mspl_nid_workspace_floats.restype = ctypes.c_int64
    lib.mspl_eesp_dw_exp_pack_floats.restype = ctypes.c_int64
    lib.mspl_bn_fused_workspace_bytes.restype = ctypes.c_int64
    lib.mspl_eesp_dw_exp_next_pack_floats.restype = ctypes.c_int64
    lib.mspl_label_epilogue_hist_workspace_bytes.restype = ctypes.c_int64
    lib.mspl_png_writer_create.restype = ctypes.c_void_p          # a handle
    lib.mspl_png_writer_submit.restype = ctypes.c_int64           # a ticket (or a negative status)
    lib.mspl_version.restype = ctypes.c_char_p
    if lib.mspl_abi_version() != ABI_VERSION:
        raise ImportError('mspl_amd: %s has ABI revision %d, this package binds revision %d: rebuild the library (make -C mspl_amd/csrc)'
                          % (LIB_PATH, lib.mspl_abi_version(), ABI_VERSION))
    lib.mspl_last_error.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    lib.mspl_last_error.restype = ctypes.c_size_t
    return lib


lib = _load()


def last_error():
    buf = ctypes.create_string_buffer(512)
    lib.mspl_last_error(buf, 512)
    return buf.value.decode()


def check(status):
    if status != 0:
        raise RuntimeError('mspl_hip (%d): %s' % (status, last_error()))


def version():
    return lib.mspl_version().decode()
