import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(_HERE)
# VD_HIP_LIB: development override (kernel A/B experiments with variant builds of the same ABI)
_LIB_PATH = os.environ.get("VD_HIP_LIB") or os.path.join(_PKG, "libvd_hip.so")
_lib = None


class VdHipError(RuntimeError):
    pass


class VdGemmDesc(ctypes.Structure):
    _fields_ = [
        ("a0", ctypes.c_void_p), ("a1", ctypes.c_void_p), ("w", ctypes.c_void_p), ("bias", ctypes.c_void_p),
        ("rowvec", ctypes.c_void_p), ("res", ctypes.c_void_p), ("out", ctypes.c_void_p), ("ws", ctypes.c_void_p),
        ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32),
        ("c0", ctypes.c_int32), ("c1", ctypes.c_int32), ("lda0", ctypes.c_int32), ("lda1", ctypes.c_int32),
        ("ldw", ctypes.c_int32), ("ldc", ctypes.c_int32), ("ldr", ctypes.c_int32),
        ("Hin", ctypes.c_int32), ("Win", ctypes.c_int32), ("Hout", ctypes.c_int32), ("Wout", ctypes.c_int32),
        ("ksize", ctypes.c_int32), ("stride", ctypes.c_int32), ("pad", ctypes.c_int32), ("ups", ctypes.c_int32),
        ("rows_per_batch", ctypes.c_int32), ("flags", ctypes.c_int32), ("act", ctypes.c_int32),
        ("alpha", ctypes.c_float), ("batch", ctypes.c_int32), ("split_k", ctypes.c_int32),
        ("stride_a", ctypes.c_int64), ("stride_w", ctypes.c_int64), ("stride_out", ctypes.c_int64),
        ("stride_res", ctypes.c_int64),
        ("colsum", ctypes.c_void_p), ("ln_eps", ctypes.c_float), ("reserved", ctypes.c_int32),
        ("sync", ctypes.c_void_p), ("ln_stats", ctypes.c_void_p),
        ("out_stats", ctypes.c_void_p), ("stat_img_rows", ctypes.c_int32), ("gn_groups", ctypes.c_int32),
        ("gn_gamma", ctypes.c_void_p), ("gn_beta", ctypes.c_void_p), ("gn_eps", ctypes.c_float), ("reserved3", ctypes.c_int32),
        ("skip_a0", ctypes.c_void_p), ("skip_a1", ctypes.c_void_p), ("skip_w", ctypes.c_void_p),
        ("skip_c0", ctypes.c_int32), ("skip_c1", ctypes.c_int32), ("skip_lda0", ctypes.c_int32), ("skip_lda1", ctypes.c_int32),
        ("skip_ldw", ctypes.c_int32), ("reserved4", ctypes.c_int32),
        ("row_sums", ctypes.c_void_p),
        ("stat_sums", ctypes.c_void_p),
    ]


class VdFfChain(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("a", ctypes.c_void_p), ("wo", ctypes.c_void_p), ("bo", ctypes.c_void_p),
                ("x1_scratch", ctypes.c_void_p), ("w1_packed", ctypes.c_void_p), ("b1_packed", ctypes.c_void_p),
                ("w2", ctypes.c_void_p), ("b2", ctypes.c_void_p), ("wp", ctypes.c_void_p), ("bp", ctypes.c_void_p),
                ("res", ctypes.c_void_p), ("out", ctypes.c_void_p), ("out_stats", ctypes.c_void_p),
                ("stat_sums", ctypes.c_void_p), ("stat_img_rows", ctypes.c_int64),
                ("M", ctypes.c_int64), ("C", ctypes.c_int32), ("ln_eps", ctypes.c_float), ("alpha", ctypes.c_float),
                ("reserved", ctypes.c_int32)]


# name -> (restype, argtypes); mirrors include/vd_hip.h one to one (checked by
# tests/test_host_cpu.py::test_capi_exports_every_declared_symbol)
_P, _I, _F, _L, _Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64, ctypes.c_size_t
PROTOTYPES = {
    "vd_gemm_f16": (_I, [ctypes.POINTER(VdGemmDesc), _P]),
    "vd_gemm_workspace_bytes": (_Z, [ctypes.POINTER(VdGemmDesc)]),
    "vd_gemm_plan": (_I, [ctypes.POINTER(VdGemmDesc), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "vd_gemm_stat_rows": (_I, [ctypes.POINTER(VdGemmDesc), ctypes.POINTER(ctypes.c_int)]),
    "vd_gemm_skip_ok": (_I, [ctypes.POINTER(VdGemmDesc)]),
    "vd_gemm_row_sums_ok": (_I, [ctypes.POINTER(VdGemmDesc)]),
    "vd_conv3x3_wstream_f16": (_I, [ctypes.POINTER(VdGemmDesc), _P, _P]),
    "vd_gemm_wstream_supported": (_I, [ctypes.POINTER(VdGemmDesc)]),
    "vd_conv3x3_wstream_plan": (_I, [ctypes.POINTER(VdGemmDesc), ctypes.POINTER(ctypes.c_int)]),
    "vd_gemm_wstream_plan": (_I, [ctypes.POINTER(VdGemmDesc), ctypes.POINTER(ctypes.c_int)]),
    "vd_gemm_wstream_f16": (_I, [ctypes.POINTER(VdGemmDesc), _P, _P]),
    "vd_conv3x3_wstream_supported": (_I, [ctypes.POINTER(VdGemmDesc)]),
    "vd_conv3x3_wstream_set_variant": (_I, [_I, _I]),
    "vd_gemm_config_name": (ctypes.c_char_p, [_I]),
    "vd_gemm_num_configs": (_I, []),
    "vd_gemm_set_override": (_I, [_I]),
    "vd_conv_halo_set_variant": (_I, [_I]),
    "vd_ff_geglu_f16": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _I, _F, _P]),
    "vd_ff_geglu_supported": (_I, [_I]),
    "vd_ff_chain_f16": (_I, [ctypes.POINTER(VdFfChain), _P]),
    "vd_ff_chain_supported": (_I, [_I]),
    "vd_gemm_row320_f16": (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _F, _P]),
    "vd_gemm_row320_supported": (_I, [_L, _I, _I]),
    "vd_gemm_row320_chain_f16": (_I, [_P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _L, _I, _F, _P]),
    "vd_groupnorm_affine_f16": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "vd_gemm_tune_set": (_I, [_I, _I, _I, _I, _I, _I, _I]),
    "vd_gemm_tune_clear": (_I, []),
    "vd_groupnorm_silu_f16": (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _I, _I, _I, _F, _I, _P]),
    "vd_groupnorm_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "vd_groupnorm_from_stats_f16": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _F, _I, _P]),
    "vd_chan_stats_f16": (_I, [_P, ctypes.c_long, _I, _I, _I, _P, _P]),
    "vd_gn_table_f32": (_I, [_P, _I, _I, _P, _I, _I, _I, _I, _P, _P, _I, _F, _P, _P]),
    "vd_gn_affine_from_stats_f16": (_I, [_P, _I, _I, _P, _I, _I, _I, _I, _P, _P, _I, _F, _P, _P, _P, _P]),
    "vd_gn_apply_table_f16": (_I, [_P, _I, _P, _I, _I, _I, _P, _I, _P, _P]),
    "vd_gn_apply_sums_f16": (_I, [_P, _I, _P, _P, _I, _P, _I, _I, _P, _P, _I, _F, _I, _P, _P]),
    "vd_groupnorm0d_silu_f16": (_I, [_P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _F, _I, _P]),
    "vd_layernorm_f16": (_I, [_P, _P, _P, _P, _I, _I, _F, _P]),
    "vd_row_stats_f16": (_I, [_P, _P, _L, _I, _L, _F, _P]),
    "vd_attention_f16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _L, _L, _L, _L, _F, _I, _P]),
    "vd_xattn_f16": (_I, [_P, _P, _P, _P, _F, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _L, _L, _F, _P]),
    "vd_xattn_supported": (_I, [_I, _I]),
    "vd_softmax_rows_f32_f16": (_I, [_P, _P, _L, _I, _P]),
    "vd_softmax_rows_f32_f32": (_I, [_P, _P, _L, _I, _F, _P]),
    "vd_timestep_embedding_f16": (_I, [_P, _P, _I, _I, _F, _P]),
    "vd_cfg_ddim_step_f16": (_I, [_P, _P, _P, _P, _P, _L, _I, _F, _F, _F, _F, _F, _P]),
    "vd_cfg_ddim_step_dev_f16": (_I, [_P, _P, _P, _P, _P, _L, _I, _P, _P]),
    "vd_q_sample_f16": (_I, [_P, _P, _P, _P, _P, _I, _L, _P]),
    "vd_nchw_to_nhwc_f16": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "vd_nhwc_to_nchw_f16": (_I, [_P, _P, _I, _I, _I, _I, _F, _F, _I, _P]),
    "vd_im2col_small_f16": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _L, _L, _L, _L, _I, _F, _F, _P]),
    "vd_diag_gaussian_sample_f16": (_I, [_P, _P, _P, _I, _I, _I, _F, _P]),
    "vd_axpby_f16": (_I, [_P, _P, _P, _F, _F, _L, _P]),
    "vd_unary_f16": (_I, [_P, _P, _I, _L, _P]),
    "vd_embed_tokens_f16": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "vd_clip_vision_embed_f16": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "vd_patchify_f16": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "vd_scale_by_row_norm_f16": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "vd_last_error": (ctypes.c_char_p, []),
    "vd_abi_version": (_I, []),
    "vd_probe_mfma_layout": (_I, [_P, _P, _P, _P]),
    "vd_probe_lds_tr16": (_I, [_P, _P, _P]),
    "vd_probe_xcc_ids": (_I, [_P, _I, _I, _P]),
    "vd_image_to_u8": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "vd_adjust_rank_f16": (_I, [_P, _P, _I, _I, _I, _I, _P, _F, _I, _P, _P]),
    "vd_adjust_rank_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "vd_mask_patch_weights": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "vd_color_adjust_f16": (_I, [_P, _P, _P, _I, _I, _I, _L, _P]),
    "vd_clip_preprocess_f16": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P]),
}


def lib_path():
    return _LIB_PATH


def lib_digest():
    """First 16 hex digits of the SHA-256 of the library file that is (or would be) loaded: stamps measurements that depend on
    the exact kernels (profiles/rNN_pmc_traffic.json) so bench.py can tell a stale file from a current one."""
    import hashlib
    h = hashlib.sha256()
    with open(_LIB_PATH, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()[:16]


def available():
    return os.path.exists(_LIB_PATH)


def lib():
    """Load (once) and return the ctypes handle; raises VdHipError when the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise VdHipError(
            "libvd_hip.so not found at %s -- build it with `python versatile-diffusion_amd/build.py` "
            "(there is no CPU fallback for the product path)" % _LIB_PATH)
    try:
        h = ctypes.CDLL(_LIB_PATH)
    except OSError as e:  # e.g. no ROCm runtime on this host
        raise VdHipError("cannot load %s: %s" % (_LIB_PATH, e))
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(h, name)
        except AttributeError:
            if os.environ.get("VD_HIP_LIB"):   # development A/B against an older build of the ABI: tolerate missing entry points
                continue
            raise VdHipError("%s does not export %s: rebuild it (python versatile-diffusion_amd/build.py --force)" % (_LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    _lib = h
    return h
