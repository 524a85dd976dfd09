"""ctypes binding of the C-ABI library (include/ldm_hip.h).

The product path has NO fallback: if ``libldm_hip.so`` is missing or cannot be
loaded, every op raises ``LdmHipUnavailable`` (importing the package itself stays
possible on a CPU-only box so that host logic and symbol checks can be tested).
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LDM_HIP_LIB") or os.path.join(HERE, "libldm_hip.so")     # LDM_HIP_LIB: A/B another build

LDM_MAX_SEG = 4
ACT_NONE, ACT_RELU, ACT_GATE, ACT_LRELU = 0, 1, 2, 3
A_ROWS, A_CONV3X3 = 0, 1
O_ROWS, O_CONVT2X2, O_UP2 = 0, 1, 2
SEG_N, SEG_K = 0, 1

c_fp = ctypes.c_void_p      # device pointers travel as plain addresses


class LdmHipUnavailable(RuntimeError):
    pass


class LdmHipError(RuntimeError):
    pass


class GemmDesc(ctypes.Structure):
    """struct ldm_gemm_desc (include/ldm_hip.h) -- field order must match."""
    _fields_ = [
        ("a", c_fp), ("lda", ctypes.c_longlong),
        ("M", ctypes.c_int), ("N", ctypes.c_int), ("K", ctypes.c_int),
        ("a_mode", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int), ("Cin", ctypes.c_int),
        ("nseg", ctypes.c_int), ("seg_mode", ctypes.c_int), ("seg_len", ctypes.c_int),
        ("w", c_fp * LDM_MAX_SEG), ("w2", c_fp * LDM_MAX_SEG),
        ("bias", c_fp * LDM_MAX_SEG), ("bias2", c_fp * LDM_MAX_SEG),
        ("ldw", ctypes.c_longlong),
        ("act", ctypes.c_int), ("slope", ctypes.c_float),
        ("addend", c_fp), ("ldadd", ctypes.c_longlong),
        ("out", c_fp), ("ldo", ctypes.c_longlong),
        ("o_mode", ctypes.c_int), ("OH", ctypes.c_int), ("OW", ctypes.c_int), ("Cout", ctypes.c_int),
        ("groups", ctypes.c_int),
        ("a_gstride", ctypes.c_longlong), ("w_gstride", ctypes.c_longlong), ("o_gstride", ctypes.c_longlong),
        ("b_gstride", ctypes.c_longlong), ("w_table", c_fp), ("bias_table", c_fp),
        ("workspace", c_fp), ("workspace_bytes", ctypes.c_longlong),
    ]


LDM_MAX_LEVELS = 8


class UNetBlockDesc(ctypes.Structure):
    """struct ldm_unet_block"""
    _fields_ = [("attention", ctypes.c_int), ("shift", ctypes.c_int),
                ("conv_w", c_fp), ("conv_b", c_fp),
                ("enc_w1", c_fp), ("enc_b1", c_fp), ("enc_w2", c_fp), ("enc_b2", c_fp),
                ("a_w", c_fp * 5), ("a_b", c_fp * 5), ("b_w", c_fp * 5), ("b_b", c_fp * 5), ("c_w", c_fp * 5), ("c_b", c_fp * 5),
                ("in_w", c_fp), ("in_b", c_fp), ("out_w", c_fp), ("out_b", c_fp)]


class UNetPlanDesc(ctypes.Structure):
    """struct ldm_unet_plan"""
    _fields_ = [("levels", ctypes.c_int), ("input_channels", ctypes.c_int), ("window", ctypes.c_int), ("nblocks", ctypes.c_int),
                ("eps", ctypes.c_float),
                ("channels", ctypes.c_int * LDM_MAX_LEVELS), ("enc_blocks", ctypes.c_int * LDM_MAX_LEVELS),
                ("dec_blocks", ctypes.c_int * LDM_MAX_LEVELS),
                ("stem_w", c_fp), ("stem_b", c_fp), ("head_w", c_fp), ("head_b", c_fp),
                ("down_w", c_fp * LDM_MAX_LEVELS), ("down_b", c_fp * LDM_MAX_LEVELS),
                ("up_w", c_fp * LDM_MAX_LEVELS), ("up_b", c_fp * LDM_MAX_LEVELS),
                ("pos_freq", c_fp * LDM_MAX_LEVELS), ("time_freq", c_fp * LDM_MAX_LEVELS),
                ("blocks", ctypes.POINTER(UNetBlockDesc))]


class UNetBlock16Desc(ctypes.Structure):
    """struct ldm_unet_block_bf16"""
    _fields_ = [("conv_w", c_fp), ("a_w", c_fp * 5), ("b_w", c_fp * 5), ("c_w", c_fp * 5), ("in_w", c_fp), ("out_w", c_fp)]


class UNetPlan16Desc(ctypes.Structure):
    """struct ldm_unet_plan_bf16"""
    _fields_ = [("nblocks", ctypes.c_int), ("blocks", ctypes.POINTER(UNetBlock16Desc))]


class CastJob(ctypes.Structure):
    """struct ldm_cast_job"""
    _fields_ = [("src", c_fp), ("dst", c_fp), ("dst_t", c_fp), ("rows", ctypes.c_longlong), ("cols", ctypes.c_int)]


_I, _L, _F, _P = ctypes.c_int, ctypes.c_longlong, ctypes.c_float, ctypes.c_void_p

# name -> (restype, argtypes); every symbol declared in include/ldm_hip.h
SIGNATURES = {
    "ldm_version": (_I, []),
    "ldm_last_error": (ctypes.c_char_p, []),
    "ldm_device_ok": (_I, []),
    "ldm_scratch_release": (_I, []),
    "ldm_gemm_f32": (_I, [ctypes.POINTER(GemmDesc), _P]),
    "ldm_gemm_f32_gate_fwd": (_I, [_P, _P, _P, _P]),
    "ldm_gemm_f32_gate_bwd": (_I, [_P, _P, _P, _P, _P]),
    "ldm_gemm_variant": (_I, [_I]),
    "ldm_gemm_wide_epilogue": (_I, [_I]),
    "ldm_gemm_ring": (_I, [_I]),
    "ldm_unet_workspace_bytes": (ctypes.c_size_t, [ctypes.POINTER(UNetPlanDesc), _I, _I, _I, _I]),
    "ldm_unet_forward_f32": (_I, [ctypes.POINTER(UNetPlanDesc), _P, _P, _I, _P, ctypes.POINTER(ctypes.c_int), _I, _I, _I, _P,
                                  ctypes.c_size_t, _P, _P]),
    "ldm_unet_forward_ex_f32": (_I, [ctypes.POINTER(UNetPlanDesc), _P, _P, _I, _P, ctypes.POINTER(ctypes.c_int), _I, _I, _I, _P,
                                     ctypes.c_size_t, _P, _I, _P]),
    "ldm_prof_enable": (_I, [_I]),
    "ldm_prof_read": (_I, [ctypes.POINTER(_L), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "ldm_prof_read_bytes": (_I, [_I, ctypes.POINTER(ctypes.c_double)]),
    "ldm_prof_dump": (_L, [ctypes.POINTER(ctypes.c_double), _L]),
    "ldm_prof_read_class": (_I, [_I, ctypes.POINTER(_L), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "ldm_channelnorm_film_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _P]),
    "ldm_film_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "ldm_sincos_embed_f32": (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "ldm_window_attention_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ldm_avgpool2_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ldm_stem_nchw_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ldm_head_nchw_f32": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ldm_ddim_update_f32": (_I, [_P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _P]),
    "ldm_qsample_f32": (_I, [_P, _P, _P, _P, _P, _I, _L, _P]),
    "ldm_rgb_head_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ldm_rgb_head_oc_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ldm_nchw_to_nhwc_f32": (_I, [_P, _P, _I, _I, _I, _P]),
    "ldm_nhwc_to_nchw_f32": (_I, [_P, _P, _I, _I, _I, _P]),
    "ldm_to_uint8_hwc": (_I, [_P, _P, _I, _I, _I, _P]),
    "ldm_gate_fwd_f32": (_I, [_P, _P, _P, _L, _P]),
    "ldm_gate_bwd_f32": (_I, [_P, _P, _P, _P, _P, _L, _P]),
    "ldm_relu_bwd_f32": (_I, [_P, _P, _P, _L, _P]),
    "ldm_add_f32": (_I, [_P, _P, _L, _P]),
    "ldm_colsum_f32": (_I, [_P, _P, _L, _I, _I, _P]),
    "ldm_transpose_colsum_f32": (_I, [_P, _P, _P, _L, _I, _P]),
    "ldm_reduce_partials_f32": (_I, [_P, _P, _I, _L, _P]),
    "ldm_reduce_partials_pair_f32": (_I, [_P, _P, _L, _P, _P, _L, _I, _L, _L, _P]),
    "ldm_gconv3x3_wgrad_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ldm_gemm_tn_f32": (_I, [_P, _L, _P, _L, _P, _P, _I, _I, _I, _I, _P]),
    "ldm_conv3x3_wgrad_f32": (_I, [_P, _L, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ldm_conv3x3_wgrad_npad": (_I, [_I]),
    "ldm_channelnorm_film_bwd_f32": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _P]),
    "ldm_avgpool2_bwd_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "ldm_sumpool2_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ldm_stem_bwd_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "ldm_head_bwd_f32": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ldm_l1_loss_f32": (_I, [_P, _P, _L, _P, _P]),
    "ldm_l1_loss_bwd_f32": (_I, [_P, _P, _P, _P, _L, _P]),
    "ldm_im2col3x3_t_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ldm_window_attention_bwd_f32": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ldm_window_attention_bwd_mfma": (_I, [_I]),
    # bf16 training step
    "ldm_gemm_bf16": (_I, [ctypes.POINTER(GemmDesc), _I, _P]),
    "ldm_window_attention_bwd_bf16": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ldm_window_attention_bwd_bf16_core": (_I, [_I]),
    "ldm_gemm_bf16_gate_fwd": (_I, [ctypes.POINTER(GemmDesc), _P, _P, _P]),
    "ldm_gemm_bf16_gate_bwd": (_I, [ctypes.POINTER(GemmDesc), _P, _P, _P, _P]),
    "ldm_gemm_tn_bf16": (_I, [_P, _L, _P, _L, _P, _P, _I, _I, _I, _I, _P]),
    "ldm_gemm_tn_ring": (_I, [_I]),
    "ldm_cast_bf16": (_I, [_P, _P, _L, _P]),
    "ldm_uncast_bf16": (_I, [_P, _P, _L, _P]),
    "ldm_transpose_cast_bf16": (_I, [_P, _P, _L, _I, _P]),
    "ldm_gate_fwd_bf16": (_I, [_P, _P, _P, _L, _P]),
    "ldm_gate_bwd_bf16": (_I, [_P, _P, _P, _P, _P, _L, _P]),
    "ldm_relu_bwd_bf16": (_I, [_P, _P, _P, _L, _P]),
    "ldm_channelnorm_film_bf16": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _F, _P]),
    "ldm_channelnorm_film_bwd_bf16": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P]),
    "ldm_channelnorm_film16_bf16": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _F, _P]),
    "ldm_channelnorm_film16_bwd_bf16": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _P]),
    "ldm_gconv3x3_bf16": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ldm_gconv3x3_bf16_tiled": (_I, [_I]),
    "ldm_gconv3x3_wgrad_bf16_splits": (_I, [_I, _I, _I, _I]),
    "ldm_gconv3x3_wgrad_bf16": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ldm_vq_quantize_f32": (_I, [_P, _P, _P, _L, _I, _I, _P]),
    "ldm_vq_embed_f32": (_I, [_P, _P, _P, _L, _I, _P]),
    "ldm_vq_loss_f32": (_I, [_P, _P, _L, _P, _P]),
    "ldm_vq_loss_bwd_f32": (_I, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _P]),
    # VAE Decoder backward
    "ldm_lrelu_bwd_f32": (_I, [_P, _P, _P, _L, _F, _P]),
    "ldm_im2col3x3_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ldm_space_to_depth2_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ldm_rgb_head_bwd_f32": (_I, [_P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ldm_rgb_head_bwd_oc_f32": (_I, [_P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ldm_gconv_pack_bf16": (_I, [_P, _P, _P, _I, _P]),
    "ldm_replicate_f32": (_I, [_P, _P, _I, _I, _P]),
    "ldm_pack3x3_f32": (_I, [_P, _P, _P, _I, _I, _P]),
    "ldm_multi_cast_table_bytes": (ctypes.c_size_t, [_I]),
    "ldm_multi_cast_bf16": (_I, [ctypes.POINTER(CastJob), _I, _P, _I, ctypes.POINTER(_L), _P]),
    "ldm_film_hidden": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "ldm_film_hidden_bwd_chunks": (_I, [_I, _I, _I]),
    "ldm_film_hidden_bwd": (_I, [_P, _P, _I, _P, _P, _I, _I, _I, _I, _P]),
    # bf16 sampling / decode
    "ldm_window_attention_bf16io": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ldm_window_attention_bf16_core": (_I, [_I]),
    "ldm_stem_nchw_bf16": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ldm_depth_to_space2_bf16": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ldm_rgb_head_bf16": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ldm_rgb_head_oc_bf16": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "ldm_up2_add_f32": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "ldm_avgpool2_bf16": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "ldm_unet_streams": (_I, [_I]),
    "ldm_unet_forward_bf16": (_I, [ctypes.POINTER(UNetPlanDesc), ctypes.POINTER(UNetPlan16Desc), _P, _P, _I, _P, ctypes.POINTER(ctypes.c_int), _I, _I, _I,
                                   _P, ctypes.c_size_t, _P, _I, _P]),
}

_lib = None


def load():
    """Load (once) and return the ctypes handle; raise loudly if impossible."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LdmHipUnavailable(
            "%s is missing: build it with `python -m ldm_image_generator_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    # torch bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Streams and device pointers
    # are shared with torch, so both must live in ONE HIP runtime: make sure torch's copy is the
    # one already mapped before our library's NEEDED entry is resolved.
    import torch  # noqa: F401
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as exc:          # e.g. libamdhip64 not present
        raise LdmHipUnavailable("cannot load %s: %s" % (LIB_PATH, exc))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError here == ABI mismatch, let it propagate
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    v = os.environ.get("LDM_GEMM_VARIANT")            # opt-in GEMM schedule for a whole process (see ldm_gemm_variant)
    if v is not None:
        lib.ldm_gemm_variant(int(v))
    v = os.environ.get("LDM_GEMM_RING")               # A/B: 0 = never the 256-row ring kernel, 2 = whenever legal (see ldm_gemm_ring)
    if v is not None:
        lib.ldm_gemm_ring(int(v))
    return lib


def check(code, what):
    if code != 0:
        msg = load().ldm_last_error()
        raise LdmHipError("%s failed (%d): %s" % (what, code, msg.decode() if msg else "?"))
