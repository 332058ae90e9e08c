"""ctypes binding of libape_hip.so (the C-ABI declared in include/ape_hip.h).

The product path has NO fallback: if the shared library is missing or a tensor is not on a HIP
device, the ops raise.  (CPU unit tests swap `ape_amd.ops` functions for torch emulations that live
under tests/ -- never the other way round.)
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libape_hip.so")

DT_F32 = 0
DT_BF16 = 1
DT_F16 = 2
ACT_NONE, ACT_RELU, ACT_GELU, ACT_SWIGLU, ACT_SILU = 0, 1, 2, 3, 4
MASK_NONE, MASK_ZERO_INPUT, MASK_ZERO_OUTPUT = 0, 1, 2


class GemmArgs(Structure):
    _fields_ = [
        ("A", c_void_p), ("W", c_void_p), ("C", c_void_p), ("bias", c_void_p), ("residual", c_void_p),
        ("rowmask", c_void_p), ("rope_cos", c_void_p), ("rope_sin", c_void_p),
        ("M", c_int32), ("N", c_int32), ("K", c_int32),
        ("lda", c_int32), ("ldw", c_int32), ("ldc", c_int32), ("ldr", c_int32),
        ("in_dt", c_int32), ("out_dt", c_int32), ("res_dt", c_int32),
        ("act", c_int32), ("mask_mode", c_int32), ("trans_out", c_int32),
        ("rope_rows", c_int32), ("rope_hd", c_int32), ("rope_cols", c_int32), ("vec_ok", c_int32),
        ("alpha", c_float), ("clamp", c_float),
        ("splitk", c_int32), ("tile64", c_int32), ("workspace", c_void_p),
        ("rowscale", c_void_p), ("rowshift", c_void_p), ("colvec", c_void_p), ("rope_cs", c_void_p),
        ("ln_w", c_void_p), ("ln_b", c_void_p), ("ln_eps", c_float), ("reserved0", c_int32),
        ("conv_perm", c_void_p), ("conv_zero", c_void_p), ("conv_h", c_int32), ("conv_w", c_int32),
        ("rowstat_cols", c_int32), ("rowstat_eps", c_float),
    ]


class LayerNormArgs(Structure):
    _fields_ = [
        ("x", c_void_p), ("w", c_void_p), ("b", c_void_p), ("y", c_void_p), ("add", c_void_p), ("y2", c_void_p),
        ("M", c_int32), ("C", c_int32), ("Cpad", c_int32),
        ("ldx", c_int32), ("ldy", c_int32), ("ldadd", c_int32), ("ldy2", c_int32),
        ("x_dt", c_int32), ("y_dt", c_int32), ("add_dt", c_int32), ("act", c_int32), ("eps", c_float),
    ]


class GroupNormArgs(Structure):
    _fields_ = [
        ("x", c_void_p), ("w", c_void_p), ("b", c_void_p), ("y", c_void_p), ("add", c_void_p), ("workspace", c_void_p),
        ("HW", c_int32), ("C", c_int32), ("G", c_int32),
        ("ldx", c_int32), ("ldy", c_int32), ("ldadd", c_int32),
        ("x_dt", c_int32), ("y_dt", c_int32), ("add_dt", c_int32), ("act", c_int32), ("eps", c_float),
    ]


# name -> (restype, argtypes); every symbol declared in include/ape_hip.h must appear here
SIGNATURES = {
    "ape_hip_last_error": (c_char_p, []),
    "ape_hip_abi_version": (c_int, []),
    "ape_hip_sizeof_args": (c_int, [c_int]),
    "ape_hip_gemm": (c_int, [POINTER(GemmArgs), c_void_p]),
    "ape_hip_gemm_last_kernel": (c_char_p, []),
    "ape_hip_meter_begin": (c_int, []),
    "ape_hip_meter_count": (c_int, []),
    "ape_hip_meter_end": (c_int, []),
    "ape_hip_meter_read": (c_int, [c_int, POINTER(c_char_p), POINTER(c_float)]),
    "ape_hip_zero": (c_int, [c_void_p, ctypes.c_size_t, c_void_p]),
    "ape_hip_stuff_collapse": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "ape_hip_sem_class_weights": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_int, c_int, c_void_p]),
    "ape_hip_pan_class_scores": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_float, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p]),
    "ape_hip_argmax_labels": (c_int, [c_void_p, ctypes.c_size_t, c_int, ctypes.c_size_t, c_float, c_void_p, c_void_p]),
    "ape_hip_sdma_usable": (c_int, [c_void_p, c_void_p]),
    "ape_hip_sdma_d2h": (c_int, [c_void_p, c_void_p, ctypes.c_size_t]),
    "ape_hip_sdma_d2h_multi": (c_int, [c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), ctypes.POINTER(ctypes.c_size_t), c_int]),
    "ape_hip_sdma_engines": (c_int, [c_void_p, c_void_p]),
    "ape_hip_row_stats": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "ape_hip_gemv": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                             c_float, c_void_p]),
    "ape_hip_gemv_affine": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                    c_float, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "ape_hip_layernorm": (c_int, [POINTER(LayerNormArgs), c_void_p]),
    "ape_hip_postnorm_residual": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_int, c_void_p, c_int, c_int,
                                          c_int, c_int, c_void_p]),
    "ape_hip_groupnorm_workspace_floats": (c_int, [c_int, c_int]),
    "ape_hip_groupnorm": (c_int, [POINTER(GroupNormArgs), c_void_p]),
    "ape_hip_ms_deform_attn_forward": (c_int, [c_void_p, c_int, POINTER(c_int64), POINTER(c_int64), c_void_p, c_void_p,
                                               c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ape_hip_msda_fused": (c_int, [c_void_p, c_int, c_int, POINTER(c_int64), POINTER(c_int64), c_void_p, c_int, c_void_p,
                                   c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ape_hip_msda_fused_h": (c_int, [c_void_p, c_int, c_int, POINTER(c_int64), POINTER(c_int64), c_void_p, c_int, c_void_p,
                                     c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ape_hip_attention": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                  c_int, c_float, c_int, c_void_p]),
    "ape_hip_geometry": (c_int, [c_int, c_int, c_int, c_int, POINTER(c_int), c_void_p, c_int, c_void_p, c_float, c_float, c_float,
                                 c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p]),
    "ape_hip_head_gemv": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int,
                                  c_int, c_void_p]),
    "ape_hip_attention_strided": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                          c_int, c_int, c_float, c_int, c_void_p]),
    "ape_hip_attention_causal": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                         c_int, c_int, c_float, c_int, c_void_p]),
    "ape_hip_embed_tokens": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                     c_int, c_int, c_void_p]),
    "ape_hip_patchify": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, POINTER(c_float), POINTER(c_float), c_void_p,
                                 c_int, c_int, c_void_p]),
    "ape_hip_im2col3x3": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "ape_hip_maxpool2x2": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "ape_hip_gather_rows": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "ape_hip_gather_rows_i64": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "ape_hip_nms_mask_words": (c_int, [c_int]),
    "ape_hip_nms_mask": (c_int, [c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p]),
    "ape_hip_nms_scan_segments": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "ape_hip_nms_scan_classes": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "ape_hip_vl_pool_workspace_floats": (c_int, [c_int, c_int]),
    "ape_hip_vl_pool": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ape_hip_segment_softmax": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ape_hip_colstats_workspace_floats": (c_int, [c_int, c_int]),
    "ape_hip_colstats": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ape_hip_transpose": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                  c_void_p]),
    "ape_hip_mask_upsample_bits": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ape_hip_roi_align_bits": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "ape_hip_paste_bits": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ape_hip_mask_upsample_sigmoid": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int,
                                              c_void_p]),
    "ape_hip_box_refine": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "ape_hip_det_records": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ape_hip_query_init": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_float, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                   c_void_p]),
    "ape_hip_query_finish": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                     c_float, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ape_hip_bilinear_resize": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "ape_hip_panoptic_pixels": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_int, c_int, c_void_p, c_void_p,
                                        c_void_p, c_void_p]),
    "ape_hip_panoptic_decide": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_double, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ape_hip_panoptic_write": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "ape_hip_enc_finalize": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ape_hip_topk_workspace_words": (c_int, [c_int]),
    "ape_hip_proposal_topk": (c_int, [c_void_p, c_int, POINTER(c_int), POINTER(c_int), c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                      c_void_p]),
    "ape_hip_proposal_order": (c_int, [c_void_p, c_int, c_void_p, c_void_p, POINTER(c_int), POINTER(c_int), c_int, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ape_hip_proposal_quota": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, POINTER(c_int), POINTER(c_int),
                                       c_int, c_int, c_void_p, c_void_p]),
    "ape_hip_det_sort": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p]),
    "ape_hip_det_topk": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p]),
    "ape_hip_ffn_fused": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                  c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p]),
    "ape_hip_attention_ext": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                      c_int, c_float, c_int, c_void_p]),
    "ape_hip_relpos_extend": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                                      c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "ape_hip_resize_coeffs": (c_int, [c_int, c_int, c_void_p, c_void_p, c_int]),
    "ape_hip_resize_tile_rows": (c_int, [c_void_p, c_int, POINTER(c_int)]),
    "ape_hip_resize_bilinear_u8": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                           c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ape_hip_rle_workspace_words": (c_int, [c_int, c_int, c_int, c_int]),
    "ape_hip_rle_encode": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "ape_hip_rle_to_string": (c_int, [c_void_p, c_int, c_void_p, c_int]),
}

_lib = None


def load():
    """Load libape_hip.so and attach the C signatures.  Raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built (run `python -m ape_amd.build` or "
            "__graft_entry__.build()).  ape_amd has no CPU/PyTorch fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().ape_hip_last_error()
        raise RuntimeError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")
