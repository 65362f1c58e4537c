"""ctypes binding of libgandanet_hip.so (C ABI: include/gandanet.h).

The product path has NO fallback: if the shared library is missing or a call
fails, this raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"``.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libgandanet_hip.so")

PREC_FP32, PREC_BF16, PREC_X3 = 0, 1, 2   # GD_PREC_*: exact f32 MFMA / bf16 operands / split-bf16 (hi + lo, three MFMAs per product)
PAM_BWD_K64_ATOMIC, PAM_BWD_K64_PARTS, PAM_BWD_K32_PARTS, PAM_BWD_TWO_KERNEL = 0, 1, 2, 3
ACT_NONE, ACT_RELU, ACT_LEAKY02, ACT_SIGMOID = 0, 1, 2, 3

c_fp = C.c_void_p  # device pointers travel as integers


class ConvDesc(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("M", C.c_int), ("Mstore", C.c_int), ("Ck", C.c_int), ("ks", C.c_int),
        ("stride", C.c_int), ("pad", C.c_int), ("transposed", C.c_int),
        ("Hi", C.c_int), ("Wi", C.c_int), ("Ho", C.c_int), ("Wo", C.c_int),
        ("a", c_fp), ("a_bs", C.c_long), ("a_sm", C.c_long), ("a_sc", C.c_long), ("a_st", C.c_long),
        ("x", c_fp), ("x_bs", C.c_long),
        ("in_scale", c_fp), ("in_shift", c_fp), ("in_relu", C.c_int),
        ("y", c_fp), ("y_bs", C.c_long),
        ("out_layout", C.c_int), ("out_bf16", C.c_int), ("ldo", C.c_int),
        ("alpha", c_fp), ("bias", c_fp), ("res", c_fp), ("res_bs", C.c_long),
        ("act", C.c_int), ("accumulate", C.c_int), ("precision", C.c_int),
        ("sub_oy", C.c_int), ("sub_ox", C.c_int), ("sub_step", C.c_int),
    ]


class GemmNTDesc(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("M", C.c_int), ("N", C.c_int), ("kseg", C.c_int), ("klen", C.c_long),
        ("a", c_fp), ("a_bs", C.c_long), ("a_ss", C.c_long), ("lda", C.c_long),
        ("bm", c_fp), ("b_bs", C.c_long), ("b_ss", C.c_long), ("ldb", C.c_long),
        ("im2col", C.c_int), ("ks", C.c_int), ("stride", C.c_int), ("pad", C.c_int),
        ("Hi", C.c_int), ("Wi", C.c_int), ("Ho", C.c_int), ("Wo", C.c_int),
        ("in_scale", c_fp), ("in_shift", c_fp), ("in_relu", C.c_int),
        ("c", c_fp), ("c_bs", C.c_long), ("ldc", C.c_long),
        ("alpha", c_fp), ("bias", c_fp),
        ("accumulate", C.c_int), ("splits", C.c_int), ("precision", C.c_int),
    ]


_i, _l, _f, _p, _sz = C.c_int, C.c_long, C.c_float, c_fp, C.c_size_t

# name -> (restype, argtypes); every symbol declared in include/gandanet.h
SIGNATURES = {
    "gd_version": (_i, []),
    "gd_set_deterministic": (None, [_i]),
    "gd_get_deterministic": (_i, []),
    "gd_last_error": (_i, [C.c_char_p, _i]),
    "gd_sizeof_conv_desc": (_i, []),
    "gd_sizeof_gemm_nt_desc": (_i, []),
    "gd_conv2d": (_i, [C.POINTER(ConvDesc), _p]),
    "gd_gemm_nt": (_i, [C.POINTER(GemmNTDesc), _p]),
    "gd_conv3x3_ws_bytes": (_sz, [_i, _i]),
    "gd_conv3x3_eligible": (_i, [C.POINTER(ConvDesc)]),
    "gd_conv3x3": (_i, [C.POINTER(ConvDesc), _p, _sz, _p]),
    "gd_conv3x3_wgrad": (_i, [_p, _l, _p, _p, _l, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    "gd_pack_16_split": (_i, [_p, _l, _i, _i, _i, _p, _p, _i, _p, _l, _i, _l, _i, _i, _p, _l, _i, _l, _i, _i, _p]),
    "gd_pack_16_split_masked": (_i, [_p, _l, _i, _i, _i, _p, _p, _i, _p, _l, _i, _l, _i, _i, _p, _l, _i, _l, _i, _i, _p, _l, _p]),
    "gd_split3_weights": (_i, [_p, _l, _l, _l, _p, _p]),
    "gd_bn_stats_ws_floats": (_sz, [_i, _i, _l]),
    "gd_bn_stats": (_i, [_p, _l, _i, _i, _l, _f, _f, _p, _p, _p, _p, _p, _p]),
    "gd_bn_fold": (_i, [_p, _p, _p, _p, _i, _p, _p, _p]),
    "gd_bn_fold_eval": (_i, [_p, _p, _p, _p, _f, _i, _p, _p, _p, _p]),
    "gd_affine_act": (_i, [_p, _l, _p, _p, _i, _i, _l, _i, _p, _l, _p]),
    "gd_bn_act_bwd": (_i, [_p, _l, _p, _l, _p, _p, _p, _p, _p, _i, _i, _l, _i, _i, _p, _p, _p, _l, _i, _p, _p]),
    "gd_bn_stats_local": (_i, [_p, _l, _i, _i, _l, _p, _p, _p]),
    "gd_bn_stats_merge": (_i, [_p, _i, _i, _f, _f, _p, _p, _p, _p, _p]),
    "gd_bn_act_bwd_dx": (_i, [_p, _l, _p, _l, _p, _p, _p, _p, _p, _p, _f, _i, _i, _l, _i, _p, _l, _i, _p]),
    "gd_bicubic_fwd": (_i, [_p, _i, _i, _i, _p, _i, _i, _f, _f, _p]),
    "gd_bicubic_bwd": (_i, [_p, _i, _i, _i, _p, _i, _i, _f, _f, _p]),
    "gd_bilinear_fwd": (_i, [_p, _i, _i, _i, _p, _i, _i, _i, _p, _p]),
    "gd_bilinear_bwd": (_i, [_p, _i, _i, _i, _p, _i, _i, _p]),
    "gd_maxpool2_fwd": (_i, [_p, _i, _i, _i, _p, _p]),
    "gd_maxpool2_bwd": (_i, [_p, _p, _i, _i, _i, _p, _p]),
    "gd_act_fwd": (_i, [_p, _p, _l, _i, _p]),
    "gd_act_bwd": (_i, [_p, _p, _p, _l, _i, _p]),
    "gd_axpby": (_i, [_p, _f, _p, _f, _l, _p]),
    "gd_scale_dev": (_i, [_p, _p, _p, _l, _i, _p]),
    "gd_copy_rows": (_i, [_p, _l, _l, _p, _l, _l, _i, _i, _i, _p]),
    "gd_channel_sum": (_i, [_p, _l, _i, _i, _l, _p, _i, _p, _p]),
    "gd_copy_slab": (_i, [_p, _l, _p, _l, _i, _l, _i, _p]),
    "gd_softmax_rows": (_i, [_p, _p, _l, _i, _f, _p]),
    "gd_softmax_rows_bwd": (_i, [_p, _p, _p, _l, _i, _f, _p]),
    "gd_dot": (_i, [_p, _p, _l, _p, _i, _p, _p]),
    "gd_transpose": (_i, [_p, _p, _i, _i, _i, _p]),
    "gd_add_transpose": (_i, [_p, _p, _i, _i, _p]),
    "gd_bce_logits": (_i, [_p, _l, _f, _p, _p, _p, _p]),
    "gd_bce_logits_target": (_i, [_p, _p, _l, _p, _p, _p, _p, _p]),
    "gd_leaky_fwd": (_i, [_p, _p, _l, _f, _p]),
    "gd_leaky_bwd": (_i, [_p, _p, _p, _l, _f, _p]),
    "gd_mse": (_i, [_p, _p, _l, _p, _p, _p, _p]),
    "gd_l1": (_i, [_p, _p, _l, _p, _p, _p, _p]),
    "gd_tv": (_i, [_p, _i, _i, _i, _i, _f, _p, _p, _p, _p]),
    "gd_ssim": (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _p]),
    "gd_ssim_samples": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _p, _p]),
    "gd_ssim_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p]),
    "gd_adamw": (_i, [_p, _p, _p, _p, _l, _i, _f, _f, _f, _f, _f, _f, _p]),
    "gd_pam_flash_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _l, _p, _l, _p, _p, _p, _p]),
    "gd_pam_key_sqnorm_max": (_i, [_p, _i, _i, _i, _i, _p, _p]),
    "gd_pam_fwd_shift_ws_bytes": (_sz, [_i, _i]),
    "gd_pam_flash_fwd_shift": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _l, _p, _l, _p, _p, _i, _p, _sz, _p]),
    "gd_pam_flash_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _l, _p, _sz, _p]),
    "gd_pam_bwd_scratch_bytes": (_sz, [_i, _i]),
    "gd_pam_k64_variant": (None, [_i, _i]),
    "gd_pam_k64_debug": (None, [_p]),
    "gd_chan_dot": (_i, [_p, _l, _p, _l, _i, _i, _i, _p, _p, _p, _p]),
    "gd_pam_f16_scale": (_i, [_p, _l, _i, _i, _i, _p, _p, _p, _p, _p]),
    "gd_conv3x3_nhwc_pack": (_i, [_p, _i, _i, _i, _p, _sz, _p]),
    "gd_conv3x3_nhwc": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "gd_conv3x3_nhwc_f32out": (_i, [_p, _p, _p, _p, _l, _i, _i, _i, _i, _i, _i, _p]),
    "gd_disc_stem_fwd": (_i, [_p, _i, _i, _i, _i, _p, _p, _i, _f, _p, _i, _p]),
    "gd_disc_stem_wgrad": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _p, _i, _p]),
    "gd_disc_stem_dgrad": (_i, [_p, _i, _i, _i, _i, _p, _i, _p, _i, _p]),
    "gd_conv3x3_nhwc_s2": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _i, _p]),
    "gd_conv3x3_nhwc_s2_dgrad": (_i, [_p, _p, _p, _f, _p, _i, _i, _i, _i, _i, _i, _p]),
    "gd_nhwc_flatten_fwd": (_i, [_p, _i, _i, _i, _p, _i, _p]),
    "gd_nhwc_flatten_bwd": (_i, [_p, _p, _f, _i, _i, _i, _p, _i, _p]),
    "gd_nhwc_to_nchw16": (_i, [_p, _i, _i, _i, _p, _p, _i, _p]),
    "gd_nhwc_stem_fwd": (_i, [_p, _i, _i, _i, _i, _p, _p, _i, _i, _p, _i, _p]),
    "gd_nhwc_stem_bwd": (_i, [_p, _i, _i, _i, _i, _p, _i, _p, _i, _p]),
    "gd_nhwc_maxpool2_fwd": (_i, [_p, _i, _i, _i, _i, _p, _i, _p]),
    "gd_nhwc_maxpool2_bwd": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _i, _p]),
    "gd_nhwc_l1": (_i, [_p, _p, _l, _p, _i, _p, _i, _p]),
    "gd_nhwc_l1_grad": (_i, [_p, _p, _l, _p, _i, _p, _i, _p]),
    "gd_shift_sum9_fwd": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "gd_shift_sum9_bwd": (_i, [_p, _p, _i, _i, _i, _p]),
    "gd_combine_inputs": (_i, [_p, _i, _i, _i, _f, _p, _i, _i, _i, _f, _p, _i, _i, _i, _p]),
    "gd_hist_match_ws_bytes": (_sz, [_l, _l]),
    "gd_hist_match": (_i, [_p, _p, _i, _l, _l, C.c_double, _p, _p, _sz, _p]),
    "gd_blend_region": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "gd_augment_d4": (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _f, _p]),
    "gd_bcast_mul": (_i, [_p, _p, _p, _i, _i, _l, _i, _p]),
    "gd_row_dot": (_i, [_p, _p, _p, _l, _l, _p]),
    "gd_chan_maxmean_fwd": (_i, [_p, _p, _p, _i, _i, _l, _p]),
    "gd_chan_maxmean_bwd": (_i, [_p, _p, _p, _i, _i, _l, _p]),
    "gd_comm_unique_id": (_i, [C.c_char_p]),
    "gd_comm_init": (_i, [_i, _i, C.c_char_p]),
    "gd_comm_world": (_i, []),
    "gd_allreduce": (_i, [_p, _sz, _i, _p]),
    "gd_reduce_scatter": (_i, [_p, _p, _sz, _i, _p]),
    "gd_allgather": (_i, [_p, _p, _sz, _i, _p]),
    "gd_comm_destroy": (_i, []),
    "gd_pack_bf16": (_i, [_p, _l, _i, _i, _i, _p, _f, _p, _i, _i, _p, _i, _i, _i, _i, _p]),
    "gd_pack_16": (_i, [_p, _l, _i, _i, _i, _p, _f, _p, _i, _i, _p, _i, _i, _i, _i, _i, _p]),
    "gd_pack_16_affine": (_i, [_p, _l, _i, _i, _i, _p, _p, _i, _p, _i, _i, _p, _i, _i, _i, _p]),
}

_lib = None


class GandanetError(RuntimeError):
    pass


def load() -> C.CDLL:
    """dlopen the library (once) and attach the prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GandanetError(
            f"{LIB_PATH} is missing: the HIP extension is not built (run __graft_entry__.build()). "
            "There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    if lib.gd_sizeof_conv_desc() != C.sizeof(ConvDesc) or lib.gd_sizeof_gemm_nt_desc() != C.sizeof(GemmNTDesc):
        raise GandanetError("descriptor struct layout differs between _lib.py and libgandanet_hip.so")
    _lib = lib
    return lib


def last_error() -> str:
    buf = C.create_string_buffer(512)
    load().gd_last_error(buf, 512)
    return buf.value.decode()


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise GandanetError(f"{what or 'gandanet call'} failed (rc={rc}): {last_error()}")
