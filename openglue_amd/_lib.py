"""ctypes binding of libopenglue_amd.so (include/openglue_amd.h).

There is NO fallback: if the shared library is missing or a call fails the product path raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OPENGLUE_AMD_LIB") or os.path.join(HERE, "lib", "libopenglue_amd.so")   # override: A/B builds

OG_ABI_VERSION = 10
OG_FLAG_RESIDUAL, OG_FLAG_USE_OFFSET, OG_FLAG_NO_DESCRIPTORS, OG_FLAG_SIREN_ENCODER, OG_FLAG_LINEAR_ATTENTION = 1, 2, 4, 8, 16
OG_FLAG_FAVOR_RELU = 32
OG_MAX_HIDDEN = 8
OG_MAX_RAGGED = 64
OG_STAGES = ("encoder_input", "gemm_f32", "attention", "sinkhorn", "matches", "gemm_f16x3", "mlp_fused")

_ERRORS = {-1: "OG_E_INVALID (null pointer / bad size)", -2: "OG_E_SHAPE (unsupported shape)",
           -3: "OG_E_ALIGN (pointer or leading dimension not 16-byte aligned)", -4: "OG_E_FLAG (unknown flag)",
           -5: "OG_E_RANGE (a folded weight is not finite)"}

c_float_p = C.POINTER(C.c_float)


class og_shape(C.Structure):
    _fields_ = [("batch", C.c_int32), ("m", C.c_int32), ("n", C.c_int32), ("desc_dim", C.c_int32),
                ("num_heads", C.c_int32), ("num_stages", C.c_int32), ("side_info", C.c_int32),
                ("num_hidden", C.c_int32), ("hidden", C.c_int32 * OG_MAX_HIDDEN),
                ("sinkhorn_iters", C.c_int32), ("sinkhorn_reg", C.c_float), ("flags", C.c_int32),
                ("match_threshold", C.c_float)]


class og_conv(C.Structure):
    _fields_ = [("weight", C.c_void_p), ("bias", C.c_void_p)]


class og_bn(C.Structure):
    _fields_ = [("weight", C.c_void_p), ("bias", C.c_void_p), ("running_mean", C.c_void_p), ("running_var", C.c_void_p)]


class og_layer_params(C.Structure):
    _fields_ = [("in_proj_q", og_conv), ("in_proj_k", og_conv), ("in_proj_v", og_conv), ("out_proj", og_conv),
                ("fc0", og_conv), ("fc_bn", og_bn), ("fc3", og_conv), ("favor_projection", C.c_void_p)]


class og_params(C.Structure):
    _fields_ = [("enc_conv", og_conv * (OG_MAX_HIDDEN + 1)), ("enc_bn", og_bn * OG_MAX_HIDDEN),
                ("layers", C.POINTER(og_layer_params)), ("linear_proj", og_conv), ("mix_coefs", C.c_void_p),
                ("dustbin_score", C.c_float)]


class og_inputs(C.Structure):
    _fields_ = [("keypoints0", C.c_void_p), ("keypoints1", C.c_void_p), ("descriptors0", C.c_void_p),
                ("descriptors1", C.c_void_p), ("side_info0", C.c_void_p), ("side_info1", C.c_void_p),
                ("image0_wh", C.c_float * 2), ("image1_wh", C.c_float * 2)]


class og_outputs(C.Structure):
    _fields_ = [("scores", C.c_void_p), ("context_descriptors0", C.c_void_p), ("context_descriptors1", C.c_void_p),
                ("matches0", C.c_void_p), ("matching_scores0", C.c_void_p), ("matches1", C.c_void_p),
                ("matching_scores1", C.c_void_p)]


class og_packed_layout_t(C.Structure):
    _fields_ = [("n_enc", C.c_int32), ("enc_k", C.c_int32 * (OG_MAX_HIDDEN + 1)), ("enc_out", C.c_int32 * (OG_MAX_HIDDEN + 1)),
                ("enc_w", C.c_int64 * (OG_MAX_HIDDEN + 1)), ("enc_b", C.c_int64 * (OG_MAX_HIDDEN + 1))] + \
               [(k, C.c_int64) for k in ("layer0", "layer_stride", "o_wqkv", "o_bqkv", "o_w0", "o_b0", "o_w3", "o_b3",
                                         "wp", "bp", "alpha", "dustbin", "total", "o_scale", "scales", "o_wmlp", "o_wqkvs", "o_wqkvb")]


# every symbol include/openglue_amd.h declares: name -> (restype, argtypes)
_i32, _i64, _f, _vp, _sz = C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_size_t
SYMBOLS = {
    "og_abi_version": (C.c_int, []),
    "og_check_shape": (C.c_int, [C.POINTER(og_shape)]),
    "og_packed_weights_bytes": (_sz, [C.POINTER(og_shape)]),
    "og_workspace_bytes": (_sz, [C.POINTER(og_shape)]),
    "og_packed_layout": (C.c_int, [C.POINTER(og_shape), C.POINTER(og_packed_layout_t)]),
    "og_pack_weights": (C.c_int, [C.POINTER(og_shape), C.POINTER(og_params), _vp]),
    "og_forward": (C.c_int, [C.POINTER(og_shape), C.POINTER(og_inputs), _vp, _vp, C.POINTER(og_outputs), _vp]),
    "og_forward_tap": (C.c_int, [C.POINTER(og_shape), C.POINTER(og_inputs), _vp, _vp, C.POINTER(og_outputs), _vp, _i32, _vp]),
    "og_forward_ragged": (C.c_int, [C.POINTER(og_shape), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_float),
                                    C.POINTER(C.c_float), C.POINTER(og_inputs), _vp, _vp, C.POINTER(og_outputs), _vp]),
    "og_forward_profiled": (C.c_int, [C.POINTER(og_shape), C.POINTER(og_inputs), _vp, _vp, C.POINTER(og_outputs), _vp,
                                      C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    "og_forward_ragged_profiled": (C.c_int, [C.POINTER(og_shape), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_float),
                                             C.POINTER(C.c_float), C.POINTER(og_inputs), _vp, _vp, C.POINTER(og_outputs), _vp,
                                             C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    "og_gemm_nt": (C.c_int, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _vp, _i32,
                             _vp, _i64, _vp, _f, _vp]),
    "og_gemm_kmajor": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _i64, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _f, _vp]),
    "og_attention_train_lse": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _f, _vp, _vp]),
    "og_attention_backward_parts": (C.c_int, [_i32]),
    "og_attention_backward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f, _vp, _vp, _vp, _vp]),
    "og_attention_backward_ld": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f, _vp, _vp, _i64, _vp, _i64, _vp]),
    "og_attention_delta": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    "og_split_f16_rows": (C.c_int, [_vp, _i64, _i64, _i32, _i32, _f, _f, _vp, _vp, _i64, _vp]),
    "og_merge_f16": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "og_splitk_reduce": (C.c_int, [_vp, _i32, _i32, _i64, _i32, _vp, _vp, _vp]),
    "og_split_f16": (C.c_int, [_vp, _i64, _vp, _vp, _vp]),
    "og_split_f16_hl": (C.c_int, [_vp, _i64, _i32, _i64, _vp, _i64, _vp]),
    "og_gemm_nt_f16x3": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _f, _vp, _i32, _vp, _i64, _vp, _i64,
                                   _vp, _vp, _i64, _i32, _vp]),
    "og_gemm_nt_f16x3_reshl": (C.c_int, [_vp, _i64, _vp, _i64, _i32, _i32, _i32, _f, _vp, _i32, _vp, _i64, _vp, _i64,
                                         _vp, _vp, _i64, _i32, _vp]),
    "og_proj_block_stream_bytes": (_sz, [_i32, _i32]),
    "og_proj_block_pack": (C.c_int, [_i32, _i32, _vp, _vp]),
    "og_proj_block": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp]),
    "og_mlp_block_stream_bytes": (_sz, [_i32]),
    "og_mlp_block_pack": (C.c_int, [_i32, _vp, _vp, _vp]),
    "og_mlp_block": (C.c_int, [_i32, _vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    "og_attention": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _i32, _i32,
                               _i32, _vp, _vp]),
    "og_keypoint_encoder": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "og_scores": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _i64, _vp]),
    "og_sinkhorn_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "og_sinkhorn_status": (C.c_int, [_vp, _i32, _i32, _i32]),
    "og_sinkhorn_schedule": (C.c_int, [_i32, _i32, _i32, _i32]),
    "og_sinkhorn_schedule_ragged": (C.c_int, [_i32, _vp, _vp, _i32]),
    "og_sinkhorn_resident_geometry": (C.c_int, [_i32, _i32, _vp]),
    "og_sinkhorn_resident_ragged_footprint": (C.c_int, [_i32, _vp, _vp, _vp]),
    "og_sinkhorn_resident_rows_per_wave": (C.c_int, [_i32, _i32, _i32]),
    "og_forward_status": (C.c_int, [_vp, _vp]),
    "og_sinkhorn_train_workspace_bytes": (_sz, [_i32, _i32, _i32, _i32]),
    "og_sinkhorn_train_forward": (C.c_int, [_vp, _i64, _f, _vp, _i32, _i32, _i32, _i32, _f, _vp, _vp, _vp]),
    "og_sinkhorn_backward": (C.c_int, [_vp, _i64, _f, _vp, _i32, _i32, _i32, _i32, _f, _vp, _vp, _vp, _i64, _vp, _vp]),
    "og_batchnorm_train_workspace_bytes": (_sz, [_i64, _i32]),
    "og_batchnorm_train_forward": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _vp, _f, _f, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "og_batchnorm_train_backward": (C.c_int, [_vp, _i64, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _i32, _vp, _i64, _vp, _vp, _vp, _vp]),
    "og_transpose_f32": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _i64, _vp]),
    "og_colsum_f32": (C.c_int, [_vp, _i64, _i64, _i32, _vp, _vp, _vp]),
    "og_transpose_f32_batched": (C.c_int, [_vp, _i64, _i64, _i64, _i32, _vp, _i64, _i64, _i32, _vp]),
    "og_softmax_rows": (C.c_int, [_vp, _i64, _i64, _i32, _vp]),
    "og_softmax_rows_backward": (C.c_int, [_vp, _vp, _i64, _i64, _i32, _f, _vp]),
    "og_sinkhorn": (C.c_int, [_vp, _i64, _f, _i32, _i32, _i32, _i32, _f, _vp, _vp, _vp]),
    "og_matches_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "og_prepare_features": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp]),
    "og_compact_workspace_bytes": (_sz, [_i32, _i32]),
    "og_compact_matches": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "og_extract_matches": (C.c_int, [_vp, _i32, _i32, _i32, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the HIP library (built by `python -m openglue_amd.build`).  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # torch wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7).  If OUR library were
    # dlopen'ed first it would pull /opt/rocm's copy and the process would end up with TWO HIP runtimes
    # (ours then sees no device: hipErrorNoDevice at the first launch).  Importing torch first makes the
    # loader resolve our DT_NEEDED libamdhip64.so.7 to the copy torch already loaded.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -m openglue_amd.build` "
            "(hipcc --offload-arch=gfx950).  openglue_amd has no CPU / eager fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)           # AttributeError if the .so does not export it
        fn.restype, fn.argtypes = restype, argtypes
    if lib.og_abi_version() != OG_ABI_VERSION:
        raise RuntimeError(f"libopenglue_amd ABI {lib.og_abi_version()} != binding {OG_ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    if rc < 0:
        raise RuntimeError(f"{what}: {_ERRORS.get(rc, rc)}")
    raise RuntimeError(f"{what}: HIP error {rc} at kernel launch")
