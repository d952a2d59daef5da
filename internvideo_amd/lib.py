"""ctypes binding of libinternvideo_hip.so (the C ABI declared in include/internvideo_hip.h).

The library is the product: there is NO CPU / PyTorch fallback for any op.  Importing this module never
needs a GPU (so that CPU-only hosts can build, inspect symbols and run the host-logic tests), but every
compute entry point goes through `call()`, which raises if the library is missing and `require_gpu()`,
which raises if no MI355X is visible.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# IVH_LIB_PATH: measurement aid -- load another build of the SAME library (same-box A/B of a kernel change: tools/gpu_ab_libs.sh); the
# product path is the in-tree library
LIB_PATH = os.environ.get("IVH_LIB_PATH") or os.path.join(_HERE, "csrc", "libinternvideo_hip.so")

_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
_u32 = C.c_uint32


class GemmDesc(C.Structure):
    """struct ivh_gemm_desc"""
    _fields_ = [
        ("A", _vp), ("B", _vp), ("lda", _i64), ("ldb", _i64),
        ("M", _i32), ("N", _i32), ("K", _i32), ("a_kc", _i32), ("b_kc", _i32),
        ("C", _vp), ("ldc", _i64), ("c_fp32", _i32),
        ("bias", _vp), ("act", _i32),
        ("preact", _vp), ("ldp", _i64),
        ("dact_in", _vp), ("ldd", _i64),
        ("alpha", _f32), ("batch", _i32), ("colsum_part", _vp),
        ("strideA", _i64), ("strideB", _i64), ("strideC", _i64),
        ("stride_bias", _i64), ("stride_preact", _i64), ("stride_dact", _i64),
        ("split_ws", _vp), ("split_ws_bytes", _i64),
        ("m_dev", _vp), ("k_dev", _vp),                      # ABI 2: device-side row counts
    ]


# name -> argtypes (restype is int unless listed in _RESTYPES)
SIGNATURES = {
    "ivh_last_error": [],
    "ivh_version": [],
    "ivh_abi_version": [],
    "ivh_droppath_plan": [_vp, _i32, _i32, _i32, _vp, _vp, _vp],
    "ivh_rmsnorm_add_fwd_skip": [_vp, _i32, _vp, _vp, _vp, _i32, _vp, _f32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    "ivh_rmsnorm_add_bwd_skip": [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ivh_colsum_finish_dyn": [_vp, _i32, _i32, _vp, _i32, _vp, _i32, _i32, _vp],
    "ivh_qk_rmsnorm_fwd_dyn": [_vp, _vp, _vp, _f32, _i32, _i32, _vp, _vp, _vp, _vp],
    "ivh_qk_rmsnorm_bwd_dyn": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp],
    "ivh_flash_attn_fwd_dyn": [_vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp,
                               _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp],
    "ivh_flash_attn_bwd_dyn": [_vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _vp, _vp,
                               _vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp],
    "ivh_set_dropout_epoch": [_vp],
    "ivh_device_info": [C.POINTER(_i32), C.POINTER(_i32), C.c_char_p, _i32],
    "ivh_gemm_bf16": [C.POINTER(GemmDesc), _vp],
    "ivh_gemm_grouped_bf16": [C.POINTER(GemmDesc), _i32, _vp],
    "ivh_fp8_quantize": [_vp, _i64, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _vp],
    "ivh_fp8_quantize_delayed": [_vp, _i64, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp],
    "ivh_set_gemm_fp8_kernel": [_i32],
    "ivh_gemm_fp8": [C.POINTER(GemmDesc), _vp, _vp, _vp],
    "ivh_gemm_fp8_cs": [C.POINTER(GemmDesc), _vp, _vp, _vp],
    "ivh_fp8_quantize_weight": [_vp, _i64, _i32, _i32, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp],
    "ivh_set_gemm_kernel": [_i32],
    "ivh_gemm_select": [C.POINTER(GemmDesc)],
    "ivh_gemm256_debug": [_i32, _i32],
    "ivh_gemm256_debug_stamps": [_vp],
    "ivh_gemm256_debug_max_wg": [_i32],
    "ivh_gemm256_debug_ablate": [_i32],
    "ivh_gemm256_debug_split": [_i32],
    "ivh_gemm256_half_plan": [C.POINTER(GemmDesc), _i32, C.POINTER(_i32)],
    "ivh_gemm256_debug_half": [_i32],
    "ivh_gemm256_half_rounds": [C.POINTER(GemmDesc)],
    "ivh_attn32_debug_stamps": [_vp, _i64],
    "ivh_gemm_split_workspace": [C.POINTER(GemmDesc)],
    "ivh_gemm_fp8_split_workspace": [C.POINTER(GemmDesc)],
    "ivh_gemm256_debug_sched": [_i32],
    "ivh_rmsnorm_add_fwd": [_vp, _vp, _vp, _vp, _i32, _vp, _f32, _i32, _i32, _vp, _vp, _vp, _vp],
    "ivh_norm_bwd_parts": [_i32],
    "ivh_qk_norm_bwd_parts": [_i32, _i32],
    "ivh_rmsnorm_add_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    "ivh_rmsnorm_add_fwd_bf16res": [_vp, _vp, _vp, _vp, _i32, _vp, _f32, _i32, _i32, _vp, _vp, _vp, _vp],
    "ivh_rmsnorm_add_bwd_bf16res": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ivh_colsum_finish": [_vp, _i32, _i32, _vp, _i32, _vp],
    "ivh_colsum_finish_multi": [C.POINTER(_vp), C.POINTER(_vp), _i32, _i32, _i32, _i32, _vp],
    "ivh_colsum_bf16": [_vp, _i64, _i32, _i32, _vp, _vp, _vp],
    "ivh_colsum_scratch_floats": [_i32, _i32],
    "ivh_qk_rmsnorm_fwd": [_vp, _vp, _vp, _f32, _i32, _i32, _vp, _vp, _vp],
    "ivh_qk_rmsnorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp],
    "ivh_flash_attn_fwd": [_vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp,
                           _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp],
    "ivh_flash_attn_bwd": [_vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _vp, _vp,
                           _vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp],
    "ivh_flash_attn_fwd_dropout": [_vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp,
                                   _i32, _i32, _i32, _i32, _i32, _f32, _vp, _f32, _u32, _vp],
    "ivh_flash_attn_bwd_dropout": [_vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _vp, _vp,
                                   _vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _f32, _u32, _vp],
    "ivh_set_attn_kernel": [_i32],
    "ivh_mask_to_indices": [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp],
    "ivh_patch_im2col": [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
    "ivh_assemble_tokens": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp],
    "ivh_add_pos_gather": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp],
    "ivh_add_pos_gather_bf16": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp],
    "ivh_rows_shift_bf16": [_vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "ivh_frames_merge_l2": [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp],
    "ivh_pool_attn_map": [_vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _f32, _i32, _vp, _vp],
    "ivh_assemble_tokens_nocls": [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp],
    "ivh_mae_decoder_input": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp],
    "ivh_rows_window": [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
    "ivh_rows_window_bwd": [_vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
    "ivh_pixel_target": [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, C.POINTER(_f32), C.POINTER(_f32), _vp, _vp],
    "ivh_mse_rows": [_vp, _i32, _vp, _i32, _i32, _f32, _vp, _vp, _vp],
    "ivh_pool_attn_fwd": [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp],
    "ivh_pool_attn_bwd": [_vp, _vp, _vp, _i64, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _vp],
    "ivh_cosine_rows": [_vp, _i32, _vp, _i32, _i32, _i32, _f32, _vp, _vp, _vp],
    "ivh_gather_rows": [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _vp],
    "ivh_rows_to_bf16": [_vp, _i32, _i32, _i32, _i32, _vp, _vp],
    "ivh_accum_rows": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "ivh_pos_grad": [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _i32, _vp],
    "ivh_ln_l2_fwd": [_vp, _vp, _vp, _f32, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _vp],
    "ivh_ln_l2_bwd": [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _f32, _vp, _i32, _i32, _vp, _vp, _vp, _vp],
    "ivh_sum_rows": [_vp, _i32, _f32, _vp, _vp],
    "ivh_token_mean_fwd": [_vp, _i32, _i32, _i32, _vp, _vp],
    "ivh_token_mean_bwd": [_vp, _i32, _i32, _i32, _vp, _vp],
    "ivh_layernorm_fwd": [_vp, _i32, _vp, _vp, _vp, _vp, _f32, _i32, _i32, _vp, _vp, _vp, _vp],
    "ivh_layernorm_bwd": [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp],
    "ivh_adamw_step": [_vp, _vp, _vp, _vp, _i32, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i32, _f32, _vp, _vp],
    "ivh_adamw_step_scaled": [_vp, _vp, _vp, _vp, _i32, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _i32, _f32, _vp, _vp, _vp, _i32, _i64, _vp],
    "ivh_sqnorm_scratch_floats": [],
    "ivh_sqnorm": [_vp, _i32, _i64, _vp, _vp, _i32, _vp],
    "ivh_clip_coef": [_vp, _f32, _vp, _vp, _vp],
    "ivh_shard_sum_bf16": [_vp, _i32, _i64, _vp, _vp],
    "ivh_vtc_workspace_floats": [_i32, _i32],
    "ivh_vtc_abt": [_vp, _vp, _i32, _i32, _i32, _f32, _vp, _vp],
    "ivh_vtc_loss_fwd_bwd": [_vp, _vp, _vp, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ivh_vtc_loss_fwd_bwd_dev": [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ivh_bert_embed_fwd": [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _f32, _i32, _vp, _vp, _f32, _u32, _vp],
    "ivh_bert_embed_bwd": [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _f32, _u32, _vp],
    "ivh_add_layernorm_fwd": [_vp, _vp, _i32, _vp, _vp, _f32, _i32, _i32, _vp, _vp, _f32, _u32, _vp],
    "ivh_add_layernorm_bwd": [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _f32, _u32, _vp],
    "ivh_ce_rows": [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _f32, _vp, _vp, _vp, _vp, _i32, _vp],
    "ivh_probe_cu_hog": [_vp, _vp, _i64, _i32, _vp],
    "ivh_probe_tr16": [_vp, _vp, _vp],
    "ivh_probe_mfma16": [_vp, _vp, _vp, _vp],
    "ivh_probe_mfma32": [_vp, _vp, _vp, _vp],
    "ivh_probe_mfma_rate": [_i32, _i32, _vp, _vp],
    "ivh_probe_mfma_rate2": [_i32, _i32, _i32, _i32, _vp, _vp],
    "ivh_probe_attn32_pingpong": [_i32],
    "ivh_probe_attn32_unpacked": [_i32],
    "ivh_probe_attn32_fwd_qkn": [_vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _vp,
                                 _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _vp],
}
_RESTYPES = {"ivh_last_error": C.c_char_p, "ivh_vtc_workspace_floats": C.c_int64, "ivh_gemm_split_workspace": C.c_int64, "ivh_gemm_fp8_split_workspace": C.c_int64,
             "ivh_gemm256_half_rounds": C.c_double}

_lib: Optional[C.CDLL] = None


class InternVideoHipError(RuntimeError):
    pass


def load(required: bool = True) -> Optional[C.CDLL]:
    """dlopen the in-tree library and attach signatures.  Raises if it is missing (no fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        if required:
            raise InternVideoHipError(
                f"{LIB_PATH} not found: build it with `python -m internvideo_amd.csrc.build` "
                "(__graft_entry__.build()).  internvideo_amd has no CPU/PyTorch fallback path.")
        return None
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    _lib = lib
    return lib


def exported_symbols():
    return sorted(SIGNATURES)


def require_gpu() -> None:
    if not torch.cuda.is_available():
        raise InternVideoHipError("internvideo_amd needs a visible MI355X (torch.cuda.is_available() is False); "
                                  "there is no CPU fallback")


def last_error() -> str:
    lib = load()
    return (lib.ivh_last_error() or b"").decode()


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def call(name: str, *args) -> None:
    """Invoke an int-returning entry point; raise with the library's message on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise InternVideoHipError(f"{name} failed ({rc}): {last_error()}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr()
