"""ctypes binding of libturbodiffusion_amd.so (the C-ABI in include/turbodiffusion_amd.h).

PyTorch is plumbing only: it owns device memory and the HIP stream; every operator below
hands raw device pointers to hand-written HIP.  There is NO fallback: if the shared library
is missing or an entry point fails, an exception is raised.
"""
from __future__ import annotations

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# TD_LIB_PATH: development aid for A/B runs against another build of the SAME ABI (tools/gpu/*_ab.sh); never a fallback
LIB_PATH = os.environ.get("TD_LIB_PATH") or os.path.join(_HERE, "libturbodiffusion_amd.so")

TD_F16, TD_BF16, TD_F32 = 0, 1, 2
TD_EPI_NONE, TD_EPI_GELU_TANH = 0, 1
ABI_VERSION = 5

_i64, _i32, _f32, _vp = ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_void_p

# name -> argtypes, exactly the prototypes of include/turbodiffusion_amd.h
SIGNATURES = {
    "td_abi_version": [],
    "td_last_error": [],
    "td_set_tuning": [_i32, _i32],
    "td_debug_read": [_vp, _i32],
    "td_patch_embed": [_vp, _i64, _vp, _i64, _i32, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _i64, _vp],
    "td_head": [_vp, _i32, _vp, _vp, _vp, _vp, _f32, _vp, _i32, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _vp],
    "td_time_sinusoid": [_vp, _i32, _vp, _i64, _i64, _vp],
    "td_vae_conv": [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "td_vae_conv_ex": [_vp, _i64, _vp, _vp, _vp, _vp, _i64] + [_i32] * 19 + [_vp],
    "td_vae_chan_rms": [_vp, _vp, _vp, _i64, _i32, _i32, _vp],
    "td_gemm_bf16": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32] + [_i64] * 12 + [_vp],
    "td_gemm_bf16_splitk_reduce": [_vp, _i32, _vp, _vp, _vp, _i32, _i32, _i64, _i64, _i64, _i64, _vp],
    "td_softmax_rows": [_vp, _i32, _vp, _i32, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _f32, _vp],
    "td_t5_norm": [_vp, _vp, _vp, _i32, _f32, _i64, _i64, _i64, _i64, _vp],
    "td_gemv_f32": [_vp, _vp, _vp, _i32, _i32, _vp, _i64, _i64, _i64, _vp],
    "td_bcast_add": [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp],
    "td_rcm_step": [_vp, _vp, _vp, _vp, _i32, ctypes.c_double, ctypes.c_double, _i64, _vp],
    "td_calib_mfma_i8": [_i32, _i32, _vp, _vp],
    "td_calib_hbm_read": [_vp, _i64, _vp, _vp],
    "td_calib_clock_probe": [_i64, _vp, _vp],
    "td_quant_i8_block128": [_vp, _i32, _vp, _vp, _i64, _i64, _vp],
    "td_gemm_w8a8": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i64, _i64, _i64, _i64, _vp],
    "td_gemm_w8a8_quant": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i64, _i64, _i64, _vp],
    "td_gemm_w8a8_residual": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _i64, _i64, _vp],
    "td_rmsnorm": [_vp, _i32, _vp, _vp, _i32, _f32, _i64, _i64, _vp],
    "td_layernorm": [_vp, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _i32, _f32, _i64, _i64, _i64, _vp],
    "td_layernorm_quant": [_vp, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _f32, _i64, _i64, _i64, _vp],
    "td_gated_residual": [_vp, _vp, _vp, _i64, _i32, _i64, _i64, _vp],
    "td_qk_norm_rope": [_vp, _i64, _vp, _vp, _vp, _vp, _i32, _f32, _i64, _i32, _i32, _vp],
    "td_v_transpose": [_vp, _i32, _i64, _i64, _vp, _i32, _i64, _i32, _i32, _vp],
    "td_seq_mean": [_vp, _vp, _vp, _i32, _i64, _i32, _i32, _vp],
    "td_sage_quant_pool": [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _i64, _i32, _i32, _vp],
    "td_seq_sum_partial": [_vp, _vp, _i32, _i64, _i32, _i32, _vp],
    "td_seq_mean_final": [_vp, _i32, _i64, _i64, _vp, _i32, _i64, _i32, _i32, _vp],
    "td_sla_topk": [_vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "td_attn_i8": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i64, _i64, _f32, _i64, _i64, _i64, _i32, _vp],
    "td_attn_16": [_vp, _vp, _vp, _vp, _i32, _vp, _i32, _i64, _i64, _f32, _i64, _i64, _i64, _i32, _vp],
    "td_attn_i8_ex": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i64, _i64, _f32, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _vp],
    "td_gemm_w8a8_stats": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i64, _i64, _i64, _i64, _vp, _vp],
    "td_gemm_w8a8_vt": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _i64, _i64, _i64, _vp, _i32, _vp],
    "td_row_stats_finalize": [_vp, _i32, _i64, _f32, _i64, _i32, _vp, _i64, _vp],
    "td_layernorm_quant_stats": [_vp, _i32, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i64, _vp],
    "td_attn_i8_sp": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i64, _i64, _f32, _i64, _i64, _i32, _i32, _i64, _i64, _i64,
                      _vp, _vp, _vp, _i32, _vp],
    "td_attn_16_sp": [_vp, _vp, _vp, _vp, _i32, _vp, _i32, _i64, _i64, _f32, _i64, _i64, _i32, _i32, _i64, _i64, _vp, _vp, _vp,
                      _i32, _vp],
    "td_seq_sum": [_vp, _vp, _vp, _i32, _i64, _i32, _i32, _vp],
    "td_qk_norm_rope_pair": [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _f32, _i64, _i32, _i32, _vp],
    "td_sage_quant_pool_packed_kmsum": [_vp, _vp, _vp, _i32, _i64, _i64, _i32, _i32, _vp, _i32, _i64, _vp, _vp, _i64, _i64, _i32, _i64,
                                        _i32, _i32, _vp],
    "td_sla_topk_sp": [_vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i64, _i32, _i32, _vp],
    "td_v_fp8_tiles": [_vp, _i32, _i64, _i64, _vp, _vp, _vp, _f32, _i64, _i32, _i32, _vp],
    "td_attn_i8_fp8pv": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i64, _i64, _f32, _i64, _i64, _i64, _i32, _vp, _vp,
                         _vp, _vp],
    "td_rms_stats": [_vp, _i64, _i32, _vp, _f32, _i64, _i64, _vp],
    "td_attn_16_qnorm": [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i64, _i64, _f32, _i64, _i64, _i64, _i32,
                         _vp, _vp, _vp, _vp],
    "td_attn_16_qnorm_pieces": [_vp, _i64, _vp, _i32, _f32, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _i64, _i64, _f32, _i64, _i64, _i64, _i32,
                                _vp, _vp, _vp, _vp],
    "td_attn_16_ex": [_vp, _vp, _vp, _vp, _i32, _vp, _i32, _i64, _i64, _f32, _i64, _i64, _i64, _i32, _vp, _vp, _vp, _vp],
    "td_sla_linear_out_t": [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp],
    "td_sla_linear_kv_partial": [_vp, _i32, _vp, _i32, _vp, _vp, _i64, _i32, _i32, _vp],
    "td_sla_linear_kv_final": [_vp, _vp, _i32, _i64, _i64, _i64, _i64, _vp, _vp, _i32, _i32, _i32, _vp],
    "td_sage_quant_pool_packed": [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _i64, _i64, _i32, _i64, _i32, _i32, _vp],
    "td_v_transpose_packed": [_vp, _i32, _i64, _i64, _vp, _i32, _i64, _i64, _i32, _i64, _i32, _i32, _vp],
    "td_sla_linear_kv_partial_packed": [_vp, _i32, _vp, _i32, _vp, _vp, _i64, _i64, _i32, _i64, _i32, _i32, _vp],
    "td_sla_linear_kv_final_packed": [_vp, _vp, _i32, _i64, _i64, _i64, _i64, _vp, _vp, _i32, _i32, _i64, _i32, _i32, _vp],
    "td_sla_linear_kv": [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _vp],
    "td_sla_linear_out": [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _vp],
    "td_sla_linear_kv_fm": [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _vp],
    "td_sla_linear_out_fm": [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i64, _i32, _i32, _vp],
}

_lib = None


class TurboDiffusionAMDError(RuntimeError):
    pass


def load():
    """Load the shared library (once). Raises if it is not built — no CPU/eager fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TurboDiffusionAMDError(
            f"{LIB_PATH} not found: build it with `python -m turbodiffusion_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no fallback path."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.argtypes = argtypes
        fn.restype = ctypes.c_char_p if name == "td_last_error" else ctypes.c_int
    ver = lib.td_abi_version()
    if ver != ABI_VERSION:
        raise TurboDiffusionAMDError(f"ABI version mismatch: library {ver}, binding {ABI_VERSION}")
    _lib = lib
    return lib


def dt_code(dtype: torch.dtype) -> int:
    if dtype == torch.float16:
        return TD_F16
    if dtype == torch.bfloat16:
        return TD_BF16
    if dtype == torch.float32:
        return TD_F32
    raise TurboDiffusionAMDError(f"unsupported dtype {dtype}")


def ptr(t):
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise TurboDiffusionAMDError(
                "turbodiffusion_amd operators run on the MI355X only (got a CPU tensor); "
                "there is no CPU fallback — the CPU oracle lives under oracle/ and is test-only."
            )


def call(name: str, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.td_last_error()
        raise TurboDiffusionAMDError(f"{name} failed (status {rc}): {msg.decode() if msg else ''}")
