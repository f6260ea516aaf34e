"""Importable stand-in for the reference's pybind module ``turbo_diffusion_ops``
(``/root/reference/turbodiffusion/ops/bindings.cpp:6-16``): the four functions with the pybind signatures, on top of
``libturbodiffusion_amd.so``.  Put this directory on ``sys.path`` (or ``sys.modules["turbo_diffusion_ops"] = this
module``) and the reference's ``ops/core.py`` — ``from turbo_diffusion_ops import quant_cuda, gemm_cuda`` (:9) — runs
on the MI355X unchanged.

Semantics kept from the .cu wrappers:
  quant_cuda(Input[m,n] f16|bf16, Output|None, Output_S|None) -> (Output int8 [m,n], Output_S f32 [ceil(m/128), ceil(n/128)])
      allocates what is None (ops/quant/quant.cu:28-43, common/common.hpp:65-84), raises on another dtype (:64-67)
  gemm_cuda(A int8[m,k], A_S f32, B int8[n,k], B_S f32, C f16|bf16 [m,n]) -> None, writes C in place
      (ops/gemm/gemm.cu:27-68); raises on another output dtype (:62-65)
  rms_norm_cuda(Input f32[m,n], eps, Weight|None, Output|None) -> Output f32   (ops/norm/rmsnorm.cu:12-59)
  layer_norm_cuda(Input f32[m,n], eps, W|None, B|None, Output|None) -> Output f32   (ops/norm/layernorm.cu:10-62)
Differences: an unsupported shape (k % 128, n % 8) raises instead of silently skipping the launch
(ops/gemm/launch.hpp:34-35); nothing ever calls exit().  Kernels run on torch's current HIP stream.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .. import kernels as K
from .._lib import call, dt_code, ptr, require_gpu, stream_ptr

__all__ = ["quant_cuda", "gemm_cuda", "gemm_cuda_swizzle", "gemm_cuda_swizzle_bias", "rms_norm_cuda", "layer_norm_cuda"]


def quant_cuda(Input: torch.Tensor, Output: Optional[torch.Tensor] = None,
               Output_S: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    require_gpu(Input, Output, Output_S)
    if Input.dtype not in (torch.float16, torch.bfloat16):
        raise RuntimeError("Unsupported input data type for quantize_to_fp4.")   # the reference's message, quant.cu:66
    assert Input.dim() == 2 and Input.is_contiguous()
    m, n = Input.shape
    if Output is None:
        Output = torch.empty((m, n), dtype=torch.int8, device=Input.device)
    if Output_S is None:
        Output_S = torch.empty((K.cdiv(m, 128), K.cdiv(n, 128)), dtype=torch.float32, device=Input.device)
    assert Output.dtype == torch.int8 and Output.shape == (m, n) and Output.is_contiguous()
    assert Output_S.dtype == torch.float32 and Output_S.numel() == K.cdiv(m, 128) * K.cdiv(n, 128) and Output_S.is_contiguous()
    call("td_quant_i8_block128", ptr(Input), dt_code(Input.dtype), ptr(Output), ptr(Output_S), m, n, stream_ptr())
    return Output, Output_S


def gemm_cuda(A: torch.Tensor, A_S: torch.Tensor, B: torch.Tensor, B_S: torch.Tensor, C: torch.Tensor) -> None:
    require_gpu(A, A_S, B, B_S, C)
    if C.dtype not in (torch.float16, torch.bfloat16):
        raise RuntimeError("Unsupported output data type for int8 gemm.")          # gemm.cu:64
    assert A.dtype == torch.int8 and B.dtype == torch.int8 and A_S.dtype == torch.float32 and B_S.dtype == torch.float32
    assert A.is_contiguous() and B.is_contiguous() and A_S.is_contiguous() and B_S.is_contiguous()
    m, k = A.shape
    n = B.shape[0]
    assert B.shape[1] == k and C.shape == (m, n) and C.stride(1) == 1
    call("td_gemm_w8a8", ptr(A), ptr(A_S), ptr(B), ptr(B_S), None, ptr(C), dt_code(C.dtype), 0, m, n, k, C.stride(0),
         stream_ptr())


def gemm_cuda_swizzle(A: torch.Tensor, A_S: torch.Tensor, B: torch.Tensor, B_S: torch.Tensor, C: torch.Tensor,
                      swizzle_dir: int = 0, swizzle_log: int = 0) -> None:
    """The call TurboT2AV's ``_TurboDiffusionInt8Linear`` makes (LTX-2 ``ltx_distillation/acceleration.py:753-769``; the
    binding is newer than ``ops/bindings.cpp`` in the reference tree — API drift).  ``swizzle_dir`` /
    ``swizzle_log`` are the CUTLASS threadblock-raster hint of that kernel: results do not depend on them, and this
    library's raster is chosen per launch (XCD-aware m-grouped walk, csrc/gemm_w8a8_fi.hip) — accepted and ignored."""
    gemm_cuda(A, A_S, B, B_S, C)


def gemm_cuda_swizzle_bias(A: torch.Tensor, A_S: torch.Tensor, B: torch.Tensor, B_S: torch.Tensor, C: torch.Tensor,
                           bias: torch.Tensor, swizzle_dir: int = 0, swizzle_log: int = 0) -> None:
    """``gemm_cuda_swizzle`` with the bias in the epilogue: C = cast(cast(acc) + bias), the rounding sequence of
    ``Int8Linear.forward`` (ops/core.py:408-412: the GEMM result is rounded to the activation dtype, the bias add rounds again)
    — the source of the fused kernel is not in the reference tree, so this is the unfused module's arithmetic."""
    require_gpu(A, A_S, B, B_S, C, bias)
    if C.dtype not in (torch.float16, torch.bfloat16):
        raise RuntimeError("Unsupported output data type for int8 gemm.")
    assert A.dtype == torch.int8 and B.dtype == torch.int8 and A_S.dtype == torch.float32 and B_S.dtype == torch.float32
    assert A.is_contiguous() and B.is_contiguous() and A_S.is_contiguous() and B_S.is_contiguous()
    m, k = A.shape
    n = B.shape[0]
    assert B.shape[1] == k and C.shape == (m, n) and C.stride(1) == 1
    assert bias.dtype == C.dtype and bias.shape == (n,) and bias.is_contiguous()
    call("td_gemm_w8a8", ptr(A), ptr(A_S), ptr(B), ptr(B_S), ptr(bias), ptr(C), dt_code(C.dtype), 0, m, n, k, C.stride(0),
         stream_ptr())


def _norm_out(Input, Output):
    assert Input.dtype == torch.float32 and Input.dim() == 2 and Input.is_contiguous(), "fp32 [m, n] input (rmsnorm.cu:19)"
    if Output is None:
        Output = torch.empty(Input.shape, dtype=torch.float32, device=Input.device)
    assert Output.dtype == torch.float32 and Output.shape == Input.shape and Output.is_contiguous()
    return Output


def rms_norm_cuda(Input: torch.Tensor, eps: float, Weight: Optional[torch.Tensor] = None,
                  Output: Optional[torch.Tensor] = None) -> torch.Tensor:
    require_gpu(Input, Weight, Output)
    Output = _norm_out(Input, Output)
    m, n = Input.shape
    w = torch.ones(n, dtype=torch.float32, device=Input.device) if Weight is None else Weight.float().contiguous()
    call("td_rmsnorm", ptr(Input), dt_code(torch.float32), ptr(w), ptr(Output), dt_code(torch.float32), float(eps), m, n,
         stream_ptr())
    return Output


def layer_norm_cuda(Input: torch.Tensor, eps: float, W: Optional[torch.Tensor] = None, B: Optional[torch.Tensor] = None,
                    Output: Optional[torch.Tensor] = None) -> torch.Tensor:
    require_gpu(Input, W, B, Output)
    Output = _norm_out(Input, Output)
    m, n = Input.shape
    w = None if W is None else W.float().contiguous()
    b = None if B is None else B.float().contiguous()
    if w is None and b is not None:      # BIAS without AFFINE (layernorm.cu:41-42 allows it): unit weight
        w = torch.ones(n, dtype=torch.float32, device=Input.device)
    if w is not None and b is None:
        b = torch.zeros(n, dtype=torch.float32, device=Input.device)
    call("td_layernorm", ptr(Input), dt_code(torch.float32), ptr(w), ptr(b), None, None, 0, ptr(Output),
         dt_code(torch.float32), float(eps), 0, m, n, stream_ptr())   # pad_cols 0: the CUDA twin's textbook variance
    return Output
