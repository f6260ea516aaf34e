"""Host-side mirror of the reference operator API ``turbodiffusion.ops``
(``/root/reference/turbodiffusion/ops/__init__.py:1-2``, ``ops/core.py``): same names,
argument meaning, buffer names/shapes (the checkpoint contract) and error behaviour —
but every operator runs hand-written HIP for gfx950 through the C-ABI library.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn as nn

from . import kernels as K

__all__ = ["int8_quant", "int8_linear", "rmsnorm", "layernorm", "Int8Linear", "FastRMSNorm", "FastLayerNorm"]


def int8_quant(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """ops/core.py:12-25 — per-128x128-block INT8 quantisation of a float16/bfloat16 [m,n] tensor.
    Returns (x_q int8 [m,n], x_scale f32 [ceil(m/128), ceil(n/128)])."""
    return K.quant_i8_block128(x.contiguous())


def int8_linear(x: torch.Tensor, w_q: torch.Tensor, w_s: torch.Tensor, **kwargs) -> torch.Tensor:
    """ops/core.py:28-57 — dynamic activation quant + W8A8 GEMM. x [..., K] f16|bf16,
    w_q int8 [N, K], w_s f32 [ceil(N/128), ceil(K/128)].  Extra (MI355X) kwargs: ``bias`` and
    ``gelu_tanh`` fuse Int8Linear's bias add / the FFN activation into the GEMM epilogue with the
    reference's rounding order; ``x_q``/``x_s`` reuse an existing quantisation of x."""
    assert w_q.dtype == torch.int8, "Weight tensor must be int8."
    shape = x.shape
    x2 = x.reshape(-1, shape[-1])
    n = w_q.shape[0]
    x_q, x_s = kwargs.get("x_q"), kwargs.get("x_s")
    if x_q is None:
        x_q, x_s = int8_quant(x2)
    y = K.gemm_w8a8(x_q, x_s, w_q, w_s, out_dtype=x.dtype, bias=kwargs.get("bias"),
                    gelu_tanh=bool(kwargs.get("gelu_tanh", False)))
    return y.reshape(*shape[:-1], n)


def rmsnorm(x, w, eps):
    """ops/core.py:139-191 — RMSNorm over the last dim, fp32 math, output in x's dtype."""
    assert x.is_contiguous(), "Input must be contiguous"
    return K.rmsnorm(x, w, eps)


def layernorm(x, w, b, eps, elementwise_affine=True, triton_variance=True):
    """ops/core.py:380-386 -> the Triton kernels :193-242 / :293-335.  Those sum (x - mean)^2 over next_power_of_2(n)
    columns with the masked ones loaded as 0, i.e. var = (sum (x-mean)^2 + (N2 - n) mean^2) / n — reproduced here by
    default (``triton_variance``; pinned to the reference's own kernels run on the MI355X, tests/golden/triton_leaves.pt).
    ``triton_variance=False`` gives the textbook LayerNorm of the eager WanLayerNorm / layer_norm_cuda."""
    pad = K.triton_ln_pad_cols(x.shape[-1]) if triton_variance else 0
    if elementwise_affine:
        assert w is not None and b is not None
        return K.layernorm(x.contiguous(), w, b, eps, pad_cols=pad)
    assert w is None and b is None
    return K.layernorm(x.contiguous(), None, None, eps, pad_cols=pad)


def cdiv(a: int, b: int):
    return (a + b - 1) // b


class Int8Linear(nn.Module):
    """ops/core.py:391-432 — buffers ``int8_weight [out,in] int8``, ``scale [ceil(out/128),
    ceil(in/128)] f32``, ``bias [out]`` are the published-checkpoint contract."""

    def __init__(self, in_features, out_features, bias=True, dtype=torch.bfloat16):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        row_blocks = cdiv(out_features, b=128)
        col_blocks = cdiv(in_features, b=128)
        self.register_buffer("int8_weight", torch.empty((out_features, in_features), dtype=torch.int8))
        self.register_buffer("scale", torch.empty((row_blocks, col_blocks), dtype=torch.float32))
        if bias:
            self.register_buffer("bias", torch.empty(out_features, dtype=dtype))
        else:
            self.bias = None

    def forward(self, x):
        # bias is added inside the GEMM epilogue, after the GEMM result has been rounded to
        # x.dtype — bit-identical to the reference's separate ``out + self.bias``
        b = self.bias
        if b is not None and b.dtype != x.dtype:
            b = b.to(x.dtype)
        return int8_linear(x, self.int8_weight, self.scale, bias=b)

    @classmethod
    def from_linear(cls, original_linear: nn.Linear, quantize: bool = True):
        int8_layer = cls(
            original_linear.in_features,
            original_linear.out_features,
            bias=original_linear.bias is not None,
            dtype=original_linear.weight.dtype,
        )
        if quantize:
            w_data = original_linear.weight.data.cuda()
            if w_data.dtype == torch.float32:
                w_data = w_data.to(torch.bfloat16)
            int8_w, scale = int8_quant(w_data)
            int8_layer.int8_weight = int8_w
            int8_layer.scale = scale
            if original_linear.bias is not None:
                int8_layer.bias = original_linear.bias.data.cuda().clone()
        return int8_layer


class FastRMSNorm(nn.Module):
    """ops/core.py:434-452 (buffer ``weight`` fp32)."""

    def __init__(self, dim: int, eps: float = 1e-5):
        super().__init__()
        self.dim = dim
        self.eps = eps
        self.register_buffer("weight", torch.ones(dim))

    def forward(self, x):
        # reference: rmsnorm(x.float(), w, eps).to(x.dtype); the fused kernel reads x's dtype
        # directly and rounds once at the end — the same value, without the fp32 round trip
        return K.rmsnorm(x.contiguous(), self.weight, self.eps)

    @classmethod
    def from_rmsnorm(cls, original_rmsnorm):
        layer = cls(dim=original_rmsnorm.dim, eps=original_rmsnorm.eps)
        if original_rmsnorm.weight.device != torch.device("meta"):
            layer.weight = original_rmsnorm.weight.float().data.clone()
        return layer


class FastLayerNorm(nn.Module):
    """ops/core.py:454-492."""

    def __init__(self, dim: int, eps: float = 1e-5, elementwise_affine: bool = False, bias: bool = True):
        super().__init__()
        self.dim = dim
        self.eps = eps
        self.elementwise_affine = elementwise_affine
        if self.elementwise_affine:
            self.register_buffer("weight", torch.empty(self.dim))
            if bias:
                self.register_buffer("bias", torch.empty(self.dim))
            else:
                self.bias = None
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)

    triton_variance = True   # False: the textbook variance (see ``layernorm``)

    def forward(self, x):
        return layernorm(x, self.weight, self.bias, self.eps, self.elementwise_affine, self.triton_variance)

    @classmethod
    def from_layernorm(cls, original_layernorm):
        layer = cls(
            dim=original_layernorm.normalized_shape[0],
            eps=original_layernorm.eps,
            elementwise_affine=False if original_layernorm.weight is None else True,
            bias=original_layernorm.bias is not None,
        )
        if original_layernorm.weight is not None and original_layernorm.weight.device != torch.device("meta"):
            layer.weight = original_layernorm.weight.data.clone()
        if original_layernorm.bias is not None and original_layernorm.bias.device != torch.device("meta"):
            layer.bias = original_layernorm.bias.data.clone()
        return layer
