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


def _blocks(n: int) -> int:
    return -(-n // 128)


def _on_meta(t) -> bool:
    return t is not None and t.device.type == "meta"


class _ContractModule(nn.Module):
    """The three operator modules share one shape: a few BUFFERS whose names / shapes / dtypes are the published-checkpoint
    contract (``modify_model.replace_*`` + ``load_state_dict(assign=True)``, inference/modify_model.py:56-81,137-138) and a
    ``from_<original>`` constructor that adopts an eager module's tensors.  ``_slots`` declares the buffers; absent ones are
    plain ``None`` attributes (the reference's ``self.bias = None`` / ``register_parameter(name, None)``)."""

    def _declare(self, **slots):
        for name, spec in slots.items():
            if spec is None:
                setattr(self, name, None)
            else:
                shape, dtype, fill = spec
                t = torch.ones(shape, dtype=dtype) if fill == "ones" else torch.empty(shape, dtype=dtype)
                self.register_buffer(name, t)

    def _adopt(self, name, tensor, dtype=None, device=None):
        """Take ``tensor``'s values as buffer ``name`` (a clone; skipped for meta tensors: the checkpoint fills them later)."""
        if tensor is None or _on_meta(tensor):
            return
        t = tensor.detach()
        if device is not None:
            t = t.to(device)
        setattr(self, name, (t.to(dtype) if dtype is not None else t).clone())


class Int8Linear(_ContractModule):
    """``turbodiffusion.ops.Int8Linear`` (ops/core.py:391-432): ``int8_weight [out, in] int8``, ``scale [ceil(out/128),
    ceil(in/128)] f32``, ``bias [out]`` in the model dtype.  forward = dynamic per-128x128-block activation quantisation +
    W8A8 GEMM; the bias rides in the GEMM's epilogue, added after the result has been rounded to ``x.dtype`` — the bits of
    the reference's separate ``out + self.bias`` (ops/core.py:408-412)."""

    def __init__(self, in_features, out_features, bias=True, dtype=torch.bfloat16):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self._declare(int8_weight=((out_features, in_features), torch.int8, None),
                      scale=((_blocks(out_features), _blocks(in_features)), torch.float32, None),
                      bias=((out_features,), dtype, None) if bias else None)

    def forward(self, x):
        b = self.bias
        return int8_linear(x, self.int8_weight, self.scale, bias=None if b is None else (b if b.dtype == x.dtype else b.to(x.dtype)))

    @classmethod
    def from_linear(cls, original_linear: nn.Linear, quantize: bool = True):
        """``quantize=False``: shapes only (the quantised checkpoint is loaded afterwards, modify_model.py:137); True: the
        offline quantiser's job (:156-183) on the GPU — fp32 weights go through bf16 first, as there."""
        lin = original_linear
        layer = cls(lin.in_features, lin.out_features, bias=lin.bias is not None, dtype=lin.weight.dtype)
        if quantize:
            w = lin.weight.data.cuda()
            layer.int8_weight, layer.scale = int8_quant(w.to(torch.bfloat16) if w.dtype == torch.float32 else w)
            layer._adopt("bias", lin.bias, device="cuda")
        return layer


class FastRMSNorm(_ContractModule):
    """``turbodiffusion.ops.FastRMSNorm`` (ops/core.py:434-452): buffer ``weight`` fp32 [dim].  The reference up-casts
    (``rmsnorm(x.float(), w, eps).to(x.dtype)``); the kernel reads x's dtype, computes in fp32 and rounds once — the same value
    without the fp32 round trip through HBM."""

    def __init__(self, dim: int, eps: float = 1e-5):
        super().__init__()
        self.dim, self.eps = dim, eps
        self._declare(weight=((dim,), torch.float32, "ones"))

    def forward(self, x):
        return K.rmsnorm(x.contiguous(), self.weight, self.eps)

    @classmethod
    def from_rmsnorm(cls, original_rmsnorm):
        layer = cls(dim=original_rmsnorm.dim, eps=original_rmsnorm.eps)
        layer._adopt("weight", original_rmsnorm.weight, dtype=torch.float32)
        return layer


class FastLayerNorm(_ContractModule):
    """``turbodiffusion.ops.FastLayerNorm`` (ops/core.py:454-492): optional ``weight`` / ``bias`` buffers [dim].
    ``triton_variance`` (class default True) selects the reference Triton kernel's variance, see ``layernorm``."""

    triton_variance = True

    def __init__(self, dim: int, eps: float = 1e-5, elementwise_affine: bool = False, bias: bool = True):
        super().__init__()
        self.dim, self.eps, self.elementwise_affine = dim, eps, elementwise_affine
        self._declare(weight=((dim,), torch.get_default_dtype(), None) if elementwise_affine else None,
                      bias=((dim,), torch.get_default_dtype(), None) if (elementwise_affine and bias) else None)

    def forward(self, x):
        return layernorm(x, self.weight, self.bias, self.eps, self.elementwise_affine, self.triton_variance)

    @classmethod
    def from_layernorm(cls, original_layernorm):
        ln = original_layernorm
        layer = cls(dim=ln.normalized_shape[0], eps=ln.eps, elementwise_affine=ln.weight is not None, bias=ln.bias is not None)
        layer._adopt("weight", ln.weight)
        layer._adopt("bias", ln.bias)
        return layer
