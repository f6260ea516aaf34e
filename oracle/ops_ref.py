"""CPU restatement of the reference's linear / norm / glue operators.

TEST INFRASTRUCTURE (see oracle/__init__.py).  torch-CPU, IEEE fp32.
"""
from __future__ import annotations

import torch

BLOCK = 128  # turbodiffusion/ops/quant/quant.hpp:38 (BlockSize), gemm tile K


def _cdiv(a: int, b: int) -> int:
    return (a + b - 1) // b


# --------------------------------------------------------------------------- #
# a16  block-128 INT8 quantiser
# --------------------------------------------------------------------------- #
def quant_block128(x: torch.Tensor):
    """Per-128x128-block symmetric INT8 quantisation.

    Follows turbodiffusion/ops/quant/quant.hpp:91-98 (amax -> q, scale),
    :122-154 (amax = max(1e-8, max|x|) over the block), common/load.hpp:24-47
    (out-of-range rows/cols of tail blocks are zero-filled, so a tail block's
    amax covers only valid elements) and quant.hpp:48 (int8_max = 128.f):

        mult  = 128.f / amax            (fp32 division)
        q     = sat_s8(rne(x * mult))   (cutlass NumericConverter RNE, saturating)
        scale = amax / 128.f

    x: [m, n] fp16/bf16 (or fp32 holding such values).  Returns (q int8 [m, n],
    scale fp32 [ceil(m/128), ceil(n/128)]).
    """
    assert x.dim() == 2
    m, n = x.shape
    mb, nb = _cdiv(m, BLOCK), _cdiv(n, BLOCK)
    xf = torch.zeros(mb * BLOCK, nb * BLOCK, dtype=torch.float32)
    xf[:m, :n] = x.float()
    blk = xf.view(mb, BLOCK, nb, BLOCK).permute(0, 2, 1, 3)  # [mb, nb, 128, 128]
    amax = blk.abs().amax(dim=(2, 3)).clamp_min(1e-8)  # fp32
    mult = (torch.tensor(128.0, dtype=torch.float32) / amax)  # fp32 division
    q = torch.round(blk * mult[:, :, None, None])  # RNE
    q = q.clamp_(-128, 127).to(torch.int8)
    q = q.permute(0, 2, 1, 3).reshape(mb * BLOCK, nb * BLOCK)[:m, :n].contiguous()
    scale = amax / torch.tensor(128.0, dtype=torch.float32)
    return q, scale.contiguous()


# --------------------------------------------------------------------------- #
# a17  block-scaled W8A8 GEMM
# --------------------------------------------------------------------------- #
def gemm_w8a8(a_q, a_s, b_q, b_s, out_dtype=torch.bfloat16, bias=None, gelu_tanh=False):
    """D[m,n] = sum_kb fma(float(sum_{k in kb} A[m,k] B[n,k]), AS[m/128,kb]*BS[n/128,kb], D).

    Follows turbodiffusion/ops/gemm/kernel.hpp:390-427 (int32 tile accumulate per
    128-deep K block, then dequant into an fp32 accumulator, K blocks ascending) and
    gemm/utils.hpp:116-121 (``acc += __int2float_rn(i32) * (sa*sb)`` — nvcc contracts
    this to one FMA; ``sa*sb`` is formed first in fp32, kernel.hpp:418).  The fp32
    accumulator is cast to the output dtype with RNE (kernel.hpp:445-518).

    ``bias`` (same dtype as out) reproduces Int8Linear.forward (ops/core.py:408-412):
    the GEMM result is first rounded to out_dtype, *then* bias is added in that dtype.
    ``gelu_tanh`` additionally applies nn.GELU(approximate='tanh') on the rounded
    output (the FFN's activation, wan2pt1.py:375), evaluated in fp32 and rounded.
    """
    m, k = a_q.shape
    n = b_q.shape[0]
    assert k % BLOCK == 0, "reference can_implement: k % 128 == 0 (kernel.hpp:181-186)"
    kb_n = k // BLOCK
    acc = torch.zeros(m, n, dtype=torch.float64)
    row_blk = torch.arange(m) // BLOCK
    col_blk = torch.arange(n) // BLOCK
    af = a_q.float()
    bf = b_q.float()
    for kb in range(kb_n):
        sl = slice(kb * BLOCK, (kb + 1) * BLOCK)
        # |sum| <= 128*128*128 < 2^24: exact in fp32
        i32 = af[:, sl] @ bf[:, sl].t()
        s = (a_s[row_blk, kb][:, None] * b_s[col_blk, kb][None, :])  # fp32 product
        # fma(i32, s, acc): product exact in fp64, one fp32 rounding of the sum
        acc = (acc + i32.double() * s.double()).float().double()
    out = acc.float().to(out_dtype)
    if bias is not None:
        out = out + bias.to(out_dtype)
    if gelu_tanh:
        out = torch.nn.functional.gelu(out.float(), approximate="tanh").to(out_dtype)
    return out


def int8_linear(x, w_q, w_s, bias=None, gelu_tanh=False):
    """ops/core.py:28-57 + Int8Linear.forward :408-412."""
    shape = x.shape
    x2 = x.reshape(-1, shape[-1])
    x_q, x_s = quant_block128(x2)
    y = gemm_w8a8(x_q, x_s, w_q, w_s, out_dtype=x.dtype, bias=bias, gelu_tanh=gelu_tanh)
    return y.reshape(*shape[:-1], w_q.shape[0])


# --------------------------------------------------------------------------- #
# a5 / a6  Fast norms (the Triton kernels of ops/core.py)
# --------------------------------------------------------------------------- #
def rmsnorm_fast(x, w, eps):
    """FastRMSNorm.forward (ops/core.py:441-442) -> rmsnorm (:139-191) -> Triton
    _rms_norm_fwd_fused (:96-136): fp32 in, var = sum(x*x)/N, rstd = 1/sqrt(var+eps),
    y = (x*rstd)*w in fp32, then ``.to(x.dtype)``."""
    xf = x.float()
    var = (xf * xf).sum(-1, keepdim=True) / xf.shape[-1]
    rstd = 1.0 / torch.sqrt(var + eps)
    y = (xf * rstd) * w.float()
    return y.to(x.dtype)


def layernorm_fast(x, w, b, eps, triton_variance=True):
    """FastLayerNorm.forward (ops/core.py:477-478) -> layernorm (:380-386) -> Triton
    _layer_norm_{param,noparam}_fwd_fused (:193-242, :293-335): two-pass mean/var in
    fp32, (x-mean)*rstd [*w + b], fp32 out, then ``.to(x.dtype)``.

    ``triton_variance``: those kernels load N2 = next_power_of_2(N) columns with the masked ones as 0.0
    (``tl.load(..., mask=mask[None, :], other=0.0)``, :213 / :313) and form ``_var = (x - mean) * (x - mean)`` over ALL
    N2 columns (:219-220 / :319-320): every phantom column contributes mean^2, so
        var = (sum_valid (x-mean)^2 + (N2 - N) * mean^2) / N
    — not the textbook variance unless N is a power of two or the row mean is 0.  PINNED: this restatement agrees with the
    reference kernels themselves, executed on the MI355X under Triton-ROCm, to 1e-7 (tests/golden/triton_leaves.pt,
    tests/test_oracle_cpu.py::test_oracle_norms_match_the_reference_triton_kernels); without the term the difference is
    1e-3.  False: the textbook form (the CUDA twin ops/norm/layernorm.hpp:68-77)."""
    xf = x.float()
    n = xf.shape[-1]
    mean = xf.sum(-1, keepdim=True) / n
    d = xf - mean
    ss = (d * d).sum(-1, keepdim=True)
    if triton_variance:
        n2 = 1 << (n - 1).bit_length()
        ss = ss + (n2 - n) * (mean * mean)
    var = ss / n
    rstd = 1.0 / torch.sqrt(var + eps)
    y = d * rstd
    if w is not None:
        y = y * w.float() + b.float()
    return y.to(x.dtype)


def rmsnorm_eager(x, w, eps):
    """WanRMSNorm.forward (rcm/networks/wan2pt1.py:191-199): fp32 norm, cast to x
    dtype, then multiply by the (fp32 parameter) weight — type promotion makes the
    product fp32 unless the module was cast to bf16, in which case it is bf16."""
    xf = x.float()
    y = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).type_as(x)
    return y * w


def layernorm_eager(x, w, b, eps):
    """WanLayerNorm.forward (wan2pt1.py:206-212): F.layer_norm in fp32, type_as(x)."""
    xf = x.float()
    y = torch.nn.functional.layer_norm(
        xf, (xf.shape[-1],), None if w is None else w.float(), None if b is None else b.float(), eps
    )
    return y.type_as(x)


# --------------------------------------------------------------------------- #
# a7  AdaLN glue
# --------------------------------------------------------------------------- #
def modulate(xn, scale, shift):
    """(norm(x).float() * (1 + e_scale) + e_shift).type_as(x)  — wan2pt1.py:404,411.
    ``xn`` is the *already rounded* (bf16) norm output; scale/shift fp32 [.., dim]."""
    return (xn.float() * (1 + scale.float()) + shift.float()).type_as(xn)


def gated_residual(x, y, gate):
    """x + y * gate.type_as(x)  — wan2pt1.py:405-406,412-413 (elementwise ops are not
    autocast targets: bf16*bf16 rounds to bf16, then the bf16 add rounds again)."""
    return x + y * gate.type_as(x)


# --------------------------------------------------------------------------- #
# a8  RoPE
# --------------------------------------------------------------------------- #
def rope_freqs(T, H, W, head_dim, theta=10000.0):
    """VideoRopePosition3DEmb.generate_embeddings (wan2pt1.py:86-137) with the default
    extrapolation ratios (ntk factors = 1): angles [T*H*W, head_dim/2] fp32,
    concatenated as (t: dim_t/2, h: dim_h/2, w: dim_w/2)."""
    dim_h = head_dim // 6 * 2
    dim_w = dim_h
    dim_t = head_dim - 2 * dim_h
    seq = torch.arange(max(T, H, W)).float()
    sp = torch.arange(0, dim_h, 2)[: dim_h // 2].float() / dim_h
    tp = torch.arange(0, dim_t, 2)[: dim_t // 2].float() / dim_t
    h_f = 1.0 / (theta ** sp)
    w_f = 1.0 / (theta ** sp)
    t_f = 1.0 / (theta ** tp)
    fh = torch.outer(seq[:H], h_f)
    fw = torch.outer(seq[:W], w_f)
    ft = torch.outer(seq[:T], t_f)
    f = torch.cat(
        [
            ft[:, None, None, :].expand(T, H, W, -1),
            fh[None, :, None, :].expand(T, H, W, -1),
            fw[None, None, :, :].expand(T, H, W, -1),
        ],
        dim=-1,
    )
    return f.reshape(T * H * W, -1).float().contiguous()


def rope_apply(x, freqs):
    """rope_apply (wan2pt1.py:156-178) with flash-attn's interleaved rotary semantics
    (pairs (x[2i], x[2i+1]) -> (x0*c - x1*s, x0*s + x1*c)), fp32 math, cast back.
    x: [B, L, H, D]; freqs: [L, D/2] fp32 angles."""
    b, l, h, d = x.shape
    cos = torch.cos(freqs).float()[None, :, None, :]
    sin = torch.sin(freqs).float()[None, :, None, :]
    xf = x.float().reshape(b, l, h, d // 2, 2)
    x0, x1 = xf[..., 0], xf[..., 1]
    o0 = x0 * cos - x1 * sin
    o1 = x0 * sin + x1 * cos
    return torch.stack([o0, o1], dim=-1).reshape(b, l, h, d).to(x.dtype)


def sinusoidal_embedding_1d(dim, position):
    """wan2pt1.py:144-153 (fp64)."""
    half = dim // 2
    position = position.type(torch.float64)
    sinusoid = torch.outer(position, torch.pow(10000, -torch.arange(half).to(position).div(half)))
    return torch.cat([torch.cos(sinusoid), torch.sin(sinusoid)], dim=1)
