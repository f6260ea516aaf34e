"""f4: the umT5-XXL **encoder** that turns prompt token ids into the [B, 512, 4096] text embedding the
DiT's cross-attention reads (``crossattn_emb``).

Reference: ``rcm/utils/umt5.py`` — ``T5Encoder`` (:308-337), ``T5SelfAttention`` (:217-238), ``T5Attention`` (:145-194,
no 1/sqrt(d) scaling, additive position bias, fp32 softmax), ``T5FeedForward`` (:197-214, gated tanh-GELU),
``T5LayerNorm`` (:131-142), ``T5RelativeEmbedding`` (:268-305, one table PER layer for umT5: ``shared_pos=False``),
``umt5_xxl`` (:451-465), ``UMT5EncoderModel.__call__`` (:501-521: rows past a prompt's length are returned as zeros).

What is different here, and why (results are the reference's — ``tests/test_vae_umt5_cpu.py`` pins this module to a live
``T5Encoder`` with random weights):
  * the reference pads every prompt to 512 tokens and runs all 512 rows through 24 layers, then zeroes the padding.
    Masked keys get weight exp(min - max) = 0 exactly and padded rows are discarded, so only the ``len`` valid rows are
    computed here (a typical prompt is 20-120 tokens: 4-25x less work) and the zeros are written once;
  * q | k | v are ONE [3 * dim_attn, dim] GEMM and gate | fc1 ONE [2 * dim_ffn, dim] GEMM (fatter launches; the weights are
    concatenated once at load time);
  * the relative-position bucket table is built once per length and shared by the layers (each layer keeps its own
    embedding, as umT5 does).
The rounding points of the reference's 16-bit path are kept: T5LayerNorm scales in fp32, casts, then multiplies by the
weight in the weight's dtype; scores are rounded before the bias is added; the softmax runs in fp32 and is cast back; the
GELU is the same chain of element-wise operations, each rounded.  Compute is hand-written HIP (round 4; csrc/gemm_bf16.hip):
every Linear is ``td_gemm_bf16`` (the residual adds in the o / fc2 epilogues, the gated GELU in the gate|fc1 epilogue, whose
weight rows are interleaved once at load time), the per-head score and value products are batched ``td_gemm_bf16`` calls on
strided views of the fused q|k|v output, ``td_softmax_rows`` adds the position bias (rounded, as the reference's 16-bit
add) and normalises in fp32, ``td_t5_norm`` is T5LayerNorm with its two roundings.  torch is left with the token-embedding
and position-table gathers (index plumbing).  HIP only: 16-bit on a GPU; the library-operator restatement the CPU pins run
is ``oracle/f4_ref.py``.  Tokenisation (umt5.py:58-98) is ``tokenizer.HuggingfaceTokenizer``; ``UMT5EncoderModel`` /
``get_umt5_embedding`` / ``clear_umt5_memory`` at the end of this file are the reference's entry points of the same names
(umt5.py:479-545) on top of both."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _fp16_clamp(x):
    if x.dtype == torch.float16 and torch.isinf(x).any():
        c = torch.finfo(x.dtype).max - 1000
        x = torch.clamp(x, min=-c, max=c)
    return x


def relative_buckets(L: int, num_buckets: int, max_dist: int, device) -> torch.Tensor:
    """[L, L] bucket ids of (key - query), bidirectional (T5RelativeEmbedding._relative_position_bucket, umt5.py:289-305)."""
    rel = torch.arange(L, device=device).unsqueeze(0) - torch.arange(L, device=device).unsqueeze(1)
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    rel = rel.abs()
    exact = nb // 2
    large = exact + (torch.log(rel.float() / exact) / math.log(max_dist / exact) * (nb - exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return out + torch.where(rel < exact, rel, large)


def synthetic_state_dict(layers=24, dim=4096, dim_ffn=10240, heads=64, vocab=256384, num_buckets=32, seed=0,
                         dtype=torch.bfloat16, device="cpu"):
    """Random-init encoder weights in the reference's key layout (``umt5_xxl`` defaults, umt5.py:451-465; per-layer position
    tables) for timing at full size and toy-size tests — generated on ``device`` (the XXL embedding alone is 2 GB)."""
    g = torch.Generator(device=device).manual_seed(seed)

    def w(o, i, std=None):
        return (torch.randn(o, i, device=device, generator=g) * (std if std is not None else i ** -0.5)).to(dtype)

    sd = {"token_embedding.weight": w(vocab, dim, 1.0), "norm.weight": torch.ones(dim, dtype=dtype, device=device)}
    for i in range(layers):
        p = f"blocks.{i}."
        sd[p + "norm1.weight"] = torch.ones(dim, dtype=dtype, device=device)
        sd[p + "norm2.weight"] = torch.ones(dim, dtype=dtype, device=device)
        sd[p + "attn.q.weight"] = w(dim, dim, (dim * dim) ** -0.5)       # init_weights, umt5.py:118-122
        sd[p + "attn.k.weight"], sd[p + "attn.v.weight"], sd[p + "attn.o.weight"] = w(dim, dim), w(dim, dim), w(dim, dim)
        sd[p + "ffn.gate.0.weight"], sd[p + "ffn.fc1.weight"], sd[p + "ffn.fc2.weight"] = w(dim_ffn, dim), w(dim_ffn, dim), w(dim, dim_ffn)
        sd[p + "pos_embedding.embedding.weight"] = w(num_buckets, heads, (2 * num_buckets * heads) ** -0.5)
    return sd


class Umt5Encoder:
    """``encoder(ids, mask)`` -> [B, L_pad, dim]: ``ids`` / ``mask`` [B, L_pad] as the reference's tokenizer returns them
    (padding on the right).  ``state_dict``: the reference encoder's (``models_t5_umt5-xxl-enc-bf16.pth`` layout); the
    configuration is read off the tensors.  16-bit on a GPU (the reference loads the bf16 checkpoint, umt5.py:478-488)."""

    def __init__(self, state_dict, dtype=torch.bfloat16, device="cuda", max_dist=128, eps=1e-6):
        self.dtype, self.device, self.max_dist, self.eps = dtype, torch.device(device), max_dist, eps
        if dtype not in (torch.bfloat16, torch.float16) or self.device.type != "cuda":
            raise ValueError("Umt5Encoder runs on the HIP kernels only: bf16 / fp16 on a GPU; the library-operator restatement "
                             "for CPU checks is oracle/f4_ref.py")
        from . import kernels as K_     # raises if the HIP library is missing
        self.K = K_

        def t(k):
            return state_dict[k].detach().to(device=self.device, dtype=dtype).contiguous()

        self.emb = t("token_embedding.weight")
        self.dim = self.emb.shape[1]
        n = 0
        while f"blocks.{n}.norm1.weight" in state_dict:
            n += 1
        if n == 0:
            raise ValueError("not a T5 encoder state dict: blocks.0.norm1.weight missing")
        self.shared_pos = "pos_embedding.embedding.weight" in state_dict
        self.pos = t("pos_embedding.embedding.weight") if self.shared_pos else None
        self.dim_ffn = state_dict["blocks.0.ffn.fc2.weight"].shape[1]
        if self.dim_ffn % 32:
            raise ValueError(f"dim_ffn = {self.dim_ffn}: the gated-GELU epilogue interleaves gate / fc1 in blocks of 32 columns")
        self.layers = []
        for i in range(n):
            p = f"blocks.{i}."
            lay = {
                "n1": t(p + "norm1.weight"), "n2": t(p + "norm2.weight"),
                "qkv": torch.cat([t(p + "attn.q.weight"), t(p + "attn.k.weight"), t(p + "attn.v.weight")], 0).contiguous(),
                "o": t(p + "attn.o.weight"),
                "gf": K_.geglu_interleave(t(p + "ffn.gate.0.weight"), t(p + "ffn.fc1.weight")),
                "fc2": t(p + "ffn.fc2.weight"),
                "pos": None if self.shared_pos else t(p + "pos_embedding.embedding.weight"),
            }
            self.layers.append(lay)
        self.final_norm = t("norm.weight")
        pos0 = self.pos if self.shared_pos else self.layers[0]["pos"]
        self.num_buckets, self.num_heads = pos0.shape
        self.dim_attn = self.layers[0]["o"].shape[1]
        assert self.dim_attn % self.num_heads == 0

    @classmethod
    def from_reference(cls, module_or_state_dict, **kw):
        sd = module_or_state_dict if isinstance(module_or_state_dict, dict) else module_or_state_dict.state_dict()
        return cls(sd, **kw)

    def _rows(self, ids):
        """ids [n] (one prompt's valid tokens) -> [n, dim]"""
        K = self.K
        n, H, c = ids.shape[0], self.num_heads, self.dim_attn // self.num_heads
        n64 = K.cdiv(n, 64) * 64
        x = F.embedding(ids, self.emb)                                             # gather (plumbing)
        buckets = relative_buckets(n, self.num_buckets, self.max_dist, self.device)

        def bias_of(table):   # [H, n, n] -> rows of the score matrix [H * n, n] (gather: plumbing)
            return F.embedding(buckets, table).permute(2, 0, 1).contiguous().view(H * n, n)

        shared = bias_of(self.pos) if self.shared_pos else None
        s_buf = torch.empty((H, n, n64), dtype=self.dtype, device=self.device)     # scores, then probabilities, in place
        vt = torch.zeros((H, c, n64), dtype=self.dtype, device=self.device)        # V^T, zero behind the n valid keys
        for lay in self.layers:
            bias = shared if shared is not None else bias_of(lay["pos"])
            qkv = K.gemm_bf16(K.t5_norm(x, lay["n1"], self.eps), lay["qkv"]).view(n, 3, H, c)
            q, k = qkv[:, 0].transpose(0, 1), qkv[:, 1].transpose(0, 1)            # [H, n, c] strided views
            vt[:, :, :n] = qkv[:, 2].permute(1, 2, 0)
            s = K.gemm_bf16_batched(q, k, out=s_buf[:, :, :n])                     # no 1/sqrt(d) scaling (umt5.py:183)
            K.softmax_rows(s_buf.view(H * n, n64)[:, :n], 1.0, bias=bias, out=s_buf.view(H * n, n64)[:, :n])
            o = K.gemm_bf16_batched(s_buf, vt)                                     # [H, n, c]  (k = n64: zero columns x zero rows)
            o = o.transpose(0, 1).reshape(n, H * c)
            x = _fp16_clamp(K.gemm_bf16(o, lay["o"], res=x))                       # x + o(attn)
            h = K.gemm_bf16(K.t5_norm(x, lay["n2"], self.eps), lay["gf"], epilogue="geglu")   # fc1(x) * gelu(gate(x)), umt5.py:210
            x = _fp16_clamp(K.gemm_bf16(h, lay["fc2"], res=x))
        return K.t5_norm(x, self.final_norm, self.eps)

    @torch.no_grad()
    def __call__(self, ids, mask=None):
        ids = ids.to(self.device)
        B, Lp = ids.shape
        lens = [Lp] * B if mask is None else mask.to(self.device).gt(0).sum(dim=1).tolist()
        out = torch.zeros(B, Lp, self.dim, dtype=self.dtype, device=self.device)
        for b, n in enumerate(lens):
            if mask is not None and n > 0 and not bool(mask[b, :n].to(self.device).gt(0).all()):
                raise ValueError("mask must be right-padded (valid tokens first), as the reference's tokenizer produces it")
            if n > 0:
                out[b, :n] = self._rows(ids[b, :n])
        return out


class UMT5EncoderModel:
    """The reference's ``UMT5EncoderModel`` (umt5.py:479-521): checkpoint + tokenizer -> ``model(texts)`` = [B, text_len, 4096]
    with the rows past each prompt's length zero.  ``checkpoint_path``: the reference's ``models_t5_umt5-xxl-enc-bf16.pth`` (a
    plain state dict) or a state dict already in memory; ``tokenizer_path``: a local ``tokenizer.json`` / directory
    (``tokenizer.HuggingfaceTokenizer``; the reference's default is the hub id, there is no network here)."""

    def __init__(self, text_len=512, dtype=torch.bfloat16, device="cuda", checkpoint_path="models_t5_umt5-xxl-enc-bf16.pth",
                 tokenizer_path="google/umt5-xxl"):
        from .tokenizer import HuggingfaceTokenizer
        self.text_len, self.dtype, self.device = text_len, dtype, device
        if isinstance(checkpoint_path, dict):
            sd = checkpoint_path
        else:
            assert str(checkpoint_path).endswith(".pth")                       # umt5.py:496
            sd = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
        self.model = Umt5Encoder(sd, dtype=dtype, device=device)
        self.tokenizer = HuggingfaceTokenizer(name=tokenizer_path, seq_len=text_len, clean="whitespace")

    def __call__(self, texts, device=None):
        ids, mask = self.tokenizer(texts, return_mask=True, add_special_tokens=True)
        out = self.model(ids, mask)              # valid rows computed, padding rows written as zeros (umt5.py:510-521)
        return out if device is None else out.to(device)


t5_encoder = None


def get_umt5_embedding(checkpoint_path, prompts, device="cuda", max_length=512, tokenizer_path="google/umt5-xxl"):
    """umt5.py:524-533: one process-wide encoder, built on first use"""
    global t5_encoder
    if t5_encoder is None:
        t5_encoder = UMT5EncoderModel(text_len=max_length, device=device, checkpoint_path=checkpoint_path,
                                      tokenizer_path=tokenizer_path)
    return t5_encoder(prompts, device=device)


def clear_umt5_memory():
    """umt5.py:536-545"""
    global t5_encoder
    if t5_encoder is not None:
        t5_encoder = None
    import gc
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
