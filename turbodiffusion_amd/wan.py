"""MI355X-native Wan DiT forward — the reference's ``WanModel.forward`` surface
(``rcm/networks/wan2pt1.py:598-721``; ``wan2pt2.py`` = same + ``y`` concatenated on channels and a
plain text cross-attention) re-built as a fused HIP kernel pipeline.

Drop-in properties kept:
  * constructor arguments and ``forward(x_B_C_T_H_W, timesteps_B_T, crossattn_emb,
    frame_cond_crossattn_emb_B_L_D=None, y_B_C_T_H_W=None, **kw)`` -> ``[B, C_out, T, H, W]`` velocity,
    so the reference sampler loop (``inference/wan2.1_t2v_infer.py:129-139``) can call it unchanged;
  * module tree / state-dict keys identical to the reference model *after* ``modify_model.replace_*``
    (``blocks.N.self_attn.q.int8_weight|scale|bias``, ``...norm_q.weight``,
    ``...self_attn.attn_op.local_attn.proj_l.*``, ``blocks.N.modulation``, ``head.head.*`` ...), so
    published ``*-quant.pth`` checkpoints load with ``load_state_dict``.

What is different underneath (one pass per arrow, all hand-written HIP unless marked torch):
  x --LayerNorm+AdaLN-modulate--> h --quant--> int8 --W8A8 GEMM (q|k|v fused, bias in epilogue)-->
  qkv --RMSNorm(dim)+RoPE+head-major--> q,k ; v --transposed MFMA tiles--> vt ;
  smooth-K mean, block pool + INT8 quant, top-k LUT, block-sparse Sage attention, linear branch (+=),
  --quant + W8A8 GEMM(o)--> gated residual ; norm3 -> cross-attention (dense, 512 keys) -> residual ;
  LayerNorm+modulate -> W8A8 GEMM (+bias+GELU-tanh epilogue) -> W8A8 GEMM -> gated residual.
Embeddings and the head (outside ``blocks``; not quantised by the reference either,
``modify_model.py:63``) are plain library GEMMs through torch.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn

from . import kernels as K
from .ops import FastLayerNorm, FastRMSNorm, Int8Linear
from .sla import SageSparseLinearAttention, SparseLinearAttention, sagesla_split_projection, sparse_linear_attention_hld

ATTENTION_TYPES = ("original", "sage", "sla", "sagesla")


class AttnOp(nn.Module):
    """Stands in for ``MinimalA2AAttnOp`` (rcm/utils/a2a_cp.py:189-200): only its ``local_attn``
    attribute is part of the contract (``modify_model.py:50-52``)."""

    def __init__(self, local_attn: Optional[nn.Module] = None):
        super().__init__()
        if local_attn is not None:
            self.local_attn = local_attn
        self.pg = None
        self.stream = None

    def set_context_parallel_group(self, process_group, ranks=None, stream=None):
        """a2a_cp.py:184-196: remembered only — the exchange itself is the model's (``WanModel.enable_context_parallel``
        -> seqpar: one packed all-gather of the K side per layer instead of the reference's four all-to-alls)."""
        del ranks
        self.pg, self.stream = process_group, stream


def _linear(in_f, out_f, quant, dtype):
    if quant:
        return Int8Linear(in_f, out_f, bias=True, dtype=dtype)
    return nn.Linear(in_f, out_f, dtype=dtype)


class WanSelfAttention(nn.Module):
    def __init__(self, dim, num_heads, eps, quant, dtype, attention_type=None, sla_topk=0.1, image_branch=False):
        super().__init__()
        self.dim, self.num_heads, self.head_dim, self.eps = dim, num_heads, dim // num_heads, eps
        self.q = _linear(dim, dim, quant, dtype)
        self.k = _linear(dim, dim, quant, dtype)
        self.v = _linear(dim, dim, quant, dtype)
        self.o = _linear(dim, dim, quant, dtype)
        self.norm_q = FastRMSNorm(dim, eps=eps)
        self.norm_k = FastRMSNorm(dim, eps=eps)
        local = None
        if attention_type == "sla":
            local = SparseLinearAttention(self.head_dim, sla_topk, BLKQ=128, BLKK=64)
        elif attention_type == "sagesla":
            local = SageSparseLinearAttention(self.head_dim, sla_topk)
        self.attn_op = AttnOp(local)
        if image_branch:   # WanI2VCrossAttention (wan2pt1.py:303-313): a second K / V pair for the CLIP image tokens
            self.k_img = _linear(dim, dim, quant, dtype)
            self.v_img = _linear(dim, dim, quant, dtype)
            self.norm_k_img = FastRMSNorm(dim, eps=eps)
            self.attn_op_image = AttnOp(None)


class MLPProj(nn.Module):
    """wan2pt1.py:457-486 (without the first-last-frame position table): LayerNorm, Linear, exact GELU, Linear, LayerNorm on the
    CLIP image tokens; state-dict keys ``img_emb.proj.{0,1,3,4}.*``."""

    def __init__(self, in_dim, out_dim, dtype):
        super().__init__()
        self.proj = nn.Sequential(nn.LayerNorm(in_dim, dtype=dtype), nn.Linear(in_dim, in_dim, dtype=dtype), nn.GELU(),
                                  nn.Linear(in_dim, out_dim, dtype=dtype), nn.LayerNorm(out_dim, dtype=dtype))


class WanAttentionBlock(nn.Module):
    def __init__(self, dim, ffn_dim, num_heads, cross_attn_norm, eps, quant, dtype, attention_type, sla_topk, image_branch=False):
        super().__init__()
        self.norm1 = FastLayerNorm(dim, eps)
        self.self_attn = WanSelfAttention(dim, num_heads, eps, quant, dtype, attention_type, sla_topk)
        self.norm3 = FastLayerNorm(dim, eps, elementwise_affine=True) if cross_attn_norm else nn.Identity()
        self.cross_attn = WanSelfAttention(dim, num_heads, eps, quant, dtype, image_branch=image_branch)
        self.norm2 = FastLayerNorm(dim, eps)
        self.ffn = nn.Sequential(_linear(dim, ffn_dim, quant, dtype), nn.GELU(approximate="tanh"),
                                 _linear(ffn_dim, dim, quant, dtype))
        self.modulation = nn.Parameter(torch.randn(1, 6, dim) / dim ** 0.5)


class Head(nn.Module):
    def __init__(self, dim, out_dim, patch_size, eps, dtype):
        super().__init__()
        self.norm = FastLayerNorm(dim, eps)
        self.head = nn.Linear(dim, math.prod(patch_size) * out_dim, dtype=dtype)
        self.modulation = nn.Parameter(torch.randn(1, 2, dim) / dim ** 0.5)


def sinusoidal_embedding_1d(dim, position):
    """wan2pt1.py:144-153 (fp64)."""
    half = dim // 2
    position = position.type(torch.float64)
    sinusoid = torch.outer(position, torch.pow(10000, -torch.arange(half, device=position.device).to(position).div(half)))
    return torch.cat([torch.cos(sinusoid), torch.sin(sinusoid)], dim=1)


def rope_angles(T, H, W, head_dim, device, theta=10000.0):
    """VideoRopePosition3DEmb.generate_embeddings (wan2pt1.py:86-137), default ntk factors."""
    dim_h = head_dim // 6 * 2
    dim_t = head_dim - 2 * dim_h
    seq = torch.arange(max(T, H, W), device=device).float()
    sp = torch.arange(0, dim_h, 2, device=device)[: dim_h // 2].float() / dim_h
    tp = torch.arange(0, dim_t, 2, device=device)[: dim_t // 2].float() / dim_t
    fh = torch.outer(seq[:H], 1.0 / (theta ** sp))
    fw = torch.outer(seq[:W], 1.0 / (theta ** sp))
    ft = torch.outer(seq[:T], 1.0 / (theta ** tp))
    f = torch.cat([ft[:, None, None, :].expand(T, H, W, -1), fh[None, :, None, :].expand(T, H, W, -1),
                   fw[None, None, :, :].expand(T, H, W, -1)], dim=-1)
    return f.reshape(T * H * W, -1).float().contiguous()


class WanModel(nn.Module):
    """Wan2.1 T2V (1.3B / 14B) and Wan2.2 A14B I2V (``y`` concatenated on channels) denoiser."""

    def __init__(self, model_type="t2v", patch_size=(1, 2, 2), text_len=512, in_dim=16, dim=2048, ffn_dim=8192,
                 freq_dim=256, text_dim=4096, out_dim=16, num_heads=16, num_layers=32, qk_norm=True,
                 cross_attn_norm=True, eps=1e-6, attention_type="sagesla", sla_topk=0.1, quant_linear=True,
                 dtype=torch.bfloat16, default_norm=None, clip_dim=None, **_unused):
        super().__init__()
        assert model_type in ("t2v", "i2v") and qk_norm
        assert attention_type in ATTENTION_TYPES
        if dim // num_heads != 128 or dim % num_heads:
            raise ValueError(f"dim={dim}, num_heads={num_heads}: the MI355X attention kernels are built for head_dim 128")
        self.model_type, self.patch_size, self.text_len = model_type, patch_size, text_len
        self.in_dim, self.dim, self.ffn_dim, self.freq_dim = in_dim, dim, ffn_dim, freq_dim
        self.text_dim, self.out_dim, self.num_heads, self.num_layers = text_dim, out_dim, num_heads, num_layers
        self.eps, self.attention_type, self.sla_topk, self.quant_linear = eps, attention_type, sla_topk, quant_linear
        self.dtype = dtype
        # The reference's ``--default_norm`` (wan2.1_t2v_infer.py:53; ``replace_norm = not default_norm``, modify_model.py:
        # 56-81): True = the eager WanLayerNorm arithmetic in the blocks (textbook variance); False = FastLayerNorm, whose
        # Triton kernel sums (x - mean)^2 over next_power_of_2(dim) columns (ops/core.py:213-224; K.triton_ln_pad_cols) —
        # reproduced so that this model returns what the reference's accelerated path returns.  None: False whenever an
        # accelerated operator is selected, True for the plain eager-equivalent configuration (original attention, bf16
        # linears).  The head's norm is never replaced by the reference (only ``model.blocks`` is walked).
        self.default_norm = (attention_type == "original" and not quant_linear) if default_norm is None else bool(default_norm)
        self.patch_embedding = nn.Linear(in_dim * math.prod(patch_size), dim, dtype=dtype)
        self.text_embedding = nn.Sequential(nn.Linear(text_dim, dim, dtype=dtype), nn.GELU(approximate="tanh"),
                                            nn.Linear(dim, dim, dtype=dtype))
        self.time_embedding = nn.Sequential(nn.Linear(freq_dim, dim, dtype=dtype), nn.SiLU(),
                                            nn.Linear(dim, dim, dtype=dtype))
        self.time_projection = nn.Sequential(nn.SiLU(), nn.Linear(dim, dim * 6, dtype=dtype))
        self.blocks = nn.ModuleList([
            WanAttentionBlock(dim, ffn_dim, num_heads, cross_attn_norm, eps, quant_linear, dtype, attention_type,
                              sla_topk, image_branch=bool(clip_dim)) for _ in range(num_layers)])
        self.head = Head(dim, out_dim, patch_size, eps, dtype)
        # Wan2.1 I2V (rcm/networks/wan2pt1.py with model_type "i2v": :575, :590-591): CLIP image tokens [B, 257, clip_dim = 1280]
        # arrive as ``frame_cond_crossattn_emb_B_L_D``, go through ``img_emb`` and are attended by a second K / V pair of
        # every block's cross-attention.  Wan2.2-A14B I2V (wan2pt2.py) has no such branch: clip_dim = None.
        self.clip_dim = clip_dim
        if clip_dim:
            self.img_emb = MLPProj(clip_dim, dim, dtype)
        self._fused = {}
        self._rope_cache = {}
        self.seq_parallel = None  # set by turbodiffusion_amd.seqpar.enable(...)
        self.fuse_norm_quant = True
        self.fuse_cross_q_norm = True
        self.batch_text_kv = True
        self._ckv_all = None
        self.fuse_row_stats = True  # LayerNorm / cross-q RMSNorm row statistics from the producing GEMM's epilogue
        self.fuse_vt = True         # self-attention V leaves the q|k|v GEMM as the attention kernel's V^T tiles (K.gemm_w8a8_vt)
        self.fuse_stats_finalize = True   # the row-statistics finalisers inside their consumers where that is possible (round 6)
        self.two_streams = True     # SageSLA self-attention: Q-side chain on a second stream beside the K-side chain (sla.py)
        self.split_qkv = True       # ... and the q|k|v projection as K|V then Q, the K-side chain under the Q GEMM (round 5)
        self.split_tokens = True    # everything after self-attention is token-local: two token halves on two streams (_block)
        self._side_streams = {}
        self.sage_pv = "fp16"      # "fp8": SageAttention's FP8-PV variant for self-attention (the reference's sm89+ branch)
        self.cache_text_kv = True  # cross-attention K / V^T of the text are a function of the text only: once per video
        self._text_states = {}     # data_ptr -> (key, source tensor, context, [per batch entry: [per block: (k, vt)]], generation)
        self._text_gen = 0         # bumped whenever a text entry gets NEW persistent buffers (captured graphs point into them)
        self._weights_epoch = 0   # bumped whenever derived weight copies are dropped (GraphedModel re-captures on a change)
        self.register_load_state_dict_post_hook(lambda module, incompatible_keys: module.invalidate_caches())

    def __getstate__(self):
        """copy.deepcopy / pickle: HIP stream objects and cached derived tensors do not travel."""
        d = self.__dict__.copy()
        d["_side_streams"] = {}
        d["_text_states"] = {}
        return d

    # ------------------------------------------------------------------ context (sequence) parallelism: the reference's hooks
    def enable_context_parallel(self, process_group=None):
        """``WanModel.enable_context_parallel`` (rcm/networks/wan2pt1.py:786-792): shard the flattened (t h w) token axis over
        the ranks of ``process_group``.  Same hook name and the same observable behaviour — inputs are broadcast from the
        group's first rank (wan2pt1.py:627-636), every rank returns the full ``[B, C, T, H, W]`` output (``cat_outputs_cp``,
        :703-707) — with this repo's exchange underneath (``seqpar``: per self-attention layer one packed RCCL all-gather of
        the rank's quantised K side; no ``H % cp`` or ``L % cp`` constraint, where the reference's Ulysses all-to-all needs
        both, a2a_cp.py:49-51, wan2pt1.py:663)."""
        from . import seqpar
        seqpar.enable(self, process_group, broadcast_inputs=True)
        for blk in self.blocks:
            blk.self_attn.attn_op.set_context_parallel_group(process_group, None, None)
        return self

    def disable_context_parallel(self):
        """wan2pt1.py:774-784."""
        from . import seqpar
        seqpar.disable(self)
        for blk in self.blocks:
            blk.self_attn.attn_op.set_context_parallel_group(None, None, None)
        return self

    @property
    def is_context_parallel_enabled(self):
        """wan2pt1.py:794-796."""
        return self.seq_parallel is not None

    # ------------------------------------------------------------------ derived weight copies
    def invalidate_caches(self):
        """Forget every tensor derived from the weights (the q|k|v / cross k|v concatenations, the all-blocks text K|V
        weight, proj_l copies) and make ``GraphedModel`` re-capture.  Called automatically after ``load_state_dict`` and
        after ``.to()`` / ``.cuda()`` / ``.half()`` ...; call it yourself after replacing a weight tensor by assignment.
        (In-place updates need nothing: the per-module weights are VIEWS of the concatenations, see ``_fused_weights``.)"""
        self._fused.clear()
        self._ckv_all = None
        self._text_states = {}
        self._weights_epoch += 1

    _FP32_BUFFERS = ("weight", "bias", "scale")

    def _apply(self, fn, recurse=True):
        """``net.to(torch.bfloat16)`` / ``.half()`` would also convert the fp32 buffers the HIP kernels read as fp32
        (FastRMSNorm / FastLayerNorm weights, Int8Linear scales — the hazard wan2pt1.py warns about): keep those fp32
        (device moves still apply), and drop the derived copies."""
        keep = []
        for m in self.modules():
            if isinstance(m, (FastRMSNorm, FastLayerNorm, Int8Linear)):
                for name in self._FP32_BUFFERS:
                    t = m._buffers.get(name)
                    if t is not None and t.dtype == torch.float32:
                        keep.append((m, name, t))
        out = super()._apply(fn, recurse)
        for m, name, t in keep:
            new = m._buffers.get(name)
            if new is not None and new.dtype != torch.float32:
                m._buffers[name] = t.to(device=new.device)
        self.invalidate_caches()
        return out

    # ------------------------------------------------------------------ weights
    @torch.no_grad()
    def load_from_float_state_dict(self, sd: dict):
        """Take an UNQUANTISED reference state dict (keys of rcm.networks.wan2pt1.WanModel) and do what
        ``modify_model.py:156-183`` does offline: per-128x128-block INT8 quantisation of every Linear
        inside ``blocks`` except ``proj_l`` — with the HIP quantiser."""
        dev = next(self.parameters()).device
        own = self.state_dict()
        new = {}
        for k, v in sd.items():
            if k in own:
                new[k] = v
                continue
            if k.endswith(".weight") and k[:-7] + ".int8_weight" in own:
                w = v.to(dev)
                if w.dtype == torch.float32:
                    w = w.to(self.dtype)
                q, s = K.quant_i8_block128(w.contiguous())
                new[k[:-7] + ".int8_weight"] = q
                new[k[:-7] + ".scale"] = s
                continue
            raise KeyError(f"unexpected key {k}")
        missing = [k for k in own if k not in new]
        if missing:
            raise KeyError(f"missing keys {missing[:5]} ...")
        self.load_state_dict(new, assign=False)   # (the post-hook drops the derived copies)

    def _lin(self, mod, x, gelu=False):
        """One Linear of a block on a [M, K] activation: W8A8 (HIP) or the plain bf16 library GEMM."""
        if isinstance(mod, Int8Linear):
            xq, xs = K.quant_i8_block128(x)
            return K.gemm_w8a8(xq, xs, mod.int8_weight, mod.scale, x.dtype, bias=mod.bias, gelu_tanh=gelu)
        return self._lin16(x, mod.weight, mod.bias, gelu)

    def _lin16(self, x, w, b, gelu=False):
        """A plain 16-bit Linear (BASELINE config 3's "bf16 linears", the text MLP, C1's arithmetic on the GPU) on
        ``td_gemm_bf16`` — bias and GELU-tanh in the GEMM's epilogue with the operator sequence's rounding points.  Any
        width (``K.gemm_bf16`` zero-pads a reduction that is not a multiple of 64: toy models); there is no library-GEMM
        path beside it — what the kernel does not take raises."""
        if x.dtype not in (torch.bfloat16, torch.float16) or w.dtype != x.dtype or (b is not None and b.dtype != x.dtype):
            raise TypeError(f"plain Linear on the MI355X path: activation {x.dtype}, weight {w.dtype}, bias "
                            f"{None if b is None else b.dtype} — td_gemm_bf16 takes one 16-bit dtype (bfloat16 or float16) for all "
                            "three; build the model with dtype=torch.bfloat16 (fp32 models are the CPU oracle's, oracle/wan_ref.py)")
        if w.shape[0] % 8:
            raise ValueError(f"plain Linear with {w.shape[0]} output features: td_gemm_bf16 stores 16-byte row pieces (a multiple of 8)")
        x2 = x.reshape(-1, x.shape[-1])
        y = K.gemm_bf16(x2, w.detach(), None if b is None else b.detach(), epilogue="gelu_tanh" if gelu else "none")
        return y.view(*x.shape[:-1], w.shape[0])

    def _lin_q(self, mod, xq, xs, dtype, gelu=False):
        """Int8Linear on an already block-quantised activation."""
        return K.gemm_w8a8(xq, xs, mod.int8_weight, mod.scale, dtype, bias=mod.bias, gelu_tanh=gelu)

    def _ffn_q(self, lin1, lin2, xq, xs, dtype):
        hq, hs = K.gemm_w8a8_quant(xq, xs, lin1.int8_weight, lin1.scale, dtype, bias=lin1.bias, gelu_tanh=True)
        return K.gemm_w8a8(hq, hs, lin2.int8_weight, lin2.scale, dtype, bias=lin2.bias)

    def _residual_lin_(self, x2, mod, y, gate, stats=False):
        """x2 += Linear(y) * gate.type_as(x2)  (gate fp32 [B, dim] or None), in place (wan2pt1.py:405-406,412-413).
        W8A8 and one batch entry: the GEMM's epilogue applies the residual (same bits, one pass less over [L, dim]).
        stats: also return the LayerNorm statistics (mean, rstd) [L, 2] of the updated rows, from the same epilogue."""
        if isinstance(mod, Int8Linear) and (gate is None or gate.shape[0] == 1):
            yq, ys = y if isinstance(y, tuple) else K.quant_i8_block128(y)
            if stats:
                _, ws = K.gemm_w8a8_stats(yq, ys, mod.int8_weight, mod.scale, mod.bias, x=x2, gate=gate)
                return K.row_stats_finalize(ws, x2.shape[1], self.eps, pad_cols=self._ln_pad)
            K.gemm_w8a8_residual_(x2, yq, ys, mod.int8_weight, mod.scale, bias=mod.bias, gate=gate)
            return None
        K.gated_residual_(x2, self._lin(mod, y), gate)
        return None

    def _ffn_residual_(self, x2, lin1, lin2, h, gate, stats=False):
        """x2 += FFN(h) * gate: Linear -> GELU(tanh) -> Linear with both fusions (epilogue quantiser, epilogue residual)."""
        if isinstance(lin1, Int8Linear) and isinstance(lin2, Int8Linear) and gate.shape[0] == 1:
            xq, xs = h if isinstance(h, tuple) else K.quant_i8_block128(h)
            hq, hs = K.gemm_w8a8_quant(xq, xs, lin1.int8_weight, lin1.scale, x2.dtype, bias=lin1.bias, gelu_tanh=True)
            if stats:
                _, ws = K.gemm_w8a8_stats(hq, hs, lin2.int8_weight, lin2.scale, lin2.bias, x=x2, gate=gate)
                return K.row_stats_finalize(ws, x2.shape[1], self.eps, pad_cols=self._ln_pad)
            K.gemm_w8a8_residual_(x2, hq, hs, lin2.int8_weight, lin2.scale, bias=lin2.bias, gate=gate)
            return None
        f2 = self._ffn_q(lin1, lin2, h[0], h[1], x2.dtype) if isinstance(h, tuple) else self._ffn(lin1, lin2, h)
        K.gated_residual_(x2, f2, gate)
        return None

    def _ffn(self, lin1, lin2, x):
        """Linear -> GELU(tanh) -> Linear (wan2pt1.py:375).  W8A8: the first GEMM's epilogue also block-quantises its
        output for the second (same bits as Int8Linear -> GELU -> int8_quant, without the [L, ffn] 16-bit round trip)."""
        if isinstance(lin1, Int8Linear) and isinstance(lin2, Int8Linear):
            xq, xs = K.quant_i8_block128(x)
            return self._ffn_q(lin1, lin2, xq, xs, x.dtype)
        return self._lin(lin2, self._lin(lin1, x, gelu=True))

    def _fused_weights(self, i, blk):
        """q|k|v of self-attention and k|v of cross-attention share their input: concatenate the
        weights (and block scales — 128-row aligned, so the concatenation keeps the block structure)
        once, so each input is quantised once and multiplied once.  The per-module tensors are then RE-POINTED to
        views of the concatenation: one copy in memory, ``state_dict()`` unchanged, and an in-place update of a module's
        weight (``load_state_dict`` without ``assign``, ``copy_``) is an update of the fused tensor."""
        f = self._fused.get(i)
        if f is not None:
            return f
        sa, ca = blk.self_attn, blk.cross_attn
        f = {}
        if isinstance(sa.q, Int8Linear):
            assert self.dim % 128 == 0
            for key, mods in (("qkv", (sa.q, sa.k, sa.v)), ("ckv", (ca.k, ca.v))):
                for suf, attr in (("_w", "int8_weight"), ("_s", "scale"), ("_b", "bias")):
                    parts = [getattr(m, attr) for m in mods]
                    cat = torch.cat(parts, 0).contiguous()
                    f[key + suf] = cat
                    o = 0
                    for m, p_ in zip(mods, parts):
                        setattr(m, attr, cat[o:o + p_.shape[0]])
                        o += p_.shape[0]
        else:
            for key, mods in (("qkv", (sa.q, sa.k, sa.v)), ("ckv", (ca.k, ca.v))):
                for suf, attr in (("_w", "weight"), ("_b", "bias")):
                    parts = [getattr(m, attr) for m in mods]
                    cat = torch.cat([p_.detach() for p_ in parts], 0).contiguous()
                    f[key + suf] = cat
                    o = 0
                    for p_ in parts:
                        p_.data = cat[o:o + p_.shape[0]]
                        o += p_.shape[0]
        la = getattr(sa.attn_op, "local_attn", None)
        if la is not None:
            # fp32, contiguous nn.Linear parameters: these are the parameters' own storage (no copy), so fine-tuning
            # updates are seen; after a dtype conversion of proj_l they would be copies -> _apply drops them
            f["proj_w"] = la.proj_l.weight.detach().float().contiguous()
            f["proj_b"] = la.proj_l.bias.detach().float().contiguous()
        self._fused[i] = f
        return f

    def _text_kv_all(self, context):
        """[Lc, dim] text tokens -> [Lc, nblk * 2 * dim]: cross-attention k|v projections of every block, block i in columns
        [i*2*dim, (i+1)*2*dim).  The concatenated weights replace the per-block copies (views), so nothing is duplicated."""
        a = self._text_kv_weights()
        return self._fused_lin(context, a["ckv_w"], a.get("ckv_s"), a["ckv_b"])

    def _text_kv_weights(self):
        if self._ckv_all is None:
            fs = [self._fused_weights(i, blk) for i, blk in enumerate(self.blocks)]
            n2 = 2 * self.dim
            allw = {k: torch.cat([f[k] for f in fs], 0).contiguous() for k in ("ckv_w", "ckv_s", "ckv_b") if k in fs[0]}
            quant = "ckv_s" in allw
            for i, (f, blk) in enumerate(zip(fs, self.blocks)):   # per-block entries (and the modules' own tensors) become views
                f["ckv_w"] = allw["ckv_w"][i * n2:(i + 1) * n2]
                f["ckv_b"] = allw["ckv_b"][i * n2:(i + 1) * n2]
                if quant:
                    f["ckv_s"] = allw["ckv_s"][i * (n2 // 128):(i + 1) * (n2 // 128)]
                ca, d = blk.cross_attn, self.dim
                for j, m in enumerate((ca.k, ca.v)):
                    if quant:
                        m.int8_weight = f["ckv_w"][j * d:(j + 1) * d]
                        m.scale = f["ckv_s"][j * (d // 128):(j + 1) * (d // 128)]
                        m.bias = f["ckv_b"][j * d:(j + 1) * d]
                    else:
                        m.weight.data = f["ckv_w"][j * d:(j + 1) * d]
                        m.bias.data = f["ckv_b"][j * d:(j + 1) * d]
            self._ckv_all = allw
        return self._ckv_all

    def _fused_lin(self, x, w, s, b):
        if s is not None:
            xq, xs = K.quant_i8_block128(x)
            return K.gemm_w8a8(xq, xs, w, s, x.dtype, bias=b)
        return self._lin16(x, w, b)

    # ------------------------------------------------------------------ one transformer block
    def _self_attention(self, i, blk, h, cos, sin, L_loc, dtype, quant_out=False):
        """h: [L_loc, dim] modulated input of this rank's tokens (or its (int8, scales) pair when the norm
        already quantised it) -> [L_loc, dim] attention output."""
        f = self._fused_weights(i, blk)
        sa = blk.self_attn
        H, D, dim = self.num_heads, 128, self.dim
        at = self.attention_type
        sage = at in ("sage", "sagesla")
        vt = None
        fuse_vt = (isinstance(h, tuple) and self.fuse_vt and self.seq_parallel is None and dim % 256 == 0
                   and f["qkv_b"] is not None and not (sage and self.sage_pv == "fp8"))
        if (fuse_vt and self.split_qkv and self.two_streams and at == "sagesla" and self.sage_pv == "fp16"
                and f.get("proj_w") is not None and H * D == dim):
            # the projection as two launches, K|V then Q, with the K-side chain of the attention glue under the second
            qkv = torch.empty((L_loc, 3 * dim), dtype=dtype, device=h[0].device)
            nb = dim // 128

            def kv_proj():
                return K.gemm_w8a8_vt(h[0], h[1], f["qkv_w"][dim:], f["qkv_s"][nb:], f["qkv_b"][dim:], dim, torch.float16,
                                      out_dtype=dtype, out=qkv[:, dim:])[1]

            def q_proj():
                K.gemm_w8a8(h[0], h[1], f["qkv_w"][:dim], f["qkv_s"][:nb], dtype, bias=f["qkv_b"][:dim], out=qkv[:, :dim])
            out = dtype if quant_out else torch.empty((L_loc, dim), dtype=dtype, device=qkv.device)
            res, _, _ = sagesla_split_projection(
                kv_proj, lambda: K.qk_norm_rope(qkv, dim, H, D, sa.norm_k.weight, cos, sin, self.eps), q_proj,
                lambda: K.qk_norm_rope(qkv, 0, H, D, sa.norm_q.weight, cos, sin, self.eps), f["proj_w"], f["proj_b"], L_loc,
                self.sla_topk, out, D, dim, quant_out, self._side())
            return res
        if fuse_vt:
            # the V third of the projection leaves the GEMM as the attention kernel's V^T tiles (no v_transpose pass;
            # qkv's V columns are not written)
            qkv, vt = K.gemm_w8a8_vt(h[0], h[1], f["qkv_w"], f["qkv_s"], f["qkv_b"], 2 * dim,
                                     torch.float16 if sage else dtype, out_dtype=dtype)
        elif isinstance(h, tuple):
            qkv = K.gemm_w8a8(h[0], h[1], f["qkv_w"], f["qkv_s"], dtype, bias=f["qkv_b"])
        else:
            qkv = self._fused_lin(h, f["qkv_w"], f.get("qkv_s"), f["qkv_b"])  # [L, 3*dim]
        two = (self.two_streams and self.seq_parallel is None and vt is not None and at == "sagesla" and self.sage_pv == "fp16"
               and f.get("proj_w") is not None)

        def q_fn():
            return K.qk_norm_rope(qkv, 0, H, D, sa.norm_q.weight, cos, sin, self.eps)
        if self.seq_parallel is not None:
            if sage and self.sage_pv != "fp16":
                raise NotImplementedError("sage_pv='fp8' is not built for the sequence-parallel path (the packed K-side exchange "
                                          "carries fp16 V^T tiles); use sage_pv='fp16' with seqpar.enable")
            # q and k in ONE launch (a rank's shard is small: every launch in front of the K-side exchange counts)
            q, k = K.qk_norm_rope_pair(qkv, 0, dim, H, D, sa.norm_q.weight, sa.norm_k.weight, cos, sin, self.eps)
            out = dtype if quant_out else torch.empty((L_loc, dim), dtype=dtype, device=qkv.device)
            return self.seq_parallel.self_attention(self, f, q, k, qkv, out, quant_out=quant_out)
        q = None if two else q_fn()
        k = K.qk_norm_rope(qkv, dim, H, D, sa.norm_k.weight, cos, sin, self.eps)
        out = dtype if quant_out else torch.empty((L_loc, dim), dtype=dtype, device=qkv.device)
        dense = at in ("original", "sage")
        # W8A8: the attention kernel's epilogue hands the o projection its INT8 activation directly
        res, _, _ = sparse_linear_attention_hld(q, k, qkv[:, 2 * dim:], f.get("proj_w") if not dense else None,
                                                f.get("proj_b") if not dense else None, self.sla_topk, sage, out, D, dim,
                                                (D, 3 * dim), dense=dense, quant_out=quant_out,
                                                pv=self.sage_pv if sage else "fp16", vt=vt,
                                                side=self._side() if two else None, q_fn=q_fn if two else None)
        return res

    def _side(self):
        """The second HIP stream of the two-stream self-attention schedule: one per (device, calling stream), so that two
        host threads driving this model on two streams never share one (the allocation-safety argument of
        sla._sagesla_two_streams is per pair of streams)."""
        key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream)
        st = self._side_streams.get(key)
        if st is None:
            if len(self._side_streams) >= 16:
                self._side_streams.clear()
            st = self._side_streams[key] = torch.cuda.Stream()
        return st

    def _text_kvt(self, i, blk, context, text_kv=None):
        """Cross-attention K (RMSNorm'ed, head-major) and V^T tiles of block i for one batch entry's text tokens."""
        f = self._fused_weights(i, blk)
        ca = blk.cross_attn
        H, D, dim = self.num_heads, 128, self.dim
        Lc = context.shape[0]
        if text_kv is not None:   # this block's columns of the all-blocks text K|V projection
            kv = text_kv[:, i * 2 * dim:(i + 1) * 2 * dim]
        else:
            kv = self._fused_lin(context, f["ckv_w"], f.get("ckv_s"), f["ckv_b"])  # [Lc, 2*dim]
        k = K.qk_norm_rope(kv, 0, H, D, ca.norm_k.weight, None, None, self.eps)
        vt = K.v_transpose(kv[:, dim:], D, kv.stride(0), Lc, H, D, context.dtype)
        return k, vt

    def _text_mlp(self, crossattn_emb):
        """text_embedding (wan2pt1.py:678): Linear(text_dim -> dim), GELU(tanh), Linear(dim -> dim) on the [B, 512, text_dim]
        prompt embedding — two td_gemm_bf16 launches (bias + GELU in the first one's epilogue)."""
        te = self.text_embedding
        x = crossattn_emb.to(self.dtype)
        h = self._lin16(x, te[0].weight, te[0].bias, gelu=True)
        return self._lin16(h, te[2].weight, te[2].bias).contiguous()

    @torch.no_grad()
    def prepare_text(self, crossattn_emb):
        """Everything on the cross-attention K side depends on the text embedding only — the text MLP, the K|V projections
        of all blocks, K's RMSNorm, the V^T tiles — and the text is constant over the 4 steps of a video
        (wan2.1_t2v_infer.py:129-139 passes the same ``condition`` every step): compute it ONCE per text and keep it in
        persistent buffers (over-written in place when the text changes, so a captured hipGraph keeps valid pointers).
        ``forward`` calls this itself on a cache miss; ``GraphedModel`` calls it eagerly before replaying."""
        key = (crossattn_emb.data_ptr(), crossattn_emb._version, tuple(crossattn_emb.shape), crossattn_emb.dtype)
        st = self._text_states.get(key[0])
        if st is not None and st[0] == key:
            return st
        context = self._text_mlp(crossattn_emb)  # [B, Lc, dim]
        B = context.shape[0]
        per_b = []
        for b in range(B):
            tkv = self._text_kv_all(context[b]) if self.batch_text_kv else None
            per_b.append([self._text_kvt(i, blk, context[b], tkv) for i, blk in enumerate(self.blocks)])
        gen = None
        if st is not None and st[2].shape == context.shape and len(st[3]) == B:
            st[2].copy_(context)                       # same buffer, new contents: refresh the persistent tensors in place
            for old_b, new_b in zip(st[3], per_b):
                for (ok, ovt), (nk, nvt) in zip(old_b, new_b):
                    ok.copy_(nk)
                    ovt.copy_(nvt)
            context, per_b, gen = st[2], st[3], st[4]
        if gen is None:     # new persistent buffers: a graph captured on an earlier generation of this entry must re-capture
            self._text_gen += 1
            gen = self._text_gen
        # the entry holds a reference to the source tensor: its address cannot be recycled for another text while cached,
        # so (address, version) identifies the contents.  A handful of entries (one per live text buffer), oldest dropped.
        self._text_states.pop(key[0], None)
        while len(self._text_states) >= 8:
            self._text_states.pop(next(iter(self._text_states)))
        self._text_states[key[0]] = (key, crossattn_emb, context, per_b, gen)
        return self._text_states[key[0]]

    def _img_mlp(self, clip_tokens):
        """``img_emb`` (MLPProj, wan2pt1.py:457-486) on the CLIP tokens [B, 257, clip_dim]: the two LayerNorms on td_layernorm
        (fp32 statistics, one rounding, eps 1e-5 — nn.LayerNorm's), the Linears on td_gemm_bf16 with the exact GELU in the
        first one's epilogue."""
        pj = self.img_emb.proj
        B, n, c = clip_tokens.shape
        x = clip_tokens.to(self.dtype).reshape(B * n, c).contiguous()
        x = K.layernorm(x, pj[0].weight.float().contiguous(), pj[0].bias.float().contiguous(), pj[0].eps)
        x = K.gemm_bf16(x, pj[1].weight.detach(), pj[1].bias.detach(), epilogue="gelu_erf")
        x = K.gemm_bf16(x, pj[3].weight.detach(), pj[3].bias.detach())
        x = K.layernorm(x, pj[4].weight.float().contiguous(), pj[4].bias.float().contiguous(), pj[4].eps)
        return x.view(B, n, self.dim)

    def _img_kvt(self, blk, img_ctx):
        """K (RMSNorm'ed, head-major) and V^T tiles of a block's image branch for one batch entry's CLIP context [257, dim]
        (wan2pt1.py:343-344)."""
        ca = blk.cross_attn
        H, D, dim = self.num_heads, 128, self.dim
        kk = self._lin(ca.k_img, img_ctx)
        vv = self._lin(ca.v_img, img_ctx)
        k = K.qk_norm_rope(kk, 0, H, D, ca.norm_k_img.weight, None, None, self.eps)
        vt = K.v_transpose(vv, D, vv.stride(0), img_ctx.shape[0], H, D, img_ctx.dtype)
        return k, vt

    def _cross_attention(self, i, blk, xn, context, quant_out=False, text_kv=None, kvt=None, img_ctx=None):
        """xn [L, dim] (norm3 output), context [Lc, dim] -> [L, dim] (before the o projection).  img_ctx [257, dim]: Wan2.1
        I2V's image branch — the same queries against the CLIP tokens' K / V, the two results added (wan2pt1.py:345-351)."""
        ca = blk.cross_attn
        H, D, dim = self.num_heads, 128, self.dim
        L_ = xn[0].shape[0] if isinstance(xn, tuple) else xn.shape[0]
        rstd = None
        if isinstance(xn, tuple) and self.fuse_cross_q_norm and self._stats_ok(L_):
            # q projection whose epilogue also yields the RMSNorm statistic of its own output rows (no td_rms_stats pass)
            qc, ws = K.gemm_w8a8_stats(xn[0], xn[1], ca.q.int8_weight, ca.q.scale, ca.q.bias, out_dtype=context.dtype)
            # round 6: the pieces go to the attention kernel as they are (td_attn_16_qnorm_pieces forms the row statistic on
            # load, in td_row_stats_finalize's own order of additions): one launch less per cross-attention, the same bits
            rstd = (ws, self.eps) if self.fuse_stats_finalize else K.row_stats_finalize(ws, dim, self.eps, rms=True)
        else:
            qc = self._lin_q(ca.q, xn[0], xn[1], context.dtype) if isinstance(xn, tuple) else self._lin(ca.q, xn)
        k, vt = kvt if kvt is not None else self._text_kvt(i, blk, context, text_kv)
        out = None if quant_out else torch.empty((L_, dim), dtype=context.dtype, device=context.device)
        if self.fuse_cross_q_norm:
            # RMSNorm(q) applied where the attention kernel loads Q: the head-major normalised copy is never written
            # (one statistics pass over q instead of td_qk_norm_rope's read + write; bit-identical)
            if rstd is None:
                rstd = K.rms_stats(qc, dim, self.eps)
            res = K.attn_16_qnorm(qc, rstd, ca.norm_q.weight, k, vt, None, out, D, dim, quant_out=quant_out)
            if img_ctx is not None:
                ki, vti = self._img_kvt(blk, img_ctx)
                oi = torch.empty_like(res)
                K.attn_16_qnorm(qc, rstd, ca.norm_q.weight, ki, vti, None, oi, D, dim)
                K.gated_residual_(res, oi, None)          # x = x + img_x in the activation dtype (wan2pt1.py:350)
            return res
        q = K.qk_norm_rope(qc, 0, H, D, ca.norm_q.weight, None, None, self.eps)
        res = K.attn_16(q, k, vt, None, out, D, dim, quant_out=quant_out)
        if img_ctx is not None:
            ki, vti = self._img_kvt(blk, img_ctx)
            oi = torch.empty_like(res)
            K.attn_16(q, ki, vti, None, oi, D, dim)
            K.gated_residual_(res, oi, None)
        return res

    @property
    def _ln_pad(self):
        """pad_cols of the blocks' LayerNorms (see ``default_norm``)."""
        return 0 if self.default_norm else K.triton_ln_pad_cols(self.dim)

    def _stats_ok(self, rows):
        """Row statistics from the GEMM epilogues: W8A8 bf16 model, fused norm->INT8 path, rows worth a 256x256-tile GEMM."""
        return (self.fuse_row_stats and self.fuse_norm_quant and self.quant_linear and self.dtype == torch.bfloat16
                and self.dim % 64 == 0 and self.dim <= K.LNQ_MAX_N and rows >= 1024)

    def _split_rows(self, L_loc):
        """Row at which the token-local tail of a block is cut in two (0 = do not split): a multiple of 256 — the GEMM tile,
        hence also of the 128-row quantiser / Q blocks — so that each half computes exactly the bits of the whole."""
        if not (self.split_tokens and self.seq_parallel is None and L_loc >= 4096):
            return 0
        return (K.cdiv(L_loc, 256) // 2) * 256

    def _tail_half(self, i, blk, x2h, yh, ec, context, kvt, ws_h, rows_per_batch):
        """o projection -> cross-attention -> FFN of ONE row range (a view of the residual stream, updated in place): the
        W8A8 / fused-norm / row-statistics path of ``_block`` on those rows.  ws_h: this range's slice of the FFN output's
        row-statistics partials (the next block's norm1 finalises them for all rows at once)."""
        pad = self._ln_pad
        st3 = self._residual_lin_(x2h, blk.self_attn.o, yh, ec[2], stats=True)
        xn = K.layernorm_quant(x2h, blk.norm3.weight, blk.norm3.bias, self.eps, stats=st3, pad_cols=pad)
        c = self._cross_attention(i, blk, xn, context, quant_out=True, kvt=kvt)
        st2 = self._residual_lin_(x2h, blk.cross_attn.o, c, None, stats=True)
        h2 = K.layernorm_quant(x2h, None, None, self.eps, scale=ec[4], shift=ec[3], rows_per_batch=rows_per_batch, stats=st2,
                               pad_cols=pad)
        lin1, lin2 = blk.ffn[0], blk.ffn[2]
        hq, hs = K.gemm_w8a8_quant(h2[0], h2[1], lin1.int8_weight, lin1.scale, x2h.dtype, bias=lin1.bias, gelu_tanh=True)
        K.gemm_w8a8_stats(hq, hs, lin2.int8_weight, lin2.scale, lin2.bias, x=x2h, gate=ec[5], ws=ws_h)

    def _tail_two_halves(self, i, blk, x2, y, ec, context, kvt, ms, tkv=None):
        """After self-attention a block is token-local (o projection, cross-attention against the 512 text keys, FFN; SURVEY
        §8e): rows [0, ms) run on the current stream, rows [ms, L) on the model's second stream, fork / join with events
        (graph edges under capture).  One video then keeps two kernels in flight most of the time — the launch ramps, store
        tails and HBM-bound norm passes of one half run under the other half's GEMMs (what a second video in flight used to
        fill: +5.7 % in round 2's ``two_videos_in_flight`` leg).  Same kernels on row ranges cut at a tile boundary: bit-identical.
        Allocation safety without record_stream: tensors made on the second stream are used there only; ``y``, ``x2`` and
        ``ws`` belong to the current stream and outlive the join."""
        L_loc, dim = x2.shape
        yq, ys = y
        if kvt is None:     # uncached text: the block's text K / V^T ONCE, before the fork (not once per half)
            kvt = self._text_kvt(i, blk, context, tkv)
        ws = torch.empty((L_loc, dim // 64, 2), dtype=torch.float32, device=x2.device)
        main, side = torch.cuda.current_stream(), self._side()
        e_fork = torch.cuda.Event()
        e_fork.record(main)
        side.wait_event(e_fork)
        # (the two halves' GEMMs share the chip: the launch planner must not price a launch as the chip's only tenant)
        K.set_tuning(K.TUNE_GEMM_COTENANT, 1)
        try:
            with torch.cuda.stream(side):
                self._tail_half(i, blk, x2[ms:], (yq[ms:], ys[ms // 128:]), ec, context, kvt, ws[ms:], L_loc)
                e_join = torch.cuda.Event()
                e_join.record(side)
            self._tail_half(i, blk, x2[:ms], (yq[:ms], ys[:ms // 128]), ec, context, kvt, ws[:ms], L_loc)
        finally:
            K.set_tuning(K.TUNE_GEMM_COTENANT, 0)
        main.wait_event(e_join)
        return K.row_stats_finalize(ws, dim, self.eps, pad_cols=self._ln_pad)

    def _block(self, i, blk, x, e_B_6_D, cos, sin, context, tkv=None, kvts=None, img_ctx=None):
        """x: [B, L_loc, dim] (updated in place); e fp32 [B, 6, dim] = this block's modulation + e0 (wan2pt1.py:400,
        formed for all blocks at once in forward); context [B, Lc, dim]."""
        B, L_loc, dim = x.shape
        ec = [e_B_6_D[:, j].contiguous() for j in range(6)]   # (views when B == 1: no copy kernels)
        x2 = x.view(B * L_loc, dim)
        dt = x.dtype
        # the norms emit the INT8 activation of their consumer directly (td_layernorm_quant == td_layernorm +
        # td_quant_i8_block128 bit for bit; row statistics pass + per-128x128-block apply/quantise pass: 18 + 28 us at
        # [32760, 1536] on MI355X against 36 + 25 us for the two operators)
        fuse = self.fuse_norm_quant and self.quant_linear and dim <= K.LNQ_MAX_N
        rows = [slice(b * L_loc, (b + 1) * L_loc) for b in range(B)]
        pad = self._ln_pad
        # ---- self attention ----
        fstats = fuse and B == 1 and self._stats_ok(L_loc)   # LayerNorm statistics ride on the GEMM that produced x
        st1, self._carry_stats = getattr(self, "_carry_stats", None), None
        if fuse:
            hs_ = [K.layernorm_quant(x2[r], None, None, self.eps, scale=ec[1][b:b + 1], shift=ec[0][b:b + 1],
                                     rows_per_batch=L_loc, stats=st1 if fstats else None, pad_cols=pad)
                   for b, r in enumerate(rows)]
        else:
            h = K.layernorm(x2, None, None, self.eps, scale=ec[1], shift=ec[0], rows_per_batch=L_loc, pad_cols=pad)
            hs_ = [h[r] for r in rows]
        qo = self.quant_linear and B == 1   # attention epilogue quantises for the o proj (one GPU and sequence-parallel alike)
        ys = [self._self_attention(i, blk, hb, cos, sin, L_loc, dt, quant_out=qo) for hb in hs_]
        y = ys[0] if B == 1 else torch.cat(ys, 0)
        ms = self._split_rows(L_loc) if (fstats and qo and isinstance(blk.norm3, FastLayerNorm) and isinstance(y, tuple)
                                         and isinstance(blk.ffn[0], Int8Linear) and img_ctx is None) else 0
        if ms:
            self._carry_stats = self._tail_two_halves(i, blk, x2, y, ec, context[0], None if kvts is None else kvts[0][i], ms,
                                                      None if tkv is None else tkv[0])
            return x
        st3 = self._residual_lin_(x2, blk.self_attn.o, y, ec[2], stats=fstats)
        # ---- cross attention ----
        if isinstance(blk.norm3, FastLayerNorm):
            if fuse:
                xns = [K.layernorm_quant(x2[r], blk.norm3.weight, blk.norm3.bias, self.eps, stats=st3, pad_cols=pad)
                       for r in rows]
            else:
                xn = K.layernorm(x2, blk.norm3.weight, blk.norm3.bias, self.eps, pad_cols=pad)
                xns = [xn[r] for r in rows]
        else:
            xns = [x2[r] for r in rows]
        cs = [self._cross_attention(i, blk, xns[b], context[b], quant_out=self.quant_linear and B == 1 and img_ctx is None,
                                    text_kv=None if tkv is None else tkv[b],
                                    kvt=None if kvts is None else kvts[b][i],
                                    img_ctx=None if img_ctx is None else img_ctx[b]) for b in range(B)]
        c = cs[0] if B == 1 else torch.cat(cs, 0)
        st2 = self._residual_lin_(x2, blk.cross_attn.o, c, None, stats=fstats)
        # ---- FFN ----
        if fuse:
            h2 = K.layernorm_quant(x2, None, None, self.eps, scale=ec[4], shift=ec[3], rows_per_batch=L_loc, stats=st2,
                                   pad_cols=pad)
        else:
            h2 = K.layernorm(x2, None, None, self.eps, scale=ec[4], shift=ec[3], rows_per_batch=L_loc, pad_cols=pad)
        self._carry_stats = self._ffn_residual_(x2, blk.ffn[0], blk.ffn[2], h2, ec[5], stats=fstats)   # -> next block's norm1
        return x

    # ------------------------------------------------------------------ forward
    def _rope(self, T, H, W, device):
        key = (T, H, W, str(device))
        if key not in self._rope_cache:
            ang = rope_angles(T, H, W, 128, device)
            self._rope_cache[key] = (torch.cos(ang).float().contiguous(), torch.sin(ang).float().contiguous())
        return self._rope_cache[key]

    @torch.no_grad()
    def forward(self, x_B_C_T_H_W, timesteps_B_T, crossattn_emb, frame_cond_crossattn_emb_B_L_D=None,
                y_B_C_T_H_W=None, **kwargs):
        return_tokens = bool(kwargs.pop("_return_tokens", False))   # parity tests: the [B, L, dim] tokens after the blocks
        del kwargs
        K.require_gpu(x_B_C_T_H_W)
        if (frame_cond_crossattn_emb_B_L_D is not None) != bool(self.clip_dim):
            raise ValueError("frame_cond_crossattn_emb_B_L_D (CLIP image tokens) goes with a model built with clip_dim (Wan2.1 I2V, "
                             "wan2pt1.py:642); Wan2.2-A14B I2V conditions through y_B_C_T_H_W only")
        assert timesteps_B_T.shape[1] == 1
        if self.seq_parallel is not None and getattr(self.seq_parallel, "broadcast_inputs", False):
            x_B_C_T_H_W, timesteps_B_T, crossattn_emb, y_B_C_T_H_W = self.seq_parallel.broadcast(
                x_B_C_T_H_W, timesteps_B_T, crossattn_emb, y_B_C_T_H_W)     # wan2pt1.py:629-636
        t_B = timesteps_B_T[:, 0]
        kt, kh, kw = self.patch_size
        B, C1, T_in, H_in, W_in = x_B_C_T_H_W.shape
        C = C1 + (0 if y_B_C_T_H_W is None else y_B_C_T_H_W.shape[1])
        assert T_in % kt == 0 and H_in % kh == 0 and W_in % kw == 0
        T, H, W = T_in // kt, H_in // kh, W_in // kw
        L_ = T * H * W
        dt = self.dtype
        cos, sin = self._rope(T, H, W, x_B_C_T_H_W.device)
        sp = self.seq_parallel
        # f3 in HIP (csrc/embed_head.hip): patchify + patch_embedding, the time MLPs, the AdaLN vectors, head + unpatchify —
        # no library GEMM and no torch elementwise kernel inside a captured forward.  ONE path: what those kernels do not
        # take is refused here, by name (there is no library-operator branch beside them; the operator sequence they are
        # checked against lives in the tests and in oracle/wan_ref.py)
        self._check_embed_head_support(C, dt, timesteps_B_T.dtype)
        row0 = 0
        if sp is not None:
            row0, stop = sp.shard_range(L_)
            cos, sin = cos[row0:stop].contiguous(), sin[row0:stop].contiguous()
        else:
            stop = L_
        x = K.patch_embed(x_B_C_T_H_W.to(dt).contiguous(), None if y_B_C_T_H_W is None else y_B_C_T_H_W.to(dt).contiguous(),
                          self.patch_embedding.weight, self.patch_embedding.bias, row0, stop - row0)   # [B, L_loc, dim]
        # time embeddings in fp32 (the reference's autocast(float32) island, wan2pt1.py:671-674)
        te, tp = self.time_embedding, self.time_projection
        e = K.gemv_f32(K.time_sinusoid(t_B.contiguous(), self.freq_dim), te[0].weight, te[0].bias)
        e_B_D = K.gemv_f32(e, te[2].weight, te[2].bias, silu_input=True)
        e0 = K.gemv_f32(e_B_D, tp[1].weight, tp[1].bias, silu_input=True).unflatten(1, (6, self.dim))
        tkv = kvts = None
        if self.cache_text_kv:
            context, kvts = self.prepare_text(crossattn_emb)[2:4]   # once per text (keyed on the tensor's identity + version)
        else:
            context = self._text_mlp(crossattn_emb)  # [B, Lc, dim]
            if self.batch_text_kv:
                # the cross-attention K|V projections of ALL blocks read the same 512 text tokens: one [Lc, nblk*2*dim] GEMM
                # (2880 tiles) and one quantisation of the text instead of one 96-tile GEMM + quantisation per block
                tkv = [self._text_kv_all(context[b]) for b in range(B)]
        # (modulation + e0) of every block in ONE add instead of one tiny kernel per block (+ 6 slice copies each)
        ver = sum(blk.modulation._version for blk in self.blocks)   # (trainable parameters: a stacked copy must notice updates)
        mods = self._fused.get("mods")
        if mods is None or mods[1] != ver:
            mods = self._fused["mods"] = (torch.stack([blk.modulation.detach().float() for blk in self.blocks], 0), ver)
        e_all = K.bcast_add(mods[0].view(len(self.blocks), 6, self.dim), e0.contiguous())   # fp32 [nblk, B, 6, dim]
        self._carry_stats = None
        img_ctx = None
        if frame_cond_crossattn_emb_B_L_D is not None:
            if sp is not None:
                frame_cond_crossattn_emb_B_L_D = sp.broadcast(frame_cond_crossattn_emb_B_L_D)[0] if getattr(sp, "broadcast_inputs", False) \
                    else frame_cond_crossattn_emb_B_L_D
            img_ctx = self._img_mlp(frame_cond_crossattn_emb_B_L_D)      # [B, 257, dim]
        tap = getattr(self, "_tap_tokens", None)   # tools/drift.py: list that receives the tokens after every block
        for i, blk in enumerate(self.blocks):
            x = self._block(i, blk, x, e_all[i], cos, sin, context, tkv, kvts, img_ctx)
            if tap is not None:
                tap.append(x.clone())
        if return_tokens:
            return x if sp is None else sp.gather_tokens(x, L_)
        # head (wan2pt1.py:444-454): fp32 modulate of the (bf16) norm, fp32 Linear
        hw = self._fused.get("head")
        hver = (self.head.head.weight._version, self.head.head.bias._version, self.head.modulation._version)
        if hw is None or hw[3] != hver:   # the fp32 island's up-cast of the (bf16) head parameters; in-place updates noticed
            hw = self._fused["head"] = (self.head.head.weight.detach().float().contiguous(),
                                       self.head.head.bias.detach().float().contiguous(),
                                       self.head.modulation.detach().float().contiguous().view(1, 2, self.dim), hver)
        em = K.bcast_add(hw[2], e_B_D.view(B, 1, self.dim))[0]    # [B, 2, dim] = modulation + e
        out = K.head(x, em[:, 1].contiguous(), em[:, 0].contiguous(), hw[0], hw[1], self.eps, self.out_dim, T, H, W,
                     unpatchify=sp is None, row0=row0)
        if sp is None:
            return out
        out = sp.gather_tokens(out, L_)
        # unpatchify "b (t h w) (kt kh kw d) -> b d (t kt) (h kh) (w kw)"   (wan2pt1.py:710-721)
        out = out.view(B, T, H, W, kt, kh, kw, self.out_dim).permute(0, 7, 1, 4, 2, 5, 3, 6)
        return out.reshape(B, self.out_dim, T * kt, H * kh, W * kw)

    def _check_embed_head_support(self, C, dt, t_dtype):
        """What csrc/embed_head.hip is built for — checked once per forward, refused by name."""
        kt, kh, kw = self.patch_size
        bad = []
        if (kt, kh, kw) != (1, 2, 2):
            bad.append(f"patch_size {self.patch_size} (need (1, 2, 2), the Wan2.1 / 2.2 patch)")
        if C % 4 or C > 64:
            bad.append(f"{C} input channels (need a multiple of 4, at most 64: 16 for T2V, 36 for I2V)")
        if dt not in (torch.bfloat16, torch.float16):
            bad.append(f"model dtype {dt} (need bfloat16 or float16)")
        pe = self.patch_embedding
        if pe.weight.dtype != dt or pe.bias.dtype != dt:
            bad.append(f"patch_embedding parameters in {pe.weight.dtype} (need the model dtype {dt})")
        for name, m in (("time_embedding.0", self.time_embedding[0]), ("time_embedding.2", self.time_embedding[2]),
                        ("time_projection.1", self.time_projection[1])):
            if m.weight.dtype not in (torch.bfloat16, torch.float16) or m.bias.dtype != m.weight.dtype:
                bad.append(f"{name} parameters in {m.weight.dtype} (td_gemv_f32 reads 16-bit weights and widens them, as the "
                           "reference's autocast(float32) island does)")
        if t_dtype not in (torch.bfloat16, torch.float16):
            bad.append(f"timesteps in {t_dtype} (the reference passes them in the model dtype, wan2.1_t2v_infer.py:131)")
        if self.dim % 8 or self.out_dim * 4 > 64:
            bad.append(f"dim {self.dim} / out_dim {self.out_dim} (need dim % 8 == 0, out_dim <= 16)")
        if bad:
            raise ValueError("WanModel.forward on the MI355X kernels does not take: " + "; ".join(bad))


MODEL_CONFIGS = {  # inference/modify_model.py:86-127
    "Wan2.1-1.3B": dict(dim=1536, eps=1e-6, ffn_dim=8960, freq_dim=256, in_dim=16, model_type="t2v",
                        num_heads=12, num_layers=30, out_dim=16, text_len=512),
    "Wan2.1-14B": dict(dim=5120, eps=1e-6, ffn_dim=13824, freq_dim=256, in_dim=16, model_type="t2v",
                       num_heads=40, num_layers=40, out_dim=16, text_len=512),
    "Wan2.2-A14B": dict(dim=5120, eps=1e-6, ffn_dim=13824, freq_dim=256, in_dim=36, model_type="i2v",
                        num_heads=40, num_layers=40, out_dim=16, text_len=512),
}


def select_model(model_name: str, **overrides) -> WanModel:
    """inference/modify_model.py:86-127."""
    if model_name not in MODEL_CONFIGS:
        raise ValueError(f"Unknown model name: {model_name}")
    cfg = dict(MODEL_CONFIGS[model_name])
    cfg.update(overrides)
    return WanModel(**cfg)
