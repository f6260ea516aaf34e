"""CPU restatement of the Wan DiT forward (one denoising step) and of the 4-step rCM sampler.

TEST INFRASTRUCTURE (see oracle/__init__.py).  torch-CPU.

``wan_forward(sd, cfg, x, t, ctx, mode=...)`` follows ``WanModel.forward``
(rcm/networks/wan2pt1.py:598-721; wan2pt2.py: same with ``y`` concatenated on channels) on a plain
state dict with the reference's key names.  Two arithmetic modes:

  mode="eager"  the reference's *original* path (config C1): plain Linear in the model dtype, eager
                WanRMSNorm / WanLayerNorm (wan2pt1.py:181-212), dense SDPA (rcm/utils/attention.py).
                Pinned against the real reference on CPU by tests/test_oracle_vs_reference.py.
  mode="turbo"  what modify_model.py installs: Fast norms (ops/core.py), optional Int8Linear on every
                Linear inside blocks except proj_l, attention in {"original","sage","sla","sagesla"}.

dtype semantics reproduce the CUDA run of a bf16 checkpoint: activations bf16; the
``amp.autocast("cuda", dtype=float32)`` islands (time embedding :671-674, modulation :399-400,
gated residuals :405-406/:412-413 — elementwise ops are not autocast targets — and the head
:451-454) are written out explicitly.
"""
from __future__ import annotations

import math
import torch
import torch.nn.functional as F

from . import ops_ref as O
from . import sla_ref as S


def _lin(x, w, b, act_dtype):
    """nn.Linear in the activation dtype (bf16 checkpoint, bf16 input)."""
    return F.linear(x.to(act_dtype), w.to(act_dtype), None if b is None else b.to(act_dtype))


class _Linears:
    """Linear layers of the blocks: plain or W8A8 (weights quantised once, cached)."""

    def __init__(self, sd, quant, act_dtype):
        self.sd, self.quant, self.dt = sd, quant, act_dtype
        self.cache = {}

    def __call__(self, name, x, gelu=False):
        w, b = self.sd[name + ".weight"], self.sd.get(name + ".bias")
        if not self.quant:
            y = _lin(x, w, b, self.dt)
            return F.gelu(y, approximate="tanh") if gelu else y
        if name not in self.cache:
            self.cache[name] = O.quant_block128(w.to(self.dt))
        wq, ws = self.cache[name]
        return O.int8_linear(x.to(self.dt), wq, ws, bias=None if b is None else b.to(self.dt), gelu_tanh=gelu)


def _attention(q, k, v, kind, sd, prefix, topk, dt):
    """q,k,v [B, L, H, D] -> [B, L, H*D]   (MinimalA2AAttnOp.forward a2a_cp.py:198-200)."""
    if kind == "original":
        o = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2))
        o = o.transpose(1, 2).contiguous()
    elif kind == "sage":  # dense SageAttention INT8-QK / FP16-PV (config C2), no linear branch
        qh, kh, vh = (t.transpose(1, 2).contiguous() for t in (q, k, v))
        km = S.seq_mean(kh)
        q8, qs = S.quant_per_block_int8(qh, 128)
        k8, ks = S.quant_per_block_int8(kh, 64, km)
        lut = S.dense_lut(qh.shape[0], qh.shape[1], qh.shape[2], 128, 64)
        o = S.sage_sparse_attn(q8, qs, k8, ks, vh, lut, out_dtype=dt).transpose(1, 2).contiguous()
    else:
        wp = sd[prefix + ".attn_op.local_attn.proj_l.weight"].float()
        bp = sd[prefix + ".attn_op.local_attn.proj_l.bias"].float()
        if kind == "sla":
            o = S.sla_forward(q, k, v, wp, bp, topk, 128, 64, dt)
        else:
            o = S.sagesla_forward(q, k, v, wp, bp, topk, dt)
    return o.flatten(2)


def wan_forward(sd, cfg, x_B_C_T_H_W, timesteps_B_T, crossattn_emb, y_B_C_T_H_W=None, mode="eager",
                attention="original", quant=False, topk=0.1, act_dtype=torch.bfloat16, num_layers=None,
                return_tokens=False, tap=None, clip_emb=None, on_layer=None):
    """``tap``: optional dict that receives the intermediates of block 0 (fixture generation only).
    ``return_tokens``: True -> the [B, L, dim] tokens after the last block; "both" -> (tokens, velocity) from ONE pass.
    ``sd`` may be any mapping with ``__getitem__`` / ``get`` / ``__contains__`` (make_golden_r04.LazyLayers generates a
    14B-width model's layers one at a time).
    ``on_layer(i, x)``: optional callback with the tokens after block ``i`` (fixture generation: depth profiles, progress)."""
    dt = act_dtype
    dim, H = cfg["dim"], cfg["num_heads"]
    D = dim // H
    eps = cfg.get("eps", 1e-6)
    freq_dim = cfg.get("freq_dim", 256)
    out_dim = cfg.get("out_dim", 16)
    kt, kh, kw = cfg.get("patch_size", (1, 2, 2))
    nl = cfg["num_layers"] if num_layers is None else num_layers
    turbo = mode == "turbo"
    lin = _Linears(sd, quant and turbo, dt)

    def rms(x, w):
        return O.rmsnorm_fast(x, w.float(), eps) if turbo else O.rmsnorm_eager(x, w.to(dt), eps)

    def ln(x, w=None, b=None):
        return O.layernorm_fast(x, w, b, eps) if turbo else O.layernorm_eager(x, w, b, eps)

    x = x_B_C_T_H_W
    if y_B_C_T_H_W is not None:
        x = torch.cat([x, y_B_C_T_H_W], dim=1)
    B, C, T_in, H_in, W_in = x.shape
    T, Hh, Ww = T_in // kt, H_in // kh, W_in // kw
    L = T * Hh * Ww
    x = x.to(dt).view(B, C, T, kt, Hh, kh, Ww, kw).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B, L, -1)
    x = _lin(x, sd["patch_embedding.weight"], sd["patch_embedding.bias"], dt)
    # time embeddings (fp32 island)
    t_B = timesteps_B_T[:, 0]
    e = O.sinusoidal_embedding_1d(freq_dim, t_B).float()
    e = F.linear(e, sd["time_embedding.0.weight"].float(), sd["time_embedding.0.bias"].float())
    e_B_D = F.linear(F.silu(e), sd["time_embedding.2.weight"].float(), sd["time_embedding.2.bias"].float())
    e0 = F.linear(F.silu(e_B_D), sd["time_projection.1.weight"].float(),
                  sd["time_projection.1.bias"].float()).unflatten(1, (6, dim))
    ctx = _lin(crossattn_emb, sd["text_embedding.0.weight"], sd["text_embedding.0.bias"], dt)
    ctx = _lin(F.gelu(ctx, approximate="tanh"), sd["text_embedding.2.weight"], sd["text_embedding.2.bias"], dt)
    freqs = O.rope_freqs(T, Hh, Ww, D)
    ctx_img = None
    if clip_emb is not None:
        # Wan2.1 I2V: CLIP image tokens [B, 257, 1280] -> MLPProj (wan2pt1.py:457-486: LayerNorm, Linear, exact GELU, Linear,
        # LayerNorm, all in the model dtype) -> context_clip, attended by every block's image branch (:303-352)
        h_ = F.layer_norm(clip_emb.to(dt), (clip_emb.shape[-1],), sd["img_emb.proj.0.weight"].to(dt), sd["img_emb.proj.0.bias"].to(dt), 1e-5)
        h_ = F.gelu(_lin(h_, sd["img_emb.proj.1.weight"], sd["img_emb.proj.1.bias"], dt))
        h_ = _lin(h_, sd["img_emb.proj.3.weight"], sd["img_emb.proj.3.bias"], dt)
        ctx_img = F.layer_norm(h_, (dim,), sd["img_emb.proj.4.weight"].to(dt), sd["img_emb.proj.4.bias"].to(dt), 1e-5)
        # the reference concatenates [clip tokens, text tokens] and every block slices the two parts back out (:332-334, :682);
        # the slices are VIEWS with the concatenation's batch stride, and the CPU library GEMM's summation order depends on
        # that layout — kept, so that the bit-for-bit pin against the live module holds
        n_img = ctx_img.shape[1]
        ctx_all = torch.cat([ctx_img, ctx], dim=1)
        ctx_img, ctx = ctx_all[:, :n_img], ctx_all[:, n_img:]

    for i in range(nl):
        p = f"blocks.{i}"
        em = (sd[p + ".modulation"].float() + e0).chunk(6, dim=1)  # each [B,1,dim] fp32
        # self attention
        h = O.modulate(ln(x), em[1], em[0])
        sa = p + ".self_attn"
        q = rms(lin(sa + ".q", h), sd[sa + ".norm_q.weight"]).view(B, L, H, D)
        k = rms(lin(sa + ".k", h), sd[sa + ".norm_k.weight"]).view(B, L, H, D)
        v = lin(sa + ".v", h).view(B, L, H, D)
        qr, kr = O.rope_apply(q, freqs), O.rope_apply(k, freqs)
        a = _attention(qr, kr, v, attention if turbo else "original", sd, sa, topk, dt)
        if tap is not None and i == 0:
            tap.update(h1=h, q_rope=qr, k_rope=kr, v=v, attn=a)
        x = O.gated_residual(x, lin(sa + ".o", a), em[2])
        if tap is not None and i == 0:
            tap["x_after_sa"] = x
        # cross attention
        ca = p + ".cross_attn"
        xn = ln(x, sd.get(p + ".norm3.weight"), sd.get(p + ".norm3.bias")) if (p + ".norm3.weight") in sd else x
        q = rms(lin(ca + ".q", xn), sd[ca + ".norm_q.weight"]).view(B, L, H, D)
        k = rms(lin(ca + ".k", ctx), sd[ca + ".norm_k.weight"]).view(B, -1, H, D)
        v = lin(ca + ".v", ctx).view(B, -1, H, D)
        a = _attention(q, k, v, "original", sd, ca, topk, dt)
        if ctx_img is not None:   # WanI2VCrossAttention.forward (wan2pt1.py:340-352): the same q against the image keys, summed
            k_i = rms(lin(ca + ".k_img", ctx_img), sd[ca + ".norm_k_img.weight"]).view(B, -1, H, D)
            v_i = lin(ca + ".v_img", ctx_img).view(B, -1, H, D)
            a = a + _attention(q, k_i, v_i, "original", sd, ca, topk, dt)
        x = x + lin(ca + ".o", a)
        if tap is not None and i == 0:
            tap["x_after_ca"] = x
        # ffn
        h = O.modulate(ln(x), em[4], em[3])
        f = lin(p + ".ffn.2", lin(p + ".ffn.0", h, gelu=True))
        x = O.gated_residual(x, f, em[5])
        if on_layer is not None:
            on_layer(i, x)
        if getattr(sd, "evict_finished_layers", False):
            lin.cache = {k_: v_ for k_, v_ in lin.cache.items() if not k_.startswith(p + ".")}

    if return_tokens and return_tokens != "both":
        return x
    # head (fp32 island): norm -> type_as(x) -> *(1+e1)+e0 in fp32 -> fp32 Linear
    hm = (sd["head.modulation"].float() + e_B_D.unsqueeze(1)).chunk(2, dim=1)
    hn = O.layernorm_eager(x, None, None, eps).float() * (1 + hm[1]) + hm[0]
    out = F.linear(hn, sd["head.head.weight"].float(), sd["head.head.bias"].float())
    out = out.view(B, T, Hh, Ww, kt, kh, kw, out_dim).permute(0, 7, 1, 4, 2, 5, 3, 6)
    out = out.reshape(B, out_dim, T * kt, Hh * kh, Ww * kw)
    return (x, out) if return_tokens == "both" else out


# --------------------------------------------------------------------------- #
# a18  sampler (inference/wan2.1_t2v_infer.py:111-140)
# --------------------------------------------------------------------------- #
def rcm_timesteps(num_steps=4, sigma_max=80.0):
    mid_t = [1.5, 1.4, 1.0][: num_steps - 1]
    t = torch.tensor([math.atan(sigma_max), *mid_t, 0], dtype=torch.float64)
    return torch.sin(t) / (torch.cos(t) + torch.sin(t))


def rcm_sample(net_fn, init_noise, noises, num_steps=4, sigma_max=80.0, act_dtype=torch.bfloat16, ode=False,
               net_low_fn=None, boundary=0.9):
    """net_fn(x_bf16, t_bf16[B,1]) -> velocity; ``noises``: list of the per-step N(0,1) tensors (the
    reference draws them from a seeded CUDA generator; passing them in makes CPU/GPU runs comparable).
    ``ode``: x - (t_cur - t_next) v instead of the SDE step (wan2.2_i2v_infer.py:202-203); ``net_low_fn`` / ``boundary``:
    the Wan2.2 expert switch — the high-noise net until the first step with t_cur < boundary, the low-noise one from
    there on (:190-197)."""
    t_steps = rcm_timesteps(num_steps, sigma_max)
    x = init_noise.to(torch.float64) * t_steps[0]
    ones = torch.ones(x.size(0), 1, dtype=torch.float64)
    fn, switched = net_fn, False
    for i, (t_cur, t_next) in enumerate(zip(t_steps[:-1], t_steps[1:])):
        if net_low_fn is not None and t_cur.item() < boundary and not switched:
            fn, switched = net_low_fn, True
        v = fn(x.to(act_dtype), (t_cur.float() * ones * 1000).to(act_dtype)).to(torch.float64)
        if ode:
            x = x - (t_cur - t_next) * v
        else:
            x = (1 - t_next) * (x - t_cur * v) + t_next * noises[i].to(torch.float64)
    return x.float()


def make_state_dict(cfg, seed=0, with_proj_l=True):
    """Seeded synthetic weights with the reference's key names and init scales
    (WanModel.init_weights wan2pt1.py:723-764; head / proj_l / biases given small non-zero values so
    every branch contributes — BASELINE.md §3).  All values are bf16-representable."""
    g = torch.Generator().manual_seed(seed)
    dim, ffn, H = cfg["dim"], cfg["ffn_dim"], cfg["num_heads"]
    in_dim = cfg.get("in_dim", 16) * math.prod(cfg.get("patch_size", (1, 2, 2)))
    out = cfg.get("out_dim", 16) * math.prod(cfg.get("patch_size", (1, 2, 2)))
    text_dim, freq = cfg.get("text_dim", 4096), cfg.get("freq_dim", 256)
    sd = {}

    def lin(name, o, i, std=None):
        std = std if std is not None else math.sqrt(2.0 / (i + o))
        sd[name + ".weight"] = (torch.randn(o, i, generator=g) * std).bfloat16().float()
        sd[name + ".bias"] = (torch.randn(o, generator=g) * 0.02).bfloat16().float()

    clip_dim = cfg.get("clip_dim")
    lin("patch_embedding", dim, in_dim)
    if clip_dim:
        sd["img_emb.proj.0.weight"] = (1 + 0.1 * torch.randn(clip_dim, generator=g)).bfloat16().float()
        sd["img_emb.proj.0.bias"] = (0.02 * torch.randn(clip_dim, generator=g)).bfloat16().float()
        lin("img_emb.proj.1", clip_dim, clip_dim, 1.0 / math.sqrt(clip_dim))
        lin("img_emb.proj.3", dim, clip_dim, 1.0 / math.sqrt(clip_dim))
        sd["img_emb.proj.4.weight"] = (1 + 0.1 * torch.randn(dim, generator=g)).bfloat16().float()
        sd["img_emb.proj.4.bias"] = (0.02 * torch.randn(dim, generator=g)).bfloat16().float()
    lin("text_embedding.0", dim, text_dim, 0.02)
    lin("text_embedding.2", dim, dim, 0.02)
    lin("time_embedding.0", dim, freq, 0.02)
    lin("time_embedding.2", dim, dim, 0.02)
    lin("time_projection.1", 6 * dim, dim, 0.02)
    for i in range(cfg["num_layers"]):
        p = f"blocks.{i}"
        for a in ("self_attn", "cross_attn"):
            for n in ("q", "k", "v", "o"):
                lin(f"{p}.{a}.{n}", dim, dim, 1.0 / math.sqrt(dim))
            for n in ("norm_q", "norm_k"):
                sd[f"{p}.{a}.{n}.weight"] = (1 + 0.1 * torch.randn(dim, generator=g)).bfloat16().float()
        if clip_dim:
            for n in ("k_img", "v_img"):
                lin(f"{p}.cross_attn.{n}", dim, dim, 1.0 / math.sqrt(dim))
            sd[f"{p}.cross_attn.norm_k_img.weight"] = (1 + 0.1 * torch.randn(dim, generator=g)).bfloat16().float()
        if with_proj_l:
            sd[f"{p}.self_attn.attn_op.local_attn.proj_l.weight"] = (torch.randn(128, 128, generator=g) * 0.02).bfloat16().float()
            sd[f"{p}.self_attn.attn_op.local_attn.proj_l.bias"] = (torch.randn(128, generator=g) * 0.02).bfloat16().float()
        sd[f"{p}.norm3.weight"] = (1 + 0.1 * torch.randn(dim, generator=g)).bfloat16().float()
        sd[f"{p}.norm3.bias"] = (0.02 * torch.randn(dim, generator=g)).bfloat16().float()
        lin(f"{p}.ffn.0", ffn, dim)
        lin(f"{p}.ffn.2", dim, ffn)
        sd[f"{p}.modulation"] = (torch.randn(1, 6, dim, generator=g) / math.sqrt(dim)).bfloat16().float()
    lin("head.head", out, dim, 0.02)
    sd["head.modulation"] = (torch.randn(1, 2, dim, generator=g) / math.sqrt(dim)).bfloat16().float()
    return sd
