"""f4 on the CPU: the umT5 encoder from PyTorch library operators (``Umt5EncoderRef``, bit-identical in bf16 to the
reference's ``T5Encoder``: same rounding points) and the WHOLE-CLIP restatement of the reference's Wan2.1 VAE decoder / encoder from PyTorch library operators
(NCDHW, any dtype, any device) — what ``turbodiffusion_amd.vae_decode`` / ``vae_encode`` compute with hand-written HIP kernels.

TEST INFRASTRUCTURE (oracle/__init__.py): only tests/, the fixture generators and ``__graft_entry__.smoke()`` may import this.
Until round 4 these graphs lived inside the product classes as a second ``"torch"`` backend selected off-GPU; the product
classes are HIP-only now and raise without a GPU like every other operator.

Reference: ``rcm/tokenizers/wan2pt1.py`` — ``Decoder3d`` (:343-435), ``Encoder3d`` (:251-340), ``ResidualBlock`` (:177-209),
``AttentionBlock`` (:212-248), ``Resample`` (:83-151), ``RMS_norm`` (:58-70), ``CausalConv3d`` (:37-55), ``WanVAE_.decode``
(:520-537), ``WanVAE_.encode`` (:479-518), ``WanVAE`` (:601-681, the latent statistics).

The reference decodes ONE latent frame per pass carrying two frames of every causal convolution in a Python-side cache; a
causal convolution fed chunk by chunk with that cache IS the causal convolution of the whole clip, so the clip is processed
in ONE pass here.  The temporal re-samplers are the one special case, reproduced exactly (first frame untouched; see the
classes).  Pinned to the reference's own chunked ``WanVAE_.decode`` / ``.encode`` (live import, random weights, fp32) by
tests/test_vae_umt5_cpu.py."""
from __future__ import annotations

import re

import torch
import torch.nn.functional as F

import math

from turbodiffusion_amd.text_encoder import relative_buckets
from turbodiffusion_amd.vae_decode import LATENT_MEAN, LATENT_STD


def _chan_rms(x, gamma):
    """RMS_norm, channel first (wan2pt1.py:69-70): x / max(||x||_2 over C, 1e-12) * sqrt(C) * gamma."""
    return F.normalize(x, dim=1) * (x.shape[1] ** 0.5) * gamma


def _causal_conv(x, w, b):
    """CausalConv3d over the WHOLE clip: (kt - 1) zero frames on the left, symmetric spatial padding (wan2pt1.py:42-55)."""
    kt, kh, kw = w.shape[2:]
    if kt > 1 or kh > 1 or kw > 1:
        x = F.pad(x, (kw // 2, kw // 2, kh // 2, kh // 2, kt - 1, 0))
    return F.conv3d(x, w, b)


class _Res:
    def __init__(self, g):
        self.n1, self.w1, self.b1 = g("residual.0.gamma"), g("residual.2.weight"), g("residual.2.bias")
        self.n2, self.w2, self.b2 = g("residual.3.gamma"), g("residual.6.weight"), g("residual.6.bias")
        self.ws, self.bs = g("shortcut.weight", None), g("shortcut.bias", None)

    def __call__(self, x):
        h = x if self.ws is None else F.conv3d(x, self.ws, self.bs)
        y = _causal_conv(F.silu(_chan_rms(x, self.n1)), self.w1, self.b1)
        y = _causal_conv(F.silu(_chan_rms(y, self.n2)), self.w2, self.b2)
        return y + h


class _FrameAttention:
    """single-head self-attention over the h*w positions of every frame (wan2pt1.py:229-248)"""

    def __init__(self, g):
        self.n, self.wq, self.bq, self.wp, self.bp = g("norm.gamma"), g("to_qkv.weight"), g("to_qkv.bias"), g("proj.weight"), g("proj.bias")

    def __call__(self, x):
        B, C, T, H, W = x.shape
        f = x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W)
        qkv = F.conv2d(_chan_rms(f, self.n), self.wq, self.bq).reshape(B * T, 1, 3 * C, H * W).transpose(2, 3)
        q, k, v = qkv.contiguous().chunk(3, dim=-1)
        o = F.scaled_dot_product_attention(q, k, v).squeeze(1).transpose(1, 2).reshape(B * T, C, H, W)
        o = F.conv2d(o, self.wp, self.bp)
        return o.reshape(B, T, C, H, W).permute(0, 2, 1, 3, 4) + x


class _Up:
    """Resample 'upsample2d' / 'upsample3d' (wan2pt1.py:94-131): [time: frame 0 untouched, frames 1.. -> causal (3,1,1)
    convolution to 2C channels, the two halves interleaved in time]; then per frame nearest x2 + 3x3 conv to C/2."""

    def __init__(self, g):
        self.w, self.b = g("resample.1.weight"), g("resample.1.bias")
        self.wt, self.bt = g("time_conv.weight", None), g("time_conv.bias", None)

    def __call__(self, x):
        B, C, T, H, W = x.shape
        if self.wt is not None and T > 1:
            y = _causal_conv(x[:, :, 1:], self.wt, self.bt)                       # [B, 2C, T-1, H, W]
            y = y.reshape(B, 2, C, T - 1, H, W).permute(0, 2, 3, 1, 4, 5).reshape(B, C, 2 * (T - 1), H, W)
            x = torch.cat([x[:, :, :1], y], dim=2)
            T = x.shape[2]
        f = x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W)
        f = F.interpolate(f, scale_factor=2.0, mode="nearest-exact")
        f = F.conv2d(f, self.w, self.b, padding=1)
        return f.reshape(B, T, f.shape[1], 2 * H, 2 * W).permute(0, 2, 1, 3, 4)



class _Down:
    """Resample 'downsample2d' / 'downsample3d' (wan2pt1.py:133-149), NCDHW"""

    def __init__(self, g):
        self.w, self.b = g("resample.1.weight"), g("resample.1.bias")
        self.wt, self.bt = g("time_conv.weight", None), g("time_conv.bias", None)

    def __call__(self, x):
        B, C, T, H, W = x.shape
        f = x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W)
        f = F.conv2d(F.pad(f, (0, 1, 0, 1)), self.w, self.b, stride=2)
        x = f.reshape(B, T, C, f.shape[2], f.shape[3]).permute(0, 2, 1, 3, 4)
        if self.wt is not None and T > 1:
            x = torch.cat([x[:, :, :1], F.conv3d(x, self.wt, self.bt, stride=(2, 1, 1))], dim=2)
        return x




def _getter(sd, prefix):
    def g(name, *default):
        key = prefix + name
        if key in sd:
            return sd[key]
        if default:
            return default[0]
        raise KeyError(key)
    return g


def _stats(mean, std, z_dim, dtype, device):
    """(mean, 1 / std) as the reference forms them (wan2pt1.py:643-645): the statistics are rounded to ``dtype`` FIRST and
    the reciprocal is taken in ``dtype`` — in bf16 that differs from rounding 1/std by one ulp on 6 of the 16 channels."""
    if len(mean) != z_dim or len(std) != z_dim:   # (toy fixtures with another channel count)
        mean, std = (0.0,) * z_dim, (1.0,) * z_dim
    m = torch.tensor(mean, dtype=dtype, device=device).view(1, -1, 1, 1, 1)
    inv = (1.0 / torch.tensor(std, dtype=dtype, device=device)).view(1, -1, 1, 1, 1)
    return m, inv


class VaeDecoderRef:
    """``decode(z)``: normalised latents -> video (WanVAE.decode, wan2pt1.py:674-681), whole clip, library operators."""

    def __init__(self, state_dict, dtype=torch.float32, device="cpu", mean=LATENT_MEAN, std=LATENT_STD):
        self.dtype, self.device = dtype, torch.device(device)
        sd = {k: v.detach().to(device=self.device, dtype=dtype) for k, v in state_dict.items()
              if k.startswith(("decoder.", "conv2."))}
        if "decoder.conv1.weight" not in sd or "conv2.weight" not in sd:
            raise ValueError("not a Wan VAE state dict: decoder.conv1.weight / conv2.weight missing")
        self.sd = sd
        self.z_dim = sd["conv2.weight"].shape[0]
        self.mean, self.inv_scale = _stats(mean, std, self.z_dim, dtype, self.device)
        self.stages = [_Res(_getter(sd, "decoder.middle.0.")), _FrameAttention(_getter(sd, "decoder.middle.1.")),
                       _Res(_getter(sd, "decoder.middle.2."))]
        for i in sorted({int(m.group(1)) for k in sd for m in [re.match(r"decoder\.upsamples\.(\d+)\.", k)] if m}):
            p = f"decoder.upsamples.{i}."
            if p + "residual.0.gamma" in sd:
                self.stages.append(_Res(_getter(sd, p)))
            elif p + "resample.1.weight" in sd:
                self.stages.append(_Up(_getter(sd, p)))
            elif p + "to_qkv.weight" in sd:
                self.stages.append(_FrameAttention(_getter(sd, p)))
            else:
                raise ValueError(f"unrecognised decoder stage {p}*")
        self.t_up = sum(1 for s in self.stages if isinstance(s, _Up) and s.wt is not None)

    @classmethod
    def from_reference(cls, vae_module_or_state_dict, **kw):
        sd = vae_module_or_state_dict if isinstance(vae_module_or_state_dict, dict) else vae_module_or_state_dict.state_dict()
        return cls(sd, **kw)

    def pixel_frames(self, latent_frames: int) -> int:
        return (latent_frames - 1) * 2 ** self.t_up + 1

    @torch.no_grad()
    def decode(self, z):
        in_dtype, sd = z.dtype, self.sd
        x = z.to(device=self.device, dtype=self.dtype)
        x = x / self.inv_scale + self.mean                                     # WanVAE_.decode, wan2pt1.py:523-526
        x = F.conv3d(x, sd["conv2.weight"], sd["conv2.bias"])
        x = _causal_conv(x, sd["decoder.conv1.weight"], sd["decoder.conv1.bias"])
        for st in self.stages:
            x = st(x)
        x = F.silu(_chan_rms(x, sd["decoder.head.0.gamma"]))
        x = _causal_conv(x, sd["decoder.head.2.weight"], sd["decoder.head.2.bias"])
        return x.to(in_dtype)


class VaeEncoderRef:
    """``encode(video)`` -> normalised latent mean (WanVAE.encode, wan2pt1.py:661-672; WanVAE_.encode :479-511), whole clip."""

    def __init__(self, state_dict, dtype=torch.float32, device="cpu", mean=LATENT_MEAN, std=LATENT_STD):
        self.dtype, self.device = dtype, torch.device(device)
        sd = {k: v.detach().to(device=self.device, dtype=dtype) for k, v in state_dict.items()
              if k.startswith(("encoder.", "conv1."))}
        if "encoder.conv1.weight" not in sd or "conv1.weight" not in sd:
            raise ValueError("not a Wan VAE state dict: encoder.conv1.weight / conv1.weight missing")
        self.sd = sd
        self.z_dim = sd["conv1.weight"].shape[0] // 2
        self.mean, self.inv_std = _stats(mean, std, self.z_dim, dtype, self.device)
        self.stages = []
        for i in sorted({int(m.group(1)) for k in sd for m in [re.match(r"encoder\.downsamples\.(\d+)\.", k)] if m}):
            p = f"encoder.downsamples.{i}."
            if p + "residual.0.gamma" in sd:
                self.stages.append(_Res(_getter(sd, p)))
            elif p + "resample.1.weight" in sd:
                self.stages.append(_Down(_getter(sd, p)))
            elif p + "to_qkv.weight" in sd:
                self.stages.append(_FrameAttention(_getter(sd, p)))
            else:
                raise ValueError(f"unrecognised encoder stage {p}*")
        self.stages += [_Res(_getter(sd, "encoder.middle.0.")), _FrameAttention(_getter(sd, "encoder.middle.1.")),
                        _Res(_getter(sd, "encoder.middle.2."))]
        self.t_down = sum(1 for s in self.stages if isinstance(s, _Down) and s.wt is not None)

    @classmethod
    def from_reference(cls, vae_module_or_state_dict, **kw):
        sd = vae_module_or_state_dict if isinstance(vae_module_or_state_dict, dict) else vae_module_or_state_dict.state_dict()
        return cls(sd, **kw)

    def latent_frames(self, pixel_frames: int) -> int:
        return 1 + (pixel_frames - 1) // 2 ** self.t_down

    @torch.no_grad()
    def encode(self, video):
        in_dtype, sd = video.dtype, self.sd
        T = video.shape[2]
        if (T - 1) % 2 ** self.t_down:
            raise ValueError(f"{T} frames: the encoder takes 1 + {2 ** self.t_down} k frames (the reference's chunking, wan2pt1.py:483-499)")
        x = video.to(device=self.device, dtype=self.dtype)
        x = _causal_conv(x, sd["encoder.conv1.weight"], sd["encoder.conv1.bias"])
        for st in self.stages:
            x = st(x)
        x = F.silu(_chan_rms(x, sd["encoder.head.0.gamma"]))
        x = _causal_conv(x, sd["encoder.head.2.weight"], sd["encoder.head.2.bias"])
        x = F.conv3d(x, sd["conv1.weight"], sd["conv1.bias"])[:, :self.z_dim]
        mu = (x - self.mean) * self.inv_std                                                # wan2pt1.py:505-508
        return mu.contiguous().to(in_dtype)


# ---------------------------------------------------------------------------------------------------------------------
# umT5 encoder from library operators (rcm/utils/umt5.py: T5Encoder :308-337, T5SelfAttention :217-238, T5Attention :145-194,
# T5FeedForward :197-214, T5LayerNorm :131-142, T5RelativeEmbedding :268-305; UMT5EncoderModel.__call__ :501-521 zeroes the
# rows past a prompt's length).  Only the valid rows are computed (masked keys get weight exactly 0, padded rows are
# discarded); q|k|v and gate|fc1 are fused GEMMs.  The rounding points of the reference's 16-bit path are kept, so in bf16
# on the CPU this is bit-identical to the reference module (tests/test_vae_umt5_cpu.py).
def _t5_norm(x, w, eps=1e-6):
    y = x * torch.rsqrt(x.float().pow(2).mean(dim=-1, keepdim=True) + eps)
    if w.dtype in (torch.float16, torch.bfloat16):
        y = y.type_as(w)
    return w * y


def _gelu_tanh(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def _fp16_clamp(x):
    if x.dtype == torch.float16 and torch.isinf(x).any():
        c = torch.finfo(x.dtype).max - 1000
        x = torch.clamp(x, min=-c, max=c)
    return x



class Umt5EncoderRef:
    """``encoder(ids, mask)`` -> [B, L_pad, dim]: ``ids`` / ``mask`` [B, L_pad] as the reference's tokenizer returns them
    (padding on the right).  ``state_dict``: the reference encoder's (``models_t5_umt5-xxl-enc-bf16.pth`` layout); the
    configuration is read off the tensors."""

    def __init__(self, state_dict, dtype=torch.bfloat16, device="cuda", max_dist=128, eps=1e-6):
        self.dtype, self.device, self.max_dist, self.eps = dtype, torch.device(device), max_dist, eps

        def t(k):
            return state_dict[k].detach().to(device=self.device, dtype=dtype)

        self.emb = t("token_embedding.weight")
        self.dim = self.emb.shape[1]
        n = 0
        while f"blocks.{n}.norm1.weight" in state_dict:
            n += 1
        if n == 0:
            raise ValueError("not a T5 encoder state dict: blocks.0.norm1.weight missing")
        self.shared_pos = "pos_embedding.embedding.weight" in state_dict
        self.pos = t("pos_embedding.embedding.weight") if self.shared_pos else None
        self.layers = []
        for i in range(n):
            p = f"blocks.{i}."
            lay = {
                "n1": t(p + "norm1.weight"), "n2": t(p + "norm2.weight"),
                "qkv": torch.cat([t(p + "attn.q.weight"), t(p + "attn.k.weight"), t(p + "attn.v.weight")], 0).contiguous(),
                "o": t(p + "attn.o.weight"),
                "gf": torch.cat([t(p + "ffn.gate.0.weight"), t(p + "ffn.fc1.weight")], 0).contiguous(),
                "fc2": t(p + "ffn.fc2.weight"),
                "pos": None if self.shared_pos else t(p + "pos_embedding.embedding.weight"),
            }
            self.layers.append(lay)
        self.final_norm = t("norm.weight")
        pos0 = self.pos if self.shared_pos else self.layers[0]["pos"]
        self.num_buckets, self.num_heads = pos0.shape
        self.dim_attn = self.layers[0]["o"].shape[1]
        self.dim_ffn = self.layers[0]["fc2"].shape[1]
        assert self.dim_attn % self.num_heads == 0

    @classmethod
    def from_reference(cls, module_or_state_dict, **kw):
        sd = module_or_state_dict if isinstance(module_or_state_dict, dict) else module_or_state_dict.state_dict()
        return cls(sd, **kw)

    def _rows(self, ids):
        """ids [n] (one prompt's valid tokens) -> [n, dim]"""
        n, H, c = ids.shape[0], self.num_heads, self.dim_attn // self.num_heads
        x = F.embedding(ids, self.emb)
        buckets = relative_buckets(n, self.num_buckets, self.max_dist, self.device)
        shared = None if not self.shared_pos else F.embedding(buckets, self.pos).permute(2, 0, 1)
        for lay in self.layers:
            bias = shared if shared is not None else F.embedding(buckets, lay["pos"]).permute(2, 0, 1)   # [H, n, n]
            qkv = F.linear(_t5_norm(x, lay["n1"], self.eps), lay["qkv"]).view(n, 3, H, c)
            q, k, v = qkv[:, 0].transpose(0, 1), qkv[:, 1].transpose(0, 1), qkv[:, 2].transpose(0, 1)   # [H, n, c]
            s = torch.matmul(q, k.transpose(1, 2)) + bias                      # no scaling (umt5.py:183)
            a = F.softmax(s.float(), dim=-1).type_as(s)
            o = torch.matmul(a, v).transpose(0, 1).reshape(n, H * c)
            x = _fp16_clamp(x + F.linear(o, lay["o"]))
            gf = F.linear(_t5_norm(x, lay["n2"], self.eps), lay["gf"])
            h = gf[:, self.dim_ffn:] * _gelu_tanh(gf[:, :self.dim_ffn])        # fc1(x) * gelu(gate(x)), umt5.py:210
            x = _fp16_clamp(x + F.linear(h, lay["fc2"]))
        return _t5_norm(x, self.final_norm, self.eps)

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
