"""f4, the other direction of the tokenizer interface: the Wan2.1 VAE **encoder** — video [B, 3, 1 + 4k, H, W] in [-1, 1] ->
normalised latents [B, 16, 1 + k, H/8, W/8] (what I2V feeds the DiT as its conditioning channels ``y``,
``inference/wan2.2_i2v_infer.py``; ``Wan2pt1VAEInterface.encode``, rcm/tokenizers/wan2pt1.py:702-703).

Reference: ``Encoder3d`` (wan2pt1.py:251-340), ``Resample`` 'downsample2d' / 'downsample3d' (:98-102, :133-149),
``WanVAE_.encode`` (:479-518: one frame, then chunks of four, feature caches in between; ``mu`` only, normalised with the
latent statistics).

As in vae_decode.py the clip is encoded in ONE pass.  Every causal convolution of the chunked schedule is the causal
convolution of the whole clip; the temporal down-samplers are the one special case, reproduced exactly: the first frame
passes through unconvolved and frames 1.. are produced by the stride-2, unpadded (3,1,1) convolution over the whole
sequence — windows (0,1,2), (2,3,4), ... — which is what the reference's "last frame of the previous chunk + this chunk"
computes when the chunks are 1, 4, 4, ... frames (so T must be 1 + 4k, the only lengths the reference's chunking accepts).
The spatial down-sampler is ZeroPad2d((0, 1, 0, 1)) + 3x3 stride 2.  HIP only, as vae_decode.py (bf16 on a GPU: td_vae_conv /
td_vae_conv_ex / td_vae_chan_rms / td_gemm_bf16, channels-last); the library-operator restatement for the CPU pins is
``oracle/f4_ref.py``.
Weights: the reference's ``state_dict`` (keys ``encoder.*`` and ``conv1.*``)."""
from __future__ import annotations

import math
import re

import torch

from .vae_decode import LATENT_MEAN, LATENT_STD, _HipFrameAttention, _HipRes, _k2d, latent_stats, pointwise_conv


class _HipDown:
    def __init__(self, g, K):
        self.K = K
        self.w, self.b = _k2d(g("resample.1.weight")), g("resample.1.bias")
        wt = g("time_conv.weight", None)
        self.wt, self.bt = (None, None) if wt is None else (_k2d(wt), g("time_conv.bias"))

    def __call__(self, x):
        K = self.K
        x = K.vae_conv_strided(x, self.w, self.b, 1, 3, 3, stride_hw=2, pad_hw=0)
        if self.wt is not None and x.shape[1] > 1:
            y = K.vae_conv_strided(x, self.wt, self.bt, 3, 1, 1, stride_t=2, pad_t=0)
            x = torch.cat([x[:, :1], y], dim=1)
        return x


def synthetic_state_dict(dim=96, z_dim=16, seed=0, dtype=torch.bfloat16, device="cpu"):
    """Random-init encoder weights of the named architecture in the reference's key layout (``_video_vae``, wan2pt1.py:565-574:
    dim 96, dim_mult [1, 2, 4, 4], 2 residual blocks per level, temporal down-sampling at the last two of the three
    down-samplers); bf16-representable values."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd = {}

    def put(name, t):
        sd[name] = t.bfloat16().to(dtype).to(device)

    def conv(name, o, i, k):
        put(name + ".weight", torch.randn(o, i, *k, generator=g) / (i * math.prod(k)) ** 0.5)
        put(name + ".bias", 0.1 * torch.randn(o, generator=g))

    def gamma(name, c, nd):
        put(name, (1 + 0.2 * torch.randn(c, generator=g)).reshape(c, *([1] * nd)))

    def res(p, i, o):
        gamma(p + "residual.0.gamma", i, 3)
        conv(p + "residual.2", o, i, (3, 3, 3))
        gamma(p + "residual.3.gamma", o, 3)
        conv(p + "residual.6", o, o, (3, 3, 3))
        if i != o:
            conv(p + "shortcut", o, i, (1, 1, 1))

    dims = [dim, dim, dim * 2, dim * 4, dim * 4]
    conv("encoder.conv1", dims[0], 3, (3, 3, 3))
    n = 0
    for i, (a, b) in enumerate(zip(dims[:-1], dims[1:])):
        for _ in range(2):
            res(f"encoder.downsamples.{n}.", a, b)
            a = b
            n += 1
        if i != 3:
            conv(f"encoder.downsamples.{n}.resample.1", b, b, (3, 3))
            if i >= 1:
                conv(f"encoder.downsamples.{n}.time_conv", b, b, (3, 1, 1))
            n += 1
    top = dims[-1]
    res("encoder.middle.0.", top, top)
    gamma("encoder.middle.1.norm.gamma", top, 2)
    conv("encoder.middle.1.to_qkv", 3 * top, top, (1, 1))
    conv("encoder.middle.1.proj", top, top, (1, 1))
    res("encoder.middle.2.", top, top)
    gamma("encoder.head.0.gamma", top, 3)
    conv("encoder.head.2", 2 * z_dim, top, (3, 3, 3))
    conv("conv1", 2 * z_dim, 2 * z_dim, (1, 1, 1))
    return sd


class WanVaeEncoder:
    """``encode(video)`` -> normalised latent mean (``WanVAE.encode``, wan2pt1.py:661-672; ``WanVAE_.encode`` :479-511)."""

    def __init__(self, state_dict, dtype=torch.bfloat16, device="cuda", mean=LATENT_MEAN, std=LATENT_STD):
        self.dtype, self.device = dtype, torch.device(device)
        if dtype != torch.bfloat16 or self.device.type != "cuda":
            raise ValueError("WanVaeEncoder runs on the HIP kernels only: bf16 on a GPU; the library-operator restatement for "
                             "CPU checks is oracle/f4_ref.py")
        from . import kernels as K_     # raises if the HIP library is missing
        self.K = K_
        sd = {k: v.detach().to(device=self.device, dtype=dtype) for k, v in state_dict.items()
              if k.startswith(("encoder.", "conv1."))}
        if "encoder.conv1.weight" not in sd or "conv1.weight" not in sd:
            raise ValueError("not a Wan VAE state dict: encoder.conv1.weight / conv1.weight missing")
        self.sd = sd
        self.z_dim = sd["conv1.weight"].shape[0] // 2
        self.mean, self.inv_std = latent_stats(mean, std, self.z_dim, dtype, self.device)

        def getter(prefix):
            def g(name, *default):
                key = prefix + name
                if key in sd:
                    return sd[key]
                if default:
                    return default[0]
                raise KeyError(key)
            return g

        self.stages = []
        for i in sorted({int(m.group(1)) for k in sd for m in [re.match(r"encoder\.downsamples\.(\d+)\.", k)] if m}):
            p = f"encoder.downsamples.{i}."
            if p + "residual.0.gamma" in sd:
                self.stages.append(_HipRes(getter(p), K_))
            elif p + "resample.1.weight" in sd:
                self.stages.append(_HipDown(getter(p), K_))
            elif p + "to_qkv.weight" in sd:
                self.stages.append(_HipFrameAttention(getter(p), K_))
            else:
                raise ValueError(f"unrecognised encoder stage {p}*")
        self.stages += [_HipRes(getter("encoder.middle.0."), K_), _HipFrameAttention(getter("encoder.middle.1."), K_),
                        _HipRes(getter("encoder.middle.2."), K_)]
        self.t_down = sum(1 for s in self.stages if isinstance(s, _HipDown) and s.wt is not None)
        w1 = sd["encoder.conv1.weight"]                    # [dim, 3, 3, 3, 3]: td_vae_conv wants C_in % 32 == 0
        w1p = torch.zeros(w1.shape[0], 32, *w1.shape[2:], dtype=dtype, device=self.device)
        w1p[:, :w1.shape[1]] = w1
        self.h_conv1 = (_k2d(w1p), sd["encoder.conv1.bias"])
        self.h_head = (sd["encoder.head.0.gamma"].reshape(-1).contiguous(), _k2d(sd["encoder.head.2.weight"]), sd["encoder.head.2.bias"])
        wc = sd["conv1.weight"]
        self.h_out = (wc.reshape(wc.shape[0], -1).contiguous(), sd["conv1.bias"])

    @classmethod
    def from_reference(cls, vae_module_or_state_dict, **kw):
        sd = vae_module_or_state_dict if isinstance(vae_module_or_state_dict, dict) else vae_module_or_state_dict.state_dict()
        return cls(sd, **kw)

    def latent_frames(self, pixel_frames: int) -> int:
        return 1 + (pixel_frames - 1) // 2 ** self.t_down      # get_latent_num_frames, wan2pt1.py:708-709

    @torch.no_grad()
    def encode(self, video):
        in_dtype = video.dtype
        T = video.shape[2]
        if (T - 1) % 2 ** self.t_down:
            raise ValueError(f"{T} frames: the encoder takes 1 + {2 ** self.t_down} k frames (the reference's chunking, wan2pt1.py:483-499)")
        x = video.to(device=self.device, dtype=self.dtype)
        K = self.K
        B, C, _, H, W = x.shape
        xc = torch.zeros((B, T, H, W, 32), dtype=self.dtype, device=self.device)     # channels-last, 3 -> 32 zero-padded
        xc[..., :C] = x.permute(0, 2, 3, 4, 1)
        x = K.vae_conv(xc, self.h_conv1[0], self.h_conv1[1], 3, 3, 3)
        for st in self.stages:
            x = st(x)
        g, w, b = self.h_head
        x = K.vae_conv(K.vae_chan_rms(x, g), w, b, 3, 3, 3)
        x = pointwise_conv(K, x, *self.h_out)[..., :self.z_dim].permute(0, 4, 1, 2, 3)     # conv1 (1x1x1), mu = first half
        mu = (x - self.mean) * self.inv_std                                                # wan2pt1.py:505-508
        return mu.contiguous().to(in_dtype)
