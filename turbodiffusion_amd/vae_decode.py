"""f4: the Wan2.1 VAE **decoder** — latents [B, 16, T, H/8, W/8] -> video [B, 3, 1 + 4 (T - 1), H, W].

Reference: ``rcm/tokenizers/wan2pt1.py`` — ``Decoder3d`` (:343-435), ``ResidualBlock`` (:177-209), ``AttentionBlock``
(:212-248), ``Resample`` (:83-151), ``RMS_norm`` (:58-70), ``CausalConv3d`` (:37-55), ``WanVAE_.decode`` (:520-537),
``WanVAE`` (:601-681, the latent statistics), ``Wan2pt1VAEInterface`` (:684-740, bf16 weights and activations).

What is different here, and why.  The reference decodes ONE latent frame per pass (21 passes for 81 frames), carrying the
last two input frames of each of its 30-odd causal convolutions from pass to pass in a Python-side cache — a schedule made
for 24-80 GB cards.  A causal convolution fed chunk by chunk with that cache IS the causal convolution of the whole clip
(two zero frames on the left), so with 288 GB of HBM the clip is decoded in ONE pass: every convolution runs once over all
frames (a 480p clip peaks at ~40 GB in bf16), no cache, no per-frame launches.  The one place where the chunked schedule
is not a plain causal convolution is reproduced exactly: the temporal up-samplers leave the FIRST frame alone (no time
convolution, no doubling — the reference marks its cache ``"Rep"``, wan2pt1.py:108-131) and run their time convolution over
frames 1.. with zeros to the left (frame 0 is not part of that window).  ``tests/test_vae_umt5_cpu.py`` pins this module to
the reference's own chunked ``WanVAE_.decode`` (live import, random weights, fp32).

HIP only (bf16 on a GPU; without the HIP library or off the GPU the constructor raises, like every operator of this
package — the library-operator restatement of the same whole-clip graph that the CPU pins run is test infrastructure:
``oracle/f4_ref.py``).  Activations channels-last [B, T, H, W, C]; every 3x3x3 / 3x3 / (3,1,1) convolution is ``td_vae_conv`` —
one implicit-GEMM kernel on the bf16 matrix pipe (csrc/vae_conv.hip) with the causal / spatial zero padding, the nearest x2
up-sampling and the time up-sampler's frame interleave folded into its gather and store, bias and the residual add in its
epilogue; the channel RMS-norm + SiLU is ``td_vae_chan_rms``; the 1x1 convolutions (shortcuts, to_qkv, proj, conv2) run on
the same layout through ``td_gemm_bf16`` (csrc/gemm_bf16.hip) and the single-head per-frame attention of the middle block as
two of those GEMMs around ``td_softmax_rows`` (the [hw, hw] score matrix of a frame is 78 MB at 480p: it simply lives in HBM).
(MIOpen's 3-D convolutions are not an option here: a 480p decode did not finish in 8 minutes on a fresh box — per-shape
solver search and kernel compilation, then naive fallbacks.)  Weights: the reference's own ``state_dict`` (keys ``decoder.*`` and ``conv2.*``; the
architecture is read off the keys, no config needed)."""
from __future__ import annotations

import math
import re

import torch

# per-channel statistics of the 16 latent channels (rcm/tokenizers/wan2pt1.py:607-642): z_model = z * std + mean
LATENT_MEAN = (-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
               0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921)
LATENT_STD = (2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
              3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160)


# channels-last stages on csrc/vae_conv.hip
def latent_stats(mean, std, z_dim, dtype, device):
    """(mean, 1 / std) [1, C, 1, 1, 1] as the reference forms them (wan2pt1.py:643-645: ``torch.tensor(std, dtype=dtype)``,
    then ``1.0 / self.std``): the statistics are rounded to ``dtype`` FIRST and the reciprocal is taken IN ``dtype`` — in
    bf16 that is one ulp away from the rounded fp64 reciprocal on 6 of the 16 channels (0.4-0.8 %)."""
    if len(mean) != z_dim or len(std) != z_dim:   # (tests use a 4-channel toy)
        mean, std = (0.0,) * z_dim, (1.0,) * z_dim
    m = torch.tensor(mean, dtype=dtype, device=device).view(1, -1, 1, 1, 1)
    inv = (1.0 / torch.tensor(std, dtype=dtype, device=device)).view(1, -1, 1, 1, 1)
    return m, inv


def pointwise_conv(K, x, w, b, res=None):
    """1x1(x1) convolution on channels-last activations [B, T, H, W, Ci] (Ci % 32 == 0): td_vae_conv with ONE tap — the same
    implicit-GEMM kernel as the 3x3x3 convolutions (bias and the residual add in its epilogue)."""
    return K.vae_conv(x, w, b, 1, 1, 1, res=res)


def _k2d(w):
    """[Co, Ci, (kt,) kh, kw] -> [Co, kt*kh*kw*Ci], K ordered (dt, dh, dw, c) (td_vae_conv)"""
    if w.dim() == 4:
        w = w.unsqueeze(2)
    return w.permute(0, 2, 3, 4, 1).reshape(w.shape[0], -1).contiguous()


class _HipRes:
    def __init__(self, g, K):
        self.K = K
        self.n1, self.n2 = g("residual.0.gamma").reshape(-1).contiguous(), g("residual.3.gamma").reshape(-1).contiguous()
        self.w1, self.b1 = _k2d(g("residual.2.weight")), g("residual.2.bias")
        self.w2, self.b2 = _k2d(g("residual.6.weight")), g("residual.6.bias")
        ws = g("shortcut.weight", None)
        self.ws, self.bs = (None, None) if ws is None else (ws.reshape(ws.shape[0], -1).contiguous(), g("shortcut.bias"))

    def __call__(self, x):
        K = self.K
        h = x if self.ws is None else pointwise_conv(K, x, self.ws, self.bs)
        y = K.vae_conv(K.vae_chan_rms(x, self.n1), self.w1, self.b1, 3, 3, 3)
        return K.vae_conv(K.vae_chan_rms(y, self.n2), self.w2, self.b2, 3, 3, 3, res=h)


class _HipFrameAttention:
    """single-head self-attention over the h*w positions of every frame (AttentionBlock, wan2pt1.py:212-248): S = q k^T per
    frame in fp32 (td_gemm_bf16, batched over the frames: the [hw, hw] score matrix of a frame is 156 MB at 480p and simply
    lives in HBM), row softmax with the 1/sqrt(C) scale (td_softmax_rows -> bf16 P, zero-padded to a multiple of 64 keys),
    O = P v (td_gemm_bf16 against v^T, which a batched GEMM with the roles swapped produces directly: v^T = W_v x^T), proj +
    residual in td_vae_conv's epilogue.  The v bias rides on the P.v product's epilogue (rows of P sum to 1: P (v + b) =
    P v + b; one rounding of v + b less than the reference's bf16 path)."""

    def __init__(self, g, K):
        self.K = K
        self.n = g("norm.gamma").reshape(-1).contiguous()
        wq, bq, wp = g("to_qkv.weight"), g("to_qkv.bias"), g("proj.weight")
        C = wp.shape[0]
        wq = wq.reshape(3 * C, C)
        self.wq, self.wk, self.wv = (wq[i * C:(i + 1) * C].contiguous() for i in range(3))
        self.bq, self.bk, self.bv = (bq[i * C:(i + 1) * C].contiguous() for i in range(3))
        self.wp, self.bp = wp.reshape(C, C).contiguous(), g("proj.bias")

    def __call__(self, x):
        K = self.K
        B, T, H, W, C = x.shape
        n, hw = B * T, H * W
        hwp = K.cdiv(hw, 64) * 64
        xn = K.vae_chan_rms(x, self.n, silu=False)
        q = pointwise_conv(K, xn, self.wq, self.bq).view(n, hw, C)
        k = pointwise_conv(K, xn, self.wk, self.bk).view(n, hw, C)
        vt = torch.zeros((n, C, hwp), dtype=x.dtype, device=x.device)         # V^T per frame, zero behind the hw positions
        K.gemm_bf16_batched(self.wv.unsqueeze(0).expand(n, C, C), xn.view(n, hw, C), out=vt[:, :, :hw])
        # scores / probabilities of a few frames at a time through ONE reused pair of buffers: fp32 [c, hw, hw] + 16-bit
        # [c, hw, hwp] — about 1 GiB together whatever the clip (720p: one frame = 0.8 + 0.4 GB, where all 21 frames at once were
        # 17 + 9 GB transient, growing with batch and frames, beside two resident A14B experts).  One frame's GEMMs already
        # fill the chip (480p: 6240 x 6240 x 384 = 600 tiles).
        per = 6 * hw * hwp                                                     # bytes of one frame's scores + probabilities
        c = max(1, min(n, self.chunk_bytes // per))
        s = torch.empty((c, hw, hw), dtype=torch.float32, device=x.device)
        p = torch.empty((c * hw, hwp), dtype=x.dtype, device=x.device)
        o = torch.empty((n, hw, C), dtype=x.dtype, device=x.device)
        for f0 in range(0, n, c):
            m = min(c, n - f0)
            K.gemm_bf16_batched(q[f0:f0 + m], k[f0:f0 + m], out_dtype=torch.float32, out=s[:m])
            K.softmax_rows(s[:m].view(m * hw, hw), C ** -0.5, out=p[:m * hw, :hw], out_dtype=x.dtype)   # zero-fills the padded tail
            K.gemm_bf16_batched(p[:m * hw].view(m, hw, hwp), vt[f0:f0 + m], bias=self.bv, out=o[f0:f0 + m])
        del s, p
        return pointwise_conv(K, o.reshape(B, T, H, W, C), self.wp, self.bp, res=x)

    chunk_bytes = 1 << 30


class _HipUp:
    def __init__(self, g, K):
        self.K = K
        self.w, self.b = _k2d(g("resample.1.weight")), g("resample.1.bias")
        wt = g("time_conv.weight", None)
        self.wt, self.bt = (None, None) if wt is None else (_k2d(wt), g("time_conv.bias"))

    def __call__(self, x):
        K = self.K
        B, T, H, W, C = x.shape
        if self.wt is not None and T > 1:
            y = torch.empty((B, 1 + 2 * (T - 1), H, W, C), dtype=x.dtype, device=x.device)
            y[:, 0] = x[:, 0]
            K.vae_conv(x[:, 1:], self.wt, self.bt, 3, 1, 1, interleave=True, out=y[:, 1:])
            x = y
        return K.vae_conv(x, self.w, self.b, 1, 3, 3, up2=True)


def synthetic_state_dict(dim=96, z_dim=16, seed=0, dtype=torch.bfloat16, device="cpu"):
    """Random-init decoder weights of the named architecture in the reference's key layout (``_video_vae``: dim 96,
    dim_mult [1, 2, 4, 4], 2 (+1) residual blocks per level, temporal up-sampling at the first two levels,
    wan2pt1.py:565-574) — what the timing tool and the HIP-size fixture use (there is no network for checkpoints); values are
    representable in bf16 so that an fp32 reference and the bf16 kernels see the same weights."""
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

    conv("conv2", z_dim, z_dim, (1, 1, 1))
    dims = [dim * 4, dim * 4, dim * 4, dim * 2, dim]
    conv("decoder.conv1", dims[0], z_dim, (3, 3, 3))
    res("decoder.middle.0.", dims[0], dims[0])
    gamma("decoder.middle.1.norm.gamma", dims[0], 2)
    conv("decoder.middle.1.to_qkv", 3 * dims[0], dims[0], (1, 1))
    conv("decoder.middle.1.proj", dims[0], dims[0], (1, 1))
    res("decoder.middle.2.", dims[0], dims[0])
    n = 0
    for i, (a, b) in enumerate(zip(dims[:-1], dims[1:])):
        if i in (1, 2, 3):
            a = a // 2
        for _ in range(3):
            res(f"decoder.upsamples.{n}.", a, b)
            a = b
            n += 1
        if i != 3:
            conv(f"decoder.upsamples.{n}.resample.1", b // 2, b, (3, 3))
            if i < 2:
                conv(f"decoder.upsamples.{n}.time_conv", 2 * b, b, (3, 1, 1))
            n += 1
    gamma("decoder.head.0.gamma", dims[-1], 3)
    conv("decoder.head.2", 3, dims[-1], (3, 3, 3))
    return sd


class WanVaeDecoder:
    """``decode(z)``: normalised latents (what the sampler returns) -> video in about [-1, 1], dtype of ``z``
    (WanVAE.decode, wan2pt1.py:674-681).  ``state_dict``: the reference VAE's (``WanVAE_``); only ``conv2.*`` and
    ``decoder.*`` are used.  bf16 on a GPU (the reference interface's dtype, wan2pt1.py:686-690); anything else raises."""

    def __init__(self, state_dict, dtype=torch.bfloat16, device="cuda", mean=LATENT_MEAN, std=LATENT_STD):
        self.dtype, self.device = dtype, torch.device(device)
        if dtype != torch.bfloat16 or self.device.type != "cuda":
            raise ValueError("WanVaeDecoder runs on the HIP kernels only: bf16 on a GPU (the reference interface's dtype, "
                             "wan2pt1.py:686-690); the library-operator restatement for CPU checks is oracle/f4_ref.py")
        from . import kernels as K_     # raises if the HIP library is missing
        sd = {k: v.detach().to(device=self.device, dtype=dtype) for k, v in state_dict.items()
              if k.startswith(("decoder.", "conv2."))}
        if "decoder.conv1.weight" not in sd or "conv2.weight" not in sd:
            raise ValueError("not a Wan VAE state dict: decoder.conv1.weight / conv2.weight missing")
        self.z_dim = sd["conv2.weight"].shape[0]
        self.mean, self.inv_scale = latent_stats(mean, std, self.z_dim, dtype, self.device)
        self.sd = sd

        def getter(prefix):
            def g(name, *default):
                key = prefix + name
                if key in sd:
                    return sd[key]
                if default:
                    return default[0]
                raise KeyError(key)
            return g

        self.stages = [_HipRes(getter("decoder.middle.0."), K_), _HipFrameAttention(getter("decoder.middle.1."), K_),
                       _HipRes(getter("decoder.middle.2."), K_)]
        idx = sorted({int(m.group(1)) for k in sd for m in [re.match(r"decoder\.upsamples\.(\d+)\.", k)] if m})
        for i in idx:
            p = f"decoder.upsamples.{i}."
            if p + "residual.0.gamma" in sd:
                self.stages.append(_HipRes(getter(p), K_))
            elif p + "resample.1.weight" in sd:
                self.stages.append(_HipUp(getter(p), K_))
            elif p + "to_qkv.weight" in sd:
                self.stages.append(_HipFrameAttention(getter(p), K_))
            else:
                raise ValueError(f"unrecognised decoder stage {p}*")
        self.t_up = sum(1 for s in self.stages if isinstance(s, _HipUp) and s.wt is not None)
        self.s_up = sum(1 for s in self.stages if isinstance(s, _HipUp))
        self.K = K_
        zc = self.z_dim
        zp = -(-zc // 32) * 32              # td_vae_conv wants C_in % 32 == 0: zero channels up to there, in and out of conv2
        w2 = torch.zeros(zp, zp, dtype=dtype, device=self.device)
        b2 = torch.zeros(zp, dtype=dtype, device=self.device)
        w2[:zc, :zc], b2[:zc] = sd["conv2.weight"].reshape(zc, zc), sd["conv2.bias"]
        self.zp = zp
        w1 = sd["decoder.conv1.weight"]
        w1p = torch.zeros(w1.shape[0], zp, *w1.shape[2:], dtype=dtype, device=self.device)
        w1p[:, :zc] = w1
        self.h_conv2, self.h_conv1 = (w2, b2), (_k2d(w1p), sd["decoder.conv1.bias"])
        self.h_head = (sd["decoder.head.0.gamma"].reshape(-1).contiguous(), _k2d(sd["decoder.head.2.weight"]), sd["decoder.head.2.bias"])

    @classmethod
    def from_reference(cls, vae_module_or_state_dict, **kw):
        sd = vae_module_or_state_dict if isinstance(vae_module_or_state_dict, dict) else vae_module_or_state_dict.state_dict()
        return cls(sd, **kw)

    def pixel_frames(self, latent_frames: int) -> int:
        return (latent_frames - 1) * 2 ** self.t_up + 1     # get_pixel_num_frames, wan2pt1.py:711-712

    @torch.no_grad()
    def decode(self, z):
        in_dtype = z.dtype
        K = self.K
        x = z.to(device=self.device, dtype=self.dtype)
        x = x / self.inv_scale + self.mean                                     # WanVAE_.decode, wan2pt1.py:523-526
        xz = torch.zeros(x.shape[0], *x.shape[2:], self.zp, dtype=self.dtype, device=self.device)
        xz[..., :self.z_dim] = x.permute(0, 2, 3, 4, 1)                        # channels-last from here on, 16 -> 32 zero-padded
        x = pointwise_conv(K, xz, *self.h_conv2)                               # conv2 is 1x1x1
        x = K.vae_conv(x, self.h_conv1[0], self.h_conv1[1], 3, 3, 3)
        for st in self.stages:
            x = st(x)
        g, w, b = self.h_head
        x = K.vae_conv(K.vae_chan_rms(x, g), w, b, 3, 3, 3)                # [B, T, H, W, 3]
        return x.permute(0, 4, 1, 2, 3).contiguous().to(in_dtype)
