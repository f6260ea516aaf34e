"""Prompt token ids -> video: the three stages of the reference's one-shot script (``inference/wan2.1_t2v_infer.py``:
text embedding :73-76, sampling loop :95-140, VAE decode :141-149) on one GPU, all three resident (288 GB of HBM: no
``clear_umt5_memory`` / ``net.cpu()`` paging between the stages).

    text = Umt5Encoder(...)          # turbodiffusion_amd.text_encoder  (f4)
    net = GraphedModel(WanModel...)  # the hot path
    vae = WanVaeDecoder(...)         # turbodiffusion_amd.vae_decode    (f4)
    video = t2v(text, net, vae, ids, mask, height=480, width=832)

``ids`` may also be the prompt TEXT (a str or a list of str) when ``text_encoder`` is a ``text_encoder.UMT5EncoderModel`` (tokenizer +
encoder, the reference's class of that name): ``t2v(UMT5EncoderModel(...), net, vae, "a cat surfing a wave")``.  Video file
writing stays with the caller."""
from __future__ import annotations

from typing import Callable, Optional

import torch

from .sampler import rcm_sample


def latent_shape(num_frames: int, height: int, width: int, latent_ch: int = 16):
    """[C, T, H, W] of the latent for a clip (Wan2pt1VAEInterface: 4x in time after the first frame, 8x in space)."""
    return (latent_ch, 1 + (num_frames - 1) // 4, height // 8, width // 8)


def _embed(text_encoder, ids, mask):
    """token ids (+ mask) -> encoder(ids, mask); prompt text -> encoder(texts) (UMT5EncoderModel tokenises itself)"""
    if isinstance(ids, str) or (isinstance(ids, (list, tuple)) and ids and isinstance(ids[0], str)):
        return text_encoder([ids] if isinstance(ids, str) else list(ids))
    return text_encoder(ids, mask)


@torch.no_grad()
def t2v(text_encoder: Callable, net: Callable, vae, ids: torch.Tensor, mask: Optional[torch.Tensor] = None, *,
        height: int = 480, width: int = 832, num_frames: int = 81, num_steps: int = 4, sigma_max: float = 80.0,
        seed: int = 0, num_samples: int = 1, dtype=torch.bfloat16, device="cuda"):
    """Returns video [num_samples * B, 3, num_frames, height, width] in [0, 1] (the script's ``(1 + clamp(v, -1, 1)) / 2``)."""
    emb = _embed(text_encoder, ids, mask).to(device=device, dtype=dtype)          # [B, L_text, text_dim]
    emb = emb.repeat(num_samples, 1, 1)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    noise = torch.randn(emb.shape[0], *latent_shape(num_frames, height, width), dtype=torch.float32, device=device, generator=gen)
    z = rcm_sample(net, noise, emb, num_steps=num_steps, sigma_max=sigma_max, generator=gen, dtype=dtype)
    video = vae.decode(z)
    return (1.0 + video.float().clamp(-1, 1)) / 2.0


@torch.no_grad()
def i2v_condition(vae_enc, image: torch.Tensor, num_frames: int = 81, dtype=torch.bfloat16):
    """The conditioning channels ``y`` of Wan2.2 I2V (``inference/wan2.2_i2v_infer.py:139-152``): the first frame = the image
    ([B, 3, H, W] in [-1, 1]), the other ``num_frames - 1`` frames zero, through the VAE encoder; 4 mask channels (1 on the
    first latent frame) in front -> [B, 4 + 16, T_lat, H/8, W/8]."""
    B, C, H, W = image.shape
    frames = torch.zeros(B, C, num_frames, H, W, dtype=torch.float32, device=image.device)
    frames[:, :, 0] = image
    lat = vae_enc.encode(frames)
    msk = torch.zeros(B, 4, *lat.shape[2:], dtype=dtype, device=lat.device)
    msk[:, :, 0] = 1.0
    return torch.cat([msk, lat.to(dtype)], dim=1)


@torch.no_grad()
def i2v(text_encoder: Callable, net_high: Callable, net_low: Callable, vae_enc, vae_dec, ids: torch.Tensor,
        mask: Optional[torch.Tensor], image: torch.Tensor, *, num_frames: int = 81, num_steps: int = 4, sigma_max: float = 200.0,
        boundary: float = 0.9, seed: int = 0, dtype=torch.bfloat16, device="cuda"):
    """Image + prompt ids -> video in [0, 1]: the stages of ``inference/wan2.2_i2v_infer.py`` (text :96-99, conditioning
    :139-152, the loop with the expert switch at ``boundary`` :173-213, decode :214-222), both experts resident."""
    emb = _embed(text_encoder, ids, mask).to(device=device, dtype=dtype)
    y = i2v_condition(vae_enc, image.to(device), num_frames, dtype)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    noise = torch.randn(emb.shape[0], 16, *y.shape[2:], dtype=torch.float32, device=device, generator=gen)
    z = rcm_sample(net_high, noise, emb, num_steps=num_steps, sigma_max=sigma_max, generator=gen, y=y, net_low=net_low,
                   boundary=boundary, dtype=dtype)
    return (1.0 + vae_dec.decode(z).float().clamp(-1, 1)) / 2.0
