"""Generate tests/golden/f4_vae_umt5.pt: toy-size instances of the REFERENCE's own VAE decoder (rcm/tokenizers/wan2pt1.py,
chunked ``WanVAE_.decode``) and umT5 encoder (rcm/utils/umt5.py ``T5Encoder`` + the zero padding of
``UMT5EncoderModel.__call__``), run here on the CPU with seeded random weights.  The fixture carries the weights (the
reference's state-dict layout), the inputs and the reference's outputs, so that the GPU box — which has no /root/reference —
can check turbodiffusion_amd.vae_decode / text_encoder against them.

    python -m oracle.make_golden_f4

TEST INFRASTRUCTURE (oracle/__init__.py)."""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import ref_harness as rh  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "f4_vae_umt5.pt")


def randomise(module, seed):
    """non-degenerate parameters everywhere (the reference zero-initialises e.g. the attention output projection)"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in module.named_parameters():
            if "gamma" in n or ("norm" in n and p.dim() == 1):
                p.copy_(1 + 0.2 * torch.randn(p.shape, generator=g))
            elif "pos_embedding" in n:
                p.copy_(0.5 * torch.randn(p.shape, generator=g))
            elif p.dim() > 1:
                p.copy_(torch.randn(p.shape, generator=g) / p[0].numel() ** 0.5)
            else:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))


def main():
    v = rh.load_aux("tokenizers.wan2pt1")
    u = rh.load_aux("utils.umt5")
    g = torch.Generator().manual_seed(7)
    fx = {}
    # ---- VAE: dim 4 (channels 16, 16, 16, 8, 4), 16 latent channels with the real statistics, 4 latent frames -> 13 frames
    vae = v.WanVAE_(dim=4, z_dim=16, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
                    temperal_downsample=[False, True, True], dropout=0.0).eval()
    randomise(vae, 11)
    # the latent statistics of WanVAE (wan2pt1.py:607-645; its constructor wants a checkpoint, so they are repeated here)
    mean = torch.tensor([-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508, 0.4134, -0.0715, 0.5517,
                         -0.3632, -0.1922, -0.9497, 0.2503, -0.2921])
    std = torch.tensor([2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743, 3.2687, 2.1526, 2.8652, 1.5579,
                        1.6382, 1.1253, 2.8251, 1.9160])
    z = torch.randn(1, 16, 4, 5, 6, generator=g)
    with torch.no_grad():
        video = vae.decode(z, [mean, 1.0 / std])
    fx["vae"] = {"state_dict": {k: t.clone() for k, t in vae.state_dict().items() if k.startswith(("decoder.", "conv2."))},
                 "z": z, "video": video}
    # ---- VAE at the smallest size the HIP kernels take (channel counts multiples of 32: dim 32 -> 128, 128, 128, 64, 32):
    # weights = turbodiffusion_amd.vae_decode.synthetic_state_dict(dim=32, seed=21) (bf16-representable, rebuilt by the test
    # from the seed: 5 M parameters are not committed), loaded into the reference module; 3 latent frames of 6 x 5 -> 9 frames
    # of 48 x 40: position counts that are not multiples of the 256-row tile at every level
    from turbodiffusion_amd.vae_decode import synthetic_state_dict
    sd32 = synthetic_state_dict(dim=32, z_dim=16, seed=21, dtype=torch.float32)
    vae2 = v.WanVAE_(dim=32, z_dim=16, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
                     temperal_downsample=[False, True, True], dropout=0.0).eval()
    missing, unexpected = vae2.load_state_dict(sd32, strict=False)
    assert not unexpected and all(k.startswith(("encoder.", "conv1.")) for k in missing), (missing[:4], unexpected[:4])
    z2 = torch.randn(2, 16, 3, 6, 5, generator=g).bfloat16().float()
    with torch.no_grad():
        video2 = vae2.decode(z2, [mean, 1.0 / std])
    fx["vae_hip_size"] = {"dim": 32, "seed": 21, "z": z2, "video": video2}
    # ---- the encoder at the HIP size: weights = turbodiffusion_amd.vae_encode.synthetic_state_dict(dim=32, seed=23); 5 frames of
    # 48 x 40 -> 2 latent frames of 6 x 5 (reference: chunked WanVAE_.encode)
    from turbodiffusion_amd.vae_encode import synthetic_state_dict as enc_sd
    sde = enc_sd(dim=32, z_dim=16, seed=23, dtype=torch.float32)
    missing, unexpected = vae2.load_state_dict(sde, strict=False)
    assert not unexpected and all(k.startswith(("decoder.", "conv2.")) for k in missing), (missing[:4], unexpected[:4])
    xv = torch.randn(2, 3, 5, 48, 40, generator=g).clamp(-1, 1).bfloat16().float()
    with torch.no_grad():
        lat = vae2.encode(xv, [mean, 1.0 / std])
    fx["vae_enc_hip_size"] = {"dim": 32, "seed": 23, "video": xv, "latent": lat}
    # ---- umT5: 3 layers, dim 64, 4 heads, per-layer position tables (umT5), prompts of 17 / 40 / 1 / 5 tokens padded to 40
    enc = u.T5Encoder(vocab=97, dim=64, dim_attn=48, dim_ffn=160, num_heads=4, num_layers=3, num_buckets=32,
                      shared_pos=False, dropout=0.1).eval()
    randomise(enc, 13)
    lens, Lp = [17, 40, 1, 5], 40
    ids = torch.randint(1, 97, (4, Lp), generator=g)
    mask = torch.zeros(4, Lp, dtype=torch.long)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
        ids[b, n:] = 0
    outs = {}
    sd32 = {k: t.clone() for k, t in enc.state_dict().items()}
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        e = copy.deepcopy(enc).to(dt)             # (Module.to converts in place)
        with torch.no_grad():
            ctx = e(ids, mask)
        ref = torch.zeros_like(ctx)
        for b, n in enumerate(lens):
            ref[b, :n] = ctx[b, :n]                 # UMT5EncoderModel.__call__, umt5.py:510-521
        outs[name] = ref
    fx["umt5"] = {"state_dict": sd32, "ids": ids, "mask": mask,
                  "out_f32": outs["f32"], "out_bf16": outs["bf16"]}
    torch.save(fx, OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes; video", tuple(video.shape))


if __name__ == "__main__":
    main()
