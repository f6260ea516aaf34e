"""f4 on the CPU: the ORACLE's library-operator restatements (oracle/f4_ref.py: whole-clip VAE decoder /
encoder, valid-rows-only umT5 encoder — the graphs turbodiffusion_amd.vae_decode / vae_encode / text_encoder run on HIP kernels) against
  (a) the LIVE reference — ``WanVAE_.decode`` (chunked, rcm/tokenizers/wan2pt1.py:520-537) and ``T5Encoder``
      (rcm/utils/umt5.py:308-337) imported unmodified through oracle/ref_harness.py — where /root/reference exists, and
  (b) the committed fixture those same reference modules produced (oracle/make_golden_f4.py), everywhere.
fp32: equal to summation order (the whole-clip convolutions and the fused q|k|v / gate|fc1 GEMMs add in another order);
bf16 umT5: the reference's rounding points are kept, so the CPU result is bit-identical."""
import copy
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_harness as rh  # noqa: E402
from oracle.make_golden_f4 import randomise  # noqa: E402
from oracle.f4_ref import Umt5EncoderRef as Umt5Encoder, VaeDecoderRef as WanVaeDecoder, VaeEncoderRef as WanVaeEncoder  # noqa: E402
from turbodiffusion_amd.text_encoder import relative_buckets  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "f4_vae_umt5.pt")
needs_ref = pytest.mark.skipif(not rh.available(), reason="/root/reference not present (GPU box)")


@needs_ref
@pytest.mark.parametrize("dim,zd,T,H,W", [(8, 4, 3, 4, 4), (8, 4, 1, 4, 6), (16, 16, 5, 6, 4), (8, 4, 2, 3, 5)])
def test_whole_clip_vae_decode_equals_the_reference_chunked_decode(dim, zd, T, H, W):
    """One pass over all frames == the reference's frame-by-frame decode with its feature caches, incl. the first-frame rule of
    the temporal up-samplers (1 + 4 (T - 1) frames) and a single-frame clip."""
    v = rh.load_aux("tokenizers.wan2pt1")
    vae = v.WanVAE_(dim=dim, z_dim=zd, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
                    temperal_downsample=[False, True, True], dropout=0.0).eval()
    randomise(vae, dim + T)
    g = torch.Generator().manual_seed(T)
    z = torch.randn(2, zd, T, H, W, generator=g)
    mean, std = 0.5 * torch.randn(zd, generator=g), torch.rand(zd, generator=g) + 0.5
    with torch.no_grad():
        ref = vae.decode(z, [mean, 1.0 / std])
    dec = WanVaeDecoder.from_reference(vae, dtype=torch.float32, device="cpu", mean=mean.tolist(), std=std.tolist())
    out = dec.decode(z)
    assert out.shape == ref.shape == (2, 3, dec.pixel_frames(T), 8 * H, 8 * W) and dec.pixel_frames(T) == 1 + 4 * (T - 1)
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=2e-5)


@needs_ref
@pytest.mark.parametrize("dim,zd,T,H,W", [(8, 4, 9, 32, 32), (8, 4, 1, 16, 24), (16, 16, 5, 32, 16), (8, 4, 13, 16, 16)])
def test_whole_clip_vae_encode_equals_the_reference_chunked_encode(dim, zd, T, H, W):
    """One pass over all frames == the reference's 1 + 4 + 4 + ... chunked encode with its feature caches, incl. the temporal
    down-samplers' first-frame rule (the first frame passes, frames 1.. come from stride-2 windows starting at even frames)."""
    v = rh.load_aux("tokenizers.wan2pt1")
    vae = v.WanVAE_(dim=dim, z_dim=zd, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
                    temperal_downsample=[False, True, True], dropout=0.0).eval()
    randomise(vae, dim + T)
    g = torch.Generator().manual_seed(T)
    x = torch.randn(2, 3, T, H, W, generator=g)
    mean, std = 0.5 * torch.randn(zd, generator=g), torch.rand(zd, generator=g) + 0.5
    with torch.no_grad():
        ref = vae.encode(x, [mean, 1.0 / std])
    enc = WanVaeEncoder.from_reference(vae, dtype=torch.float32, device="cpu", mean=mean.tolist(), std=std.tolist())
    out = enc.encode(x)
    assert out.shape == ref.shape == (2, zd, enc.latent_frames(T), H // 8, W // 8) and enc.latent_frames(T) == 1 + (T - 1) // 4
    torch.testing.assert_close(out, ref, rtol=1e-5, atol=2e-5)
    with pytest.raises(ValueError):
        enc.encode(torch.zeros(1, 3, T + 2, H, W))          # not 1 + 4k frames


@needs_ref
@pytest.mark.parametrize("shared", [False, True])
def test_umt5_encoder_equals_the_reference_encoder(shared):
    u = rh.load_aux("utils.umt5")
    enc = u.T5Encoder(vocab=97, dim=64, dim_attn=48, dim_ffn=160, num_heads=4, num_layers=3, num_buckets=32,
                      shared_pos=shared, dropout=0.1).eval()
    randomise(enc, 5)
    g = torch.Generator().manual_seed(3)
    lens, Lp = [17, 40, 1, 5], 40
    ids = torch.randint(1, 97, (4, Lp), generator=g)
    mask = torch.zeros(4, Lp, dtype=torch.long)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
    for dt in (torch.float32, torch.bfloat16):
        e = copy.deepcopy(enc).to(dt)             # (Module.to converts in place)
        with torch.no_grad():
            ctx = e(ids, mask)
        ref = torch.zeros_like(ctx)
        for b, n in enumerate(lens):
            ref[b, :n] = ctx[b, :n]
        out = Umt5Encoder.from_reference(e, dtype=dt, device="cpu")(ids, mask)
        if dt == torch.bfloat16:
            assert torch.equal(out, ref)          # same rounding points: bit for bit
        else:
            torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)
        # the reference's own position table for this length
        rel = e.blocks[0].pos_embedding if not shared else e.pos_embedding
        ar = torch.arange(Lp)
        assert torch.equal(relative_buckets(Lp, 32, 128, "cpu"), rel._relative_position_bucket(ar[None, :] - ar[:, None]))


def test_f4_modules_against_the_reference_fixture():
    fx = torch.load(GOLD)
    dec = WanVaeDecoder(fx["vae"]["state_dict"], dtype=torch.float32, device="cpu")
    torch.testing.assert_close(dec.decode(fx["vae"]["z"]), fx["vae"]["video"], rtol=1e-5, atol=2e-5)
    from turbodiffusion_amd.vae_encode import synthetic_state_dict as enc_sd
    from turbodiffusion_amd.vae_decode import synthetic_state_dict as dec_sd
    hd, he = fx["vae_hip_size"], fx["vae_enc_hip_size"]     # the seeded weights the GPU tests rebuild reproduce the fixture
    dec2 = WanVaeDecoder(dec_sd(dim=hd["dim"], seed=hd["seed"], dtype=torch.float32), dtype=torch.float32, device="cpu")
    torch.testing.assert_close(dec2.decode(hd["z"]), hd["video"], rtol=1e-5, atol=2e-5)
    enc2 = WanVaeEncoder(enc_sd(dim=he["dim"], seed=he["seed"], dtype=torch.float32), dtype=torch.float32, device="cpu")
    torch.testing.assert_close(enc2.encode(he["video"]), he["latent"], rtol=1e-5, atol=2e-5)
    t5 = fx["umt5"]
    enc = Umt5Encoder(t5["state_dict"], dtype=torch.float32, device="cpu")
    assert (enc.num_heads, enc.num_buckets, enc.dim, enc.dim_ffn, len(enc.layers), enc.shared_pos) == (4, 32, 64, 160, 3, False)
    torch.testing.assert_close(enc(t5["ids"], t5["mask"]), t5["out_f32"], rtol=1e-5, atol=1e-5)
    enc16 = Umt5Encoder(t5["state_dict"], dtype=torch.bfloat16, device="cpu")
    assert torch.equal(enc16(t5["ids"], t5["mask"]), t5["out_bf16"])
    with pytest.raises(ValueError):
        bad = t5["mask"].clone()
        bad[0, 0] = 0                         # a hole inside the valid prefix: not what the tokenizer produces
        enc(t5["ids"], bad)
    with pytest.raises(ValueError):
        WanVaeDecoder({"x": torch.zeros(1)}, device="cpu")


def test_product_f4_classes_are_hip_only():
    """The product classes have ONE backend: without a GPU (or in a dtype the kernels do not take) the constructors raise —
    the library-operator graphs above live under oracle/ (round-3 verdict: no silent second backend)."""
    from turbodiffusion_amd.text_encoder import Umt5Encoder as E
    from turbodiffusion_amd.vae_decode import WanVaeDecoder as D, synthetic_state_dict as dec_sd
    from turbodiffusion_amd.vae_encode import WanVaeEncoder as V, synthetic_state_dict as enc_sd
    fx = torch.load(GOLD)
    for make in (lambda: D(dec_sd(dim=32), device="cpu"), lambda: V(enc_sd(dim=32), device="cpu"),
                 lambda: E(fx["umt5"]["state_dict"], device="cpu"),
                 lambda: D(dec_sd(dim=32), dtype=torch.float32, device="cuda"),
                 lambda: E(fx["umt5"]["state_dict"], dtype=torch.float32, device="cuda")):
        with pytest.raises(ValueError, match="HIP kernels only"):
            make()


def test_latent_statistics_are_formed_like_the_reference():
    """WanVAE builds (mean, 1 / std) as ``torch.tensor(std, dtype=dtype)`` then ``1.0 / self.std`` (wan2pt1.py:643-645): in
    bf16 the reciprocal of the ROUNDED std, taken in bf16 — not the rounded fp64 reciprocal (6 of the 16 real channels differ by
    one ulp).  Product helper and oracle agree with that construction bit for bit."""
    from oracle.f4_ref import _stats
    from turbodiffusion_amd.vae_decode import LATENT_MEAN, LATENT_STD, latent_stats
    ref_inv = 1.0 / torch.tensor(LATENT_STD, dtype=torch.bfloat16)
    naive = torch.tensor([1.0 / s for s in LATENT_STD], dtype=torch.bfloat16)
    assert int((ref_inv != naive).sum()) == 6
    for fn in (latent_stats, _stats):
        m, inv = fn(LATENT_MEAN, LATENT_STD, 16, torch.bfloat16, "cpu")
        assert torch.equal(inv.flatten(), ref_inv) and torch.equal(m.flatten(), torch.tensor(LATENT_MEAN, dtype=torch.bfloat16))
    if rh.available():     # and the reference's own decode / encode in bf16 with those statistics use exactly these tensors
        z = torch.randn(1, 16, 1, 2, 2).bfloat16()
        assert torch.equal(z / ref_inv.view(1, 16, 1, 1, 1), z / inv)
