"""f4 at full size on the MI355X: the whole-clip VAE decode of one 81-frame clip (480p and 720p latents, bf16, random weights
of the real architecture: dim 96, 16 latent channels) and the umT5-XXL encoder (24 layers, dim 4096, 64 heads, ffn 10240,
bf16, random weights) on prompts of 32 / 128 / 512 tokens.  Prints one JSON line per measurement."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from turbodiffusion_amd.text_encoder import Umt5Encoder, synthetic_state_dict as t5_state_dict  # noqa: E402
from turbodiffusion_amd.vae_decode import WanVaeDecoder, synthetic_state_dict as vae_state_dict  # noqa: E402

DEV = "cuda"


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return min(ts)


def main():
    which = sys.argv[1:] or ["vae480", "vae720", "enc480", "umt5"]
    if os.environ.get("F4_VAE_CONV"):        # A/B of the convolution kernels: TD_TUNE_VAE_CONV value
        from turbodiffusion_amd import kernels as K0
        K0.set_tuning(K0.TUNE_VAE_CONV, int(os.environ["F4_VAE_CONV"]))
        print(f"TD_TUNE_VAE_CONV = {os.environ['F4_VAE_CONV']}", flush=True)
    if any(w.startswith("vae") for w in which):
        dec = WanVaeDecoder(vae_state_dict(), dtype=torch.bfloat16, device=DEV)          # HIP backend
        for tag, (h, w) in (("vae480", (480, 832)), ("vae720", (720, 1280))):
            if tag not in which:
                continue
            z = torch.randn(1, 16, 21, h // 8, w // 8, device=DEV)
            torch.cuda.reset_peak_memory_stats()
            t = timed(lambda: dec.decode(z), reps=2)
            # the convolutions' FLOPs (2 * positions * C_out * taps * C_in) and time, from one decode with every call timed
            from turbodiffusion_amd import kernels as K
            fl, evs, real = [0.0], [], K.vae_conv

            def counted(x, w2d, bias, kt, kh, kw, res=None, up2=False, interleave=False, out=None):
                B_, T_, H_, W_, _ = x.shape
                fl[0] += 2.0 * B_ * T_ * (4 if up2 else 1) * H_ * W_ * w2d.shape[0] * w2d.shape[1]
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                r = real(x, w2d, bias, kt, kh, kw, res=res, up2=up2, interleave=interleave, out=out)
                b.record()
                evs.append((a, b, (tuple(x.shape), w2d.shape[0], (kt, kh, kw), bool(up2), bool(interleave), res is not None),
                            2.0 * B_ * T_ * (4 if up2 else 1) * H_ * W_ * w2d.shape[0] * w2d.shape[1]))
                return r

            K.vae_conv = counted
            v = dec.decode(z)
            torch.cuda.synchronize()
            K.vae_conv = real
            conv_s = sum(e[0].elapsed_time(e[1]) for e in evs) * 1e-3
            if os.environ.get("F4_DETAIL"):
                for e in evs:
                    ms = e[0].elapsed_time(e[1])
                    print(f"  conv x{e[2][0]} -> Co {e[2][1]} k{e[2][2]} up2={e[2][3]} interleave={e[2][4]} res={e[2][5]}: {ms:.2f} ms, {e[3] / ms / 1e9:.0f} TFLOP/s", flush=True)
            print(json.dumps({"what": f"WanVaeDecoder.decode (HIP kernels), whole clip, bf16, latent {tuple(z.shape)} -> video {tuple(v.shape)}",
                              "seconds": round(t, 3), "conv_launches": len(evs), "conv_seconds": round(conv_s, 3),
                              "conv_TFLOP": round(fl[0] / 1e12, 1), "conv_TFLOP_per_s": round(fl[0] / conv_s / 1e12, 1),
                              "conv_frac_of_bf16_peak_2500": round(fl[0] / conv_s / 2.5e15, 3), "peak_GiB": round(torch.cuda.max_memory_allocated() / 2**30, 1),
                              "finite": bool(torch.isfinite(v).all())}), flush=True)
            del v, z
            torch.cuda.empty_cache()
        del dec
        torch.cuda.empty_cache()
    if "enc480" in which:
        from turbodiffusion_amd.vae_encode import WanVaeEncoder, synthetic_state_dict as enc_state_dict
        enc = WanVaeEncoder(enc_state_dict(), dtype=torch.bfloat16, device=DEV)
        vid = torch.rand(1, 3, 81, 480, 832, device=DEV) * 2 - 1
        torch.cuda.reset_peak_memory_stats()
        t = timed(lambda: enc.encode(vid), reps=2)
        lat = enc.encode(vid)
        print(json.dumps({"what": f"WanVaeEncoder.encode (HIP kernels), whole clip, bf16, video {tuple(vid.shape)} -> latent {tuple(lat.shape)}",
                          "seconds": round(t, 3), "peak_GiB": round(torch.cuda.max_memory_allocated() / 2**30, 1),
                          "finite": bool(torch.isfinite(lat).all())}), flush=True)
        del enc, vid, lat
        torch.cuda.empty_cache()
    if "umt5" in which:
        enc = Umt5Encoder(t5_state_dict(device=DEV), dtype=torch.bfloat16, device=DEV)
        for n in (32, 128, 512):
            ids = torch.randint(1, 256384, (1, 512), device=DEV)
            mask = torch.zeros(1, 512, dtype=torch.long, device=DEV)
            mask[0, :n] = 1
            t = timed(lambda: enc(ids, mask))
            out = enc(ids, mask)
            print(json.dumps({"what": f"Umt5Encoder (XXL: 24 layers, dim 4096), prompt of {n} tokens padded to 512, bf16",
                              "ms": round(t * 1e3, 2), "finite": bool(torch.isfinite(out).all()),
                              "weights_GiB": round(torch.cuda.memory_allocated() / 2**30, 1)}), flush=True)


if __name__ == "__main__":
    main()
