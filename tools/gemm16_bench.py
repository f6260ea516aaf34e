"""td_gemm_bf16 against the library GEMM (hipBLASLt through F.linear) on the shapes of this repo's 16-bit linears: BASELINE
config 3's blocks at L = 32 760 (q|k|v, o, ffn.0 + GELU, ffn.2), the text MLP, umT5-XXL's linears at 64 / 512 tokens.
One JSON line per shape: µs, TFLOP/s, fraction of the 2.5 PFLOP/s bf16 peak."""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from turbodiffusion_amd import kernels as K  # noqa: E402

DEV = "cuda"


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    g = torch.Generator(device=DEV).manual_seed(0)
    shapes = [("c3 qkv", 32760, 4608, 1536, "none"), ("c3 o", 32760, 1536, 1536, "none"), ("c3 ffn.0+gelu", 32760, 8960, 1536, "gelu_tanh"),
              ("c3 ffn.2", 32760, 1536, 8960, "none"), ("text mlp 0", 512, 1536, 4096, "gelu_tanh"), ("umt5 qkv n=64", 64, 12288, 4096, "none"),
              ("umt5 gate|fc1 n=512", 512, 20480, 4096, "none"), ("umt5 fc2 n=512", 512, 4096, 10240, "none"),
              ("14B ffn.0", 75600, 13824, 5120, "gelu_tanh")]
    for name, m, n, k, epi in shapes:
        a = torch.randn(m, k, device=DEV, generator=g).bfloat16()
        w = (torch.randn(n, k, device=DEV, generator=g) / k ** 0.5).bfloat16()
        b = (0.1 * torch.randn(n, device=DEV, generator=g)).bfloat16()
        ts = {}
        for variant in (1, 2):      # eight waves of 128x64 | four waves of 128x128 (forced; production picks by problem size)
            K.set_tuning(K.TUNE_GEMM16, variant)
            ts[variant] = timed(lambda: K.gemm_bf16(a, w, b, epilogue=epi))
        K.set_tuning(K.TUNE_GEMM16, 0)
        t_h = timed(lambda: K.gemm_bf16(a, w, b, epilogue=epi))
        if epi == "gelu_tanh":
            t_l = timed(lambda: F.gelu(F.linear(a, w, b), approximate="tanh"))
        else:
            t_l = timed(lambda: F.linear(a, w, b))
        fl = 2.0 * m * n * k
        y, yl = K.gemm_bf16(a, w, b, epilogue=epi).float(), (F.gelu(F.linear(a, w, b), approximate="tanh") if epi == "gelu_tanh" else F.linear(a, w, b)).float()
        print(json.dumps({"shape": name, "m": m, "n": n, "k": k, "epilogue": epi, "td_gemm_bf16_us": round(t_h, 1),
                          "eight_wave_us": round(ts[1], 1), "four_wave_us": round(ts[2], 1),
                          "library_us": round(t_l, 1), "td_TFLOPs": round(fl / t_h / 1e6, 1), "td_frac_of_2500": round(fl / t_h / 1e6 / 2500, 3),
                          "library_TFLOPs": round(fl / t_l / 1e6, 1), "rel_l2_vs_library": float(((y - yl).norm() / yl.norm()).item())}), flush=True)
        del a, w, y, yl


if __name__ == "__main__":
    main()
