"""Where ffn.0's time goes: the same [32760 x 8960 x 1536] W8A8 GEMM with each epilogue — plain 16-bit store, + GELU, fused
quantiser without / with GELU (inline and table) — and the [32760 x 4608 x 1536] q|k|v shape for the per-FLOP comparison.

    python tools/ffn0_epilogue_split.py
"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from turbodiffusion_amd import kernels as K  # noqa: E402


def t_us(fn, n=10, reps=5):
    out = []
    for _ in range(reps):
        fn(); fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        out.append(e0.elapsed_time(e1) / n * 1e3)
    return statistics.median(out)


def main():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    m, k = 32760, 1536
    x = torch.randn(m, k, device=dev, generator=g).bfloat16()
    xq, xs = K.quant_i8_block128(x)
    for n in (8960, 4608, 8192, 9216):
        w = (torch.randn(n, k, device=dev, generator=g) / k ** 0.5).bfloat16()
        b = (torch.randn(n, device=dev, generator=g) * 0.5).bfloat16()
        wq, ws = K.quant_i8_block128(w)
        fl = 2.0 * m * n * k
        rows = [("plain 16-bit store", lambda: K.gemm_w8a8(xq, xs, wq, ws, torch.bfloat16, bias=b)),
                ("16-bit store + GELU", lambda: K.gemm_w8a8(xq, xs, wq, ws, torch.bfloat16, bias=b, gelu_tanh=True)),
                ("fused quantiser, no GELU", lambda: K.gemm_w8a8_quant(xq, xs, wq, ws, torch.bfloat16, bias=b)),
                ("fused quantiser + GELU (table)", lambda: K.gemm_w8a8_quant(xq, xs, wq, ws, torch.bfloat16, bias=b, gelu_tanh=True))]
        for name, fn in rows:
            t = t_us(fn)
            print(f"[{m} x {n} x {k}] {name:34s} {t:8.1f} us  {fl / t / 1e9:7.1f} TOP/s  {t / n * 8960:8.1f} us per 8960 columns", flush=True)
        K.set_tuning(K.TUNE_GELU_TABLE, 1)
        t = t_us(rows[3][1])
        K.set_tuning(K.TUNE_GELU_TABLE, 0)
        print(f"[{m} x {n} x {k}] {'fused quantiser + GELU (inline)':34s} {t:8.1f} us  {fl / t / 1e9:7.1f} TOP/s  {t / n * 8960:8.1f} us per 8960 columns", flush=True)


if __name__ == "__main__":
    main()
