"""A/B of the fused FFN GEMM's GELU (TD_TUNE_GELU_TABLE): device-built table lookup (default) vs inline evaluation — bit
identity of the outputs at the C1 and the rank-of-8 shapes and the launch time of ffn.0, interleaved, medians.

    python tools/gelu_table_ab.py [--reps 5]
"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from turbodiffusion_amd import kernels as K  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    for (m, n, k) in ((32760, 8960, 1536), (4096, 8960, 1536), (75600, 13824, 5120), (1000, 512, 256)):
        x = torch.randn(m, k, device=dev, generator=g).bfloat16()
        w = (torch.randn(n, k, device=dev, generator=g) / k ** 0.5).bfloat16()
        b = (torch.randn(n, device=dev, generator=g) * 0.5).bfloat16()
        xq, xs = K.quant_i8_block128(x)
        wq, ws = K.quant_i8_block128(w)
        outs = {}
        modes = (1, 0)
        times = {mo: [] for mo in modes}
        for rep in range(args.reps):
            for mode in modes:
                K.set_tuning(K.TUNE_GELU_TABLE, mode)
                for _ in range(2):
                    q, s = K.gemm_w8a8_quant(xq, xs, wq, ws, torch.bfloat16, bias=b, gelu_tanh=True)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    q, s = K.gemm_w8a8_quant(xq, xs, wq, ws, torch.bfloat16, bias=b, gelu_tanh=True)
                e1.record()
                e1.synchronize()
                times[mode].append(e0.elapsed_time(e1) / 10 * 1e3)
                outs[mode] = (q.clone(), s.clone())
        K.set_tuning(K.TUNE_GELU_TABLE, 0)
        fl = 2.0 * m * n * k
        t1 = statistics.median(times[1])
        for mo in modes:
            same = torch.equal(outs[mo][0], outs[1][0]) and torch.equal(outs[mo][1], outs[1][1])
            t0 = statistics.median(times[mo])
            name = {1: "inline", 0: "table (default)"}[mo]
            print(f"[{m} x {n} x {k}] {name:34s} {t0:8.1f} us ({fl / t0 / 1e12:6.3f} POP/s)  {(t1 / t0 - 1) * 100:+5.1f} % vs inline   "
                  f"bit-identical: {same}", flush=True)
            assert same


if __name__ == "__main__":
    main()
