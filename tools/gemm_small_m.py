"""The W8A8 GEMMs of one DiT block at the per-rank row counts of a sequence split (M = L / N ranks), fused epilogues included:
the 256x256-tile kernel (TD_TUNE_GEMM_VARIANT = 4), its 128-row form (6, round 5), the 128x128 kernel (1; csrc/gemm_w8a8.hip:
G_RES / G_STATS / G_QOUT) and the automatic choice (0).  DESIGN §6's table comes from here.

    python tools/gemm_small_m.py [--M 4096,8192,16384]
"""
import argparse
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from turbodiffusion_amd import kernels as K  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--M", default="4096,8192,16384,32760")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--dim", type=int, default=1536)
ap.add_argument("--ffn", type=int, default=8960)
args = ap.parse_args()
ap2 = None
dev, dim, ffn = "cuda", args.dim, args.ffn
# 4 = 256x256 tiles only, 6 = the 128-row form of the same kernel (round 5), 1 = the 128x128 kernel, 0 = the shipped choice
VARIANTS = ((4, "tile256"), (6, "tile128x256"), (1, "tile128"), (0, "auto"))


def timeit(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3     # us


def wq(n, k):
    return K.quant_i8_block128((torch.randn(n, k, device=dev) / math.sqrt(k)).bfloat16())


w_qkv, w_o, w_f0, w_f2 = wq(3 * dim, dim), wq(dim, dim), wq(ffn, dim), wq(dim, ffn)
b = {n: (torch.randn(n, device=dev) * 0.1).bfloat16() for n in (dim, 3 * dim, ffn)}
gate = torch.randn(1, dim, device=dev)
for M in [int(v) for v in args.M.split(",")]:
    a = K.quant_i8_block128(torch.randn(M, dim, device=dev).bfloat16())
    h = K.quant_i8_block128(torch.randn(M, ffn, device=dev).bfloat16())
    x = torch.randn(M, dim, device=dev).bfloat16()
    ops = {
        "qkv (plain)": lambda: K.gemm_w8a8(a[0], a[1], w_qkv[0], w_qkv[1], torch.bfloat16, bias=b[3 * dim]),
        "o (RES+STATS)": lambda: K.gemm_w8a8_stats(a[0], a[1], w_o[0], w_o[1], b[dim], x=x, gate=gate),
        "cross-q (STATS)": lambda: K.gemm_w8a8_stats(a[0], a[1], w_o[0], w_o[1], b[dim]),
        "cross-o (RES+STATS)": lambda: K.gemm_w8a8_stats(a[0], a[1], w_o[0], w_o[1], b[dim], x=x, gate=None),
        "ffn.0 (GELU+QOUT)": lambda: K.gemm_w8a8_quant(a[0], a[1], w_f0[0], w_f0[1], torch.bfloat16, bias=b[ffn], gelu_tanh=True),
        "ffn.2 (RES+STATS)": lambda: K.gemm_w8a8_stats(h[0], h[1], w_f2[0], w_f2[1], b[dim], x=x, gate=gate),
    }
    row = {"M": M}
    best = {}
    for rnd in range(2):                 # the two kernels interleaved per operator, best of two rounds (clock ramps)
        for name, fn in ops.items():
            for variant, tag in VARIANTS:
                K.set_tuning(K.TUNE_GEMM_VARIANT, variant)
                t = timeit(fn, args.iters)
                key = f"{name} {tag} us"
                best[key] = min(best.get(key, 1e9), t)
    K.set_tuning(K.TUNE_GEMM_VARIANT, 0)
    for _, tag in VARIANTS:
        for name in ops:
            row[f"{name} {tag} us"] = round(best[f"{name} {tag} us"], 1)
        row[f"sum {tag} us"] = round(sum(best[f"{name} {tag} us"] for name in ops), 1)
    print(json.dumps(row), flush=True)
