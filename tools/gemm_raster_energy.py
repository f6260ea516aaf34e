"""Round-3 verdict, item 3 (ii): the W8A8 GEMM's tile raster judged by JOULES per launch and by the clock the NEXT kernel gets,
not by its own microseconds.  The m-grouped raster (TD_TUNE_GEMM_GROUP_M: consecutive workgroup ids walk `group_m` M tiles
for one N tile before moving to the next N tile; every XCD owns a contiguous M range) decides how often the A panel
(activations) is re-read against how often the B panel (weights) is: group_m = 1 is the N-major walk (A read once per N
tile = 35 x for ffn.0), group_m = tiles_m the M-major walk (B re-read per M tile).  The counters say ffn.0 fetches 699 MB
against 343 MB of operands (profiles/r04_pmc_hbm_traffic.json).  For every group_m: ffn.0 (GELU + quantiser epilogue) and
ffn.2 back to back, each launch timed with its own events (medians of 30), and the socket energy of ~1.2 s of ffn.0 alone."""
import math
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from turbodiffusion_amd import kernels as K  # noqa: E402
from tools.energy_per_launch import energy_uj  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    L, dim, ffn = 32760, 1536, 8960
    a0q, a0s = K.quant_i8_block128(torch.randn(L, dim, device=dev).bfloat16())
    w0q, w0s = K.quant_i8_block128((torch.randn(ffn, dim, device=dev) / math.sqrt(dim)).bfloat16())
    b0 = torch.zeros(ffn, device=dev).bfloat16()
    w2q, w2s = K.quant_i8_block128((torch.randn(dim, ffn, device=dev) / math.sqrt(ffn)).bfloat16())
    b2 = torch.zeros(dim, device=dev).bfloat16()
    x = torch.zeros(L, dim, device=dev).bfloat16()
    gate = torch.ones(1, dim, device=dev)

    def ffn0():
        return K.gemm_w8a8_quant(a0q, a0s, w0q, w0s, torch.bfloat16, bias=b0, gelu_tanh=True)

    def ffn2(hq, hs):
        K.gemm_w8a8_residual_(x, hq, hs, w2q, w2s, bias=b2, gate=gate)

    hq, hs = ffn0()
    print(f"{'group_m':>8s} {'ffn.0 us':>9s} {'ffn.2 after it us':>17s} {'sum us':>8s} {'ffn.0 W':>8s} {'ffn.0 J/launch':>14s}")
    for gm in (4, 1, 2, 8, 16, 32, 128):
        K.set_tuning(K.TUNE_GEMM_GROUP_M, gm)
        t0s, t2s = [], []
        for _ in range(33):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
            ffn0()
            e[1].record()
            K.set_tuning(K.TUNE_GEMM_GROUP_M, 0)        # the follower always runs the default raster
            ffn2(hq, hs)
            K.set_tuning(K.TUNE_GEMM_GROUP_M, gm)
            e[2].record()
            torch.cuda.synchronize()
            t0s.append(e[0].elapsed_time(e[1]) * 1e3)
            t2s.append(e[1].elapsed_time(e[2]) * 1e3)
        t0, t2 = statistics.median(t0s[3:]), statistics.median(t2s[3:])
        time.sleep(0.5)
        reps = int(1.2e6 / t0)
        e_a, ta = energy_uj(), time.perf_counter()
        for _ in range(reps):
            ffn0()
        torch.cuda.synchronize()
        dt = time.perf_counter() - ta
        e_b = energy_uj()
        print(f"{gm:8d} {t0:9.1f} {t2:17.1f} {t0 + t2:8.1f} {(e_b - e_a) * 1e-6 / dt:8.0f} {(e_b - e_a) * 1e-6 / reps:14.4f}", flush=True)
    K.set_tuning(K.TUNE_GEMM_GROUP_M, 0)


if __name__ == "__main__":
    main()
