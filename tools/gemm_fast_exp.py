"""One-VALU ("fast") dequant of the W8A8 GEMM against the exact kernels: time, error and the stated bound, for the
four GEMM shapes of a Wan2.1-1.3B block, both 256x256 kernels (variant 4 = 16x16x64 MFMA, 5 = 32x32x32 MFMA).

    python tools/gemm_fast_exp.py [--iters 10] [--L 32760]
"""
import argparse
import ctypes
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from turbodiffusion_amd import kernels as K, _lib as L  # noqa: E402
from tools.kbench import timeit  # noqa: E402

dev = "cuda"


def act(m, k, seed, outliers):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(m, k, device=dev, generator=g)
    if outliers:
        idx = torch.randperm(k, device=dev, generator=g)[: max(1, k // 1000 + 1)]
        x[:, idx] *= 20.0
    return x.bfloat16()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--L", type=int, default=32760)
    ap.add_argument("--dim", type=int, default=1536)
    ap.add_argument("--ffn", type=int, default=8960)
    args = ap.parse_args()
    Lr, dim, ffn = args.L, args.dim, args.ffn
    out = []
    for (n, k, nm, kind) in ((dim, dim, "o-proj", "res"), (3 * dim, dim, "qkv", "plain"), (ffn, dim, "ffn1", "quant"),
                             (dim, ffn, "ffn2", "res")):
        for outl in (False, True):
            a = act(Lr, k, 1, outl)
            aq, as_ = K.quant_i8_block128(a)
            wq, ws = K.quant_i8_block128((torch.randn(n, k, device=dev) / math.sqrt(k)).bfloat16())
            b = (torch.randn(n, device=dev) * 0.05).bfloat16()
            x0 = torch.randn(Lr, n, device=dev).bfloat16()
            gate = torch.randn(1, n, device=dev) * 0.5

            def run():
                if kind == "plain":
                    return K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b)
                if kind == "quant":
                    return K.gemm_w8a8_quant(aq, as_, wq, ws, torch.bfloat16, bias=b, gelu_tanh=True)
                return K.gemm_w8a8_residual_(x0.clone(), aq, as_, wq, ws, bias=b, gate=gate)

            def run_t():   # timing: no clone in the timed call
                if kind == "res":
                    return K.gemm_w8a8_residual_(x0, aq, as_, wq, ws, bias=b, gate=gate)
                return run()

            # the plain 16-bit result of the exact kernel and the bound 0.75 (G+1) sum_k s_k per output element
            K.set_tuning(K.TUNE_GEMM_VARIANT, 4); K.set_tuning(K.TUNE_GEMM_FAST, 1)
            y_ex = K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b).float()
            ssum = (as_[:, None, :] * ws[None, :, :]).sum(-1)          # [mb, nb]
            ref = run()
            for var in (4, 5):
                for G in (1, 2, 4, 8):
                    if outl and G == 1:
                        continue
                    K.set_tuning(K.TUNE_GEMM_VARIANT, var); K.set_tuning(K.TUNE_GEMM_FAST, G)
                    r = {"gemm": nm, "M": Lr, "N": n, "K": k, "kind": kind, "outliers": outl, "variant": var, "G": G}
                    try:
                        got = run()
                        y = K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b).float()
                        d = (y - y_ex).abs()
                        bound = 0.75 * (G + 1) * ssum.repeat_interleave(128, 0)[:Lr].repeat_interleave(128, 1)[:, :n]
                        ulp = torch.maximum(y_ex.abs(), torch.tensor(1e-30, device=dev)) * 2.0 ** -7
                        r.update({"plain_max_abs": d.max().item(), "plain_rel_l2": (d.norm() / y_ex.norm()).item(),
                                  "plain_frac_differ": (d > 0).float().mean().item(),
                                  "bound_violations": int((d > bound + ulp).sum().item()),
                                  "max_diff_over_bound": (d / (bound + ulp)).max().item()})
                        if kind == "quant":
                            r["codes_differ_frac"] = (got[0] != ref[0]).float().mean().item()
                            r["codes_max_diff"] = (got[0].int() - ref[0].int()).abs().max().item()
                            r["scales_differ_frac"] = (got[1] != ref[1]).float().mean().item()
                        elif kind == "res":
                            r["fused_frac_differ"] = (got != ref).float().mean().item()
                        if not outl:
                            t = timeit(run_t, args.iters)
                            r["us"] = round(t * 1e6, 1)
                            r["POPs"] = round(2.0 * Lr * n * k / t / 1e15, 3)
                    except Exception as e:  # noqa: BLE001
                        r["error"] = repr(e)[:300]
                    print(json.dumps(r), flush=True)
                    out.append(r)
            K.set_tuning(K.TUNE_GEMM_VARIANT, 0); K.set_tuning(K.TUNE_GEMM_FAST, 0)
    # phase stamps (prologue / main loop / epilogue) of variant 5: exact and G = 4
    for (abl, G, tag) in ((6, 1, "m32 exact"), (16, 4, "m32 fast G=4")):
        n, k = 4608, 1536
        a = act(Lr, k, 1, False)
        aq, as_ = K.quant_i8_block128(a)
        wq, ws = K.quant_i8_block128((torch.randn(n, k, device=dev) / math.sqrt(k)).bfloat16())
        b = torch.zeros(n, device=dev).bfloat16()
        K.set_tuning(K.TUNE_GEMM_VARIANT, 5); K.set_tuning(K.TUNE_GEMM_FAST, G); K.set_tuning(K.TUNE_GEMM_ABLATE, abl)
        for _ in range(3):
            K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 64)()
        L.call("td_debug_read", ctypes.cast(buf, ctypes.c_void_p), 64)
        K.set_tuning(K.TUNE_GEMM_ABLATE, 0); K.set_tuning(K.TUNE_GEMM_VARIANT, 0); K.set_tuning(K.TUNE_GEMM_FAST, 0)
        for r_ in range(4):
            t = [buf[r_ * 5 + i] for i in range(5)]
            print(json.dumps({"phases": tag, "round": r_, "prologue": t[1] - t[0], "main": t[2] - t[1],
                              "main_per_kblock": (t[2] - t[1]) / (k // 128), "epi_issue": t[3] - t[2],
                              "drain": t[4] - t[3], "total": t[4] - t[0]}), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/gemm_fast_exp.json", "w"), indent=1)


if __name__ == "__main__":
    main()
