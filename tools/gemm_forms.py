"""Round 6: the block's six W8A8 GEMMs (C1 shapes, the epilogues the model launches them with) on each tile form and
dequant mode — microseconds AND joules per launch (the step is power-bound: a form that is faster alone but draws more
is paid back by the kernels after it, profiles/NOTES_r03.md).

    python tools/gemm_forms.py [--rows 32760] [--forms 4,8] [--fast 1,4] [--seconds 1.0] [--only ffn2,crossq]

form = TD_TUNE_GEMM_VARIANT (4: eight waves, 256 x 256 tile; 6: eight waves, 128 x 256; 8: FOUR waves, 128 x 256, two
workgroups per CU; 0: the launch planner), fast = TD_TUNE_GEMM_FAST (1 exact, 8 one-VALU dequant: the period instantiated for every epilogue).  One JSON line per
(GEMM, form, fast); a summary table at the end.  Also checks that every form gives the bits of form 4 (per dequant mode).
"""
import argparse
import json
import math
import os
import re
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from turbodiffusion_amd import kernels as K  # noqa: E402


def energy_uj():
    out = subprocess.run(["rocm-smi", "--showenergycounter"], capture_output=True, text=True).stdout
    m = re.search(r"Accumulated Energy \(uJ\):\s*([0-9.]+)", out)
    return float(m.group(1)) if m else float("nan")


def measure(fn, seconds):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(10):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    per = ev[0].elapsed_time(ev[1]) * 1e-3 / 10
    reps = max(20, int(seconds / per))
    time.sleep(0.3)
    e0 = energy_uj()
    t0 = time.perf_counter()
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    e1 = energy_uj()
    us = ev[0].elapsed_time(ev[1]) * 1e3 / reps
    return us, (e1 - e0) * 1e-6 / reps, (e1 - e0) * 1e-6 / dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=32760)
    ap.add_argument("--forms", default="4,8")
    ap.add_argument("--fast", default="1,8")
    ap.add_argument("--seconds", type=float, default=1.0)
    ap.add_argument("--only", default="")
    ap.add_argument("--dim", type=int, default=1536)
    ap.add_argument("--ffn", type=int, default=8960)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    L, dim, ffn = a.rows, a.dim, a.ffn
    forms = [int(v) for v in a.forms.split(",")]
    fasts = [int(v) for v in a.fast.split(",")]

    def operands(n, k):
        x = torch.randn(L, k, device=dev).bfloat16()
        xq, xs = K.quant_i8_block128(x)
        wq, ws = K.quant_i8_block128((torch.randn(n, k, device=dev) / math.sqrt(k)).bfloat16())
        b = (torch.randn(n, device=dev) * 0.1).bfloat16()
        return xq, xs, wq, ws, b

    gate = (torch.randn(1, dim, device=dev) * 0.5)
    x0 = torch.randn(L, dim, device=dev).bfloat16()
    cases = {}
    o_qkv = operands(3 * dim, dim)
    cases["qkv"] = (3 * dim, dim, lambda: K.gemm_w8a8_vt(*o_qkv, 2 * dim, torch.float16))
    o_q = operands(dim, dim)
    cases["crossq"] = (dim, dim, lambda: K.gemm_w8a8_stats(*o_q))
    xr = x0.clone()
    cases["o"] = (dim, dim, lambda: K.gemm_w8a8_stats(*o_q, x=xr, gate=gate))
    o_f0 = operands(ffn, dim)
    cases["ffn0"] = (ffn, dim, lambda: K.gemm_w8a8_quant(*o_f0[:4], torch.bfloat16, bias=o_f0[4], gelu_tanh=True))
    o_f2 = operands(dim, ffn)
    xr2 = x0.clone()
    cases["ffn2"] = (dim, ffn, lambda: K.gemm_w8a8_stats(*o_f2, x=xr2, gate=gate))
    only = [s for s in a.only.split(",") if s]
    rows = []
    e0, t0 = energy_uj(), time.perf_counter()
    time.sleep(1.0)
    print(json.dumps({"idle_W": round((energy_uj() - e0) * 1e-6 / (time.perf_counter() - t0))}), flush=True)
    for name, (n, k, fn) in cases.items():
        if only and name not in only:
            continue
        ref = {}
        for fast in fasts:
            for form in forms:
                K.set_tuning(K.TUNE_GEMM_VARIANT, form)
                K.set_tuning(K.TUNE_GEMM_FAST, fast)
                try:
                    # bits: the residual forms accumulate in place -> compare one launch from the same start
                    if name in ("o", "ffn2"):
                        xs_ = x0.clone()
                        args = o_q if name == "o" else o_f2
                        out = K.gemm_w8a8_stats(*args, x=xs_, gate=gate)
                    else:
                        out = fn()
                    out = [t.clone() for t in (out if isinstance(out, tuple) else (out,))]
                    if name == "qkv":     # (the V columns of d are not written by the V^T epilogue: compare what is)
                        out[0] = out[0][:, :2 * dim].clone()
                    same = None
                    if fast in ref:
                        same = all(torch.equal(p.view(torch.uint8), q.view(torch.uint8)) for p, q in zip(ref[fast], out))
                    else:
                        ref[fast] = out
                    us, j, w = measure(fn, a.seconds)
                finally:
                    K.set_tuning(K.TUNE_GEMM_VARIANT, 0)
                    K.set_tuning(K.TUNE_GEMM_FAST, 0)
                ops = 2.0 * L * n * k
                rec = {"gemm": name, "m": L, "n": n, "k": k, "form": form, "fast": fast, "us": round(us, 1), "J": round(j, 4),
                       "W": round(w), "POPs": round(ops / us * 1e-9, 3), "frac_of_5POPs": round(ops / us * 1e-9 / 5, 4),
                       "same_bits_as_first_form": same}
                rows.append(rec)
                print(json.dumps(rec), flush=True)
    print()
    print(f"{'gemm':8s} {'form':>4s} {'fast':>4s} {'us':>8s} {'J':>8s} {'W':>6s} {'frac':>7s} bits")
    for r in rows:
        print(f"{r['gemm']:8s} {r['form']:4d} {r['fast']:4d} {r['us']:8.1f} {r['J']:8.4f} {r['W']:6d} {r['frac_of_5POPs']:7.4f} {r['same_bits_as_first_form']}")
    for fast in fasts:
        for form in forms:
            sel = [r for r in rows if r["form"] == form and r["fast"] == fast]
            if len(sel) >= 5:
                # a block = qkv + crossq + 2 x (o shape) + ffn0 + ffn2
                tot = sum(r["us"] * (2 if r["gemm"] == "o" else 1) for r in sel)
                totj = sum(r["J"] * (2 if r["gemm"] == "o" else 1) for r in sel)
                print(f"block (6 GEMMs) form {form} fast {fast}: {tot:8.1f} us  {totj:7.4f} J")


if __name__ == "__main__":
    main()
