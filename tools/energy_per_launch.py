"""Joules per launch of the big kernels (round 3: the denoising loop runs at the board's power cap, so energy per launch is
what its speed is made of).  Each kernel is launched back to back for ~1.5 s; the socket's accumulated-energy counter
(``rocm-smi --showenergycounter``) is read before and after.  C1 shapes.  One line per kernel: us per launch, W, J per
launch, pJ per int8-equivalent op where that means something."""
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


def measure(name, fn, ops=None, seconds=1.5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 0.15:      # how many launches make `seconds`
        fn()
        n += 1
    torch.cuda.synchronize()
    per = (time.perf_counter() - t0) / n
    reps = max(10, int(seconds / per))
    time.sleep(0.5)                              # let the budget settle the same way before every kernel
    e0 = energy_uj()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    e1 = energy_uj()
    j = (e1 - e0) * 1e-6 / reps
    line = f"{name:58s} {dt / reps * 1e6:9.1f} us  {(e1 - e0) * 1e-6 / dt:7.0f} W  {j:8.4f} J/launch"
    if ops:
        line += f"  {j / ops * 1e12:6.3f} pJ/op"
    print(line, flush=True)


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    L, dim, H, D, ffn = 32760, 1536, 12, 128, 8960
    print(f"idle socket power: ", end="")
    e0, t0 = energy_uj(), time.perf_counter()
    time.sleep(1.0)
    print(f"{(energy_uj() - e0) * 1e-6 / (time.perf_counter() - t0):.0f} W")
    sink = torch.zeros(16, dtype=torch.float32, device=dev)
    measure("int8 MFMA only (calibration kernel: operands in registers)",
            lambda: K.call("td_calib_mfma_i8", 4096, 1024, K.ptr(sink), K.stream_ptr()), ops=1024 * 4 * 4096 * 4 * 2.0 * 32 * 32 * 32)
    a = torch.randn(L, ffn, device=dev).bfloat16()
    aq, as_ = K.quant_i8_block128(a)
    wq, ws = K.quant_i8_block128((torch.randn(dim, ffn, device=dev) / math.sqrt(ffn)).bfloat16())
    b = torch.zeros(dim, device=dev).bfloat16()
    ops = 2.0 * L * dim * ffn
    measure("W8A8 GEMM ffn.2, exact dequant (default kernel)", lambda: K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b), ops)
    K.set_tuning(K.TUNE_GEMM_VARIANT, 5)
    measure("W8A8 GEMM ffn.2, exact dequant (32x32x32 kernel)", lambda: K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b), ops)
    K.set_tuning(K.TUNE_GEMM_FAST, 4)
    K.set_tuning(4, 3)
    measure("W8A8 GEMM ffn.2, one-VALU dequant G = 4 (opt-in)", lambda: K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b), ops)
    K.set_tuning(K.TUNE_GEMM_FAST, 0)
    K.set_tuning(4, 0)
    K.set_tuning(K.TUNE_GEMM_VARIANT, 0)
    a0 = torch.randn(L, dim, device=dev).bfloat16()
    a0q, a0s = K.quant_i8_block128(a0)
    w0q, w0s = K.quant_i8_block128((torch.randn(ffn, dim, device=dev) / math.sqrt(dim)).bfloat16())
    b0 = torch.zeros(ffn, device=dev).bfloat16()
    measure("W8A8 GEMM ffn.0 + GELU + output quantiser", lambda: K.gemm_w8a8_quant(a0q, a0s, w0q, w0s, torch.bfloat16, bias=b0, gelu_tanh=True), 2.0 * L * dim * ffn)
    # attention, windowed block selection (neighbouring Q blocks share K blocks, as in the model)
    qkv = torch.randn(L, 3 * dim, device=dev).bfloat16()
    w = torch.ones(dim, device=dev)
    ang = torch.rand(L, 64, device=dev) * 6
    cos, sin = torch.cos(ang), torch.sin(ang)
    q = K.qk_norm_rope(qkv, 0, H, D, w, cos, sin, 1e-6)
    k = K.qk_norm_rope(qkv, dim, H, D, w, cos, sin, 1e-6)
    vt = K.v_transpose(qkv[:, 2 * dim:], D, 3 * dim, L, H, D, torch.float16)
    km = K.seq_mean(k)
    pq, q8, qs = K.sage_quant_pool(q, None, 128)
    pk, k8, ks = K.sage_quant_pool(k, km, 64)
    kb, qb = pk.shape[1], pq.shape[1]
    topk = int(0.1 * kb)
    base = (torch.arange(qb) * 2).clamp(max=kb - topk)
    lut = (base[:, None] + torch.arange(topk)[None, :]).int()[None].repeat(H, 1, 1).contiguous().to(dev)
    out = torch.empty(L, H, D, device=dev, dtype=torch.bfloat16)
    aops = 4.0 * qb * 128 * topk * 64 * D * H
    measure("attention INT8 QK + FP16 PV (sparse 51 of 512)", lambda: K.attn_i8(q8, qs, k8, ks, vt, lut, out, D, H * D), aops)
    vt8, vsc = K.v_fp8_tiles(qkv[:, 2 * dim:], D, 3 * dim, L, H, D)
    measure("attention INT8 QK + FP8 PV (opt-in)", lambda: K.attn_i8(q8, qs, k8, ks, vt8, lut, out, D, H * D, v_scale=vsc), aops)
    x = torch.randn(L, dim, device=dev).bfloat16()
    sc = torch.zeros(1, dim, device=dev)
    measure("layernorm + modulate -> int8 (HBM-bound, 5 B per element)", lambda: K.layernorm_quant(x, None, None, 1e-6, sc, sc))
    measure("qk_norm_rope (HBM-bound)", lambda: K.qk_norm_rope(qkv, 0, H, D, w, cos, sin, 1e-6))


if __name__ == "__main__":
    main()
