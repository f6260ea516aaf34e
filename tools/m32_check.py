"""Variant 5 (32x32x32 MFMA) against variant 4: bit identity of the three entry points, timings, phase stamps."""
import ctypes, json, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from turbodiffusion_amd import kernels as K, _lib as L
dev = "cuda"


def timeit(f, it=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it * 1e3  # us


def with_var(v, f):
    K.set_tuning(K.TUNE_GEMM_VARIANT, v)
    try:
        return f()
    finally:
        K.set_tuning(K.TUNE_GEMM_VARIANT, 0)


Lr = 32760
for (n, k, nm) in ((1536, 1536, "o/cross"), (4608, 1536, "qkv"), (8960, 1536, "ffn.0"), (1536, 8960, "ffn.2")):
    a = torch.randn(Lr, k, device=dev).bfloat16()
    aq, as_ = K.quant_i8_block128(a)
    wq, ws = K.quant_i8_block128((torch.randn(n, k, device=dev) / math.sqrt(k)).bfloat16())
    b = (torch.randn(n, device=dev) * 0.1).bfloat16()
    gate = torch.randn(1, n, device=dev) * 0.5
    x0 = torch.randn(Lr, n, device=dev).bfloat16()
    r = {"shape": nm, "n": n, "k": k}
    o4 = with_var(4, lambda: K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b))
    o5 = with_var(5, lambda: K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b))
    r["plain_identical"] = bool(torch.equal(o4, o5))
    if not r["plain_identical"]:
        d = (o4 != o5)
        r["plain_diff"] = int(d.sum().item()); idx = d.nonzero()[:5].tolist(); r["plain_where"] = idx
    for v in (4, 5):
        r[f"plain_v{v}_us"] = round(with_var(v, lambda: timeit(lambda: K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b))), 1)
    q4 = with_var(4, lambda: K.gemm_w8a8_quant(aq, as_, wq, ws, torch.bfloat16, bias=b, gelu_tanh=True))
    q5 = with_var(5, lambda: K.gemm_w8a8_quant(aq, as_, wq, ws, torch.bfloat16, bias=b, gelu_tanh=True))
    r["quant_identical"] = bool(torch.equal(q4[0], q5[0]) and torch.equal(q4[1], q5[1]))
    for v in (4, 5):
        r[f"quant_v{v}_us"] = round(with_var(v, lambda: timeit(lambda: K.gemm_w8a8_quant(aq, as_, wq, ws, torch.bfloat16, bias=b, gelu_tanh=True))), 1)
    r4 = with_var(4, lambda: K.gemm_w8a8_residual_(x0.clone(), aq, as_, wq, ws, bias=b, gate=gate))
    r5 = with_var(5, lambda: K.gemm_w8a8_residual_(x0.clone(), aq, as_, wq, ws, bias=b, gate=gate))
    r["res_identical"] = bool(torch.equal(r4, r5))
    xx = x0.clone()
    for v in (4, 5):
        r[f"res_v{v}_us"] = round(with_var(v, lambda: timeit(lambda: K.gemm_w8a8_residual_(xx, aq, as_, wq, ws, bias=b, gate=gate))), 1)
    print(json.dumps(r), flush=True)
    if nm in ("o/cross", "ffn.2"):
        for v in (4, 5):
            K.set_tuning(1, 6)
            with_var(v, lambda: [K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b) for _ in range(3)])
            torch.cuda.synchronize()
            K.set_tuning(1, 0)
            buf = (ctypes.c_ulonglong * 64)()
            L.call("td_debug_read", ctypes.cast(buf, ctypes.c_void_p), 64)
            rounds = []
            for q in range(3):
                t = [buf[q * 5 + i] for i in range(5)]
                rounds.append([t[i + 1] - t[i] for i in range(4)])
            print(json.dumps({"phases_v%d" % v: nm, "prologue/main/epilogue/store_drain ticks per round": rounds}), flush=True)
    del a, aq, wq, x0
