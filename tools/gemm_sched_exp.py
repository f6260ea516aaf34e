"""m32 GEMM schedule experiments (TD_TUNE_GEMM_SCHED): timing, bit-identity with the default kernel, chain trace."""
import ctypes, json, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from turbodiffusion_amd import kernels as K, _lib as L
from tools.kbench import timeit
dev = "cuda"
Lr = 32760
SCHEDS = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,3".split(","))]
for (n, k, nm) in ((4608, 1536, "qkv"), (1536, 8960, "ffn2"), (1536, 1536, "o-proj")):
    a = torch.randn(Lr, k, device=dev).bfloat16()
    aq, as_ = K.quant_i8_block128(a)
    wq, ws = K.quant_i8_block128((torch.randn(n, k, device=dev) / math.sqrt(k)).bfloat16())
    b = (torch.randn(n, device=dev) * 0.05).bfloat16()
    K.set_tuning(K.TUNE_GEMM_VARIANT, 4)
    ref = K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b)
    t4 = timeit(lambda: K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b), 10)
    print(json.dumps({"gemm": nm, "variant": 4, "us": round(t4 * 1e6, 1), "POPs": round(2.0 * Lr * n * k / t4 / 1e15, 3)}), flush=True)
    K.set_tuning(K.TUNE_GEMM_VARIANT, 5)
    for fast in (0, 4):
        for sch in SCHEDS:
            K.set_tuning(4, sch); K.set_tuning(K.TUNE_GEMM_FAST, fast)
            out = K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b)
            t = timeit(lambda: K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b), 10)
            d = (out.float() - ref.float()).abs()
            print(json.dumps({"gemm": nm, "variant": 5, "sched": sch, "fast": fast, "us": round(t * 1e6, 1),
                              "POPs": round(2.0 * Lr * n * k / t / 1e15, 3), "identical": bool(torch.equal(out, ref)),
                              "rel_l2": (d.norm() / ref.float().norm()).item()}), flush=True)
    K.set_tuning(K.TUNE_GEMM_FAST, 0)
    if nm == "ffn2":
        for sch in SCHEDS:
            K.set_tuning(4, sch); K.set_tuning(1, 9)
            for _ in range(3):
                K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b)
            torch.cuda.synchronize()
            buf = (ctypes.c_ulonglong * 128)()
            L.call("td_debug_read", ctypes.cast(buf, ctypes.c_void_p), 128)
            for w in range(2):
                t = [x for x in (buf[w * 64 + i] for i in range(64)) if x]
                print("  sched", sch, "wave", w * 4, "chain deltas:", [t[i + 1] - t[i] for i in range(len(t) - 1)])
            K.set_tuning(1, 0)
    K.set_tuning(4, 0); K.set_tuning(K.TUNE_GEMM_VARIANT, 0)
