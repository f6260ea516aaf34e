"""Is the GEMM epilogue's store phase limited per CU or by the fabric?  Phase stamps of workgroup 0 for problem sizes
that occupy 66, 132, 252 CUs (one round) and the full 3-round o-projection."""
import ctypes, math, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from turbodiffusion_amd import kernels as K, _lib as L
dev = "cuda"
n = k = 1536
for M in (2816, 5632, 10752, 32760):
    a = torch.randn(M, k, device=dev).bfloat16()
    aq, as_ = K.quant_i8_block128(a)
    wq, ws = K.quant_i8_block128((torch.randn(n, k, device=dev) / math.sqrt(k)).bfloat16())
    b = torch.zeros(n, device=dev).bfloat16()
    for mode, nm in ((6, "stores on"), (5, "stores off")):
        K.set_tuning(0, 4); K.set_tuning(1, mode)
        for _ in range(3):
            K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 64)()
        L.call("td_debug_read", ctypes.cast(buf, ctypes.c_void_p), 64)
        K.set_tuning(1, 0); K.set_tuning(0, 0)
        t = [buf[i] for i in range(5)]
        print(f"M={M:6d} tiles={((M + 255) // 256) * 6:4d} {nm:10s}: prologue {t[1]-t[0]:6d}  main {t[2]-t[1]:7d}  epi-issue {t[3]-t[2]:6d}  drain {t[4]-t[3]:6d}")
