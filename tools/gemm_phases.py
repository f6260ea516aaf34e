"""Phase-level s_memtime trace of the fine-interleaved GEMM (profiling aid): per round on one CU slot,
cycles of {launch->first stage landed, main loop, epilogue issue, store drain}."""
import ctypes, math, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from turbodiffusion_amd import kernels as K, _lib as L
dev = "cuda"
VARIANT = int(sys.argv[1]) if len(sys.argv) > 1 else 4
MODE = int(sys.argv[2]) if len(sys.argv) > 2 else 6  # 5 = stores predicated off
Lr = 32760
for (n, k, nm) in ((1536, 1536, "o-proj"), (4608, 1536, "qkv"), (1536, 8960, "ffn2")):
    a = torch.randn(Lr, k, device=dev).bfloat16()
    aq, as_ = K.quant_i8_block128(a)
    wq, ws = K.quant_i8_block128((torch.randn(n, k, device=dev) / math.sqrt(k)).bfloat16())
    b = torch.zeros(n, device=dev).bfloat16()
    K.set_tuning(0, VARIANT); K.set_tuning(1, MODE)
    for _ in range(3):
        K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    L.call("td_debug_read", ctypes.cast(buf, ctypes.c_void_p), 64)
    K.set_tuning(1, 0); K.set_tuning(0, 0)
    rounds = (((Lr + 255) // 256) * (n // 256) + 255) // 256
    print(nm, "rounds", rounds)
    t0 = buf[0]
    for r in range(min(rounds, 12)):
        t = [buf[r * 5 + i] for i in range(5)]
        print(f"  round {r}: start@{t[0]-t0:7d}  prologue {t[1]-t[0]:6d}  main {t[2]-t[1]:7d}  epi-issue {t[3]-t[2]:6d}  drain {t[4]-t[3]:6d}  total {t[4]-t[0]:7d}")
