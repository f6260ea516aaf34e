"""Segment-level s_memtime trace of the ping-pong GEMM (profiling aid)."""
import ctypes, math, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from turbodiffusion_amd import kernels as K, _lib as L
dev = "cuda"
Lr, n, k = 32760, 1536, 8960
a = torch.randn(Lr, k, device=dev).bfloat16()
aq, as_ = K.quant_i8_block128(a)
wq, ws = K.quant_i8_block128((torch.randn(n, k, device=dev) / math.sqrt(k)).bfloat16())
b = torch.zeros(n, device=dev).bfloat16()
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 4
K.set_tuning(0, variant)
modes = ((9, "exact"), (19, "fast G=4")) if variant == 5 else ((9, "full"), (7, "full, L2 prefetch")) if variant == 4 else ((9, "full"), (10, "no DMA in loop"), (11, "no ds_read in loop"), (12, "no MFMA"))
for mode, nm in modes:
    K.set_tuning(1, mode)
    for _ in range(3):
        K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b)
    e1.record(); e1.synchronize()
    buf = (ctypes.c_ulonglong * 128)()
    L.call("td_debug_read", ctypes.cast(buf, ctypes.c_void_p), 128)
    print(nm, "ms", e0.elapsed_time(e1) / 5)
    for w in range(2):
        t = [buf[w * 64 + i] for i in range(64)]
        t = [x for x in t if x]
        print("  wave", w * 4, "deltas:", [t[i + 1] - t[i] for i in range(len(t) - 1)])
K.set_tuning(1, 0); K.set_tuning(0, 0)
