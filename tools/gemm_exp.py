"""GEMM experiments (profiling aid): row-stride padding (L2 channel spread), raster group size, DMA schedule.
Results of padded runs are garbage by construction (the operands are read with a different stride); only time matters."""
import sys, os, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from turbodiffusion_amd import kernels as K, _lib as L
dev = "cuda"
Lr = 32760
def t_ms(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it
K.set_tuning(0, 4)
for (n, k, nm) in ((1536, 1536, "o-proj"), (4608, 1536, "qkv"), (1536, 8960, "ffn2")):
    for pad in (0,):
        ld = k + pad
        aq = torch.randint(-128, 128, (Lr, ld), dtype=torch.int8, device=dev)
        wq = torch.randint(-128, 128, (n, ld), dtype=torch.int8, device=dev)
        as_ = torch.rand((Lr + 127) // 128, k // 128, device=dev) * 0.01
        ws = torch.rand((n + 127) // 128, k // 128, device=dev) * 0.01
        b = torch.zeros(n, device=dev).bfloat16()
        out = torch.empty(Lr, n, device=dev, dtype=torch.bfloat16)
        K.set_tuning(2, pad)
        def run():
            L.call("td_gemm_w8a8", L.ptr(aq), L.ptr(as_), L.ptr(wq), L.ptr(ws), L.ptr(b), L.ptr(out), L.TD_BF16, 0, Lr, n, k, n, L.stream_ptr())
        for gm in (4,):
            for sched in (0, 1, 2, 3, 0, 2):
                K.set_tuning(3, gm); K.set_tuning(4, sched)
                ms = t_ms(run)
                print(json.dumps({"shape": nm, "pad": pad, "group_m": gm, "sched": sched, "ms": round(ms, 4),
                                  "POPs": round(2.0 * Lr * n * k / ms / 1e9, 1)}), flush=True)
        K.set_tuning(2, 0); K.set_tuning(3, 0); K.set_tuning(4, 0)
        del aq, wq
K.set_tuning(0, 0)
