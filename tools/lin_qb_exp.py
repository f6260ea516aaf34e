"""Linear branch pass 2 (td_sla_linear_out_t): Q blocks per workgroup vs time at the C1 shape (grid balance over 256 CUs)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from turbodiffusion_amd import kernels as K
dev = "cuda"
H, L, D = 12, 32760, 128
q = torch.randn(H, L, D, device=dev).bfloat16()
kv = torch.randn(H, D, D, device=dev).bfloat16()
ks = torch.rand(H, D, device=dev).bfloat16()
wp = (torch.randn(D, D, device=dev) * 0.05)
bp = (torch.randn(D, device=dev) * 0.05)
ref = None
for rep in range(2):
    for qpw in (8, 4):
        K.set_tuning(K.TUNE_LIN_QB, qpw)
        out = K.sla_linear_out_t(q, kv, ks, wp, bp)
        if ref is None:
            ref = out.clone()
        for _ in range(3):
            K.sla_linear_out_t(q, kv, ks, wp, bp)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            K.sla_linear_out_t(q, kv, ks, wp, bp)
        e1.record(); e1.synchronize()
        print(f"qb_per_wg {qpw:2d}: {e0.elapsed_time(e1) / 20 * 1e3:6.1f} us  workgroups {-(-256 // qpw) * H}  bit-identical {bool(torch.equal(out, ref))}", flush=True)
# phase stamps (negative value = that many blocks per workgroup + s_memtime stamps of wave 0 of workgroup (0, 0))
import ctypes
from turbodiffusion_amd import _lib as L_
K.set_tuning(K.TUNE_LIN_QB, -4)
K.sla_linear_out_t(q, kv, ks, wp, bp)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
L_.call("td_debug_read", ctypes.cast(buf, ctypes.c_void_p), 64)
t = [buf[i] for i in range(20)]
for b in range(4):
    d = [t[5 * b + i + 1] - t[5 * b + i] for i in range(4)]
    print(f"block {b}: softmax {d[0]}  GEMM1 {d[1]}  o_l {d[2]}  GEMM2+stores {d[3]}  (cycles); block total {t[5 * b + 4] - t[5 * b]}", flush=True)
K.set_tuning(K.TUNE_LIN_QB, 0)
