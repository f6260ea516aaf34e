"""Phases of the attention kernels from their own s_memtime stamps (TD_TUNE_ATTN_OCC = 9): per workgroup {entry, Q fragments +
first tile landed, K loop done, epilogue stores issued} and the hardware id -> prologue / loop / epilogue cycles and how
many workgroups a CU overlaps.  Cross-attention (L x 512 keys, dense, 16-bit, Q normalised on load, quantised output) and the
block-sparse INT8 self-attention (random inputs: scattered block selection, see tools/README.md)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from turbodiffusion_amd import kernels as K, _lib as L_
dev = "cuda"
H, L, D, dim = 12, 32760, 128, 1536
NW = 4096

def read(nwg):
    n = min(nwg, NW)
    buf = (ctypes.c_ulonglong * (256 + 5 * n))()
    L_.call("td_debug_read", ctypes.cast(buf, ctypes.c_void_p), 256 + 5 * n)
    return np.array(buf[256:], dtype=np.uint64).reshape(n, 5)

def report(name, us, t):
    t0, t1, t2, t3 = (t[:, i].astype(np.int64) for i in range(4))
    hw = t[:, 4]
    cu = ((hw >> np.uint64(32)) & np.uint64(0xf)) * np.uint64(4096) + ((hw & np.uint64(0xffffffff)) >> np.uint64(8) & np.uint64(0xff))
    conc = []
    for c in np.unique(cu)[:64]:
        idx = np.where(cu == c)[0]
        s, e = t0[idx], t3[idx]
        conc.append(np.mean([np.sum((s <= x) & (e > x)) for x in s]))
    print(f"{name}: {us:.1f} us;  prologue (Q + first tile) {np.mean(t1 - t0):.0f}  K loop {np.mean(t2 - t1):.0f}  epilogue {np.mean(t3 - t2):.0f}  "
          f"total {np.mean(t3 - t0):.0f} cycles per workgroup (p90 {np.percentile(t3 - t0, 90):.0f});  "
          f"workgroups resident on a CU when one starts: {np.mean(conc):.2f}", flush=True)

def timed(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3

# ---- cross attention: the model's call (wan.py:_cross_attention)
xq = torch.randn(L, dim, device=dev).bfloat16()
rstd = K.rms_stats(xq, dim, 1e-6)
w = torch.ones(dim, device=dev)
kc = torch.randn(H, 512, D, device=dev).bfloat16()
vc = torch.randn(512, dim, device=dev).bfloat16()
vtc = K.v_transpose(vc, D, dim, 512, H, D, torch.bfloat16)
K.set_tuning(K.TUNE_ATTN_OCC, 9)
us = timed(lambda: K.attn_16_qnorm(xq, rstd, w, kc, vtc, None, torch.bfloat16, D, dim, lk=512, quant_out=True))
report("cross attention L x 512 (dense, 8 tiles)", us, read(H * 256))
# ---- sparse INT8 self attention
q = torch.randn(H, L, D, device=dev).bfloat16(); k = torch.randn(H, L, D, device=dev).bfloat16()
v = torch.randn(L, dim, device=dev).bfloat16()
vt = K.v_transpose(v, D, dim, L, H, D, torch.float16)
km = K.seq_mean(k)
pq, q8, qs = K.sage_quant_pool(q, None, 128); pk, k8, ks = K.sage_quant_pool(k, km, 64)
lut = K.sla_topk(pq, pk, 51)
out = torch.empty(L, dim, device=dev, dtype=torch.bfloat16)
us = timed(lambda: K.attn_i8(q8, qs, k8, ks, vt, lut, torch.bfloat16, D, dim, quant_out=True))
report("self attention INT8/FP16-PV sparse 51 of 512 (random inputs)", us, read(H * 256))
K.set_tuning(K.TUNE_ATTN_OCC, 0)
