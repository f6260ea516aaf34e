"""Where a GEMM launch's wall time goes beyond its workgroups' own cycles: {start, end, hardware id} of EVERY workgroup of
the phase-stamp instantiation (TD_TUNE_GEMM_ABLATE = 6) -> per CU: the workgroups it ran in order, the gaps between one
workgroup's last store and the next one's first instruction, and the idle time before the first / after the last."""
import ctypes, math, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from turbodiffusion_amd import kernels as K, _lib as L
dev = "cuda"
Lr = 32760
for (n, k, nm) in ((1536, 1536, "attn proj"), (4608, 1536, "fused qkv"), (1536, 8960, "ffn2 shape, plain")):
    a = torch.randn(Lr, k, device=dev).bfloat16()
    aq, as_ = K.quant_i8_block128(a)
    wq, ws = K.quant_i8_block128((torch.randn(n, k, device=dev) / math.sqrt(k)).bfloat16())
    b = (torch.randn(n, device=dev) * 0.1).bfloat16()
    K.set_tuning(K.TUNE_GEMM_ABLATE, 6)
    for _ in range(3):
        K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b)
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    nwg = -(-Lr // 256) * -(-n // 256)
    buf = (ctypes.c_ulonglong * (256 + 3 * nwg))()
    L.call("td_debug_read", ctypes.cast(buf, ctypes.c_void_p), 256 + 3 * nwg)
    t = np.array(buf[256:], dtype=np.uint64).reshape(nwg, 3)
    start, end, hw = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64), t[:, 2]
    cu = ((hw >> np.uint64(32)) & np.uint64(0xf)) * np.uint64(4096) + ((hw & np.uint64(0xffffffff)) >> np.uint64(8) & np.uint64(0xff))  # xcc, (se, sh, cu) bits 15:8
    # (s_memtime bases differ between XCDs: only differences within one CU are meaningful)
    dur = end - start
    gaps, spans, per_cu = [], [], []
    for c in np.unique(cu):
        idx = np.where(cu == c)[0]
        o = idx[np.argsort(start[idx])]
        per_cu.append(len(o))
        spans.append(end[o[-1]] - start[o[0]])
        gaps += list(start[o[1:]] - end[o[:-1]])
    gaps = np.array(gaps)
    print(f"{nm} M={Lr} N={n} K={k}: {us:.1f} us by events; {nwg} workgroups on {len(np.unique(cu))} distinct CU ids, "
          f"{min(per_cu)}-{max(per_cu)} per CU", flush=True)
    print(f"   workgroup duration: mean {dur.mean():.0f} min {dur.min()} max {dur.max()} cycles;  gap between consecutive workgroups of a CU: "
          f"mean {gaps.mean():.0f} median {np.median(gaps):.0f} p90 {np.percentile(gaps, 90):.0f} max {gaps.max()}", flush=True)
    print(f"   first start -> last end on a CU: mean {np.mean(spans):.0f} cycles in a {us:.1f}-us launch => >= {np.mean(spans) / us / 1e3:.2f} GHz shader clock", flush=True)
K.set_tuning(K.TUNE_GEMM_ABLATE, 0)
