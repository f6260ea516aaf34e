"""Phases of a W8A8 GEMM tile from the kernel's own stamps (workgroups 0, 256, 512, ...: start, first stage landed, main
loop done, stores issued, stores complete).  profiles/r02_gemm_store_exp.txt is this script's output at the commit that
introduced the LDS-staged full-line stores, where the old direct 16-byte-piece stores were still selectable (ABLATE 10 =
staged): epilogue 11.7 k -> 6.9 k cycles per tile.  The staged form is now the only one."""
import ctypes, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from turbodiffusion_amd import kernels as K, _lib as L
dev = "cuda"
Lr = int(sys.argv[1]) if len(sys.argv) > 1 else 32760
for (n, k, nm) in ((1536, 1536, "attn proj"), (4608, 1536, "fused qkv"), (1536, 8960, "ffn2 shape, plain")):
    a = torch.randn(Lr, k, device=dev).bfloat16()
    aq, as_ = K.quant_i8_block128(a)
    wq, ws = K.quant_i8_block128((torch.randn(n, k, device=dev) / math.sqrt(k)).bfloat16())
    b = (torch.randn(n, device=dev) * 0.1).bfloat16()
    K.set_tuning(K.TUNE_GEMM_ABLATE, 0)
    ref = K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b)
    for rep in range(2):
        for mode, label in ((0, "default"), (6, "default + stamps")):
            K.set_tuning(K.TUNE_GEMM_ABLATE, mode)
            out = K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b)
            same = bool(torch.equal(out, ref))
            for _ in range(3):
                K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b)
            e1.record(); e1.synchronize()
            line = f"{nm} M={Lr} N={n} K={k} [{label}] {e0.elapsed_time(e1) / 20 * 1e3:.1f} us bit-identical={same}"
            if mode:
                buf = (ctypes.c_ulonglong * 128)()
                L.call("td_debug_read", ctypes.cast(buf, ctypes.c_void_p), 128)
                ph = []
                for r in range(3):
                    t = [buf[r * 5 + i] for i in range(5)]
                    if all(t):
                        ph.append([t[i + 1] - t[i] for i in range(4)])
                line += f"  phases (first-stage, main loop, epilogue issue, store drain) x100ns-clock: {ph}"
            print(line, flush=True)
K.set_tuning(K.TUNE_GEMM_ABLATE, 0)
