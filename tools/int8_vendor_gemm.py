"""Context for the W8A8 roofline: what the vendor library reaches on a PLAIN int8 GEMM (int32 out, no per-128-K block
scales, no epilogue) at the same shapes, through torch._int_mm (hipBLASLt).  Not on the product path."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kbench import timeit
dev = "cuda"
for (m, n, k, nm) in ((32760, 1536, 1536, "attn proj"), (32760, 4608, 1536, "qkv"), (32760, 8960, 1536, "ffn1"), (32760, 1536, 8960, "ffn2")):
    a = torch.randint(-128, 128, (m // 8 * 8, k), dtype=torch.int8, device=dev)
    b = torch.randint(-128, 128, (k, n), dtype=torch.int8, device=dev)
    try:
        t = timeit(lambda: torch._int_mm(a, b), 10)
        print(json.dumps({"vendor_int8_gemm": nm, "M": a.shape[0], "N": n, "K": k, "us": round(t * 1e6, 1), "POPs": round(2.0 * a.shape[0] * n * k / t / 1e15, 3)}), flush=True)
    except Exception as e:  # noqa: BLE001
        print(json.dumps({"vendor_int8_gemm": nm, "error": repr(e)[:200]}), flush=True)
