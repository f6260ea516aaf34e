"""Round 6: the HBM-bound family of a block, one launch each at C1's extents — microseconds and fraction of the 8 TB/s
roofline against the algorithmic bytes bench.py's `roofline_hbm` uses (HBM_FAMILY).  A quick loop for kernel work:

    python tools/glue_bench.py [--L 32760] [--iters 20]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from turbodiffusion_amd import kernels as K  # noqa: E402


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=int, default=32760)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = "cuda"
    torch.manual_seed(0)
    L, H, D, dim = a.L, 12, 128, 1536
    qkv = torch.randn(L, 3 * dim, device=dev).bfloat16()
    w = torch.ones(dim, device=dev)
    ang = torch.rand(L, 64, device=dev) * 6
    cos, sin = torch.cos(ang), torch.sin(ang)
    q = K.qk_norm_rope(qkv, 0, H, D, w, cos, sin, 1e-6)
    k = K.qk_norm_rope(qkv, dim, H, D, w, cos, sin, 1e-6)
    vt = K.v_transpose(qkv[:, 2 * dim:], D, 3 * dim, L, H, D, torch.float16)
    kv_t, ksum, km = K.sla_linear_kv(k, vt, want_kmean=True)
    wp = (torch.randn(128, 128, device=dev) * 0.05)
    bp = torch.zeros(128, device=dev)
    x = torch.randn(L, dim, device=dev).bfloat16()
    stats = torch.stack([x.float().mean(1), x.float().var(1, unbiased=False).add(1e-6).rsqrt()], 1).contiguous()
    sc = torch.randn(1, dim, device=dev) * 0.1
    ld = L * dim
    cases = [
        ("qk_norm_rope (q)", lambda: K.qk_norm_rope(qkv, 0, H, D, w, cos, sin, 1e-6), 4 * ld),
        ("sage_quant_pool<128> (q)", lambda: K.sage_quant_pool(q, None, 128), 3 * ld),
        ("sage_quant_pool<64> (k, smooth-K)", lambda: K.sage_quant_pool(k, km, 64), 3 * ld),
        ("sla_linear_kv (+ smooth-K mean)", lambda: K.sla_linear_kv(k, vt, want_kmean=True), 4 * ld),
        ("sla_linear_kv", lambda: K.sla_linear_kv(k, vt), 4 * ld),
        ("sla_linear_out_t", lambda: K.sla_linear_out_t(q, kv_t, ksum, wp, bp), 4 * ld),
        ("layernorm_quant (modulate, given statistics)", lambda: K.layernorm_quant(x, None, None, 1e-6, scale=sc, shift=sc, rows_per_batch=L, stats=stats), 3 * ld),
        ("sla_topk", lambda: K.sla_topk(*[K.sage_quant_pool(t_, km_, b_)[0] for t_, km_, b_ in ((q, None, 128),)], K.sage_quant_pool(k, km, 64)[0], 51), 0),
    ]
    for name, fn, byts in cases:
        try:
            us = timeit(fn, a.iters)
        except Exception as e:   # a wrapper signature that moved: report, go on
            print(json.dumps({"kernel": name, "error": repr(e)[:200]}), flush=True)
            continue
        rec = {"kernel": name, "us": round(us, 1)}
        if byts:
            rec.update({"GBps": round(byts / us * 1e-3), "frac_of_8TBps": round(byts / us * 1e-3 / 8000, 3)})
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
