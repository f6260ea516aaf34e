"""Ceilings for glue fusions, measured before building them: one process, C1, hipGraph replay; variants in which a class of
small kernels is LEFT OUT of the captured graph (its outputs are taken from the eager warm-up: same shapes, stale values —
results are garbage, only the clock is read).  Interleaved repetitions.

    python tools/ablate_glue.py [reps]

variants:
  base            the production forward
  no_finalize     td_row_stats_finalize (4 launches per layer) returns its warm-up result
  no_topk         td_sla_topk returns its warm-up LUT
  no_qpool        td_sage_quant_pool returns its warm-up outputs (both Q and K side)
  no_linear       the linear branch's two passes (kv, out) return their warm-up outputs
  no_norm_rope    td_qk_norm_rope (q and k) returns its warm-up outputs"""
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from turbodiffusion_amd import kernels as K  # noqa: E402
from turbodiffusion_amd.graph import GraphedModel  # noqa: E402


def cached(fn_name):
    """replace K.<fn_name> by a wrapper that really calls it once per distinct call-site signature and returns that result
    afterwards (no launch)"""
    real = getattr(K, fn_name)
    memo = {}

    def wrapper(*a, **kw):
        key = tuple((tuple(t.shape), t.dtype) if isinstance(t, torch.Tensor) else (t if isinstance(t, (int, float, bool, str, type(None))) else id(t))
                    for t in list(a) + list(kw.values()))
        if key not in memo:
            memo[key] = real(*a, **kw)
        return memo[key]
    return real, wrapper


VARIANTS = {
    "base": [],
    "no_finalize": ["row_stats_finalize"],
    "no_topk": ["sla_topk"],
    "no_qpool": ["sage_quant_pool"],
    "no_linear": ["sla_linear_kv", "sla_linear_out_t"],
    "no_norm_rope": ["qk_norm_rope"],          # the ceiling of folding norm + RoPE into the kernels that read q / k next
    "no_norm_rope_no_qpool": ["qk_norm_rope", "sage_quant_pool"],
}


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device("cuda", 0)
    net, cfg = bench.build_model("Wan2.1-1.3B", bench.WORKLOADS["turbo"], dev, 0.1)
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(1, 16, 21, 60, 104, device=dev, generator=g)
    t = torch.full((1, 1), 500.0, device=dev)
    ctx = torch.randn(1, 512, cfg.get("text_dim", 4096), device=dev, generator=g).bfloat16()
    models = {}
    only = os.environ.get("ABLATE_ONLY")
    for name, fns in VARIANTS.items():
        if only and name not in ("base", only):
            continue
        print(f"building variant {name}", flush=True)
        saved = []
        for fn in fns:
            if not hasattr(K, fn):
                print(f"(no K.{fn}; variant {name} skipped)")
                saved = None
                break
            real, wrap = cached(fn)
            saved.append((fn, real))
            setattr(K, fn, wrap)
        if saved is None:
            continue
        try:
            with torch.no_grad():
                net(x, t, ctx)                    # fills the wrappers' memo eagerly
                gm = GraphedModel(net)
                gm(x, t, ctx)                     # capture (the wrappers return memo entries: no launches recorded)
            models[name] = gm
        finally:
            for fn, real in saved:
                setattr(K, fn, real)
    torch.cuda.synchronize()
    res = {n: [] for n in models}
    for r in range(reps):
        for name, gm in models.items():
            for _ in range(2):
                gm(x, t, ctx)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(8):
                gm(x, t, ctx)
            torch.cuda.synchronize()
            res[name].append((time.perf_counter() - t0) / 8 * 1e3)
    base = min(res["base"])
    for name, v in res.items():
        print(f"{name:12s} ms per DiT forward: " + " ".join(f"{u:.2f}" for u in v) + f"   best {min(v):.2f}  ({min(v) - base:+.2f} vs base)")


if __name__ == "__main__":
    main()
