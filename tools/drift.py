"""Error accumulation over the 30 blocks of Wan2.1-1.3B at the C1 shape (L = 32 760): hidden tokens after every block
of the accelerated arithmetics against the dense-bf16 path (attention 'original' + bf16 library linears — the class the
reference's eager path is in, pinned to it by tests/test_gpu_wan.py) with the SAME seeded weights.

    python tools/drift.py [--layers 30]

Prints rel-L2 of the token tensor after blocks 1, 2, 5, 10, 20, 30 and of the final velocity, for:
  turbo          W8A8 + SageSLA top-k 0.1 (exact dequant, FP16 PV)      — the benchmarked configuration
  turbo/fast     + one-VALU GEMM dequant (G = 4)                         — vs turbo: what the opt-in costs at model level
  turbo/fp8pv    + FP8-PV SageAttention                                  — vs turbo
  w8a8-dense     W8A8 + dense SageAttention (no sparsity)                — separates quantisation from sparsity
  sla-bf16       SageSLA top-k 0.1 + bf16 linears                        — separates sparsity from quantisation
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import wan_ref as W  # noqa: E402  (seeded weights only)
from turbodiffusion_amd import kernels as K  # noqa: E402
from turbodiffusion_amd.wan import MODEL_CONFIGS, WanModel  # noqa: E402

DEV = "cuda"


def rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=30)
    args = ap.parse_args()
    cfg = dict(MODEL_CONFIGS["Wan2.1-1.3B"], num_layers=args.layers, text_dim=4096)
    sd = W.make_state_dict(cfg, seed=0)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 16, 21, 60, 104, generator=g).to(DEV).bfloat16()
    ctx = torch.randn(1, 512, 4096, generator=g).to(DEV).bfloat16()
    t = torch.tensor([[933.781]], device=DEV).bfloat16()

    def run(attention, quant, fast=0, pv="fp16"):
        with torch.device(DEV):
            net = WanModel(attention_type=attention, sla_topk=0.1, quant_linear=quant, **cfg)
        sdd = {k: v.to(DEV) for k, v in sd.items() if attention in ("sla", "sagesla") or "proj_l" not in k}
        own = net.state_dict()
        sdd = {k: (v.to(own[k].dtype) if k in own else v) for k, v in sdd.items()}
        net.load_from_float_state_dict(sdd)
        net.eval()
        net.sage_pv = pv
        net._tap_tokens = []
        if fast:
            K.set_tuning(K.TUNE_GEMM_VARIANT, 5); K.set_tuning(K.TUNE_GEMM_FAST, fast); K.set_tuning(4, 3)
        try:
            with torch.no_grad():
                y = net(x, t, ctx)
        finally:
            K.set_tuning(K.TUNE_GEMM_VARIANT, 0); K.set_tuning(K.TUNE_GEMM_FAST, 0); K.set_tuning(4, 0)
        toks = net._tap_tokens
        del net
        torch.cuda.empty_cache()
        return toks, y

    marks = [m for m in (1, 2, 5, 10, 20, 30) if m <= args.layers]
    ref_t, ref_y = run("original", False)
    tur_t, tur_y = run("sagesla", True)
    rows = [("turbo vs dense-bf16", tur_t, tur_y, ref_t, ref_y)]
    for name, kw, base in (("turbo/fast(G=4) vs turbo", dict(attention="sagesla", quant=True, fast=4), "turbo"),
                           ("turbo/fp8pv vs turbo", dict(attention="sagesla", quant=True, pv="fp8"), "turbo"),
                           ("w8a8-dense vs dense-bf16", dict(attention="sage", quant=True), "ref"),
                           ("sla-bf16 vs dense-bf16", dict(attention="sagesla", quant=False), "ref")):
        tt, yy = run(**kw)
        bt, by = (tur_t, tur_y) if base == "turbo" else (ref_t, ref_y)
        rows.append((name, tt, yy, bt, by))
        if name.startswith("turbo/"):
            rows.append((name.split(" vs ")[0] + " vs dense-bf16", tt, yy, ref_t, ref_y))
    for name, tt, yy, bt, by in rows:
        print(json.dumps({"pair": name, "rel_l2_after_block": {str(m): round(rel(tt[m - 1], bt[m - 1]), 5) for m in marks},
                          "velocity_rel_l2": round(rel(yy, by), 5)}), flush=True)


if __name__ == "__main__":
    main()
