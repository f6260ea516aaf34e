"""Generate tests/golden/*.pt from the REAL reference (imported on CPU via oracle/ref_harness.py).

    python -m oracle.make_golden

Run in the build container only (needs /root/reference).  The fixtures travel to the GPU box; the
reference does not.  Contents of wan_tiny.pt:
  cfg, x, t, ctx                      seeded inputs (BASELINE.md §3 conventions, tiny shapes)
  sd_seed                             the state dict is re-created with oracle.wan_ref.make_state_dict(cfg, sd_seed)
  ref_fp32, ref_bf16                  reference WanModel outputs (fp32 model / bf16-checkpoint emulation)
  ref_sample_bf16                     latents after the 4-step rCM sampler driven by the reference net
  noises                              the per-step N(0,1) tensors used by that sampler run
"""
import os
import warnings

import torch

from . import ref_harness as rh
from . import wan_ref as W

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

TINY = dict(dim=256, eps=1e-6, ffn_dim=512, freq_dim=256, in_dim=16, model_type="t2v", num_heads=2, num_layers=2,
            out_dim=16, text_len=512, text_dim=128)


def sla_golden():
    """tests/golden/sla_tiny.pt: the reference's SparseLinearAttention / SageSparseLinearAttention modules (SLA/core.py)
    run on the CPU through oracle/ref_harness.patched_sla (only the Triton / SpargeAttn leaves replaced):
      q, k, v [1, L, H, 128] bf16, proj_w/proj_b, topk       seeded inputs (ragged L: 128- and 64-row tails)
      ref_sla, sla_lut, sparse_map                           SparseLinearAttention(BLKQ=128, BLKK=64) output, the LUT
                                                             (torch.topk order) it handed its kernel, its block map
      ref_sagesla_f16pv, ref_sagesla_fp8pv                   SageSparseLinearAttention, sm80 / sm89 branches"""
    g = torch.Generator().manual_seed(21)
    L, H = 600, 2
    q, k, v = [torch.randn(1, L, H, 128, generator=g).bfloat16() for _ in range(3)]
    wp = torch.randn(128, 128, generator=g) * 0.05
    bp = torch.randn(128, generator=g) * 0.05
    topk = 0.4
    out = {"q": q, "k": k, "v": v, "proj_w": wp, "proj_b": bp, "topk": topk}

    def mod(sla, cls, *a, **kw):
        m = getattr(sla, cls)(128, *a, **kw)
        m.proj_l.weight.copy_(wp)
        m.proj_l.bias.copy_(bp)
        return m

    with torch.no_grad():
        with rh.patched_sla("sm80") as sla:
            out["ref_sla"] = mod(sla, "SparseLinearAttention", topk, BLKQ=128, BLKK=64)(q, k, v)
            out["sla_lut"] = sla._td_recorded["lut"].clone()
            out["sparse_map"] = sla._td_recorded["sparse_map"].clone()
            out["ref_sagesla_f16pv"] = mod(sla, "SageSparseLinearAttention", topk)(q, k, v)
        with rh.patched_sla("sm89") as sla:
            out["ref_sagesla_fp8pv"] = mod(sla, "SageSparseLinearAttention", topk)(q, k, v)
    torch.save(out, os.path.join(OUT, "sla_tiny.pt"))
    print("wrote", os.path.join(OUT, "sla_tiny.pt"), {k_: (tuple(v_.shape) if hasattr(v_, "shape") else v_) for k_, v_ in out.items()})


def main():
    warnings.filterwarnings("ignore")
    os.makedirs(OUT, exist_ok=True)
    sla_golden()
    cfg, seed = TINY, 0
    sd = W.make_state_dict(cfg, seed)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 16, 5, 16, 24, generator=g)       # L = 5*8*12 = 480 tokens
    ctx = torch.randn(1, 512, cfg["text_dim"], generator=g)
    t = torch.tensor([[933.781]])
    out = {"cfg": cfg, "sd_seed": seed, "x": x, "t": t, "ctx": ctx}
    with torch.no_grad():
        net32 = rh.reference_wan_from_sd(cfg, sd, None)
        out["ref_fp32"] = net32(x, t, ctx)
        netbf = rh.reference_wan_from_sd(cfg, sd, torch.bfloat16)
        out["ref_bf16"] = netbf(x.bfloat16(), t.bfloat16(), ctx.bfloat16())
        noises = [torch.randn(x.shape, generator=g) for _ in range(4)]
        out["noises"] = noises
        out["ref_sample_bf16"] = W.rcm_sample(lambda xx, tt: netbf(xx, tt, ctx.bfloat16()), x, noises)
    torch.save(out, os.path.join(OUT, "wan_tiny.pt"))
    print("wrote", os.path.join(OUT, "wan_tiny.pt"), {k: (tuple(v.shape) if hasattr(v, "shape") else type(v).__name__)
                                                       for k, v in out.items()})


if __name__ == "__main__":
    main()
