"""Run the reference's OWN Triton leaves on the MI355X and keep what they produce as a golden fixture.

TEST INFRASTRUCTURE (oracle/__init__.py).  The reference's Triton kernels are in its tree — ``SLA/kernel.py:21-82``
(``_attn_fwd``, wrapper ``:240-274``), ``SLA/utils.py:21-52`` (``compress_kernel`` / ``mean_pool``), ``ops/core.py:96-136``
(RMSNorm), ``:193-335`` (LayerNorm with / without affine) — and Triton 3.6 has a ROCm backend, so unlike the CUDA ops
(nvcc + CUTLASS) and SpargeAttn (sources absent) they CAN execute, on the GPU box.  ``/root/reference`` does not exist
there, so this is a two-stage recipe:

    python -m oracle.triton_leaves stage     # HERE: copy SLA/*.py and ops/core.py to oracle/_ref/triton/ (git-ignored,
                                             #       travels with gpurun; never imported by the product or bench.py)
    gpurun -- python -m oracle.triton_leaves run   # GPU box: execute them, write gpurun_out/triton_leaves.pt
    python -m oracle.triton_leaves unstage   # HERE: delete the staged copies again (no reference source stays in the repo)
    cp gpurun_out/triton_leaves.pt tests/golden/

The fixture holds the INPUTS (from the integer hash of make_golden_c1, so they are machine-independent anyway) and the
reference's outputs:
  * ``_attention.forward`` at BLKQ 128 / BLKK 64, D 128, ragged L (40-row tail in both block sizes): O and the LSE;
  * ``mean_pool`` at 128 and 64 (ragged), ``get_block_map`` (its own torch.topk on the device);
  * the whole ``SparseLinearAttention.forward`` (BLKQ 128 and the reference default 64), nothing patched;
  * ``rmsnorm``, ``layernorm`` (affine / no affine) at N = 1536 and the N <= 512 path.
``tests/test_oracle_cpu.py`` compares the oracle's restatements with these tensors and ``tests/test_gpu_sla.py`` /
``test_gpu_ops.py`` compare the HIP kernels with them directly.
"""
import os
import shutil
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
STAGE = os.path.join(HERE, "_ref", "triton")
REF = os.environ.get("TD_REFERENCE_ROOT", "/root/reference")
FILES = {  # staged name -> reference path
    "SLA/__init__.py": "turbodiffusion/SLA/__init__.py",
    "SLA/core.py": "turbodiffusion/SLA/core.py",
    "SLA/kernel.py": "turbodiffusion/SLA/kernel.py",
    "SLA/utils.py": "turbodiffusion/SLA/utils.py",
    "ref_ops_core.py": "turbodiffusion/ops/core.py",
}


def stage():
    for dst, src in FILES.items():
        d = os.path.join(STAGE, dst)
        os.makedirs(os.path.dirname(d), exist_ok=True)
        shutil.copyfile(os.path.join(REF, src), d)
    print("staged", len(FILES), "reference files under", STAGE)


def unstage():
    shutil.rmtree(STAGE, ignore_errors=True)
    print("removed", STAGE)


class _Ctx:
    """Stands in for the autograd context of ``_attention.forward``: keeps what it saves (the LSE is only there)."""

    def save_for_backward(self, *t):
        self.saved = t


def inputs():
    from .make_golden_c1 import det_normal
    H, L, D = 2, 5 * 128 + 40, 128           # 6 Q blocks (tail 40), 11 K blocks (tail 40)
    q = det_normal((1, H, L, D), 11).bfloat16()
    k = (det_normal((1, H, L, D), 12) + 0.5 * det_normal((1, H, 1, D), 13)).bfloat16()   # a non-zero sequence mean (smooth-K matters)
    v = det_normal((1, H, L, D), 14).bfloat16()
    xn = det_normal((200, 1536), 15)
    xn[:, 7::500] *= 20.0
    xs = det_normal((64, 384), 16)             # the N <= 512 (BLOCK_M = 32) path; M a multiple of 32 — the reference kernels
    #                                            do not mask rows (ops/core.py:110-117), a ragged M would write out of bounds
    w = 1.0 + 0.1 * det_normal((1536,), 17)
    b = 0.1 * det_normal((1536,), 18)
    ws = 1.0 + 0.1 * det_normal((384,), 19)
    bs = 0.1 * det_normal((384,), 20)
    pw = 0.05 * det_normal((D, D), 21)
    pb = 0.05 * det_normal((D,), 22)
    return dict(q=q, k=k, v=v, xn=xn, xs=xs, w=w, b=b, ws=ws, bs=bs, proj_w=pw, proj_b=pb, topk=0.3)


def run(out_path):
    assert os.path.isdir(STAGE), "run `python -m oracle.triton_leaves stage` in the build container first"
    sys.path.insert(0, STAGE)
    stub = types.ModuleType("turbo_diffusion_ops")     # ops/core.py imports the CUDA extension at module level (core.py:9)
    stub.quant_cuda = stub.gemm_cuda = None
    sys.modules["turbo_diffusion_ops"] = stub
    import triton
    import SLA
    import SLA.kernel as SK
    import SLA.utils as SU
    import ref_ops_core as RO

    dev = torch.device("cuda")
    I = inputs()
    q, k, v = (I[n].to(dev) for n in "qkv")
    res = {"inputs": I, "triton": triton.__version__, "device": torch.cuda.get_device_name(0)}
    with torch.no_grad():
        res["pool_q128"] = SU.mean_pool(q, 128).cpu()
        res["pool_k64"] = SU.mean_pool(k, 64).cpu()
        res["pool_q64"] = SU.mean_pool(q, 64).cpu()
        for blkq in (128, 64):
            smap, lut, topk = SU.get_block_map(q, k, I["topk"], BLKQ=blkq, BLKK=64)
            ctx = _Ctx()
            o = SK._attention.forward(ctx, q, k, v, smap, lut, topk, blkq, 64)
            lse = ctx.saved[5]
            res[f"map{blkq}"] = smap.cpu()
            res[f"lut{blkq}"] = lut.cpu()
            res[f"topk{blkq}"] = topk
            res[f"attn_o{blkq}"] = o.cpu()
            res[f"attn_lse{blkq}"] = lse.cpu()
            # the same kernel with the LUT in ASCENDING order (what SpargeAttn and the HIP path visit)
            lut_s = lut.sort(-1).values.contiguous()
            ctx = _Ctx()
            res[f"attn_o{blkq}_sorted"] = SK._attention.forward(ctx, q, k, v, smap, lut_s, topk, blkq, 64).cpu()
            res[f"attn_lse{blkq}_sorted"] = ctx.saved[5].cpu()
            # the whole module, nothing patched (inputs [B, L, H, D], SLA/core.py:93-95)
            m = SLA.SparseLinearAttention(128, I["topk"], BLKQ=blkq, BLKK=64).to(dev)
            m.proj_l.weight.copy_(I["proj_w"].to(dev))
            m.proj_l.bias.copy_(I["proj_b"].to(dev))
            om, sp = m(q.transpose(1, 2).contiguous(), k.transpose(1, 2).contiguous(), v.transpose(1, 2).contiguous(),
                       return_sparsity=True)
            res[f"sla_module{blkq}"] = om.cpu()
            res[f"sla_sparsity{blkq}"] = float(sp)
        xn, xs = I["xn"].to(dev), I["xs"].to(dev)
        w, b, ws, bs = (I[n].to(dev) for n in ("w", "b", "ws", "bs"))
        res["rms_n1536"] = RO.rmsnorm(xn, w, 1e-6).cpu()
        res["rms_n384"] = RO.rmsnorm(xs, ws, 1e-6).cpu()
        res["ln_affine_n1536"] = RO.layernorm(xn, w, b, 1e-6, True).cpu()
        res["ln_plain_n1536"] = RO.layernorm(xn, None, None, 1e-6, False).cpu()
        res["ln_affine_n384"] = RO.layernorm(xs, ws, bs, 1e-6, True).cpu()
        res["ln_plain_n384"] = RO.layernorm(xs, None, None, 1e-6, False).cpu()
        # FastRMSNorm / FastLayerNorm as the model calls them (ops/core.py:441-442,477-478): bf16 in, .float(), .to(bf16)
        xb = xn.bfloat16()
        res["fast_rms_bf16"] = RO.rmsnorm(xb.float(), w, 1e-6).to(torch.bfloat16).cpu()
        res["fast_ln_bf16"] = RO.layernorm(xb.float(), None, None, 1e-6, False).to(torch.bfloat16).cpu()
    torch.cuda.synchronize()
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    torch.save(res, out_path)
    print("wrote", out_path, "with", sorted(kk for kk in res if kk != "inputs"))


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else ""
    if cmd == "stage":
        stage()
    elif cmd == "unstage":
        unstage()
    elif cmd == "run":
        try:
            run(sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "triton_leaves.pt"))
        except Exception:   # "Triton refuses gfx950" is also an answer: keep the error text
            import traceback
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "triton_leaves_error.txt"), "w") as f:
                traceback.print_exc(file=f)
            raise
    else:
        print(__doc__)
