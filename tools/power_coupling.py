"""Does the attention kernel's efficiency change the speed of the GEMMs that follow it?  (round 3: three interleaved
end-to-end A/Bs of attention builds moved time between attention and the GEMMs without moving their sum.)

Sequence per repetition, on one stream, C1 shapes: [X] -> ffn.0 GEMM (GELU + quantiser epilogue) -> ffn.2 GEMM, with
X = nothing | the production attention build | the Q64 build (half the LDS reads, slower) | the prefetch build (OCC2) |
an idle gap as long as the attention kernel.  Every kernel is timed with its own pair of events; the table gives medians."""
import math
import statistics
import sys
import os

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from turbodiffusion_amd import kernels as K  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    L, dim, H, D, ffn = 32760, 1536, 12, 128, 8960
    qkv = torch.randn(L, 3 * dim, device=dev).bfloat16()
    w = torch.ones(dim, device=dev)
    ang = torch.rand(L, 64, device=dev) * 6
    cos, sin = torch.cos(ang), torch.sin(ang)
    q = K.qk_norm_rope(qkv, 0, H, D, w, cos, sin, 1e-6)
    k = K.qk_norm_rope(qkv, dim, H, D, w, cos, sin, 1e-6)
    vt = K.v_transpose(qkv[:, 2 * dim:], D, 3 * dim, L, H, D, torch.float16)
    km = K.seq_mean(k)
    pq, q8, qs = K.sage_quant_pool(q, None, 128)
    pk, k8, ks = K.sage_quant_pool(k, km, 64)
    kb = pk.shape[1]
    topk = int(0.1 * kb)
    # neighbouring Q blocks share most of their K blocks in the model; imitate that: a sliding window + a few far blocks
    qb = pq.shape[1]
    g = torch.Generator().manual_seed(1)
    base = (torch.arange(qb) * 2).clamp(max=kb - topk)
    lut = (base[:, None] + torch.arange(topk)[None, :]).int()
    lut = lut[None].repeat(H, 1, 1).contiguous().to(dev)
    out = torch.empty(L, H, D, device=dev, dtype=torch.bfloat16)
    a0 = torch.randn(L, dim, device=dev).bfloat16()
    a0q, a0s = K.quant_i8_block128(a0)
    w0q, w0s = K.quant_i8_block128((torch.randn(ffn, dim, device=dev) / math.sqrt(dim)).bfloat16())
    b0 = torch.zeros(ffn, device=dev).bfloat16()
    w2q, w2s = K.quant_i8_block128((torch.randn(dim, ffn, device=dev) / math.sqrt(ffn)).bfloat16())
    b2 = torch.zeros(dim, device=dev).bfloat16()

    def attn():
        K.attn_i8(q8, qs, k8, ks, vt, lut, out, D, H * D)

    def gemms(evs):
        evs[0].record()
        hq, hs = K.gemm_w8a8_quant(a0q, a0s, w0q, w0s, torch.bfloat16, bias=b0, gelu_tanh=True)
        evs[1].record()
        K.gemm_w8a8(hq, hs, w2q, w2s, torch.bfloat16, bias=b2)
        evs[2].record()

    ev = lambda: torch.cuda.Event(enable_timing=True)   # noqa: E731
    variants = [("no attention before", None), ("production build", 0), ("Q64 build", 3), ("prefetch build (OCC2)", 2), ("idle gap", "sleep")]
    res = {n: {"attn": [], "ffn0": [], "ffn2": []} for n, _ in variants}
    # idle gap ~ the attention kernel's duration in GPU cycles of torch.cuda._sleep (a spin on the shader clock)
    for rnd in range(4):
        for name, v in variants:
            if isinstance(v, int):
                K.set_tuning(K.TUNE_ATTN_OCC, v)
            for rep in range(12):
                e = [ev() for _ in range(5)]
                e[3].record()
                if v == "sleep":
                    torch.cuda._sleep(1_000_000)
                elif v is not None:
                    attn()
                e[4].record()
                gemms(e)
                torch.cuda.synchronize()
                if rep >= 2:
                    res[name]["attn"].append(e[3].elapsed_time(e[4]) * 1e3)
                    res[name]["ffn0"].append(e[0].elapsed_time(e[1]) * 1e3)
                    res[name]["ffn2"].append(e[1].elapsed_time(e[2]) * 1e3)
    K.set_tuning(K.TUNE_ATTN_OCC, 0)
    print(f"{'what ran before the two GEMMs':32s} {'X us':>9s} {'ffn.0 us':>9s} {'ffn.2 us':>9s} {'sum us':>9s}")
    for name, _ in variants:
        r = res[name]
        a, f0, f2 = (statistics.median(r[k_]) for k_ in ("attn", "ffn0", "ffn2"))
        print(f"{name:32s} {a:9.1f} {f0:9.1f} {f2:9.1f} {a + f0 + f2:9.1f}")


if __name__ == "__main__":
    main()
