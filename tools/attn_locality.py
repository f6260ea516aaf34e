"""How much of the sparse attention kernel's time is K/V fetch latency?  Same work, three LUTs: the real top-k map,
one fixed set of blocks for every Q block (K/V of a head = 1.2 MB: always in L2), and consecutive windows
(neighbouring Q blocks share most blocks)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from turbodiffusion_amd import kernels as K
dev = "cuda"
L, H, D, dim = 32760, 12, 128, 1536


def timeit(f, it=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


torch.manual_seed(0)
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
qb, kb = pq.shape[1], pk.shape[1]
topk = 51
lut_real = K.sla_topk(pq, pk, topk)
lut_fixed = torch.arange(topk, device=dev, dtype=torch.int32).expand(H, qb, topk).contiguous()
base = (torch.arange(qb, device=dev, dtype=torch.int32) * 2).clamp(max=kb - topk)
lut_win = (base[None, :, None] + torch.arange(topk, device=dev, dtype=torch.int32)[None, None, :]).expand(H, qb, topk).contiguous()
out = torch.empty(L, H, D, device=dev, dtype=torch.bfloat16)
for nm, lut in (("real top-k map", lut_real), ("same 51 blocks for every Q block (L2-resident K/V)", lut_fixed),
                ("sliding window of 51 blocks", lut_win)):
    t = timeit(lambda: K.attn_i8(q8, qs, k8, ks, vt, lut, out, D, H * D))
    print(json.dumps({"lut": nm, "us": round(t, 1)}), flush=True)
