import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sla_ref as S
from turbodiffusion_amd import kernels as K
from tests.test_gpu_sla import _sage_inputs
from tests.util import cosine, rel_l2
DEV = "cuda"
for (H, L, ratio) in ((2, 1000, 0.3), (1, 128, 1.0), (1, 64, 1.0)):
    q, k, v, q_i8, q_s, k_i8, k_s = _sage_inputs(H, L, 5)
    _, lut, topk = S.get_block_map(q, k, ratio, 128, 64)
    vt = torch.empty(1, H, 128, (L + 127) // 128 * 128, dtype=v.dtype)
    S.transpose_pad_permute(v, vt)
    v8, vs = S.v_fp8_quant(vt, L, 2.25)
    ref = S.sage_sparse_attn_fp8(q_i8, q_s, k_i8, k_s, v8, vs, lut, out_dtype=torch.bfloat16)[0]
    ref16 = S.sage_sparse_attn(q_i8, q_s, k_i8, k_s, v, lut, out_dtype=torch.bfloat16)[0]
    vl = v[0].transpose(0, 1).contiguous().to(DEV)
    vt8, vsc = K.v_fp8_tiles(vl, 128, H * 128, L, H, 128, 2.25)
    out = torch.empty(H, L, 128, dtype=torch.bfloat16, device=DEV)
    dense = ratio >= 1.0
    K.attn_i8(q_i8[0].to(DEV), q_s[0].to(DEV), k_i8[0].to(DEV), k_s[0].to(DEV), vt8,
              None if dense else lut[0].int().to(DEV), out, L * 128, 128, v_scale=vsc)
    o = out.float().cpu()
    print(H, L, ratio, "cos", cosine(out, ref), "rel", rel_l2(out, ref), "oracle fp8 vs f16", rel_l2(ref, ref16), "hip vs f16", rel_l2(out, ref16))
    # per-row error profile
    err = (o - ref.float()).norm(dim=-1) / ref.float().norm(dim=-1)
    print("  per-row rel err head0: first rows", err[0, :8].tolist(), "max row", err.max().item(), "argmax", err.argmax().item())
    d_err = (o - ref.float()).abs().mean(dim=(0, 1))
    print("  per-d mean abs err (first 16):", [round(x, 4) for x in d_err[:16].tolist()], "ratio out/ref per d:", [round(x, 3) for x in (o.abs().mean(dim=(0,1)) / ref.float().abs().mean(dim=(0,1)))[:8].tolist()])
