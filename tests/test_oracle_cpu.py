"""not-gpu tests: the oracle against the reference's golden fixtures (and the live reference when
/root/reference is present), closed-form properties of the restatements, host logic, C-ABI exports."""
import os
import re
import warnings

import pytest
import torch

from oracle import ops_ref as O
from oracle import ref_harness as rh
from oracle import sla_ref as S
from oracle import wan_ref as W
from tests.util import rel_l2

GOLD = os.path.join(os.path.dirname(__file__), "golden", "wan_tiny.pt")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gold():
    return torch.load(GOLD, weights_only=False)


def test_oracle_dit_matches_reference_golden_fp32(gold):
    sd = W.make_state_dict(gold["cfg"], gold["sd_seed"])
    y = W.wan_forward(sd, gold["cfg"], gold["x"], gold["t"], gold["ctx"], mode="eager", act_dtype=torch.float32)
    assert torch.equal(y, gold["ref_fp32"]), "oracle eager fp32 must be bit-identical to the reference"


def test_oracle_dit_matches_reference_golden_bf16(gold):
    sd = W.make_state_dict(gold["cfg"], gold["sd_seed"])
    y = W.wan_forward(sd, gold["cfg"], gold["x"], gold["t"].bfloat16(), gold["ctx"].bfloat16(), mode="eager",
                      act_dtype=torch.bfloat16)
    assert torch.equal(y, gold["ref_bf16"]), "oracle eager bf16 emulation must be bit-identical"


def test_oracle_sampler_matches_reference_golden(gold):
    sd = W.make_state_dict(gold["cfg"], gold["sd_seed"])
    ctx = gold["ctx"].bfloat16()
    out = W.rcm_sample(lambda x, t: W.wan_forward(sd, gold["cfg"], x, t, ctx, mode="eager"), gold["x"], gold["noises"])
    assert torch.equal(out, gold["ref_sample_bf16"])


def test_rcm_timesteps_values():
    # sigma_max = 80 -> [0.987654, 0.933781, 0.852895, 0.608979, 0]
    t = W.rcm_timesteps(4, 80.0)
    torch.testing.assert_close(t, torch.tensor([0.987654, 0.933781, 0.852895, 0.608979, 0.0], dtype=torch.float64),
                               atol=2e-6, rtol=0)


@pytest.mark.skipif(not rh.available(), reason="/root/reference not present (GPU box)")
def test_oracle_dit_matches_live_reference():
    warnings.filterwarnings("ignore")
    cfg = dict(dim=256, eps=1e-6, ffn_dim=384, freq_dim=256, in_dim=16, model_type="t2v", num_heads=2, num_layers=1,
               out_dim=16, text_len=512, text_dim=64)
    sd = W.make_state_dict(cfg, seed=5)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 16, 2, 8, 10, generator=g)  # B = 2 exercises the per-sample modulation
    t = torch.tensor([[987.654], [608.979]])
    ctx = torch.randn(2, 512, 64, generator=g)
    net = rh.reference_wan_from_sd(cfg, sd, torch.bfloat16)
    with torch.no_grad():
        ref = net(x.bfloat16(), t.bfloat16(), ctx.bfloat16())
    y = W.wan_forward(sd, cfg, x, t.bfloat16(), ctx.bfloat16(), mode="eager")
    assert torch.equal(y, ref)


@pytest.mark.skipif(not rh.available(), reason="/root/reference not present (GPU box)")
def test_oracle_rope_and_freqs_match_reference():
    mod = rh.load("wan2pt1")
    emb = mod.VideoRopePosition3DEmb(head_dim=128, len_h=128, len_w=128, len_t=32)
    f_ref = emb.generate_embeddings(torch.Size([1, 5, 8, 12, 256]))
    assert torch.equal(O.rope_freqs(5, 8, 12, 128), f_ref)
    x = torch.randn(1, 480, 2, 128).bfloat16()
    assert torch.equal(O.rope_apply(x, f_ref), mod.rope_apply(x, f_ref))
    assert torch.equal(O.sinusoidal_embedding_1d(256, torch.tensor([933.5])), mod.sinusoidal_embedding_1d(256, torch.tensor([933.5])))


# ---------------------------------------------------------------- closed-form properties of restatements
def test_quant_block128_properties():
    x = torch.randn(300, 384).bfloat16()
    q, s = O.quant_block128(x)
    assert q.dtype == torch.int8 and s.shape == (3, 3)
    # |dequant - x| <= scale/2 inside every block, +amax -> 127 (multiplier 128/amax saturates)
    deq = q.float() * s.repeat_interleave(128, 0)[:300].repeat_interleave(128, 1)[:, :384]
    assert ((deq - x.float()).abs() <= s.max() * 1.0 + 1e-6).all()
    blk = x[:128, :128].float()
    i = blk.abs().argmax()
    assert abs(int(q[:128, :128].flatten()[i])) in (127, 128)
    # tail block scale only covers valid rows
    assert s[2, 0].item() == pytest.approx(x[256:300, :128].float().abs().max().item() / 128, rel=1e-6)


def test_gemm_w8a8_matches_dequantised_matmul():
    x = torch.randn(200, 256).bfloat16()
    w = (torch.randn(136, 256) / 16).bfloat16()
    xq, xs = O.quant_block128(x)
    wq, ws = O.quant_block128(w)
    y = O.gemm_w8a8(xq, xs, wq, ws)
    xd = xq.float() * xs.repeat_interleave(128, 0)[:200].repeat_interleave(128, 1)
    wd = wq.float() * ws.repeat_interleave(128, 0)[:136].repeat_interleave(128, 1)
    assert rel_l2(y, xd @ wd.t()) < 4e-3   # only the bf16 output rounding differs
    assert rel_l2(y, x.float() @ w.float().t()) < 2e-2


def test_fused_adaln_closed_form():
    # the closed forms of TurboT2AV/.../test_transformer_fusion_helpers.py:25-77, on the Wan glue
    torch.manual_seed(7)
    x = torch.randn(2, 5, 64).bfloat16()
    scale, shift, gate = torch.randn(2, 1, 64), torch.randn(2, 1, 64), torch.randn(2, 1, 64)
    xn = O.layernorm_fast(x, None, None, 1e-6)
    torch.testing.assert_close(O.modulate(xn, scale, shift).float(), xn.float() * (1 + scale) + shift, atol=4e-2, rtol=2e-2)
    res = torch.randn(2, 5, 64).bfloat16()
    torch.testing.assert_close(O.gated_residual(x, res, gate).float(), x.float() + res.float() * gate, atol=6e-2, rtol=2e-2)


def test_sage_dense_equals_softmax_attention_within_tolerance():
    from tests.test_gpu_sla import qkv
    q, k, v = qkv(2, 300, 0)
    km = S.seq_mean(k)
    q8, qs = S.quant_per_block_int8(q, 128)
    k8, ks = S.quant_per_block_int8(k, 64, km)
    o = S.sage_sparse_attn(q8, qs, k8, ks, v, S.dense_lut(1, 2, 300, 128, 64))
    assert rel_l2(o, S.sdpa_ref(q, k, v)) < 3e-2
    # smooth-K invariance: the K mean only shifts each score row by a constant
    o2 = S.sla_sparse_attn(q, (k.float() - km.float()).bfloat16(), v, S.dense_lut(1, 2, 300, 128, 64))
    assert rel_l2(o2, S.sdpa_ref(q, k, v)) < 1e-2


def test_block_map_topk_rule():
    score = torch.tensor([[[[1.0, 3.0, 3.0, 2.0, 3.0]]]]).bfloat16()
    assert S.select_topk(score, 2).tolist() == [[[[1, 2]]]]  # ties -> lower index, ascending output
    assert S.select_topk(score, 4).tolist() == [[[[1, 2, 3, 4]]]]


# ---------------------------------------------------------------- host logic / C-ABI
def test_library_exports_every_declared_symbol():
    import turbodiffusion_amd._lib as L
    from turbodiffusion_amd import build
    build.build(verbose=False)
    lib = L.load()
    hdr = open(os.path.join(ROOT, "include", "turbodiffusion_amd.h")).read()
    declared = set(re.findall(r"\b(td_[a-z0-9_]+)\s*\(", hdr)) - {"td_stream_t"}
    assert declared, "no prototypes found"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/turbodiffusion_amd.h but not exported"
        assert name in L.SIGNATURES, f"{name} has no ctypes signature"
    assert lib.td_abi_version() == 1


def test_ops_fail_loudly_without_gpu():
    import turbodiffusion_amd.ops as ops
    from turbodiffusion_amd._lib import TurboDiffusionAMDError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(TurboDiffusionAMDError):
        ops.int8_quant(torch.zeros(128, 128, dtype=torch.bfloat16))
    with pytest.raises(TurboDiffusionAMDError):
        ops.rmsnorm(torch.zeros(4, 128), torch.ones(128), 1e-6)


def test_module_tree_matches_reference_checkpoint_keys():
    """State-dict keys of our WanModel == keys of the reference model after modify_model.replace_*
    (SURVEY §8b 'Checkpoint contract')."""
    from turbodiffusion_amd.wan import WanModel
    cfg = dict(dim=256, ffn_dim=512, num_heads=2, num_layers=1, text_dim=64, in_dim=16, out_dim=16)
    with torch.device("meta"):
        net = WanModel(attention_type="sagesla", quant_linear=True, **cfg)
    keys = set(net.state_dict().keys())
    expect = {"patch_embedding.weight", "text_embedding.0.weight", "text_embedding.2.bias", "time_embedding.0.weight",
              "time_projection.1.weight", "head.head.weight", "head.modulation", "blocks.0.modulation",
              "blocks.0.self_attn.q.int8_weight", "blocks.0.self_attn.q.scale", "blocks.0.self_attn.q.bias",
              "blocks.0.self_attn.norm_q.weight", "blocks.0.self_attn.attn_op.local_attn.proj_l.weight",
              "blocks.0.self_attn.attn_op.local_attn.proj_l.bias", "blocks.0.cross_attn.v.int8_weight",
              "blocks.0.cross_attn.norm_k.weight", "blocks.0.norm3.weight", "blocks.0.norm3.bias",
              "blocks.0.ffn.0.int8_weight", "blocks.0.ffn.2.scale"}
    assert expect <= keys, expect - keys
    assert not any("norm1" in k or "norm2" in k for k in keys)  # no-affine norms carry no tensors
    sd = net.state_dict()
    assert sd["blocks.0.ffn.0.int8_weight"].shape == (512, 256) and sd["blocks.0.ffn.0.scale"].shape == (4, 2)
    if rh.available():
        warnings.filterwarnings("ignore")
        ref = rh.load("wan2pt1").WanModel(model_type="t2v", text_len=512, freq_dim=256, eps=1e-6, **cfg)
        ref_keys = set(ref.state_dict().keys())
        ours_float = {k.replace(".int8_weight", ".weight") for k in keys if not k.endswith(".scale")}
        assert ref_keys <= ours_float | {k for k in ref_keys if "proj_l" in k}, ref_keys - ours_float


def test_checkpoint_converter_key_contract():
    """turbodiffusion_amd.convert: `net.` prefix / state_dict unwrapping / patch-embedding reshape of
    modify_model.py:161-171, and float -> *-quant.pth key mapping (Int8Linear for every Linear under blocks except
    proj_l) lands exactly on the model's own state-dict keys."""
    from turbodiffusion_amd import convert as C
    from turbodiffusion_amd.wan import WanModel
    cfg = dict(dim=256, ffn_dim=512, num_heads=2, num_layers=2, in_dim=16, out_dim=16, text_dim=64, freq_dim=64)
    sd = W.make_state_dict(cfg, 0)
    wrapped = {"state_dict": {"net." + k: (v.reshape(-1) if k.startswith("patch_embedding") else v) for k, v in sd.items()}}
    n = C.normalize_checkpoint(wrapped, sd["patch_embedding.weight"].shape, sd["patch_embedding.bias"].shape)
    assert set(n) == set(sd) and n["patch_embedding.weight"].shape == sd["patch_embedding.weight"].shape
    with torch.device("meta"):
        net = WanModel(attention_type="sagesla", quant_linear=True, **cfg)
    own = set(net.state_dict())
    mapped = set()
    for k, v in sd.items():
        if C._is_block_linear_weight(k, v):
            mapped |= {k[:-7] + ".int8_weight", k[:-7] + ".scale"}
        else:
            mapped.add(k)
    assert mapped == own, (sorted(mapped - own)[:4], sorted(own - mapped)[:4])
    assert not any("proj_l" in k and k.endswith("int8_weight") for k in mapped)
    assert not C.is_quantized(sd) and C.is_quantized({"blocks.0.ffn.0.int8_weight": None})
