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
@pytest.mark.parametrize("model_type,in_dim", [("i2v", 36), ("t2v", 16)])
def test_oracle_dit_matches_live_reference_wan2pt2(model_type, in_dim):
    """Wan2.2 (rcm/networks/wan2pt2.py:251-276 self-attention, :282 the plain text cross-attention for BOTH model types,
    :581-645 forward with ``y_B_C_T_H_W`` concatenated on channels — in_dim 36 = 16 + 4 mask + 16 image-latent channels,
    wan2.2_i2v_infer.py:149-152): the oracle's forward against the LIVE reference module, bit for bit."""
    warnings.filterwarnings("ignore")
    cfg = dict(dim=256, eps=1e-6, ffn_dim=384, freq_dim=256, in_dim=in_dim, model_type=model_type, num_heads=2,
               num_layers=2, out_dim=16, text_len=512, text_dim=64)
    sd = W.make_state_dict(cfg, seed=8)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 16, 2, 8, 10, generator=g)
    y = None
    if model_type == "i2v":
        y = torch.cat([torch.zeros(2, 4, 2, 8, 10), torch.randn(2, 16, 2, 8, 10, generator=g)], 1)
        y[:, :4, 0] = 1.0
    t = torch.tensor([[995.025], [852.895]])
    ctx = torch.randn(2, 512, 64, generator=g)
    net = rh.reference_wan_from_sd(cfg, sd, torch.bfloat16, which="wan2pt2")
    with torch.no_grad():
        ref = net(x.bfloat16(), t.bfloat16(), ctx.bfloat16(), y_B_C_T_H_W=None if y is None else y.bfloat16())
    out = W.wan_forward(sd, cfg, x, t.bfloat16(), ctx.bfloat16(), y_B_C_T_H_W=None if y is None else y.bfloat16(),
                        mode="eager")
    assert torch.equal(out, ref)
    if y is not None:   # the conditioning channels are really read
        out0 = W.wan_forward(sd, cfg, x, t.bfloat16(), ctx.bfloat16(), y_B_C_T_H_W=torch.zeros_like(y).bfloat16(), mode="eager")
        assert not torch.equal(out0, out)


@pytest.mark.skipif(not rh.available(), reason="/root/reference not present (GPU box)")
def test_oracle_dit_matches_live_reference_wan2pt1_i2v_clip_branch():
    """Wan2.1 I2V (rcm/networks/wan2pt1.py: ``model_type="i2v"`` -> ``WanI2VCrossAttention`` :303-352 with k_img / v_img /
    norm_k_img and ``MLPProj`` :457-486 on the 257 CLIP tokens, ``frame_cond_crossattn_emb_B_L_D`` + ``y`` both required :642):
    the oracle's forward with ``clip_emb`` against the LIVE reference module, bit for bit."""
    warnings.filterwarnings("ignore")
    cfg = dict(dim=256, eps=1e-6, ffn_dim=384, freq_dim=256, in_dim=36, model_type="i2v", num_heads=2, num_layers=2,
               out_dim=16, text_len=512, text_dim=64, clip_dim=1280)
    sd = W.make_state_dict(cfg, seed=11)
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 16, 2, 8, 10, generator=g)
    y = torch.cat([torch.zeros(2, 4, 2, 8, 10), torch.randn(2, 16, 2, 8, 10, generator=g)], 1)
    y[:, :4, 0] = 1.0
    t = torch.tensor([[995.025], [852.895]])
    ctx = torch.randn(2, 512, 64, generator=g)
    clip = torch.randn(2, 257, 1280, generator=g)
    net = rh.reference_wan_from_sd(cfg, sd, torch.bfloat16)
    with torch.no_grad():
        ref = net(x.bfloat16(), t.bfloat16(), ctx.bfloat16(), frame_cond_crossattn_emb_B_L_D=clip.bfloat16(), y_B_C_T_H_W=y.bfloat16())
    out = W.wan_forward(sd, cfg, x, t.bfloat16(), ctx.bfloat16(), y_B_C_T_H_W=y.bfloat16(), mode="eager", clip_emb=clip.bfloat16())
    assert torch.equal(out, ref)
    out0 = W.wan_forward(sd, cfg, x, t.bfloat16(), ctx.bfloat16(), y_B_C_T_H_W=y.bfloat16(), mode="eager", clip_emb=(0.5 * clip).bfloat16())
    assert not torch.equal(out0, out)      # the image tokens are really read


@pytest.mark.skipif(not rh.available(), reason="/root/reference not present (GPU box)")
def test_oracle_rope_and_freqs_match_reference():
    mod = rh.load("wan2pt1")
    emb = mod.VideoRopePosition3DEmb(head_dim=128, len_h=128, len_w=128, len_t=32)
    f_ref = emb.generate_embeddings(torch.Size([1, 5, 8, 12, 256]))
    assert torch.equal(O.rope_freqs(5, 8, 12, 128), f_ref)
    x = torch.randn(1, 480, 2, 128).bfloat16()
    assert torch.equal(O.rope_apply(x, f_ref), mod.rope_apply(x, f_ref))
    assert torch.equal(O.sinusoidal_embedding_1d(256, torch.tensor([933.5])), mod.sinusoidal_embedding_1d(256, torch.tensor([933.5])))


def _sla_case(seed, B, L, H, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    q, k, v = [torch.randn(B, L, H, 128, generator=g).to(dtype) for _ in range(3)]
    wp = torch.randn(128, 128, generator=g) * 0.05
    bp = torch.randn(128, generator=g) * 0.05
    return q, k, v, wp, bp


def _ref_module(sla, cls, wp, bp, *a, **kw):
    m = getattr(sla, cls)(128, *a, **kw)
    with torch.no_grad():
        m.proj_l.weight.copy_(wp)
        m.proj_l.bias.copy_(bp)
    return m


@pytest.mark.skipif(not rh.available(), reason="/root/reference not present (GPU box)")
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("blkq", [128, 64])
def test_oracle_sla_matches_live_reference_module(dtype, blkq):
    """The reference's own SparseLinearAttention.forward (SLA/core.py:83-119) and get_block_map (SLA/utils.py:55-67) on
    the CPU with only the Triton leaves replaced (oracle/ref_harness.patched_sla) == oracle/sla_ref.sla_forward bit for
    bit: [B,L,H,D] transposes, casts, block map, the 16-bit o_s + o_l, proj_l under autocast, return_sparsity."""
    warnings.filterwarnings("ignore")
    q, k, v, wp, bp = _sla_case(11, 2, 777, 3, dtype)
    with torch.no_grad(), rh.patched_sla("sm80") as sla:
        m = _ref_module(sla, "SparseLinearAttention", wp, bp, 0.3, BLKQ=blkq, BLKK=64)
        ref, sparsity = m(q, k, v, return_sparsity=True)
        rec = dict(sla._td_recorded)
    assert ref.dtype == dtype and ref.shape == q.shape
    # the block map: same selected SET (torch.topk(sorted=False) leaves the order open)
    smap, lut, topk = S.get_block_map(q.transpose(1, 2).contiguous(), k.transpose(1, 2).contiguous(), 0.3, blkq, 64)
    assert topk == rec["real_topk"] and sparsity == topk / smap.shape[-1]
    assert torch.equal(smap, rec["sparse_map"]), "selected block sets differ"
    # same visiting order as the reference handed its kernel -> bit-identical output
    mine = S.sla_forward(q, k, v, wp, bp, 0.3, blkq=blkq, blkk=64, lut=rec["lut"])
    assert torch.equal(mine, ref)
    # ascending order (what the HIP path and SpargeAttn use): the online softmax is order dependent at rounding level only
    asc = S.sla_forward(q, k, v, wp, bp, 0.3, blkq=blkq, blkk=64)
    assert rel_l2(asc, ref) < 2e-3


@pytest.mark.skipif(not rh.available(), reason="/root/reference not present (GPU box)")
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("arch", ["sm80", "sm89"])
def test_oracle_sagesla_matches_live_reference_module(dtype, arch):
    """SageSparseLinearAttention.forward (SLA/core.py:168-258), FP16-PV (sm80) and FP8-PV (sm89) branches, with the
    SpargeAttn / Triton leaves replaced by the oracle's statement of them: the reference's composition (smooth-K mean,
    per-block quantiser call, delta LUT + valid counts, pv threshold, v -> fp16 or transposed/padded fp8 + v_scale,
    softmax scale, linear branch, o_s + o_l) == sla_ref.sagesla_forward[_fp8] bit for bit."""
    warnings.filterwarnings("ignore")
    q, k, v, wp, bp = _sla_case(12, 2, 600, 2, dtype)
    with torch.no_grad(), rh.patched_sla(arch) as sla:
        m = _ref_module(sla, "SageSparseLinearAttention", wp, bp, 0.25)
        ref, sparsity = m(q, k, v, return_sparsity=True)
    fwd = S.sagesla_forward if arch == "sm80" else S.sagesla_forward_fp8
    assert torch.equal(fwd(q, k, v, wp, bp, 0.25), ref)
    assert sparsity == pytest.approx(int(0.25 * 10) / 10)


LEAVES = os.path.join(os.path.dirname(__file__), "golden", "triton_leaves.pt")


@pytest.fixture(scope="module")
def leaves():
    """What the reference's OWN Triton kernels produced on an MI355X (oracle/triton_leaves.py; inputs from the hash)."""
    from oracle.triton_leaves import inputs
    g = torch.load(LEAVES, weights_only=False)
    g["inputs"] = inputs()
    return g


def test_oracle_norms_match_the_reference_triton_kernels(leaves):
    """a5 / a6: ops/core.py:96-136 (RMSNorm), :193-242 / :293-335 (LayerNorm, affine and plain) as EXECUTED, N = 1536
    (N2 = 2048: 512 phantom columns in the LayerNorm variance) and the N <= 512 path.  fp32 outputs to 1e-6 (reduction
    order), the bf16 Fast-module outputs to one ulp on < 1 % of the values."""
    I = leaves["inputs"]
    xn, xs, w, b, ws, bs = I["xn"][:64], I["xs"], I["w"], I["b"], I["ws"], I["bs"]
    for got, name in ((O.rmsnorm_fast(xn, w, 1e-6), "rms_n1536"), (O.rmsnorm_fast(xs, ws, 1e-6), "rms_n384"),
                      (O.layernorm_fast(xn, w, b, 1e-6), "ln_affine_n1536"), (O.layernorm_fast(xn, None, None, 1e-6), "ln_plain_n1536"),
                      (O.layernorm_fast(xs, ws, bs, 1e-6), "ln_affine_n384"), (O.layernorm_fast(xs, None, None, 1e-6), "ln_plain_n384")):
        torch.testing.assert_close(got, leaves[name], rtol=2e-6, atol=6e-6, msg=name)
        assert rel_l2(got, leaves[name]) < 3e-7, name
    # the phantom-column term is what makes them agree: the textbook variance is 1e-3 away on these rows
    assert rel_l2(O.layernorm_fast(xn, None, None, 1e-6, triton_variance=False), leaves["ln_plain_n1536"]) > 1e-4
    xb = xn.bfloat16()
    for got, name in ((O.rmsnorm_fast(xb.float(), w, 1e-6).bfloat16(), "fast_rms_bf16"),
                      (O.layernorm_fast(xb.float(), None, None, 1e-6).bfloat16(), "fast_ln_bf16")):
        ulp = (got.view(torch.int16).int() - leaves[name].view(torch.int16).int()).abs()
        assert ulp.max().item() <= 1 and (ulp > 0).float().mean().item() < 0.01, name


def test_oracle_sla_leaves_match_the_reference_triton_kernels(leaves):
    """a11 / a12: ``compress_kernel`` / ``mean_pool`` (SLA/utils.py:21-52) and ``get_block_map`` (:55-67) bit for bit;
    ``_attn_fwd`` (SLA/kernel.py:21-82) O equal on > 99.5 % of the values and one bf16 rounding step apart on the rest
    (fp32 accumulation order inside the matrix instructions); BLKQ 128 and 64, ragged L (40-row tails), the reference's LUT order and
    ascending order."""
    I = leaves["inputs"]
    q, k, v = I["q"], I["k"], I["v"]
    assert torch.equal(S.mean_pool(q, 128), leaves["pool_q128"])
    assert torch.equal(S.mean_pool(k, 64), leaves["pool_k64"])
    assert torch.equal(S.mean_pool(q, 64), leaves["pool_q64"])
    for blkq in (128, 64):
        smap, lut, topk = S.get_block_map(q, k, I["topk"], blkq, 64)
        assert topk == leaves[f"topk{blkq}"] == 3
        assert torch.equal(smap, leaves[f"map{blkq}"])                       # same selected sets as the device topk
        for sfx, lut_ref in (("", leaves[f"lut{blkq}"]), ("_sorted", leaves[f"lut{blkq}"].sort(-1).values)):
            o = S.sla_sparse_attn(q, k, v, lut_ref, blkq, 64)
            ref = leaves[f"attn_o{blkq}{sfx}"]
            # one bf16 rounding step of the larger values (outputs are O(1): 2^-8 relative, 2e-3 absolute near zero)
            torch.testing.assert_close(o.float(), ref.float(), rtol=2 ** -7, atol=2e-3)
            assert (o != ref).float().mean().item() < 5e-3, (blkq, sfx)
            assert rel_l2(o, ref) < 2e-4


def test_oracle_sla_module_matches_the_reference_module_run_on_the_gpu(leaves):
    """The whole ``SparseLinearAttention.forward`` (SLA/core.py:83-119) as the reference ran it on the MI355X — its own
    Triton pooling / attention kernels, device topk, bf16 matmuls of the linear branch, ``proj_l`` under CUDA autocast —
    against the oracle's module restatement: the only differences are GPU-vs-CPU matmul accumulation orders."""
    I = leaves["inputs"]
    q, k, v = (I[n].transpose(1, 2).contiguous() for n in "qkv")          # [B, L, H, D]
    for blkq in (128, 64):
        o = S.sla_forward(q, k, v, I["proj_w"], I["proj_b"], I["topk"], blkq, 64, torch.bfloat16)
        ref = leaves[f"sla_module{blkq}"]
        assert o.shape == ref.shape
        assert rel_l2(o, ref) < 5e-3, (blkq, rel_l2(o, ref))
        assert abs(leaves[f"sla_sparsity{blkq}"] - 3 / 11) < 1e-9



def test_delta_lut_roundtrip():
    g = torch.Generator().manual_seed(0)
    smap = (torch.rand(1, 2, 5, 17, generator=g) < 0.3).to(torch.int8)
    smap[0, 0, 0] = 0
    smap[0, 0, 0, 16] = 1
    delta, valid = S.block_map_lut(smap)
    ids = S.lut_from_delta(delta)
    for h in range(2):
        for qb in range(5):
            n = int(valid[0, h, qb])
            assert ids[0, h, qb, :n].tolist() == torch.nonzero(smap[0, h, qb]).flatten().tolist()
            assert delta[0, h, qb, n:].abs().sum() == 0


def test_sla_golden_fixture_matches_oracle():
    """tests/golden/sla_tiny.pt holds tensors produced by the REAL reference modules (oracle/make_golden.py); the oracle
    reproduces them on any box (this is what pins the oracle on the GPU box, where /root/reference is absent)."""
    gold = torch.load(os.path.join(os.path.dirname(GOLD), "sla_tiny.pt"), weights_only=False)
    q, k, v, wp, bp, topk = gold["q"], gold["k"], gold["v"], gold["proj_w"], gold["proj_b"], gold["topk"]
    assert torch.equal(S.sla_forward(q, k, v, wp, bp, topk, lut=gold["sla_lut"]), gold["ref_sla"])
    assert torch.equal(S.sagesla_forward(q, k, v, wp, bp, topk), gold["ref_sagesla_f16pv"])
    assert torch.equal(S.sagesla_forward_fp8(q, k, v, wp, bp, topk), gold["ref_sagesla_fp8pv"])
    smap, _, _ = S.get_block_map(q.transpose(1, 2).contiguous(), k.transpose(1, 2).contiguous(), topk, 128, 64)
    assert torch.equal(smap, gold["sparse_map"])


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
    assert lib.td_abi_version() == L.ABI_VERSION == 5


def test_ops_fail_loudly_without_gpu():
    import turbodiffusion_amd.ops as ops
    from turbodiffusion_amd._lib import TurboDiffusionAMDError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(TurboDiffusionAMDError):
        ops.int8_quant(torch.zeros(128, 128, dtype=torch.bfloat16))
    with pytest.raises(TurboDiffusionAMDError):
        ops.rmsnorm(torch.zeros(4, 128), torch.ones(128), 1e-6)


def test_fused_weight_views_and_cache_invalidation():
    """the q|k|v / cross k|v concatenations must never go stale.  The per-module tensors are views of the
    concatenation (in-place updates propagate, one copy in memory); load_state_dict and .to()/.half() drop the derived
    copies and bump the epoch GraphedModel re-captures on; fp32 buffers the kernels read as fp32 survive .to(bf16)."""
    from turbodiffusion_amd.wan import WanModel
    cfg = dict(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64, in_dim=16, out_dim=16)
    for quant in (True, False):
        net = WanModel(attention_type="sagesla", quant_linear=quant, **cfg)
        with torch.no_grad():
            for b in net.buffers():
                if b.dtype == torch.int8:
                    b.copy_(torch.randint(-128, 128, b.shape, dtype=torch.int8))
                else:
                    b.copy_(torch.randn(b.shape))
        blk = net.blocks[0]
        attr = "int8_weight" if quant else "weight"
        before = getattr(blk.self_attn.k, attr).detach().clone()
        keys = set(net.state_dict())
        f = net._fused_weights(0, blk)
        allw = net._text_kv_weights()
        assert set(net.state_dict()) == keys and torch.equal(getattr(blk.self_attn.k, attr), before)
        assert torch.equal(f["qkv_w"][256:512], before)
        # views: an in-place update of a module's weight IS an update of the fused tensors
        with torch.no_grad():
            getattr(blk.self_attn.k, attr).zero_()
            getattr(net.blocks[1].cross_attn.v, attr).fill_(3)
        assert float(f["qkv_w"][256:512].float().abs().sum()) == 0.0
        assert float((allw["ckv_w"][3 * 256:4 * 256].float() - 3).abs().sum()) == 0.0
        assert getattr(blk.self_attn.k, attr).data_ptr() == f["qkv_w"][256:512].data_ptr()
        # load_state_dict drops the derived copies and bumps the epoch
        e0 = net._weights_epoch
        net.load_state_dict(net.state_dict())
        assert net._fused == {} and net._ckv_all is None and net._weights_epoch > e0
        # .to(bf16): proj_l etc. convert, the fp32 buffers the kernels read as fp32 do not
        net._fused_weights(0, blk)
        e1 = net._weights_epoch
        net.to(torch.bfloat16)
        assert net._fused == {} and net._weights_epoch > e1
        assert blk.self_attn.norm_q.weight.dtype == torch.float32 and blk.norm3.weight.dtype == torch.float32
        if quant:
            assert blk.ffn[0].scale.dtype == torch.float32 and blk.ffn[0].int8_weight.dtype == torch.int8
        assert net.patch_embedding.weight.dtype == torch.bfloat16


def test_sla_modules_name_what_is_supported():
    """The reference's constructor surface (SLA/core.py:38-77, 122-161): the class defaults construct (BLKQ = 64, the three
    feature maps); what the MI355X kernels are not built for fails at construction with a message naming the supported set
    (never a bare assert, never a silent mismatch under python -O)."""
    from turbodiffusion_amd.sla import SageSparseLinearAttention, SparseLinearAttention
    m = SparseLinearAttention(128, 0.1)                       # reference defaults: BLKQ = 64, BLKK = 64, softmax (SLA/core.py:39)
    assert (m.BLKQ, m.BLKK, m.feature_map, m.dtype) == (64, 64, "softmax", torch.bfloat16)
    for fm in ("elu", "relu", "softmax"):
        assert SparseLinearAttention(128, 0.1, feature_map=fm, BLKQ=128, BLKK=64).feature_map == fm
        assert SageSparseLinearAttention(128, 0.1, feature_map=fm, use_bf16=False).dtype == torch.float16
    with pytest.raises(ValueError, match="head_dim = 128"):
        SageSparseLinearAttention(64, 0.1)                    # SLA/core.py:207 allows 64
    with pytest.raises(ValueError, match="BLKK = 64"):
        SparseLinearAttention(128, 0.1, BLKQ=128, BLKK=128)
    with pytest.raises(NotImplementedError, match="Not supported feature map"):
        SageSparseLinearAttention(128, 0.1, feature_map="hedgehog")     # the reference raises the same (SLA/core.py:75)
    m = SparseLinearAttention(128, 0.1, BLKQ=128, BLKK=64)
    assert m.proj_l.weight.dtype == torch.float32 and float(m.proj_l.weight.abs().sum()) == 0.0


def test_compat_module_exports_the_pybind_names():
    """ops/bindings.cpp:11-16 registers quant_cuda, rms_norm_cuda, layer_norm_cuda, gemm_cuda; ops/core.py:9 imports two."""
    from turbodiffusion_amd.compat import turbo_diffusion_ops as T
    from turbodiffusion_amd._lib import TurboDiffusionAMDError
    for name in ("quant_cuda", "gemm_cuda", "rms_norm_cuda", "layer_norm_cuda"):
        assert callable(getattr(T, name))
    if not torch.cuda.is_available():
        with pytest.raises(TurboDiffusionAMDError):
            T.quant_cuda(torch.zeros(128, 128, dtype=torch.bfloat16), None, None)


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


def test_gemm_refuses_operands_beyond_32_bit_offsets():
    """The 256x256 GEMM kernels address A and B through 32-bit offsets: m*k or n*k >= 2^32 must be an error status at the
    C-ABI (checked before anything is dereferenced — no GPU needed), never a wrapped address.  C4's ffn.2 at B = 1 is
    1.05e9 (fine); five samples of it are not."""
    import ctypes
    from turbodiffusion_amd import _lib as L
    lib = L.load()
    p = ctypes.c_void_p(4096)
    big_m, n, k = 5 * 75600, 5120, 13824
    assert big_m * k >= 2 ** 32
    for name, args in (
            ("td_gemm_w8a8", (p, p, p, p, None, p, L.TD_BF16, 0, big_m, n, k, n, None)),
            ("td_gemm_w8a8_quant", (p, p, p, p, None, p, p, L.TD_BF16, 0, big_m, n, k, None)),
            ("td_gemm_w8a8_residual", (p, p, p, p, None, p, None, L.TD_BF16, big_m, n, k, n, None)),
            ("td_gemm_w8a8_stats", (p, p, p, p, p, p, None, 1, L.TD_BF16, big_m, n, k, n, p, None)),
            ("td_gemm_w8a8_vt", (p, p, p, p, p, p, L.TD_BF16, big_m, 15360, k, 15360, 10240, p, L.TD_F16, None))):
        rc = getattr(lib, name)(*args)
        assert rc != 0 and b"2^32" in lib.td_last_error(), (name, rc, lib.td_last_error())
    # the same shapes one sample at a time pass this check (they fail later only for want of a GPU, which we do not reach:
    # m = 0 returns TD_OK before any launch)
    assert lib.td_gemm_w8a8(p, p, p, p, None, p, L.TD_BF16, 0, 0, n, k, n, None) == 0
