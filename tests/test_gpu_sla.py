"""-m gpu parity tests: attention path (block map, Sage quant, sparse/dense attention, linear branch)
through the C-ABI vs the CPU oracle."""

import pytest
import torch

from oracle import sla_ref as S
from tests.util import cosine, rel_l2, ulp_diff_bf16

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def K():
    from turbodiffusion_amd import kernels
    return kernels


def qkv(H, L, seed, dtype=torch.bfloat16, D=128):
    """q,k with unit-RMS rows plus a per-head offset on k (what smooth-K is for); v ~ N(0,1)."""
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(1, H, L, D, generator=g)
    k = torch.randn(1, H, L, D, generator=g) + 2.0 * torch.randn(1, H, 1, D, generator=g)
    v = torch.randn(1, H, L, D, generator=g)
    # low-frequency structure along L so block selection is not uniform
    t = torch.linspace(0, 6.0, L)[None, None, :, None]
    q = q + 1.5 * torch.sin(t + torch.arange(D)[None, None, None, :] * 0.1)
    k = k + 1.5 * torch.sin(t + torch.arange(D)[None, None, None, :] * 0.1)
    return q.to(dtype), k.to(dtype), v.to(dtype)


@pytest.mark.parametrize("H,L", [(2, 64), (3, 700), (1, 1000)])
@pytest.mark.parametrize("odt", [torch.float16, torch.bfloat16])
def test_v_transpose(K, H, L, odt):
    _, _, v = qkv(H, L, 1)
    v = v[0].contiguous()  # [H, L, D]
    vt = K.v_transpose(v.to(DEV), L * 128, 128, L, H, 128, odt).cpu()
    kb = (L + 63) // 64
    pad = torch.zeros(H, kb * 64, 128, dtype=odt)
    pad[:, :L] = v.to(odt)
    perm = torch.tensor([0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15])
    idx = (torch.arange(4)[:, None] * 16 + perm[None, :]).reshape(-1)  # position -> key within block
    ref = pad.view(H, kb, 64, 128)[:, :, idx, :].permute(0, 1, 3, 2).contiguous()
    assert torch.equal(vt.view(torch.int16), ref.view(torch.int16))


@pytest.mark.parametrize("H,L", [(2, 300), (12, 1111)])
def test_seq_mean(K, H, L):
    _, k, _ = qkv(H, L, 2)
    km = K.seq_mean(k[0].contiguous().to(DEV))
    ref = S.seq_mean(k)[0, :, 0]
    assert ulp_diff_bf16(km, ref).max().item() <= 1


@pytest.mark.parametrize("H,L", [(2, 300), (3, 1000), (1, 64), (12, 8192)])
def test_sage_quant_pool_bit_exact(K, H, L):
    """(12, 8192): 12.6 M quotients per operand — the kernel's reciprocal + exact-remainder division must round like the
    oracle's IEEE division on every one of them."""
    q, k, _ = qkv(H, L, 3)
    km = S.seq_mean(k)  # same km on both sides -> codes/scales must be bit-exact
    for x, kmx, blk in ((q, None, 128), (k, km, 64)):
        ref_q, ref_s = S.quant_per_block_int8(x, blk, kmx)
        pooled, xq, xs = K.sage_quant_pool(x[0].contiguous().to(DEV),
                                           None if kmx is None else kmx[0, :, 0].contiguous().to(DEV), blk)
        assert torch.equal(xs.cpu(), ref_s[0]), "sage scales must be bit-exact"
        assert torch.equal(xq.cpu(), ref_q[0]), "sage int8 codes must be bit-exact"
        arg = x if kmx is None else (x - kmx)
        ref_p = S.mean_pool(arg, blk)[0]
        assert ulp_diff_bf16(pooled, ref_p).max().item() <= 1


@pytest.mark.parametrize("H,L,ratio", [(2, 1000, 0.25), (3, 2500, 0.1), (1, 640, 1.0)])
def test_block_map_agreement(K, H, L, ratio):
    q, k, _ = qkv(H, L, 4)
    _, lut_ref, topk = S.get_block_map(q, k, ratio, 128, 64)
    km = K.seq_mean(k[0].contiguous().to(DEV))
    pq, _, _ = K.sage_quant_pool(q[0].contiguous().to(DEV), None, 128, want_quant=False)
    pk, _, _ = K.sage_quant_pool(k[0].contiguous().to(DEV), km, 64, want_quant=False)
    lut = K.sla_topk(pq, pk, topk).cpu().long()
    assert lut.shape == lut_ref[0].shape
    assert (lut[..., 1:] > lut[..., :-1]).all(), "LUT must be strictly ascending"
    kb = (L + 63) // 64
    a = torch.zeros(H, lut.shape[1], kb, dtype=torch.bool).scatter_(-1, lut, True)
    b = torch.zeros(H, lut.shape[1], kb, dtype=torch.bool).scatter_(-1, lut_ref[0], True)
    agree = (a & b).sum().item() / b.sum().item()
    assert agree >= 0.99, f"selected-set agreement {agree:.4f}"
    # exactness of the selection rule itself: same scores in -> same LUT out
    score = (pq.float() @ pk.float().transpose(-1, -2)).to(torch.bfloat16).cpu()
    lut_same = S.select_topk(score, topk)
    agree2 = (torch.zeros_like(a).scatter_(-1, lut_same, True) & a).sum().item() / a.sum().item()
    assert agree2 >= 0.995


def _sage_inputs(H, L, seed):
    q, k, v = qkv(H, L, seed)
    km = S.seq_mean(k)
    q_i8, q_s = S.quant_per_block_int8(q, 128)
    k_i8, k_s = S.quant_per_block_int8(k, 64, km)
    return q, k, v, q_i8, q_s, k_i8, k_s


@pytest.mark.parametrize("H,L,ratio", [(2, 256, 1.0), (2, 1000, 0.3), (3, 777, 0.2), (1, 130, 1.0)])
def test_attn_i8_vs_oracle(K, H, L, ratio):
    q, k, v, q_i8, q_s, k_i8, k_s = _sage_inputs(H, L, 5)
    _, lut, topk = S.get_block_map(q, k, ratio, 128, 64)
    ref = S.sage_sparse_attn(q_i8, q_s, k_i8, k_s, v, lut, out_dtype=torch.bfloat16)[0]  # [H, L, D]
    vt = K.v_transpose(v[0].contiguous().to(DEV), L * 128, 128, L, H, 128, torch.float16)
    out = torch.empty(H, L, 128, dtype=torch.bfloat16, device=DEV)
    dense = ratio >= 1.0
    K.attn_i8(q_i8[0].to(DEV), q_s[0].to(DEV), k_i8[0].to(DEV), k_s[0].to(DEV), vt,
              None if dense else lut[0].int().to(DEV), out, L * 128, 128)
    assert cosine(out, ref) > 0.9999
    assert rel_l2(out, ref) < 5e-3
    # element-wise: within one bf16 ulp of the oracle (or 1e-4 of the output scale, for outputs near zero).  The
    # kernel folds scale and max into one fma and keeps a lazy running max, so P is rounded to fp16 against a
    # different — equally valid — reference point than the oracle's eager online softmax: same-size rounding noise
    o32, r32 = out.float().cpu(), ref.float()
    close = (o32 - r32).abs() <= torch.maximum(r32.abs() * 2.0 ** -7, torch.full_like(r32, 1e-4 * r32.abs().max().item()))
    assert close.float().mean().item() > 0.998   # (0.9989-0.9991 on the 130-token case: outputs near zero, rounding noise of either build)
    # and the stated fp tolerance vs fp32 softmax attention on the same selected blocks
    if dense:
        sd = S.sdpa_ref(q, k, v)[0]
        assert cosine(out, sd) > 0.999 and rel_l2(out, sd) < 3e-2


def test_attn_i8_lhd_output_layout(K):
    H, L = 2, 300
    q, k, v, q_i8, q_s, k_i8, k_s = _sage_inputs(H, L, 6)
    lut = S.dense_lut(1, H, L, 128, 64)
    ref = S.sage_sparse_attn(q_i8, q_s, k_i8, k_s, v, lut, out_dtype=torch.bfloat16)[0]
    vt = K.v_transpose(v[0].contiguous().to(DEV), L * 128, 128, L, H, 128, torch.float16)
    out = torch.zeros(L, H, 128, dtype=torch.bfloat16, device=DEV)
    K.attn_i8(q_i8[0].to(DEV), q_s[0].to(DEV), k_i8[0].to(DEV), k_s[0].to(DEV), vt, None, out, 128, H * 128)
    assert rel_l2(out.transpose(0, 1), ref) < 5e-3


@pytest.mark.parametrize("H,L,ratio", [(2, 256, 1.0), (2, 1000, 0.3), (1, 450, 0.5)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_attn_16_vs_oracle(K, H, L, ratio, dtype):
    q, k, v = qkv(H, L, 7, dtype)
    _, lut, topk = S.get_block_map(q, k, ratio, 128, 64)
    ref = S.sla_sparse_attn(q, k, v, lut, 128, 64)[0]
    vt = K.v_transpose(v[0].contiguous().to(DEV), L * 128, 128, L, H, 128, dtype)
    out = torch.empty(H, L, 128, dtype=dtype, device=DEV)
    K.attn_16(q[0].contiguous().to(DEV), k[0].contiguous().to(DEV), vt,
              None if ratio >= 1.0 else lut[0].int().to(DEV), out, L * 128, 128)
    assert cosine(out, ref) > 0.9999 and rel_l2(out, ref) < 5e-3


def test_attn_16_cross_shape(K):
    """cross-attention shape: L queries x 512 keys, dense (wan2pt1.py:280-300)."""
    H, L, Lk = 2, 333, 512
    g = torch.Generator().manual_seed(8)
    q = torch.randn(1, H, L, 128, generator=g).bfloat16()
    k = torch.randn(1, H, Lk, 128, generator=g).bfloat16()
    v = torch.randn(1, H, Lk, 128, generator=g).bfloat16()
    ref = S.sdpa_ref(q, k, v)[0]
    vt = K.v_transpose(v[0].contiguous().to(DEV), Lk * 128, 128, Lk, H, 128, torch.bfloat16)
    out = torch.empty(H, L, 128, dtype=torch.bfloat16, device=DEV)
    K.attn_16(q[0].contiguous().to(DEV), k[0].contiguous().to(DEV), vt, None, out, L * 128, 128)
    assert cosine(out, ref) > 0.9995 and rel_l2(out, ref) < 2e-2


def test_attn_softmax_rescale_spike(K):
    """Force the online-softmax rescale: one key far above the rest late in the sequence."""
    H, L = 1, 512
    q, k, v = qkv(H, L, 9)
    k[0, 0, 400] = q[0, 0, 37] * 4.0  # huge score for row 37 at K block 6
    km = S.seq_mean(k)
    q_i8, q_s = S.quant_per_block_int8(q, 128)
    k_i8, k_s = S.quant_per_block_int8(k, 64, km)
    lut = S.dense_lut(1, H, L, 128, 64)
    ref = S.sage_sparse_attn(q_i8, q_s, k_i8, k_s, v, lut, out_dtype=torch.bfloat16)[0]
    vt = K.v_transpose(v[0].contiguous().to(DEV), L * 128, 128, L, H, 128, torch.float16)
    out = torch.empty(H, L, 128, dtype=torch.bfloat16, device=DEV)
    K.attn_i8(q_i8[0].to(DEV), q_s[0].to(DEV), k_i8[0].to(DEV), k_s[0].to(DEV), vt, None, out, L * 128, 128)
    assert rel_l2(out, ref) < 5e-3
    assert rel_l2(out[0, 37], ref[0, 37]) < 1e-2


@pytest.mark.parametrize("H,L,Lk,dtype", [(2, 300, 512, torch.bfloat16), (12, 1111, 77, torch.bfloat16), (3, 640, 512, torch.float16)])
@pytest.mark.parametrize("quant_out", [False, True])
def test_attn_16_qnorm_bit_exact(K, H, L, Lk, dtype, quant_out):
    """Q normalised where the attention kernel loads it (td_rms_stats + td_attn_16_qnorm on the [L, H*128] linear
    output) == td_qk_norm_rope (no RoPE) followed by td_attn_16, bit for bit, also with the quantising epilogue."""
    g = torch.Generator().manual_seed(H * L + Lk)
    dim = H * 128
    src = (torch.randn(L, dim + 64, generator=g) * 1.7).to(dtype).to(DEV)[:, :dim]   # row stride > dim
    w = (torch.rand(dim, generator=g) + 0.5).to(DEV)
    kk = torch.randn(H, Lk, 128, generator=g).to(dtype).to(DEV)
    v = torch.randn(Lk, H, 128, generator=g).to(dtype).to(DEV)
    vt = K.v_transpose(v, 128, H * 128, Lk, H, 128, dtype)
    q = K.qk_norm_rope(src, 0, H, 128, w, None, None, 1e-6)
    out_ref = torch.empty(L, dim, dtype=dtype, device=DEV)
    out = torch.empty(L, dim, dtype=dtype, device=DEV)
    rstd = K.rms_stats(src, dim, 1e-6)
    exact = dtype == torch.bfloat16   # (the model's dtype.  fp16: the two-step path rounds q twice and loses fp16
    #                                    subnormals on the way — ~5e-5 of the elements differ, the outputs by < 1e-3)
    if quant_out:
        q_ref, s_ref = K.attn_16(q, kk, vt, None, out_ref, 128, dim, quant_out=True)
        q_new, s_new = K.attn_16_qnorm(src, rstd, w, kk, vt, None, out, 128, dim, quant_out=True)
        if exact:
            assert torch.equal(s_ref, s_new) and torch.equal(q_ref, q_new)
        else:
            assert (q_ref.int() - q_new.int()).abs().max().item() <= 1
            torch.testing.assert_close(s_new, s_ref, rtol=1e-3, atol=0)
    else:
        K.attn_16(q, kk, vt, None, out_ref, 128, dim)
        K.attn_16_qnorm(src, rstd, w, kk, vt, None, out, 128, dim)
        if exact:
            assert torch.equal(out_ref, out)
        else:
            torch.testing.assert_close(out, out_ref, rtol=0, atol=1e-3)


@pytest.mark.parametrize("H,L,Lk,k", [(12, 1500, 512, 256), (40, 1100, 512, 128), (12, 4096, 77, 1536)])
@pytest.mark.parametrize("quant_out", [False, True])
def test_attn_16_qnorm_from_the_gemm_epilogue_pieces_is_bit_identical(K, H, L, Lk, k, quant_out):
    """Round 6: td_attn_16_qnorm_pieces — the RMSNorm statistic of Q formed where the attention kernel loads Q, from the
    per-64-column (mean, M2) pieces the q projection's STATS epilogue wrote — against td_row_stats_finalize(mode 1) +
    td_attn_16_qnorm: the same additions in the same order, the same bits (outputs, and the fused INT8 codes + scales)."""
    g = torch.Generator().manual_seed(H * L + Lk)
    dim = H * 128
    a = (torch.randn(L, k, generator=g) * 1.3).to(torch.bfloat16).to(DEV)
    w_ = (torch.randn(dim, k, generator=g) / k ** 0.5).to(torch.bfloat16).to(DEV)
    b = (torch.randn(dim, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
    aq, as_ = K.quant_i8_block128(a)
    wq, ws_ = K.quant_i8_block128(w_)
    src, pieces = K.gemm_w8a8_stats(aq, as_, wq, ws_, b)
    wn = (torch.rand(dim, generator=g) + 0.5).to(DEV)
    kk = torch.randn(H, Lk, 128, generator=g).to(torch.bfloat16).to(DEV)
    v = torch.randn(Lk, H, 128, generator=g).to(torch.bfloat16).to(DEV)
    vt = K.v_transpose(v, 128, H * 128, Lk, H, 128, torch.bfloat16)
    rstd = K.row_stats_finalize(pieces, dim, 1e-6, rms=True)
    o0 = torch.empty(L, dim, dtype=torch.bfloat16, device=DEV)
    o1 = torch.empty(L, dim, dtype=torch.bfloat16, device=DEV)
    r0 = K.attn_16_qnorm(src, rstd, wn, kk, vt, None, o0, 128, dim, quant_out=quant_out)
    r1 = K.attn_16_qnorm(src, (pieces, 1e-6), wn, kk, vt, None, o1, 128, dim, quant_out=quant_out)
    if quant_out:
        assert torch.equal(r0[0], r1[0]) and torch.equal(r0[1], r1[1])
    else:
        assert torch.isfinite(o0.float()).all() and torch.equal(o0, o1)


@pytest.mark.parametrize("H,L", [(2, 300), (12, 1111), (3, 4100)])
def test_linear_kv_pass_also_yields_the_smooth_k_mean(K, H, L):
    """td_sla_linear_kv with a km output: kvsum / ksum unchanged (bit for bit) and km == td_seq_mean(k) up to the
    rounding of a different (still fixed) fp32 summation tree, and within a 16-bit ulp of the exact mean."""
    _, k, v = qkv(H, L, 4)
    kd = k[0].contiguous().to(DEV)
    vt = K.v_transpose(v[0].contiguous().to(DEV), L * 128, 128, L, H, 128, torch.float16)
    kv_t, ksum = K.sla_linear_kv(kd, vt)
    kv_t2, ksum2, km = K.sla_linear_kv(kd, vt, want_kmean=True)
    assert torch.equal(kv_t, kv_t2) and torch.equal(ksum, ksum2)
    km_ref = K.seq_mean(kd)
    tol = 2.0 ** -8 * km_ref.float().abs().clamp_min(1e-2)
    assert ((km.float() - km_ref.float()).abs() <= tol).all()
    exact = S.seq_mean(k)[0, :, 0]
    assert ulp_diff_bf16(km, exact).max().item() <= 1 or ((km.float().cpu() - exact.float()).abs() <= 1e-4).all()


@pytest.mark.parametrize("H,L,ratio", [(2, 300, 0.5), (3, 1000, 0.3), (12, 1664, 0.2)])
def test_attention_epilogue_fusions_bit_exact(K, H, L, ratio):
    """add_t (o_l added in the attention epilogue) and quant_out (epilogue block-quantiser) reproduce the separate
    passes bit for bit: attn -> linear_out_ (read-modify-write) -> quant_i8_block128."""
    q, k, v, q_i8, q_s, k_i8, k_s = _sage_inputs(H, L, 21)
    _, lut, topk = S.get_block_map(q, k, ratio, 128, 64)
    g = torch.Generator().manual_seed(5)
    wp = (torch.randn(128, 128, generator=g) * 0.05).to(DEV)
    bp = (torch.randn(128, generator=g) * 0.05).to(DEV)
    qd, kd = q[0].contiguous().to(DEV), k[0].contiguous().to(DEV)
    vt = K.v_transpose(v[0].contiguous().to(DEV), L * 128, 128, L, H, 128, torch.float16)
    kv_t, ksum = K.sla_linear_kv(kd, vt)
    args = (q_i8[0].to(DEV), q_s[0].to(DEV), k_i8[0].to(DEV), k_s[0].to(DEV), vt, lut[0].int().to(DEV))
    # reference: separate passes, token-major [L, H*128] output like the DiT block uses
    ref = torch.empty(L, H * 128, dtype=torch.bfloat16, device=DEV)
    K.attn_i8(*args, ref, 128, H * 128)
    K.sla_linear_out_(qd, kv_t, ksum, wp, bp, ref, 128, H * 128)
    rq, rs = K.quant_i8_block128(ref)
    # fused: o_l first, added + quantised in the attention epilogue
    o_l = K.sla_linear_out_t(qd, kv_t, ksum, wp, bp)
    out = torch.empty(L, H * 128, dtype=torch.bfloat16, device=DEV)
    K.attn_i8(*args, out, 128, H * 128, add_t=o_l)
    assert torch.equal(out, ref), f"add_t: {(out != ref).sum().item()} elements differ"
    oq, os_ = K.attn_i8(*args, torch.bfloat16, 128, H * 128, add_t=o_l, quant_out=True)
    assert torch.equal(os_, rs) and torch.equal(oq, rq)
    # 16-bit QK kernel (the dense cross-attention shape: few keys), quantised output
    kc, vc = kd[:, :200].contiguous(), v[0][:, :200].contiguous().to(DEV)
    vtc = K.v_transpose(vc, 200 * 128, 128, 200, H, 128, torch.bfloat16)
    ref2 = torch.empty(L, H * 128, dtype=torch.bfloat16, device=DEV)
    K.attn_16(qd, kc, vtc, None, ref2, 128, H * 128)
    rq2, rs2 = K.quant_i8_block128(ref2)
    oq2, os2 = K.attn_16(qd, kc, vtc, None, None, 128, H * 128, quant_out=True)
    assert torch.equal(os2, rs2) and torch.equal(oq2, rq2)


@pytest.mark.parametrize("H,L", [(2, 300), (3, 1000)])
def test_linear_branch(K, H, L):
    q, k, v = qkv(H, L, 10)
    g = torch.Generator().manual_seed(11)
    wp = torch.randn(128, 128, generator=g) * 0.05
    bp = torch.randn(128, generator=g) * 0.05
    o_s = torch.randn(1, H, L, 128, generator=g).bfloat16()
    ref = (o_s + S.linear_branch_exact_autocast(q, k, v, wp, bp))[0]
    for vdt in (torch.float16, torch.bfloat16):
        vt = K.v_transpose(v[0].contiguous().to(DEV), L * 128, 128, L, H, 128, vdt)
        kv_t, ksum = K.sla_linear_kv(k[0].contiguous().to(DEV), vt)
        ck = torch.softmax(k, dim=-1).bfloat16()
        kv_ref = (ck.float().transpose(-1, -2) @ v.float())[0]
        assert rel_l2(kv_t.transpose(-1, -2), kv_ref) < 1e-2
        assert rel_l2(ksum, ck.float().sum(-2)[0]) < 1e-2
        out = o_s[0].clone().to(DEV)
        K.sla_linear_out_(q[0].contiguous().to(DEV), kv_t, ksum, wp.to(DEV), bp.to(DEV), out, L * 128, 128)
        assert rel_l2(out, ref) < 1e-2 and cosine(out, ref) > 0.9999
        assert rel_l2(out.float().cpu() - o_s[0].float(), ref.float() - o_s[0].float()) < 3e-2


@pytest.mark.parametrize("sage", [True, False])
def test_sla_modules_vs_oracle(sage):
    from turbodiffusion_amd.sla import SageSparseLinearAttention, SparseLinearAttention
    B, L, H, D = 1, 900, 3, 128
    q, k, v = qkv(H, L, 12)
    q, k, v = (t.transpose(1, 2).contiguous() for t in (q, k, v))  # [B, L, H, D]
    g = torch.Generator().manual_seed(13)
    wp = torch.randn(D, D, generator=g) * 0.05
    bp = torch.randn(D, generator=g) * 0.05
    if sage:
        mod = SageSparseLinearAttention(D, 0.25)
        ref = S.sagesla_forward(q, k, v, wp, bp, 0.25)
    else:
        mod = SparseLinearAttention(D, 0.25, BLKQ=128, BLKK=64)
        ref = S.sla_forward(q, k, v, wp, bp, 0.25)
    with torch.no_grad():
        mod.proj_l.weight.copy_(wp)
        mod.proj_l.bias.copy_(bp)
    mod = mod.to(DEV)
    out, sparsity = mod(q.to(DEV), k.to(DEV), v.to(DEV), return_sparsity=True)
    assert out.shape == (B, L, H, D) and out.dtype == q.dtype
    assert sparsity == pytest.approx(int(0.25 * 15) / 15)
    # block selection may differ on near-ties (SURVEY §8c iii) -> output-level tolerance
    assert cosine(out, ref) > 0.999 and rel_l2(out, ref) < 3e-2


# ---------------------------------------------------------------- against tensors produced by the REAL reference
def _gold():
    import os
    return torch.load(os.path.join(os.path.dirname(__file__), "golden", "sla_tiny.pt"), weights_only=False)


def test_sla_modules_vs_reference_golden(K):
    """tests/golden/sla_tiny.pt = outputs of the reference's own SparseLinearAttention / SageSparseLinearAttention
    modules (SLA/core.py, leaves patched — oracle/make_golden.py): the HIP modules, fed the same [B,L,H,D] tensors,
    agree within the stated attention tolerance (cosine >= 0.999, rel-L2 <= 2e-2)."""
    from turbodiffusion_amd.sla import SageSparseLinearAttention, SparseLinearAttention
    g = _gold()
    for cls, kw, key in ((SparseLinearAttention, dict(BLKQ=128, BLKK=64), "ref_sla"),
                         (SageSparseLinearAttention, {}, "ref_sagesla_f16pv")):
        m = cls(128, g["topk"], **kw)
        with torch.no_grad():
            m.proj_l.weight.copy_(g["proj_w"])
            m.proj_l.bias.copy_(g["proj_b"])
        out, sparsity = m.to(DEV)(g["q"].to(DEV), g["k"].to(DEV), g["v"].to(DEV), return_sparsity=True)
        ref = g[key]
        assert out.shape == ref.shape and out.dtype == ref.dtype
        assert sparsity == pytest.approx(4 / 10)
        assert cosine(out, ref) > 0.9995, key
        assert rel_l2(out, ref) < 2e-2, key


def test_block_map_vs_reference_golden(K):
    """The block map of the reference's get_block_map (SLA/utils.py:55-67, pooled bf16 scores + torch.topk) against
    td_sage_quant_pool + td_sla_topk on the same q, k: identical selected sets."""
    g = _gold()
    q = g["q"][0].transpose(0, 1).contiguous().to(DEV)   # [H, L, D]
    k = g["k"][0].transpose(0, 1).contiguous().to(DEV)
    km = K.seq_mean(k)
    pq, _, _ = K.sage_quant_pool(q, None, 128, want_quant=False)
    pk, _, _ = K.sage_quant_pool(k, km, 64, want_quant=False)
    lut = K.sla_topk(pq, pk, 4).cpu().long()
    smap = torch.zeros(g["sparse_map"].shape[1:], dtype=torch.int8).scatter_(-1, lut, 1)
    agree = ((smap > 0) & (g["sparse_map"][0] > 0)).sum().item() / (g["sparse_map"][0] > 0).sum().item()
    assert agree >= 0.99, agree


# ---------------------------------------------------------------- a13, FP8-PV variant (the reference's sm89+ branch)
_POS2KEY8 = torch.tensor([32 * ((p >> 4) & 1) + ((p & 15) & 3) + 8 * ((p & 15) >> 2) + 4 * (p >> 5) for p in range(64)])


@pytest.mark.parametrize("H,L", [(2, 64), (3, 700), (12, 1000)])
def test_v_fp8_tiles_bit_exact(K, H, L):
    """td_v_fp8_tiles == the oracle's v_fp8_quant (per-channel scale = max|v| / 2.25, e4m3 RNE of v / scale): scales and
    every e4m3 byte, in the position order of the fp8 PV MFMA's B operand; tail keys zero."""
    _, _, v = qkv(H, L, 31)
    vt = torch.empty(1, H, 128, (L + 127) // 128 * 128, dtype=v.dtype)
    S.transpose_pad_permute(v, vt)
    q_ref, s_ref = S.v_fp8_quant(vt, L, 2.25)                       # [1,H,128,Lpad] e4m3, [1,H,128]
    vl = v[0].transpose(0, 1).contiguous().to(DEV)                  # [L, H, D]
    vt8, vs = K.v_fp8_tiles(vl, 128, H * 128, L, H, 128, 2.25)
    assert torch.equal(vs.cpu(), s_ref[0]), "v scales must be bit-exact"
    kb = (L + 63) // 64
    ref = torch.zeros(H, 128, kb * 64, dtype=torch.uint8)
    ref[:, :, :L] = q_ref[0].view(torch.uint8)[:, :, :L]
    ref = ref.view(H, 128, kb, 64)[:, :, :, _POS2KEY8].permute(0, 2, 1, 3).contiguous()   # [H, kb, 128, 64 positions]
    got = vt8.cpu()
    # -0 and +0 are the same value (tail keys, exact zeros)
    same = (got == ref) | (((got & 0x7f) == 0) & ((ref & 0x7f) == 0))
    assert same.all(), f"{(~same).sum().item()} e4m3 bytes differ"


@pytest.mark.parametrize("H,L,ratio", [(2, 1000, 0.3), (3, 777, 0.2), (2, 256, 1.0)])
def test_attn_i8_fp8pv_vs_oracle(K, H, L, ratio):
    """INT8-QK / FP8-PV attention against the oracle's statement of SpargeAttn's sm89 kernels (sage_sparse_attn_fp8):
    same V8 / v_scale, P rounded to e4m3 against the kernel's lazy running max instead of the exact one.  e4m3 keeps 3
    mantissa bits of every probability, so any two correct implementations that round against different reference
    points differ by that noise (measured: both sit ~3e-2 from the FP16-PV result and ~2e-2 from each other); the
    variant's distance to the FP16-PV result must be what the oracle's is."""
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
    assert cosine(out, ref) > 0.9995
    assert rel_l2(out, ref) < 3.5e-2, rel_l2(out, ref)
    gap_oracle, gap_hip = rel_l2(ref, ref16), rel_l2(out, ref16)
    assert gap_hip < gap_oracle * 1.2 + 3e-3, (gap_hip, gap_oracle)


def test_sagesla_fp8pv_module_vs_reference_golden(K):
    """SageSparseLinearAttention with pv_dtype = 'fp8' against the output of the reference module's sm89 branch
    (tests/golden/sla_tiny.pt: ref_sagesla_fp8pv, produced by SLA/core.py:217-239 with its leaves patched)."""
    from turbodiffusion_amd.sla import SageSparseLinearAttention
    g = _gold()
    m = SageSparseLinearAttention(128, g["topk"])
    m.pv_dtype = "fp8"
    with torch.no_grad():
        m.proj_l.weight.copy_(g["proj_w"])
        m.proj_l.bias.copy_(g["proj_b"])
    out = m.to(DEV)(g["q"].to(DEV), g["k"].to(DEV), g["v"].to(DEV))
    assert cosine(out, g["ref_sagesla_fp8pv"]) > 0.9995
    assert rel_l2(out, g["ref_sagesla_fp8pv"]) < 3.5e-2
    # as far from the FP16-PV branch's output as the reference's own FP8 branch is
    gap_ref = rel_l2(g["ref_sagesla_fp8pv"], g["ref_sagesla_f16pv"])
    assert rel_l2(out, g["ref_sagesla_f16pv"]) < gap_ref * 1.2 + 3e-3


# ---------------------------------------------------------------- sequence parallelism: rank-major gathered K side
@pytest.mark.parametrize("sage", [True, False])
def test_gathered_layout_kernels_are_bit_identical_to_flat(K, sage):
    """td_attn_i8_sp / td_attn_16_sp / td_sla_topk_sp read K, K scales, V^T tiles and pooled K straight from a rank-major
    all-gather output (3 ranks x 256 tokens, the last one short): same bits as the flat layout."""
    H, L, W, per = 3, 700, 3, 256
    kbp = per // 64
    q, k, v, q_i8, q_s, k_i8, k_s = _sage_inputs(H, L, 7)
    dt = torch.bfloat16
    pdt = torch.float16 if sage else dt
    qd, kd = q[0].to(DEV), k[0].to(DEV)
    vt = K.v_transpose(v[0].contiguous().to(DEV), L * 128, 128, L, H, 128, pdt)          # [H, 11, 128, 64]
    km = K.seq_mean(kd)
    pq, q8, qs = K.sage_quant_pool(qd, None, 128)
    pk, k8, ks = K.sage_quant_pool(kd, km, 64)
    kb_tot = (L + 63) // 64
    topk = 4
    lut = K.sla_topk(pq, pk, topk)
    out_flat = torch.empty(H, L, 128, dtype=dt, device=DEV)
    if sage:
        K.attn_i8(q8, qs, k8, ks, vt, lut, out_flat, L * 128, 128)
    else:
        K.attn_16(qd, kd, vt, lut, out_flat, L * 128, 128)
    # build the "all-gather output": [W, bytes] with per-rank slots k | ks | vt | pk (256-byte aligned)
    ksrc = k8 if sage else kd
    esz = ksrc.element_size()
    sizes = {"k": H * per * 128 * esz, "ks": H * kbp * 4, "vt": H * kbp * 128 * 64 * 2, "pk": H * kbp * 128 * 2}
    offs, o = {}, 0
    for n_, sz in sizes.items():
        offs[n_] = o
        o += (sz + 255) // 256 * 256
    allb = torch.zeros(W, o, dtype=torch.uint8, device=DEV)

    def slot(name, dtype, shape):
        return allb[:, offs[name]:offs[name] + sizes[name]].view(dtype).view((W,) + shape)

    kg, ksg = slot("k", ksrc.dtype, (H, per, 128)), slot("ks", torch.float32, (H, kbp))
    vtg, pkg = slot("vt", pdt, (H, kbp, 128, 64)), slot("pk", dt, (H, kbp, 128))
    for r in range(W):
        t0, t1 = r * per, min(L, (r + 1) * per)
        b0, b1 = r * kbp, min(kb_tot, (r + 1) * kbp)
        kg[r, :, : t1 - t0] = ksrc[:, t0:t1]
        ksg[r, :, : b1 - b0] = ks[:, b0:b1]
        vtg[r, :, : b1 - b0] = vt[:, b0:b1]
        pkg[r, :, : b1 - b0] = pk[:, b0:b1]
    lut_g = K.sla_topk_sp(pq, pkg, topk, kb_tot)
    assert torch.equal(lut_g, lut)
    out_g = torch.empty_like(out_flat)
    if sage:
        K.attn_i8_sp(q8, qs, kg, ksg, vtg, lut, out_g, L * 128, 128, L)
    else:
        K.attn_16_sp(qd, kg, vtg, lut, out_g, L * 128, 128, L)
    assert torch.equal(out_g, out_flat)
    # dense (no LUT) as well: every block of every rank, incl. the short tail block of the last rank
    d_flat, d_g = torch.empty_like(out_flat), torch.empty_like(out_flat)
    if sage:
        K.attn_i8(q8, qs, k8, ks, vt, None, d_flat, L * 128, 128)
        K.attn_i8_sp(q8, qs, kg, ksg, vtg, None, d_g, L * 128, 128, L)
    else:
        K.attn_16(qd, kd, vt, None, d_flat, L * 128, 128)
        K.attn_16_sp(qd, kg, vtg, None, d_g, L * 128, 128, L)
    assert torch.equal(d_g, d_flat)


@pytest.mark.parametrize("sage,dense", [(True, False), (True, True), (False, False), (False, True)])
@pytest.mark.parametrize("L_loc,per", [(512, 512), (440, 512)])
def test_sp_pack_k_side_is_bit_identical_to_flat_producers(K, sage, dense, L_loc, per):
    """kernels.sp_pack_k_side (the *_packed entry points: producers write straight into the all-gather send buffer, heads
    grouped two-level, rows padded to the rank's shard size) holds exactly the bits of the flat producers — a full shard and
    the short last rank; 6 heads = 3 groups of 2."""
    from turbodiffusion_amd.seqpar import PackLayout
    H, D = 6, 128
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(L_loc + 2 * sage + dense)
    k = (torch.randn(H, L_loc, D, generator=g) + 0.5).to(dt).cuda()
    v = torch.randn(L_loc, H * D, generator=g).to(dt).cuda()              # element (h,l,d) at h*D + l*H*D + d
    km = K.seq_mean(k) if (sage or not dense) else None
    lay = PackLayout(H, per, D, 4, sage, dense, dt)
    assert (lay.G, lay.hg) == (3, 2)
    lin_kv = torch.empty((H, D, D), dtype=torch.float32, device="cuda")
    lin_ks = torch.empty((H, D), dtype=torch.float32, device="cuda")
    pack = K.sp_pack_k_side(k, km, v, (D, H * D), L_loc, lay, lin_kv, lin_ks)
    assert pack.shape == (lay.total,) and pack.dtype == torch.uint8 and lay.total == lay.ab + lay.G * lay.gb
    kb = -(-L_loc // 64)

    def sec(name, n):
        return lay.group_section(pack, name)[:, :, :n].reshape((H, n) + lay.spec[name][1][2:])

    vt = K.v_transpose(v, D, H * D, L_loc, H, D, lay.pdt)
    assert torch.equal(sec("vt", kb), vt)
    if sage:
        pk, k_q, k_s = K.sage_quant_pool(k, km, 64, want_pool=not dense)
        assert torch.equal(sec("k", L_loc), k_q) and torch.equal(sec("ks", kb), k_s)
    else:
        assert torch.equal(sec("k", L_loc), k)
        pk = K.sage_quant_pool(k, km, 64, want_quant=False)[0] if not dense else None
    if not dense:
        assert torch.equal(lay.all_section(pack, "pk")[:, :kb], pk)          # pooled K of ALL heads: the flat all-head section
        kv32, ks32 = K.sla_linear_kv_partial_f32(k, vt)
        assert torch.equal(lin_kv, kv32) and torch.equal(lin_ks, ks32)       # linear-branch partials: the early send buffer
    if L_loc < per:   # the padding of a short rank is zero (never read as valid, but must be finite)
        assert int(lay.group_section(pack, "vt")[:, :, kb:].abs().sum()) == 0


def test_attn_i8_two_per_cu_build_is_bit_identical(K):
    """TD_TUNE_ATTN_OCC = 2 (experiment kept selectable: two workgroups per CU, three tile buffers, explicit K / V fragment
    prefetch) computes the bits of the default three-per-CU build — sparse with a ragged tail and dense."""
    H, L = 3, 1000
    q, k, v, q_i8, q_s, k_i8, k_s = _sage_inputs(H, L, 11)
    _, lut, topk = S.get_block_map(q, k, 0.3, 128, 64)
    vt = K.v_transpose(v[0].contiguous().to(DEV), L * 128, 128, L, H, 128, torch.float16)
    args = (q_i8[0].to(DEV), q_s[0].to(DEV), k_i8[0].to(DEV), k_s[0].to(DEV), vt)
    for lt in (lut[0].int().to(DEV), None):
        outs = []
        for occ in (0, 2):
            K.set_tuning(K.TUNE_ATTN_OCC, occ)
            o = torch.zeros(H, L, 128, dtype=torch.bfloat16, device=DEV)
            K.attn_i8(*args, lt, o, L * 128, 128)
            outs.append(o)
        K.set_tuning(K.TUNE_ATTN_OCC, 0)
        assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("H,L,ratio", [(2, 1000, 0.3), (3, 777, 0.2), (1, 130, 1.0), (2, 2080, 0.1)])
def test_attn_i8_rowsum_on_the_matrix_pipe_build_vs_oracle(K, H, L, ratio):
    """TD_TUNE_ATTN_OCC = 4 (round-4 experiment): the two-workgroup build with the softmax denominator accumulated by four
    extra MFMAs per tile against an all-ones operand (the sum of the fp16-ROUNDED probabilities) instead of 32 v_add per lane:
    same bar against the oracle as the production build, and rounding-level distance to it."""
    q, k, v, q_i8, q_s, k_i8, k_s = _sage_inputs(H, L, 7)
    _, lut, topk = S.get_block_map(q, k, ratio, 128, 64)
    ref = S.sage_sparse_attn(q_i8, q_s, k_i8, k_s, v, lut, out_dtype=torch.bfloat16)[0]
    vt = K.v_transpose(v[0].contiguous().to(DEV), L * 128, 128, L, H, 128, torch.float16)
    dense = ratio >= 1.0
    args = (q_i8[0].to(DEV), q_s[0].to(DEV), k_i8[0].to(DEV), k_s[0].to(DEV), vt, None if dense else lut[0].int().to(DEV))
    outs = []
    try:
        for occ in (0, 4, 5):      # 5 (round-5 experiment): the production build with the row sum as 16 v_dot2_f32_f16 of the rounded pairs
            K.set_tuning(K.TUNE_ATTN_OCC, occ)
            o = torch.full((H, L, 128), float("nan"), dtype=torch.bfloat16, device=DEV)
            K.attn_i8(*args, o, L * 128, 128)
            outs.append(o)
    finally:
        K.set_tuning(K.TUNE_ATTN_OCC, 0)
    base = outs[0]
    for out in outs[1:]:
        assert torch.isfinite(out).all()
        assert cosine(out, ref) > 0.9999 and rel_l2(out, ref) < 5e-3
        assert rel_l2(out, base.float().cpu()) < 3e-3


@pytest.mark.parametrize("H,L,ratio", [(2, 256, 1.0), (2, 1000, 0.3), (3, 777, 0.2), (1, 130, 1.0), (2, 2080, 0.1), (1, 97, 1.0)])
def test_attn_i8_q64_build_vs_oracle(K, H, L, ratio):
    """TD_TUNE_ATTN_OCC = 3: the waves of a workgroup as 2 (Q halves of 64 rows) x 2 (key halves) — every wave a split-K
    stream over its 32 keys of each tile, merged through LDS at the end.  Same bar against the oracle as the 4 x 32 build
    (test_attn_i8_vs_oracle), incl. last blocks whose second key half is empty (L % 64 <= 32) or partial, and the
    difference to the 4 x 32 build is rounding (different reference points of the lazy running max, another summation order)."""
    q, k, v, q_i8, q_s, k_i8, k_s = _sage_inputs(H, L, 5)
    _, lut, topk = S.get_block_map(q, k, ratio, 128, 64)
    ref = S.sage_sparse_attn(q_i8, q_s, k_i8, k_s, v, lut, out_dtype=torch.bfloat16)[0]
    vt = K.v_transpose(v[0].contiguous().to(DEV), L * 128, 128, L, H, 128, torch.float16)
    dense = ratio >= 1.0
    args = (q_i8[0].to(DEV), q_s[0].to(DEV), k_i8[0].to(DEV), k_s[0].to(DEV), vt, None if dense else lut[0].int().to(DEV))
    outs = []
    try:
        for occ in (0, 3):
            K.set_tuning(K.TUNE_ATTN_OCC, occ)
            o = torch.full((H, L, 128), float("nan"), dtype=torch.bfloat16, device=DEV)
            K.attn_i8(*args, o, L * 128, 128)
            outs.append(o)
    finally:
        K.set_tuning(K.TUNE_ATTN_OCC, 0)
    base, out = outs
    assert torch.isfinite(out).all()
    assert cosine(out, ref) > 0.9999 and rel_l2(out, ref) < 5e-3
    o32, r32 = out.float().cpu(), ref.float()
    close = (o32 - r32).abs() <= torch.maximum(r32.abs() * 2.0 ** -7, torch.full_like(r32, 1e-4 * r32.abs().max().item()))
    assert close.float().mean().item() > 0.998   # (0.9989-0.9991 on the 130-token case: outputs near zero, rounding noise of either build)
    assert rel_l2(out, base.float().cpu()) < 3e-3


def test_attn_i8_q64_build_epilogue_variants(K):
    """The Q64 build's epilogue is the 4 x 32 build's (linear-branch addend in the lane-private layout, block quantiser for
    the o projection): with the addend the outputs agree to rounding; the quantised form agrees in its scales to 1e-3 and
    in its codes to one step."""
    H, L = 2, 1000
    q, k, v, q_i8, q_s, k_i8, k_s = _sage_inputs(H, L, 13)
    _, lut, topk = S.get_block_map(q, k, 0.3, 128, 64)
    vt = K.v_transpose(v[0].contiguous().to(DEV), L * 128, 128, L, H, 128, torch.float16)
    g = torch.Generator().manual_seed(3)
    add_t = (0.1 * torch.randn(H, (L + 127) // 128, 4, 16, 64, 4, generator=g)).bfloat16().to(DEV)
    args = (q_i8[0].to(DEV), q_s[0].to(DEV), k_i8[0].to(DEV), k_s[0].to(DEV), vt, lut[0].int().to(DEV))
    res = []
    try:
        for occ in (0, 3):
            K.set_tuning(K.TUNE_ATTN_OCC, occ)
            o = torch.zeros(L, H * 128, dtype=torch.bfloat16, device=DEV)
            K.attn_i8(*args, o, 128, H * 128, add_t=add_t)
            oq, os_ = K.attn_i8(*args, torch.bfloat16, 128, H * 128, add_t=add_t, quant_out=True)
            res.append((o, oq.clone(), os_.clone()))
    finally:
        K.set_tuning(K.TUNE_ATTN_OCC, 0)
    (o0, q0, s0), (o1, q1, s1) = res
    assert rel_l2(o1.float().cpu(), o0.float().cpu()) < 3e-3
    torch.testing.assert_close(s1, s0, rtol=1e-2, atol=0)
    assert (q1.int() - q0.int()).abs().max().item() <= 2 and (q1 != q0).float().mean().item() < 0.05
