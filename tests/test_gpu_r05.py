"""-m gpu, round 5: the sequence-parallel layer's fused launches (each against the operator sequence it replaces, bit for
bit), the head-group form of the attention epilogue's fused quantiser, and parity at the REAL size at full depth
(oracle/make_golden_r05.py)."""
import os

import pytest
import torch

from tests.util import cosine, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def K():
    from turbodiffusion_amd import kernels
    return kernels


def _qk(H, L, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(H, L, 128, generator=g) * scale).bfloat16().to(DEV)


@pytest.mark.parametrize("H,L", [(12, 4096), (3, 1000), (40, 9472), (2, 77)])
def test_seq_sum_is_the_partials_summed_in_order(K, H, L):
    """td_seq_sum (chunk partials + one in-order pass, no library reduction / copy) == td_seq_sum_partial's 64 partials added in
    chunk order, bit for bit; into a slice of a larger buffer (the early send buffer of the sequence-parallel layer)."""
    k = _qk(H, L, H + L) + 0.25
    ws = K.seq_sum_partial(k)
    want = torch.zeros(H, 128, device=DEV)
    for c in range(64):
        want = want + ws[:, c]
    assert torch.equal(K.seq_sum(k, None), want)
    buf = torch.full((H * 128 + 77,), -1.0, device=DEV)
    got = K.seq_sum(k, None, out=buf[:H * 128].view(H, 128))
    assert torch.equal(got, want) and bool((buf[H * 128:] == -1.0).all())


@pytest.mark.parametrize("L,dim", [(4096, 1536), (333, 256), (1184, 5120)])
def test_qk_norm_rope_pair_is_the_two_single_launches(K, L, dim):
    g = torch.Generator().manual_seed(L)
    H = dim // 128
    qkv = torch.randn(L, 3 * dim, generator=g).bfloat16().to(DEV)
    wq = (1 + 0.1 * torch.randn(dim, generator=g)).to(DEV)
    wk = (1 + 0.1 * torch.randn(dim, generator=g)).to(DEV)
    ang = torch.rand(L, 64, generator=g) * 6.28
    cos, sin = torch.cos(ang).to(DEV).contiguous(), torch.sin(ang).to(DEV).contiguous()
    q, k = K.qk_norm_rope_pair(qkv, 0, dim, H, 128, wq, wk, cos, sin, 1e-6)
    assert torch.equal(q, K.qk_norm_rope(qkv, 0, H, 128, wq, cos, sin, 1e-6))
    assert torch.equal(k, K.qk_norm_rope(qkv, dim, H, 128, wk, cos, sin, 1e-6))


@pytest.mark.parametrize("W,H,per,Lloc,sage,dense", [(8, 12, 4096, 4096, True, False), (2, 4, 512, 460, True, False), (4, 6, 256, 256, False, False),
                                                     (3, 4, 256, 200, True, True)])
def test_pack_with_in_kernel_smooth_k_mean_is_the_finalised_mean(K, W, H, per, Lloc, sage, dense):
    """kernels.sp_pack_k_side fed the GATHERED per-rank column sums forms the smooth-K mean inside the quantiser: the packed
    bytes equal those of the same call fed td_seq_mean_final's mean — codes, scales, pooled K (all-head section), V^T tiles,
    linear-branch partials — for Sage and 16-bit K, sparse and dense."""
    from turbodiffusion_amd.seqpar import PackLayout
    g = torch.Generator().manual_seed(W * per + H)
    k = _qk(H, Lloc, 5) + 0.5
    v = torch.randn(Lloc, H, 128, generator=g).bfloat16().to(DEV)
    L_tot = (W - 1) * per + Lloc
    allp = (torch.randn(W, H, 128, generator=g) * 30).to(DEV)
    allp[W - 1] = K.seq_sum(k, None)
    km = K.seq_mean_final(allp, W, 128, H * 128, L_tot, H, 128, torch.bfloat16)
    lay = PackLayout(H, per, 128, 2, sage, dense, torch.bfloat16)
    lins = [(torch.empty((H, 128, 128), device=DEV), torch.empty((H, 128), device=DEV)) for _ in range(2)]
    a = K.sp_pack_k_side(k, (allp, L_tot), v, (128, H * 128), Lloc, lay, *lins[0])
    b = K.sp_pack_k_side(k, km, v, (128, H * 128), Lloc, lay, *lins[1])
    for name in ("k", "vt", "ks"):
        if lay.sizes[name]:
            sa, sb = lay.group_section(a, name), lay.group_section(b, name)
            n = Lloc if name == "k" else -(-Lloc // 64)
            assert torch.equal(sa[:, :, :n].reshape(-1).view(torch.uint8), sb[:, :, :n].reshape(-1).view(torch.uint8)), name
    if lay.asizes["pk"]:
        sa, sb = lay.all_section(a, "pk")[:, :-(-Lloc // 64)], lay.all_section(b, "pk")[:, :-(-Lloc // 64)]
        assert torch.equal(sa.reshape(-1).view(torch.uint8), sb.reshape(-1).view(torch.uint8))
        # ... and they are the flat producers' results: pooled K of td_sage_quant_pool, the partials of the flat linear-branch pass
        pk_flat, _, _ = K.sage_quant_pool(k, km, 64, want_quant=False)
        assert torch.equal(sb, pk_flat)
    if lay.linear:
        kv_flat, ks_flat = K.sla_linear_kv_partial_f32(k, K.v_transpose(v, 128, H * 128, Lloc, H, 128, lay.pdt))
        assert torch.equal(lins[0][0], kv_flat) and torch.equal(lins[0][1], ks_flat) and torch.equal(lins[1][0], kv_flat)


@pytest.mark.parametrize("sage", [True, False])
def test_gathered_attention_quantises_per_head_group_like_the_whole_row(K, sage):
    """*_sp attention with quant_out per HEAD GROUP (q_heads_total, ABI v4): two launches over heads [0, 2) and [2, 4) writing
    their columns / scale entries of one [L, 4*128] int8 row == td_quant_i8_block128 of the 16-bit output of the same two
    launches, bit for bit; with the linear branch's o_l added in the epilogue."""
    H, W, per, Lq = 4, 2, 256, 512
    g = torch.Generator().manual_seed(3 + sage)
    q, kk = _qk(H, Lq, 11), _qk(H, W * per, 12)
    v = torch.randn(W * per, H, 128, generator=g).bfloat16().to(DEV)
    o_l = K.sla_linear_out_t(q, *K.sla_linear_kv(kk, K.v_transpose(v, 128, H * 128, W * per, H, 128, torch.bfloat16)),
                             (torch.randn(128, 128, generator=g) * 0.05).to(DEV), torch.zeros(128, device=DEV))
    km = K.seq_mean(kk)
    vt = K.v_transpose(v, 128, H * 128, W * per, H, 128, torch.float16 if sage else torch.bfloat16)     # [H, kb, 128, 64]
    kb = W * per // 64
    if sage:
        _, q8, qs = K.sage_quant_pool(q, None, 128, want_pool=False)
        _, k8, ks = K.sage_quant_pool(kk, km, 64, want_pool=False)
    # rank-major [W, H, ...] copies of the K side (what an all-gather of per-rank [H, per, ...] parts produces)
    rm = lambda t, n: t.view(H, W, n, *t.shape[2:]).transpose(0, 1).contiguous()    # noqa: E731
    vt_g = rm(vt, per // 64)
    out16 = torch.empty((Lq, H * 128), dtype=torch.bfloat16, device=DEV)
    oq = torch.empty((Lq, H * 128), dtype=torch.int8, device=DEV)
    os_ = torch.empty((Lq // 128, H), dtype=torch.float32, device=DEV)
    for h0 in (0, 2):
        sl = slice(h0, h0 + 2)
        o_g = out16.view(-1)[h0 * 128:]
        if sage:
            args = (q8[sl], qs[sl], rm(k8, per)[:, sl], rm(ks, per // 64)[:, sl], vt_g[:, sl], None)
            K.attn_i8_sp(*args, o_g, 128, H * 128, W * per, add_t=o_l[sl])
            K.attn_i8_sp(*args, torch.bfloat16, 128, H * 128, W * per, add_t=o_l[sl], quant_out=(oq, os_, h0, H))
        else:
            args = (q[sl], rm(kk, per)[:, sl], vt_g[:, sl], None)
            K.attn_16_sp(*args, o_g, 128, H * 128, W * per, add_t=o_l[sl])
            K.attn_16_sp(*args, torch.bfloat16, 128, H * 128, W * per, add_t=o_l[sl], quant_out=(oq, os_, h0, H))
    q_ref, s_ref = K.quant_i8_block128(out16)
    assert torch.equal(os_, s_ref) and torch.equal(oq, q_ref)
    assert kb == vt.shape[1]


def test_sla_core_refuses_what_it_does_not_handle(K):
    """the shared core raises for BLKQ = 64 with dense / quant_out / vt instead of running the BLKQ = 128 fast path
    on a 64-row LUT, and the elu / relu feature map keeps the CALLER's QK arithmetic (16-bit SLA stays 16-bit)."""
    from turbodiffusion_amd.sla import sparse_linear_attention_hld
    H, L = 2, 512
    q, k = _qk(H, L, 1), _qk(H, L, 2)
    v = torch.randn(L, H, 128, generator=torch.Generator().manual_seed(3)).bfloat16().to(DEV)
    out = torch.empty((L, H, 128), dtype=torch.bfloat16, device=DEV)
    wp, bp = (torch.randn(128, 128, generator=torch.Generator().manual_seed(4)) * 0.05).to(DEV), torch.zeros(128, device=DEV)
    vt = K.v_transpose(v, 128, H * 128, L, H, 128, torch.bfloat16)
    for kw in (dict(dense=True), dict(quant_out=True), dict(vt=vt)):
        with pytest.raises(ValueError, match="BLKQ = 64"):
            sparse_linear_attention_hld(q, k, v, wp, bp, 0.5, False, out, 128, H * 128, (128, H * 128), blkq=64, **kw)
    # 16-bit SLA + elu with caller-supplied V^T tiles: the sparse branch must be the 16-bit kernel's result
    a, _, _ = sparse_linear_attention_hld(q, k, v, wp, bp, 0.5, False, out, 128, H * 128, (128, H * 128), vt=vt, feature_map="elu")
    a = a.clone()
    b16 = torch.empty_like(out)
    sparse_linear_attention_hld(q, k, v, None, None, 0.5, False, b16, 128, H * 128, (128, H * 128), vt=vt)
    kv_t, ksum = K.sla_linear_kv(k, vt, feature_map="elu")
    K.sla_linear_out_(q, kv_t, ksum, wp, bp, b16, 128, H * 128, feature_map="elu")
    assert torch.equal(a, b16)
