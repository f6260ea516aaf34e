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
    assert torch.equal(K.seq_sum(k), want)
    buf = torch.full((H * 128 + 77,), -1.0, device=DEV)
    got = K.seq_sum(k, out=buf[:H * 128].view(H, 128))
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
    allp[W - 1] = K.seq_sum(k)
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


@pytest.mark.parametrize("width,L_grid", [("14b", (3, 22, 36)), ("1p3b", (5, 30, 52))])
def test_projection_as_two_launches_is_bit_identical_at_the_real_widths(width, L_grid):
    """WanModel.split_qkv (the q|k|v projection as K|V then Q, the K-side chain of the attention glue under the Q GEMM,
    sla.sagesla_split_projection) against the one-launch form at the widths of C4 / C5 and C1 — one block, a token count that is
    not a multiple of any tile (L = F*H*W/4): every output element comes from the same kernel arithmetic, so the bits agree."""
    from oracle import make_golden_r04 as R4
    from turbodiffusion_amd.wan import WanModel
    cfg = dict(R4.CFG14 if width == "14b" else R4.CFG13, num_layers=1)
    with torch.device(DEV):
        net = WanModel(attention_type="sagesla", sla_topk=0.2, quant_linear=True, **cfg)
    sd = R4.hash_globals(cfg, device=DEV)
    sd.update(R4.hash_layer(cfg, 0, device=DEV))
    net.load_from_float_state_dict(sd)
    del sd
    net.eval()
    g = torch.Generator().manual_seed(11)
    F_, H_, W_ = L_grid
    x = torch.randn(1, 16, F_, H_, W_, generator=g).to(DEV).bfloat16()
    ctx = (torch.randn(1, 512, 4096, generator=g) * 0.2).to(DEV).bfloat16()
    t = torch.tensor([[700.0]], device=DEV).bfloat16()
    assert net.split_qkv and net.two_streams
    a = net(x, t, ctx, _return_tokens=True)[0].clone()
    net.split_qkv = False
    b = net(x, t, ctx, _return_tokens=True)[0].clone()
    net.two_streams = False
    c = net(x, t, ctx, _return_tokens=True)[0]
    assert torch.isfinite(a.float()).all() and torch.equal(a, b) and torch.equal(a, c)


# ---------------------------------------------------------------- parity at the REAL size, full depth (VERDICT r04 "next" 2)
DEQUANT = pytest.mark.parametrize("dequant", ["one-VALU (shipping default)", "exact"])


def _set_dequant(dequant):
    """Round 6: both W8A8 dequant forms against the same fixture — the library default (one-VALU, csrc/capi.hip) and the exact
    arithmetic of the reference (td_set_tuning(TD_TUNE_GEMM_FAST, 1)); tests/conftest.py resets the knob after the test."""
    from turbodiffusion_amd import kernels as K_
    K_.set_tuning(K_.TUNE_GEMM_FAST, 1 if dequant == "exact" else 0)


def _gold(name):
    path = os.path.join(GOLD, f"r05_{name}.pt")
    if not os.path.exists(path):
        pytest.skip(f"{path} not generated (oracle/make_golden_r05.py {name}: hours of CPU)")
    return torch.load(path, weights_only=False)


def _c1_net():
    from oracle import make_golden_r05 as R5
    from oracle import wan_ref as W
    from turbodiffusion_amd.wan import WanModel
    with torch.device(DEV):
        net = WanModel(attention_type="sagesla", sla_topk=R5.C1["topk"], quant_linear=True, **R5.C1["cfg"])
    sd = W.make_state_dict(R5.C1["cfg"], seed=R5.C1["sd_seed"])
    net.load_from_float_state_dict({k_: v_.to(DEV) for k_, v_ in sd.items()})
    return net.eval()


@DEQUANT
def test_headline_configuration_full_depth_at_the_real_length_against_the_oracle(capsys, dequant):
    """C1 as the bench runs it — Wan2.1-1.3B, 30 layers, L = 32 760 tokens, W8A8 + Fast norms + SageSLA top-k 0.1, the production
    schedule (hipGraph-able path: fused epilogues, two streams, token-half split) — ONE forward at step 2's t against the CPU
    oracle's forward of the same weights / inputs (tests/golden/r05_c1full.pt, 55 min of oracle time): tokens after the last
    block (every 32nd row + the 120-row tail block), the velocity, and the drift over depth (after blocks 1, 2, 4, 8, 16, 24).
    Bound: SURVEY §8d's 2e-2 / cosine 0.999 on the velocity would be the one-block figure; through 30 blocks of block-map
    near-ties and INT8 rounding the measured figures are printed; round 6 bounds them at 1.8e-2 (measured 1.29e-2 / 1.19e-2
    with the exact dequant: a regression of 1.5x fails)."""
    from oracle import make_golden_r05 as R5
    g = _gold("c1full")
    _set_dequant(dequant)
    net = _c1_net()
    assert net.split_tokens and net.fuse_row_stats and net.fuse_vt and net.two_streams
    x, ctx = R5.c1_inputs()
    xd, td, cd = x.to(DEV).bfloat16(), R5.c1_t(1).to(DEV), ctx.to(DEV)
    net._tap_tokens = []
    v = net(xd, td, cd)
    taps, net._tap_tokens = net._tap_tokens, None
    assert len(taps) == 30 and taps[-1].shape == (1, 32760, 1536)
    rows = g["rows"].to(DEV)
    tok = taps[-1][0][rows]
    r_tok, r_tail, r_v = rel_l2(tok, g["tok_rows"].float()), rel_l2(taps[-1][0][-120:], g["tok_rows"][-120:].float()), rel_l2(v, g["v"].float())
    depth = {i: rel_l2(taps[i][0][::g["depth_every"]], d.float()) for i, d in sorted(g["depth"].items())}
    with capsys.disabled():
        print(f"\n[C1 at full size: 30 layers x 32 760 tokens; W8A8 dequant: {dequant}] rel-L2 vs the oracle: tokens after the last block {r_tok:.4f} (tail block "
              f"{r_tail:.4f}, cosine {cosine(tok, g['tok_rows'].float()):.5f}), velocity {r_v:.4f} (cosine {cosine(v, g['v'].float()):.5f}); "
              f"after blocks " + ", ".join(f"{i + 1}: {e:.4f}" for i, e in depth.items()))
    assert torch.isfinite(v).all()
    assert r_tok < 1.8e-2 and cosine(tok, g["tok_rows"].float()) > 0.9995, r_tok
    assert r_v < 1.8e-2 and cosine(v, g["v"].float()) > 0.9995, r_v
    assert r_tail < 2.5e-2 and max(depth.values()) < 1.8e-2, (r_tail, depth)


@DEQUANT
def test_four_sampler_steps_at_the_real_size_against_the_oracle(capsys, dequant):
    """"full 4-step latent vs the eager CPU reference at identical noise: rel-L2 per step" — at the headline size: the
    same 30-layer model through the rCM loop (sampler.rcm_sample, hipGraph replay) against the oracle's loop at identical
    noise: the velocity at every step's input and the latent after every step (tests/golden/r05_c1steps.pt, ~3.6 h of
    oracle time; a partially generated fixture checks the steps it holds)."""
    from oracle import make_golden_r05 as R5
    from turbodiffusion_amd.graph import GraphedModel
    from turbodiffusion_amd.sampler import rcm_sample_iter
    g = _gold("c1steps")
    _set_dequant(dequant)
    net = GraphedModel(_c1_net())
    x0, ctx = R5.c1_inputs()
    noises = [n.to(DEV) for n in R5.c1_noises()]
    vs = []

    def model(**kw):          # the sampler's network call, with the velocity of every step kept
        v = net(**kw)
        vs.append(v.float().clone())
        return v
    lines = []
    for i, x_i in rcm_sample_iter(model, x0.to(DEV), ctx.to(DEV), num_steps=4, noises=noises, sigma_max=80.0):
        if i >= len(g["x"]):
            break
        gv, gx = (t if t.shape[-1] == R5.C1["latent"][-1] // 2 else R5._sub(t) for t in (g["v"][i], g["x"][i]))
        rv, rx = rel_l2(R5._sub(vs[i]), gv.float().to(DEV)), rel_l2(R5._sub(x_i.float()), gx.float().to(DEV))
        lines.append(f"step {i + 1}: velocity {rv:.4f}, latent {rx:.4f}")
        assert rv < 1.8e-2 and rx < 1.2e-2, (i, rv, rx)      # (measured with the exact dequant: velocity <= 1.23e-2, latent <= 0.68 %)
    with capsys.disabled():
        print(f"\n[4 rCM steps at full size, identical noise; W8A8 dequant: {dequant}] rel-L2 vs the oracle per step: " + "; ".join(lines)
              + (" (fixture partial)" if g.get("partial") else ""))
    assert lines


@DEQUANT
def test_two_blocks_at_c4_size_all_heads_against_the_oracle(capsys, dequant):
    """C4 / C5's size: dim 5120, 40 heads, ffn 13 824 at L = 75 600 tokens (720p), two blocks, all heads, top-k 0.1: tokens after the
    second block (every 128th row + the 80-row tail block) vs the oracle (tests/golden/r05_c4two.pt)."""
    from oracle import make_golden_r04 as R4
    from oracle import make_golden_r05 as R5
    from turbodiffusion_amd.wan import WanModel
    g = _gold("c4two")
    _set_dequant(dequant)
    cfg = R5.C4["cfg"]
    with torch.device(DEV):
        net = WanModel(attention_type="sagesla", sla_topk=R5.C4["topk"], quant_linear=True, **cfg)
    sd = R4.hash_globals(cfg, device=DEV)
    for i in range(cfg["num_layers"]):
        sd.update(R4.hash_layer(cfg, i, device=DEV))
    net.load_from_float_state_dict(sd)
    del sd
    x, t, ctx = R5.c4_inputs()
    tok = net.eval()(x.to(DEV).bfloat16(), t.to(DEV), ctx.to(DEV), _return_tokens=True)[0]
    assert tok.shape == (75600, 5120)
    rows = g["rows"].to(DEV)
    r, tail = rel_l2(tok[rows], g["tok_rows"].float()), rel_l2(tok[-80:], g["tok_rows"][-80:].float())
    with capsys.disabled():
        print(f"\n[two blocks at C4's size: dim 5120 x 75 600 tokens, 40 heads; W8A8 dequant: {dequant}] rel-L2 vs the oracle: sampled rows {r:.4f}, tail block {tail:.4f}")
    assert r < 1.4e-2 and tail < 1.8e-2 and cosine(tok[rows], g["tok_rows"].float()) > 0.9995      # (measured 8.8e-3 / 8.8e-3)


# ---------------------------------------------------------------- round 6: C5's OWN inputs at full size (VERDICT r05 "next" 6)
def test_two_a14b_blocks_with_i2v_inputs_at_720p_against_the_oracle(capsys):
    """C5 = Wan2.2-A14B I2V 720p: in_dim 36 — the noisy latent + y = mask | image latent concatenated on channels in front of
    the patch embedding (wan2.2_i2v_infer.py:149-152, wan2pt2.py:644-645) — dim 5120, 40 heads, ffn 13 824 at L = 75 600 tokens,
    two blocks, top-k 0.1, the library's default W8A8 dequant: tokens after the second block (every 128th row + the 80-row tail
    block) AND the velocity (head + unpatchify; every second latent row / column) vs the oracle's forward of the same hashed
    weights (tests/golden/r06_c5two.pt, oracle/make_golden_r06.py: ~1 h of CPU).  What r05_c4two left open: patch_embed with
    K = 144 and the head at this size."""
    from oracle import make_golden_r04 as R4
    from oracle import make_golden_r05 as R5
    from oracle import make_golden_r06 as R6
    from turbodiffusion_amd import kernels as K_
    from turbodiffusion_amd.wan import WanModel
    path = os.path.join(GOLD, "r06_c5two.pt")
    if not os.path.exists(path):
        pytest.skip(f"{path} not generated (oracle/make_golden_r06.py c5two: an hour of CPU)")
    g = torch.load(path, weights_only=False)
    K_.set_tuning(K_.TUNE_GEMM_FAST, 0)      # what ships
    cfg = R6.C5["cfg"]
    with torch.device(DEV):
        net = WanModel(attention_type="sagesla", sla_topk=R6.C5["topk"], quant_linear=True, **cfg)
    sd = R4.hash_globals(cfg, device=DEV)
    for i in range(cfg["num_layers"]):
        sd.update(R4.hash_layer(cfg, i, device=DEV))
    net.load_from_float_state_dict(sd)
    del sd
    x, y, t, ctx = R6.c5_inputs()
    net.eval()
    net._tap_tokens = []
    v = net(x.to(DEV).bfloat16(), t.to(DEV), ctx.to(DEV), y_B_C_T_H_W=y.to(DEV).bfloat16())
    taps, net._tap_tokens = net._tap_tokens, None
    tok = taps[-1][0]
    assert tok.shape == (75600, 5120) and v.shape == (1, 16, 21, 90, 160)
    rows = g["rows"].to(DEV)
    r, tail = rel_l2(tok[rows], g["tok_rows"].float()), rel_l2(tok[-80:], g["tok_rows"][-80:].float())
    rv = rel_l2(R5._sub(v.float()), g["v_sub"].float())
    with capsys.disabled():
        print(f"\n[two A14B blocks, I2V inputs (in_dim 36), 75 600 tokens] rel-L2 vs the oracle: sampled rows {r:.4f}, tail block {tail:.4f}, "
              f"velocity {rv:.4f} (cosine {cosine(R5._sub(v.float()), g['v_sub'].float()):.5f})")
    assert torch.isfinite(v).all()
    assert r < 1.4e-2 and tail < 1.8e-2 and cosine(tok[rows], g["tok_rows"].float()) > 0.9995
    assert rv < 1.8e-2 and cosine(R5._sub(v.float()), g["v_sub"].float()) > 0.9995
