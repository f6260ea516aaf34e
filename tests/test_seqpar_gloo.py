"""Sequence-parallel path on CPU: world_size 2, gloo, oracle compute backend (tests/oracle_ops.py).

Checks that sharding + packed all-gather + rank-padded layouts reproduce the single-rank result of the
SAME backend (bit-exact up to the smooth-K mean's summation order) and the module-level oracle."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import sla_ref as S
from tests import oracle_ops
from tests.test_gpu_sla import qkv
from tests.util import cosine, rel_l2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _single_rank(q, k, v, at, ratio, wp, bp):
    """Same pipeline, one rank, no communication (sla.sparse_linear_attention_hld's order of operations)."""
    from turbodiffusion_amd import sla as sla_mod
    import turbodiffusion_amd.sla
    H, L, D = q.shape
    out = torch.zeros(L, H, D, dtype=q.dtype)
    K_saved = sla_mod.K
    sla_mod.K = type("K", (), {n: staticmethod(getattr(oracle_ops, n)) for n in dir(oracle_ops) if not n.startswith("_")})
    sla_mod.K.cdiv = staticmethod(lambda a, b: (a + b - 1) // b)
    try:
        dense = at in ("original", "sage")
        sla_mod.sparse_linear_attention_hld(q, k, v, None if dense else wp, None if dense else bp, ratio,
                                            at in ("sage", "sagesla"), out, D, H * D, (L * D, D), dense=dense)
    finally:
        sla_mod.K = K_saved
    return out


def _worker(rank, world, port, at, L, ratio, ret, H=2):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from turbodiffusion_amd.seqpar import SeqParallel
        torch.manual_seed(0)
        D = 128
        q, k, v = (t[0].contiguous() for t in qkv(H, L, 21))  # [H, L, D], identical on every rank
        g = torch.Generator().manual_seed(3)
        wp, bp = torch.randn(D, D, generator=g) * 0.05, torch.randn(D, generator=g) * 0.05
        sp = SeqParallel(ops=oracle_ops)
        sp._groups_forced = True          # toy shards: keep the head-group pipeline (production picks 1 group for them)
        s, e = sp.plan(L)
        L_loc = e - s
        out = torch.zeros(L_loc, H, D, dtype=q.dtype)
        v_loc = v[:, s:e].contiguous()
        sp.self_attention(q[:, s:e].contiguous(), k[:, s:e].contiguous(), v_loc, (L_loc * D, D), out, D, H * D,
                          at, ratio, wp, bp)
        full = sp.gather_tokens(out.reshape(1, L_loc, H * D), L)[0].view(L, H, D)
        if rank == 0:
            ref1 = _single_rank(q, k, v, at, ratio, wp, bp)
            ret["sp_vs_single"] = rel_l2(full, ref1)
            ret["exact_frac"] = (full == ref1).float().mean().item()
            if at == "sagesla":
                refm = S.sagesla_forward(q.transpose(0, 1)[None], k.transpose(0, 1)[None], v.transpose(0, 1)[None],
                                         wp, bp, ratio)[0]
                ret["sp_vs_module_oracle"] = rel_l2(full, refm)
                ret["cos"] = cosine(full, refm)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("at,L,ratio", [("sagesla", 700, 0.3), ("sage", 450, 1.0), ("sla", 520, 0.5),
                                        ("original", 300, 1.0)])
def test_seqpar_world2_matches_single_rank(at, L, ratio):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), at, L, ratio, ret), nprocs=2, join=True)
    assert ret["sp_vs_single"] < 5e-3, dict(ret)
    assert ret["exact_frac"] > 0.98, dict(ret)
    if at == "sagesla":
        assert ret["sp_vs_module_oracle"] < 2e-2 and ret["cos"] > 0.999, dict(ret)


@pytest.mark.parametrize("world,at,L,ratio,H", [(4, "sagesla", 900, 0.3, 2),      # 8 Q blocks: 256, 256, 256, 132 tokens
                                                 (8, "sagesla", 1970, 0.25, 6),    # 16 Q blocks: 7 x 256 + a 178-token tail rank;
                                                                                   # 6 heads: 3 groups of 2 (largest divisor <= 4)
                                                 (8, "sage", 1100, 1.0, 1)])       # 9 Q blocks, per = 256: ranks 5-7 would be empty
def test_seqpar_world4_and_world8(world, at, L, ratio, H):
    """The rank-padded gathered layout, the packed exchange and the head-group pipeline at world sizes 4 and 8 (gloo,
    CPU, oracle compute backend), incl. a short tail rank; and a shape that leaves ranks without tokens is refused on
    EVERY rank before any collective (no hang)."""
    mgr = mp.Manager()
    ret = mgr.dict()
    if L == 1100:
        with pytest.raises(Exception) as ei:
            mp.spawn(_worker, args=(world, _free_port(), at, L, ratio, ret, H), nprocs=world, join=True)
        assert "use fewer ranks" in str(ei.value)
        return
    mp.spawn(_worker, args=(world, _free_port(), at, L, ratio, ret, H), nprocs=world, join=True)
    assert ret["sp_vs_single"] < 5e-3, dict(ret)
    assert ret["exact_frac"] > 0.97, dict(ret)
    assert ret["sp_vs_module_oracle"] < 2e-2 and ret["cos"] > 0.999, dict(ret)


def test_plan_c4_591_blocks_over_8_ranks():
    """C4/C5 (720p): L = 75 600 = 590 full 128-token blocks + an 80-token tail = 591 Q blocks over 8 ranks -> 74 blocks
    (9472 tokens) on ranks 0-6, 73 on rank 7 whose last block is the 80-token tail."""
    from turbodiffusion_amd.seqpar import SeqParallel

    class Fake(SeqParallel):
        def __init__(self, rank, world):
            self.rank, self.world, self.L = rank, world, None

    spans = [Fake(r, 8).plan(75600) for r in range(8)]
    assert [e - s for s, e in spans] == [9472] * 7 + [75600 - 7 * 9472]
    assert (75600 - 7 * 9472) == 72 * 128 + 80
    with pytest.raises(ValueError, match="use fewer ranks"):
        Fake(0, 8).plan(128 * 7)          # evaluated identically on every rank, rank 0 included


def test_plan_covers_all_tokens_block_aligned():
    from turbodiffusion_amd.seqpar import SeqParallel

    class Fake(SeqParallel):
        def __init__(self, rank, world):
            self.rank, self.world, self.L = rank, world, None

    for L in (32760, 75600, 1000, 129):
        for world in (1, 2, 4, 8):
            if (L + 127) // 128 < world:
                continue
            spans = [Fake(r, world).plan(L) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == L
            for (a0, a1), (b0, b1) in zip(spans[:-1], spans[1:]):
                assert a1 == b0 and a0 % 128 == 0 and b0 % 128 == 0


def test_pack_layout_groups_and_alignment():
    """PackLayout (round-5 form): equal head groups (largest divisor of H not above head_groups); ONE flat buffer = the all-head
    block (pooled K, linear-branch partials of ALL heads) followed by the group blocks (K codes, V^T tiles, K scales); 256-byte
    aligned sections addressing disjoint byte ranges; G pieces, the first = all-head block + group 0; the views of the local
    buffer and of the gathered pieces name the same bytes."""
    import torch
    from turbodiffusion_amd.seqpar import PackLayout
    for H, want in ((12, 4), (40, 4), (6, 3), (5, 1), (2, 2), (1, 1)):
        lay = PackLayout(H, 256, 128, 4, True, False, torch.bfloat16)
        assert lay.G == want and lay.hg * lay.G == H
        for offs, sizes, ext in ((lay.offs, lay.sizes, lay.gb), (lay.aoffs, lay.asizes, lay.ab)):
            assert all(o % 256 == 0 for o in offs.values()) and ext % 256 == 0
            ends = sorted((offs[n], offs[n] + sizes[n]) for n in sizes if sizes[n])
            assert all(a[1] <= b[0] for a, b in zip(ends, ends[1:])) and ends[-1][1] <= ext
        assert lay.total == lay.ab + lay.G * lay.gb and len(lay.pieces) == lay.G
        assert lay.pieces[0] == (0, lay.ab + lay.gb) and all(lay.pieces[g] == (lay.ab + g * lay.gb, lay.gb) for g in range(1, lay.G))
        assert sum(n for _, n in lay.pieces) == lay.total
        buf = torch.arange(lay.total, dtype=torch.int64).to(torch.uint8)
        assert lay.group_section(buf, "vt").shape == (lay.G, lay.hg, 4, 128, 64) and lay.group_section(buf, "vt").dtype == torch.float16
        assert lay.group_section(buf, "k").dtype == torch.int8 and lay.early_sum == H * 128 and lay.early_lin == H * 128 * 129
        assert lay.all_section(buf, "pk").shape == (H, 4, 128) and lay.all_section(buf, "pk").dtype == torch.bfloat16
        # "gathered" by one rank: the pieces of the local buffer ARE the rows of the gather outputs
        outs = [lay.piece(buf, g).view(1, -1) for g in range(lay.G)]
        assert torch.equal(lay.gathered(outs, None, "pk")[0].view(torch.uint8), lay.all_section(buf, "pk").view(torch.uint8))
        for g in range(lay.G):
            for name in ("k", "vt", "ks"):
                assert torch.equal(lay.gathered(outs, g, name)[0].reshape(-1).view(torch.uint8),
                                   lay.group_section(buf, name)[g].reshape(-1).view(torch.uint8)), (H, g, name)
    dense = PackLayout(12, 256, 128, 4, False, True, torch.bfloat16)     # "original": 16-bit K, no scales / pooled / partials
    assert dense.sizes["ks"] == dense.asizes["pk"] == dense.early_lin == 0 and dense.spec["k"][0] == torch.bfloat16 and dense.ab == 0


def _reissue_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from turbodiffusion_amd import graph
        from turbodiffusion_amd.seqpar import _Gather
        src = torch.full((5,), float(rank))
        out = torch.empty(world, 5)
        calls = []

        class Rec:   # a stand-in recorder: what graph.SegmentRecorder does at an eager point, minus the capture
            def _eager(self, fn):
                calls.append(fn)
                return fn()
        graph._ACTIVE = Rec()
        h = _Gather(dist.group.WORLD, out, src, True)
        graph.eager_point(h.issue)
        h.wait()
        graph._ACTIVE = None
        first = out.clone()
        src.add_(10.0)              # "the previous graph segment" rewrites the send buffer in place ...
        for fn in calls:            # ... and the replay re-issues the recorded collectives on the same objects
            fn()
        if rank == 0:
            ret["n_eager"] = len(calls)
            ret["first"] = first[:, 0].tolist()
            ret["second"] = out[:, 0].tolist()
    finally:
        dist.destroy_process_group()


def test_recorded_collectives_can_be_reissued_on_the_same_buffers():
    """The replay contract of graph.SegmentRecorder on the collective side: an all-gather handle's issue() and wait() are
    recorded as eager points and, called again later, move the CURRENT contents of the same send buffer into the same
    receive buffer."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_reissue_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret["n_eager"] == 2 and ret["first"] == [0.0, 1.0] and ret["second"] == [10.0, 11.0], dict(ret)


def _agree_worker(rank, world, port, scenario, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from turbodiffusion_amd.graph import agree_on_capture_outcome
        # scenario -> (failed, collectives issued) per rank
        failed, k = {"all_ok": (False, 12), "all_fail_same_point": (True, 5),
                     "one_rank_fails": (rank == 1, 5 if rank == 1 else 12)}[scenario]
        try:
            ret[rank] = ("agreed", agree_on_capture_outcome(dist.group.WORLD, failed, k, timeout_s=30))
            # a second exchange on the same group must not see the first one's keys
            ret[10 + rank] = ("agreed", agree_on_capture_outcome(dist.group.WORLD, False, 3, timeout_s=30))
        except RuntimeError as e:
            ret[rank] = ("raised", str(e)[:60])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("scenario", ["all_ok", "all_fail_same_point", "one_rank_fails"])
def test_capture_outcome_is_agreed_through_the_store(scenario):
    """graph.agree_on_capture_outcome: a failure of the segmented capture that is common to all ranks lets every rank fall
    back to eager enqueue; a rank-local one makes EVERY rank raise (their collective sequences are misaligned) — decided
    over the store, not over the communicator in question."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_agree_worker, args=(2, _free_port(), scenario, ret), nprocs=2, join=True)
    if scenario == "one_rank_fails":
        assert ret[0][0] == "raised" and ret[1][0] == "raised", dict(ret)
    else:
        want = scenario == "all_fail_same_point"
        assert ret[0] == ("agreed", want) and ret[1] == ("agreed", want), dict(ret)
        assert ret[10] == ("agreed", False) and ret[11] == ("agreed", False), dict(ret)


def test_reference_hook_names_on_wanmodel():
    """``WanModel.enable_context_parallel`` / ``disable_context_parallel`` / ``is_context_parallel_enabled`` — the reference's
    hook names (rcm/networks/wan2pt1.py:774-796) — wire the sequence-parallel adapter in and out, and hand the group to every
    block's ``attn_op`` as ``MinimalA2AAttnOp.set_context_parallel_group`` would (a2a_cp.py:193-196)."""
    from turbodiffusion_amd.seqpar import EmulatedGroup
    from turbodiffusion_amd.wan import WanModel
    net = WanModel(dim=256, ffn_dim=512, num_heads=2, num_layers=2, quant_linear=False, attention_type="sagesla")
    assert net.is_context_parallel_enabled is False
    grp = EmulatedGroup(1, 4)
    assert net.enable_context_parallel(grp) is net
    assert net.is_context_parallel_enabled is True and net.seq_parallel.broadcast_inputs is True
    assert net.seq_parallel.sp.rank == 1 and net.seq_parallel.sp.world == 4
    assert all(b.self_attn.attn_op.pg is grp for b in net.blocks)
    net.disable_context_parallel()
    assert net.is_context_parallel_enabled is False and all(b.self_attn.attn_op.pg is None for b in net.blocks)


def test_emulated_rank_runs_one_ranks_work_without_communication():
    """seqpar.EmulatedGroup (bench.py --emulate-rank): rank r of N alone — the plan, the packed layout and the gathered
    buffers have the real sizes, every peer slot holds a copy of this rank's shard, no process group exists."""
    from turbodiffusion_amd.seqpar import EmulatedGroup, SeqParallel
    assert not dist.is_initialized()
    H, L, D, W = 2, 900, 128, 4
    q, k, v = (t[0].contiguous() for t in qkv(H, L, 21))
    g = torch.Generator().manual_seed(3)
    wp, bp = torch.randn(D, D, generator=g) * 0.05, torch.randn(D, generator=g) * 0.05
    for r in (0, 3):
        sp = SeqParallel(EmulatedGroup(r, W), ops=oracle_ops)
        s, e = sp.plan(L)
        assert (s, e) == (r * 256, min(L, (r + 1) * 256)) and sp.capturable
        assert [sp.groups_for(12, p_) for p_ in (16384, 8192, 4096)] == [4, 2, 2] and sp.groups_for(40, 9472) == 4 and sp.groups_for(2, 256) == 2
        assert sp.branches_in_parallel(12, 4096, 2) and sp.branches_in_parallel(12, 16384, 4) and not sp.branches_in_parallel(40, 9472, 4)
        L_loc = e - s
        out = torch.zeros(L_loc, H, D, dtype=q.dtype)
        sp.self_attention(q[:, s:e].contiguous(), k[:, s:e].contiguous(), v[:, s:e].contiguous(), (L_loc * D, D), out, D, H * D,
                          "sagesla", 0.3, wp, bp)
        assert torch.isfinite(out.float()).all() and out.float().abs().sum() > 0
        full = sp.gather_tokens(out.reshape(1, L_loc, H * D), L)
        assert full.shape == (1, L, H * D)
        allg = sp.all_gather(torch.arange(6.0).view(2, 3))
        assert allg.shape == (W, 2, 3) and all(torch.equal(allg[i], allg[0]) for i in range(W))


def _bcast_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from turbodiffusion_amd.seqpar import SeqParallel
        sp = SeqParallel(ops=oracle_ops)
        mine = torch.full((2, 3), float(rank + 1))
        got = sp.broadcast(mine)
        ret[rank] = (got.tolist(), mine.tolist(), sp.broadcast(None) is None, sp.capturable)
        # round 5: receivers get the SAME buffer object every time; a non-contiguous input is received
        # contiguous; cached=True ships the payload only when the first rank's tensor (object / version) changed, and the
        # receivers' buffer then keeps its version — what WanModel.prepare_text keys its text K / V^T cache on
        again = sp.broadcast(torch.full((2, 3), float(rank + 1)))
        nc = sp.broadcast(torch.full((3, 2), float(rank + 5)).t(), slot=1)
        text = torch.full((4,), float(10 * (rank + 1)))
        b1 = sp.broadcast(text, slot=2, cached=True)
        v1 = b1._version
        b2 = sp.broadcast(text, slot=2, cached=True)          # same object, same version: flag only
        v2 = b2._version
        text.add_(1.0)                                        # new contents on the first rank
        b3 = sp.broadcast(text, slot=2, cached=True)
        ret[f"r5_{rank}"] = (again is got if rank else True, nc.is_contiguous(), nc.tolist(), b2 is b1 and b3 is b1, v2 == v1,
                             b3.tolist(), (b3._version > v2) if rank else True)
    finally:
        dist.destroy_process_group()


def test_input_broadcast_from_the_groups_first_rank():
    """SeqParallel.broadcast (the reference forward's input broadcast, wan2pt1.py:629-636): every rank gets the first rank's
    data; a receiving rank's own tensor is not written."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_bcast_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    ones, twos = [[1.0] * 3] * 2, [[2.0] * 3] * 2
    assert ret[0] == (ones, ones, True, False) and ret[1] == (ones, twos, True, False), dict(ret)
    for r in range(2):
        same_obj, contig, nc, same_text_obj, version_kept, b3, bumped = ret[f"r5_{r}"]
        assert same_obj and contig and nc == [[5.0] * 3] * 2 and same_text_obj and version_kept and b3 == [11.0] * 4 and bumped, (r, ret[f"r5_{r}"])


def _quiesce_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import time
        from turbodiffusion_amd.graph import quiesce_collective_watchdog
        t0 = time.perf_counter()
        ret[rank] = (quiesce_collective_watchdog(dist.group.WORLD), time.perf_counter() - t0)
    finally:
        dist.destroy_process_group()


def test_watchdog_drain_is_a_no_op_off_rccl():
    """graph.quiesce_collective_watchdog only waits where torch's NCCL (= RCCL) watchdog exists: no process group, a gloo group
    and an emulated group return at once (the GPU leg: tests/test_gpu_seqpar.py)."""
    from turbodiffusion_amd.graph import quiesce_collective_watchdog
    from turbodiffusion_amd.seqpar import EmulatedGroup
    assert quiesce_collective_watchdog() == 0.0 and quiesce_collective_watchdog(EmulatedGroup(0, 8)) == 0.0
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_quiesce_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert all(ret[r][0] == 0.0 and ret[r][1] < 0.2 for r in range(2)), dict(ret)
