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


def _worker(rank, world, port, at, L, ratio, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from turbodiffusion_amd.seqpar import SeqParallel
        torch.manual_seed(0)
        H, D = 2, 128
        q, k, v = (t[0].contiguous() for t in qkv(H, L, 21))  # [H, L, D], identical on every rank
        g = torch.Generator().manual_seed(3)
        wp, bp = torch.randn(D, D, generator=g) * 0.05, torch.randn(D, generator=g) * 0.05
        sp = SeqParallel(ops=oracle_ops)
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
