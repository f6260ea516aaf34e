"""Sequence-parallel DiT step over the GPUs of one node (BASELINE.json north star).

One process per GPU, ``torch.distributed`` backend ``nccl`` (= RCCL over xGMI).  The flattened
(t h w) token axis is sharded in contiguous, 128-token-aligned ranges (a rank owns whole Q blocks and
therefore whole 64-key K blocks).  Everything in a transformer block is token-local except
self-attention, which needs the K side of every rank.  Per self-attention layer:

  1. tiny all-gather of the per-head K column sums (-> the global smooth-K mean, SLA/core.py:197);
  2. an all-gather (issued as ``head_groups`` consecutive asynchronous pieces, one per group of heads, so that
     attention on the first heads runs while the later heads' bytes are still on the links) of a packed per-rank
     buffer holding the rank's *quantised* K-side state:
         pooled K blocks and fp32 linear-branch partials (ck^T v [H,128,128], sum ck [H,128]) of ALL heads (first piece) |
         per head group: K int8 [hg, per, 128] | V^T fp16 MFMA tiles [hg, per/64, 128, 64] | K scales
     (3 B per token-channel instead of the reference Ulysses path's 4 all-to-alls of bf16 q,k,v,o:
     rcm/utils/a2a_cp.py:146-182 — and no head-count divisibility constraint: 12 heads shard 8 ways);
  3. every rank builds the LUT for ITS Q blocks against the global pooled K, runs the block-sparse
     Sage attention of its Q blocks against the gathered K/V, and finishes the linear branch from the
     summed partials.

The DiT output is all-gathered once per step (``cat_outputs_cp``, rcm/utils/context_parallel.py:60-91).
xGMI is a full mesh, so an all-gather is one hop: each rank pushes its shard to the 7 peers in parallel.

The compute backend is injectable (``ops``): production uses the HIP kernels (``kernels``); the
world_size-2 gloo tests on CPU inject the oracle so the sharding / packing / gather logic is covered
without a GPU.  There is no silent fallback: ``ops=None`` means HIP.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .graph import eager_point


def _cdiv(a, b):
    return (a + b - 1) // b


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class EmulatedGroup:
    """Stands in for a process group when ONE rank of an N-way split is run alone on one GPU (``bench.py --emulate-rank``):
    no communication — an all-gather fills every rank's slot of its output with THIS rank's
    shard (a device copy on the compute stream: the HBM writes a real gather's incoming xGMI traffic would cause, not
    overlapped with anything, so the compute term it measures is on the pessimistic side).  Results are meaningless as
    attention outputs (every rank's K side is a copy of this rank's); shapes, launches, LUT sizes and kernel work are those of
    the real rank.  Test / measurement infrastructure: never selected implicitly."""

    def __init__(self, rank: int, world: int):
        assert 0 <= rank < world
        self.rank, self.world = rank, world


import os as _os
EMU_WIRE_STREAM = _os.environ.get("TD_EMU_WIRE_STREAM", "1") != "0"   # EmulatedGroup: transfers on their own stream (0: on the issuing stream, round 4's form)
_EMU_STREAMS = {}


def _emu_stream():
    dev = torch.cuda.current_device()
    st = _EMU_STREAMS.get(dev)
    if st is None:
        st = _EMU_STREAMS[dev] = torch.cuda.Stream()
    return st


WAIT_PROBE = None   # bench.py (N > 1): a list that receives (event before, event after, bytes gathered) of every stream-side wait
#                     for an asynchronous gather — the time the compute stream sat waiting for the wire, per collective


class _Gather:
    """One all-gather on fixed buffers: ``issue()`` / ``wait()`` may be called again (graph replay) — same tensors.
    Issue and wait are ``graph.eager_point``s: a plain call, except under a SEGMENTED capture (graph.SegmentRecorder),
    where they stay between the graph segments and are re-issued at every replay."""

    def __init__(self, group, out, src, async_op):
        self.group, self.out, self.src, self.async_op = group, out, src, async_op
        self.work = None
        self.emulated = isinstance(group, EmulatedGroup)
        # single-GPU test rig (several ranks on one device, tests/test_gpu_seqpar.py): gloo gathers host memory
        self.via_host = (not self.emulated) and src.is_cuda and dist.get_backend(group) == "gloo"

    def issue(self):
        if self.emulated:
            # the emulated transfer runs where a real one does: on a stream of its own (RCCL's communicator stream), forked
            # off the issuing stream here and joined at ``wait`` — kernels enqueued in between overlap it, as they overlap
            # the wire.  (Still CU / HBM work of THIS GPU where the fabric's DMA would do the writes: pessimistic, not free.)
            # Under a SEGMENTED capture (graph.SegmentRecorder) issue() is an eager point — re-run at every replay — while the
            # join in wait() would be recorded once, inside a captured segment, against the event of the capture pass: the next
            # segment could read the buffers before this replay's copy finished.  There the copy stays on the issuing stream.
            # The choice is made at the FIRST issue and kept: the replay re-runs this method outside any capture.
            from . import graph as _graph
            if not hasattr(self, "_on_issuing_stream"):
                self._on_issuing_stream = _graph._ACTIVE is not None
            if self.out.is_cuda and self.async_op and EMU_WIRE_STREAM and not self._on_issuing_stream:
                main = torch.cuda.current_stream()
                st = _emu_stream()
                e = torch.cuda.Event()
                e.record(main)
                st.wait_event(e)
                with torch.cuda.stream(st):
                    self.out.view(self.group.world, -1).copy_(self.src.view(1, -1).expand(self.group.world, -1))
                    self.done = torch.cuda.Event()
                    self.done.record(st)
                return
            self.out.view(self.group.world, -1).copy_(self.src.view(1, -1).expand(self.group.world, -1))
        elif self.via_host:
            host = torch.empty(self.out.shape, dtype=self.out.dtype)
            dist.all_gather_into_tensor(host.view(-1), self.src.cpu(), group=self.group)
            self.out.copy_(host)
        else:
            self.work = dist.all_gather_into_tensor(self.out.view(-1), self.src, group=self.group, async_op=self.async_op)

    def _wait(self):
        if self.work is not None:
            probe = WAIT_PROBE
            if probe is not None and self.out.is_cuda:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                self.work.wait()
                b.record()
                probe.append((a, b, self.out.numel() * self.out.element_size()))
            else:
                self.work.wait()
            self.work = None

    def wait(self):
        if self.emulated:
            d = getattr(self, "done", None)
            if d is not None:
                torch.cuda.current_stream().wait_event(d)
                self.done = None
            return
        if self.async_op and not self.via_host:
            eager_point(self._wait)


class PackLayout:
    """Byte layout of the K-side send buffer of one self-attention layer: ONE flat uint8 buffer of ``total`` bytes,

        [ all-head block (ab bytes): pk | kv | kss for ALL H heads ]  [ group 0 (gb bytes): k | vt | ks ]  [ group 1 ] ...

    sent as G pieces (one asynchronous all-gather each): piece 0 = the all-head block + group 0, piece g > 0 = group g.
    Group g = heads [g*hg, (g+1)*hg); inside a group block the sections  k (int8 codes, or the 16-bit K when not Sage) | vt (V^T
    MFMA tiles) | ks (K scales), each [hg, ...] at the rank-padded extents (``per`` rows, ``per/64`` blocks) and 256-byte
    aligned.  The all-head block holds the pooled K blocks of ALL heads: the block map is ONE launch over all heads right behind
    the first piece instead of one per head group.  The fp32 linear-branch partials (ck^T v [H,128,128], sum ck [H,128]) do not
    depend on the smooth-K mean and are small: round 5 sends them AHEAD of the pack (``early_lin``: a second tiny exchange next
    to the K column sums), so that the branch's reduction over the ranks and its second pass (-> o_l) run on a side stream
    beside the K quantiser and the pack's exchange; the attention kernel then takes o_l in its epilogue and quantises for the o
    projection, as on one GPU (no read-modify-write pass, no separate quantiser behind attention).
    G = the largest divisor of H not above ``head_groups`` (equal groups: the producer kernels address head h as group
    h // hg, member h % hg — td_common.h ``td_head_off``; the all-head sections use its flat form)."""

    def __init__(self, H, per, D, head_groups, sage, dense, dt):
        self.H, self.per, self.D, self.sage, self.dense, self.linear, self.dt = H, per, D, sage, dense, not dense, dt
        self.G = max(g for g in range(1, max(1, min(head_groups, H)) + 1) if H % g == 0)
        self.hg = hg = H // self.G
        self.kbp = kbp = per // 64
        self.pdt = torch.float16 if sage else dt
        self.spec = {   # group sections: name -> (dtype, per-group shape)
            "k": (torch.int8 if sage else dt, (hg, per, D)),
            "vt": (self.pdt, (hg, kbp, D, 64)),
            "ks": (torch.float32, (hg, kbp)) if sage else None,
        }
        self.aspec = {  # all-head sections: name -> (dtype, shape over ALL heads)
            "pk": (dt, (H, kbp, D)) if not dense else None,
        }
        # the EARLY send buffer (f32, one tiny exchange ahead of the pack): the rank's K column sums [H, D] (-> the global
        # smooth-K mean), then — linear branch — its partials sum ck [H, D] | ck^T v [H, D, D]
        self.early_sum = H * D
        self.early_lin = (H * D + H * D * D) if self.linear else 0

        def lay(spec):
            offs, sizes, o = {}, {}, 0
            for name, sp in spec.items():
                n = 0
                if sp is not None:
                    n = torch.empty((), dtype=sp[0]).element_size()
                    for d in sp[1]:
                        n *= d
                offs[name], sizes[name] = o, n
                o += _cdiv(n, 256) * 256
            return offs, sizes, o
        self.offs, self.sizes, self.gb = lay(self.spec)
        self.aoffs, self.asizes, self.ab = lay(self.aspec)
        self.total = self.ab + self.G * self.gb
        self.pieces = [(0, self.ab + self.gb)] + [(self.ab + g * self.gb, self.gb) for g in range(1, self.G)]   # (offset, bytes)

    def piece(self, pack, g):
        """pack uint8 [total] -> the bytes of piece g (a view)."""
        o, n = self.pieces[g]
        return pack[o:o + n]

    def group_section(self, pack, name):
        """pack uint8 [total] (the LOCAL send buffer) -> [G, *per-group shape] VIEW of a group section."""
        dtype, shape = self.spec[name]
        off, n = self.offs[name], self.sizes[name]
        return pack[self.ab:].view(self.G, self.gb)[:, off:off + n].view(dtype).view((self.G,) + shape)

    def all_section(self, pack, name):
        """pack uint8 [total] (the LOCAL send buffer) -> [*all-head shape] VIEW of an all-head section."""
        dtype, shape = self.aspec[name]
        off, n = self.aoffs[name], self.asizes[name]
        return pack[off:off + n].view(dtype).view(shape)

    def gathered(self, outs, g, name):
        """outs: the all-gather outputs of the pieces (uint8 [W, piece bytes] each) -> rank-major VIEW [W, *shape] of group g's
        section ``name`` (g = None: an all-head section, which travels in piece 0)."""
        if g is None:
            dtype, shape = self.aspec[name]
            off, n = self.aoffs[name], self.asizes[name]
            buf = outs[0]
        else:
            dtype, shape = self.spec[name]
            off, n = self.offs[name] + (self.ab if g == 0 else 0), self.sizes[name]
            buf = outs[g]
        return buf[:, off:off + n].view(dtype).view((buf.shape[0],) + shape)


class SeqParallel:
    def __init__(self, group=None, ops=None):
        if isinstance(group, EmulatedGroup):
            self.group, self.rank, self.world = group, group.rank, group.world
        else:
            self.group = group if group is not None else dist.group.WORLD
            self.rank = dist.get_rank(self.group)
            self.world = dist.get_world_size(self.group)
        # collectives INSIDE the captured hipGraph (graph.GraphedModel): RCCL's calls are stream-ordered, torch's nccl
        # backend forks its communicator stream off the capturing stream and joins it back at ``work.wait()``, so the whole
        # sharded forward is ONE graph with no host work between its kernels.  gloo (the CPU / one-GPU rig) moves bytes on
        # the host and cannot be captured: there the forward stays a chain of graph segments around eager collectives.
        self.capturable = isinstance(group, EmulatedGroup) or dist.get_backend(self.group) == "nccl"
        if ops is None:
            from . import kernels as ops  # HIP; raises on CPU tensors
        self.ops = ops
        self.L = None
        self.head_groups = 4   # self-attention K/V exchange and attention pipelined over this many head groups
        # the head groups' block map / attention / linear branch as PARALLEL branches (one HIP stream per group, fork / join
        # by events = graph edges under capture): a rank of an 8-way split launches 3 heads x 32 Q blocks = 96 workgroups per
        # group — a third of the chip — so queued behind each other the four groups take four under-filled waves
        # (bench.py --emulate-rank 0/8, round 4: 33.5 ms per DiT step against 84 / 8 = 10.5)
        self.parallel_groups = True
        self._group_streams = {}
        self._bcast_bufs = {}      # receiving ranks of ``broadcast``: one persistent buffer per (slot, shape, dtype)
        self._bcast_sent = {}      # first rank: per slot (tag of the tensor last sent with cached=True, the tensor itself)
        import os   # A/B switches for the measurement tools (tools/gpu/*.sh); results do not depend on them
        self._groups_forced = bool(os.environ.get("TD_SP_HEAD_GROUPS"))
        if self._groups_forced:
            self.head_groups = int(os.environ["TD_SP_HEAD_GROUPS"])
        if os.environ.get("TD_SP_PARALLEL_GROUPS"):
            self.parallel_groups = os.environ["TD_SP_PARALLEL_GROUPS"] != "0"
        # round 6: ONE early exchange per layer (K column sums | linear partials in one buffer) instead of two — one collective
        # less per layer (a launch + a rendezvous: what decides a small shard is the count, not the bytes); the sums then
        # leave a few microseconds later, behind the pack's first half (V^T tiles + linear partials), with this rank's Q
        # quantiser between the issue and the wait as before.  TD_SP_MERGE_EARLY=0: round 5's two exchanges (A/B).
        self.merge_early = os.environ.get("TD_SP_MERGE_EARLY", "1") != "0"

    def collectives_per_layer(self, G, linear):
        """all-gathers one self-attention layer issues: the early exchange(s) + one per head-group piece."""
        return (1 if (self.merge_early or not linear) else 2) + G

    def exposed_collectives_per_layer(self, linear):
        """of which the compute stream has to wait for with nothing of its own left to run: the early exchange in front of the K
        quantiser and the first piece in front of the block map (later pieces fly under the previous group's attention)."""
        return 2

    def groups_for(self, H, per):
        """Head groups of the K-side exchange for a rank with ``per`` tokens.  More groups hide more of the exchange behind
        attention (only the first group's transfer is exposed) but cut the attention / block-map / linear-branch launches into
        H/G heads x per/128 Q blocks workgroups each, and a launch far below the chip's ~768 resident workgroups takes as long
        as one workgroup's serial walk over its keys whatever its size.  Read off the rank-emulation sweep of round 4
        (profiles/r04_emu_group_sweep.txt; compute + exposed wire per DiT step): 1.3B / 480p over 2 / 4 / 8 ranks is best with
        4 / 2 / 2 groups (60.7 / 37.1 / 25.3 ms), A14B / 720p over 8 with 4 (240.7 ms) — at least two groups (half of the
        exchange hidden always pays), about one group per 384 workgroups of attention.  TD_SP_HEAD_GROUPS overrides."""
        if self._groups_forced:
            return self.head_groups
        return max(1, min(self.head_groups, max(2, (H * (per // 128)) // 384)))

    def branches_in_parallel(self, H, per, G):
        """The groups' kernels as parallel graph branches only while one group's launch under-fills the chip (< 512 workgroups):
        1.3B over 8 ranks 28.5 vs 31.1 ms with 4 groups; A14B over 8 ranks (740 workgroups per group) 243 vs 228 ms — full-size
        launches beside each other only thrash the caches (same sweep)."""
        return self.parallel_groups and G > 1 and (H // G) * (per // 128) < 512

    # ------------------------------------------------------------------ token sharding
    def plan(self, L: int):
        qb = _cdiv(L, 128)
        self.per = _cdiv(qb, self.world) * 128          # padded tokens per rank
        # evaluated identically on EVERY rank (a rank-local failure would leave the others inside an all-gather)
        if (self.world - 1) * self.per >= L:
            raise ValueError(f"sequence parallelism over {self.world} ranks needs more than {(self.world - 1) * self.per} "
                             f"tokens (128-token-aligned shards of {self.per}); L = {L}: the last rank would own none — use fewer ranks")
        self.start = min(L, self.rank * self.per)
        self.stop = min(L, self.start + self.per)
        self.L = L
        return self.start, self.stop

    def shard_tokens(self, x, cos, sin):
        """x [B, L, C] (replicated on every rank) -> this rank's [B, L_loc, C]; RoPE tables alike
        (split_inputs_cp, wan2pt1.py:662-664,685-686)."""
        s, e = self.plan(x.shape[1])
        return x[:, s:e].contiguous(), cos[s:e].contiguous(), sin[s:e].contiguous()

    def all_gather(self, t: torch.Tensor, async_op: bool = False):
        """[...] -> [world, ...] (rank-major).  ``async_op``: returns (out, handle); with RCCL the gather runs on the
        communicator's own stream and ``handle.wait()`` makes the compute stream wait for it, so kernels enqueued in
        between overlap the transfer.  Both the issue and the wait are ``graph.eager_point``s: under a segmented hipGraph
        capture they stay outside the graphs and are re-issued on the same buffers at every replay."""
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        h = _Gather(self.group, out, t.contiguous().view(-1), async_op)
        eager_point(h.issue)
        return (out, h) if async_op else out

    def broadcast(self, t, slot=0, cached=False):
        """The group's first rank's contents of ``t`` on every rank (``broadcast`` of rcm/utils/context_parallel.py:167-184 as
        ``WanModel.forward`` uses it, wan2pt1.py:629-636).  Data only: the reference also ships the SHAPE first, which costs a
        host synchronisation per input; here a shape mismatch between ranks is the caller's error.  The caller's tensor is
        not written on the receiving ranks: they receive into a PERSISTENT contiguous buffer per (slot, shape, dtype) — the
        same tensor object every forward (``dist.broadcast`` on nccl refuses non-contiguous tensors; a fresh clone per
        forward left a stale entry per step in ``WanModel.prepare_text``'s cache).
        cached (the text embedding: constant over the steps of a video): the first rank first broadcasts ONE flag — "same
        tensor object at the same version as last time?" — and the payload only when it changed; the receivers' buffer then
        keeps its identity AND its version, so their text K / V^T cache hits on the steps after the first like the first
        rank's does (every rank runs in lockstep: a miss on any rank set the step time)."""
        if t is None:
            return None
        if isinstance(self.group, EmulatedGroup):
            return t
        src = min(dist.get_process_group_ranks(self.group))
        me_src = dist.get_rank() == src
        via_host = t.is_cuda and dist.get_backend(self.group) == "gloo"
        if me_src:
            buf = t if t.is_contiguous() else t.contiguous()
        else:
            # ALIASING (documented contract): a receiving rank gets the SAME tensor object for a slot on every
            # forward with that input signature — the previous forward's x / t / text of that slot is overwritten.  WanModel.forward
            # consumes its inputs inside the call, so nothing outlives it; a caller that keeps a received tensor must clone it.
            # At most 4 signatures per slot are kept (least recently used goes; a video server alternates a handful of shapes):
            # an evicted text buffer only costs one payload broadcast + one text-cache miss when that signature returns.
            key = (slot, tuple(t.shape), t.dtype, str(t.device))
            buf = self._bcast_bufs.pop(key, None)
            if buf is None:
                buf = torch.empty(t.shape, dtype=t.dtype, device=t.device)
                mine = [k_ for k_ in self._bcast_bufs if k_[0] == slot]
                if len(mine) >= 4:
                    del self._bcast_bufs[mine[0]]          # dicts keep insertion order: the first is the least recently used
            self._bcast_bufs[key] = buf                    # (re-inserted at the end: most recently used)

        def payload():
            if via_host:
                host = buf.cpu()
                dist.broadcast(host, src, group=self.group)
                if not me_src:
                    buf.copy_(host)
            elif cached and not me_src:
                # c10d writes through the data pointer and does not bump the tensor's version counter: receive into a scratch
                # buffer and copy_ (which does), or a new prompt would look like the cached one to whatever keys on
                # (address, version) — WanModel.prepare_text
                recv = torch.empty_like(buf)
                dist.broadcast(recv, src, group=self.group)
                buf.copy_(recv)
            else:
                dist.broadcast(buf, src, group=self.group)

        def fn():
            if not cached:
                return payload()
            flag = torch.zeros(1, dtype=torch.int64, device="cpu" if (via_host or not t.is_cuda) else t.device)
            if me_src:
                tag = (t.data_ptr(), t._version, tuple(t.shape), t.dtype)
                last = self._bcast_sent.get(slot)
                changed = last is None or last[0] != tag
                self._bcast_sent[slot] = (tag, t)      # (the OBJECT is held: its address cannot be recycled for another text)
                flag.fill_(1 if changed else 0)
            dist.broadcast(flag, src, group=self.group)
            if int(flag.item()):
                payload()
        eager_point(fn)
        return buf

    def gather_tokens(self, out_loc, L):
        """[B, L_loc, C] -> [B, L, C] on every rank (cat_outputs_cp)."""
        B, L_loc, C = out_loc.shape
        buf = torch.zeros((B, self.per, C), dtype=out_loc.dtype, device=out_loc.device)
        buf[:, :L_loc] = out_loc
        allb = self.all_gather(buf)  # [W, B, per, C]
        return allb.permute(1, 0, 2, 3).reshape(B, self.world * self.per, C)[:, :L].contiguous()

    # ------------------------------------------------------------------ self attention
    def self_attention(self, q, k, v_src, v_strides, out, o_stride_h, o_stride_l, attention_type, topk_ratio,
                       proj_w=None, proj_b=None, quant_out=False):
        """q, k: [H, L_loc, D] (after RoPE); V element (h,l,d) at v_src + h*v_strides[0] + l*v_strides[1] + d;
        out: element (h,l,d) at out + h*o_stride_h + l*o_stride_l + d (this rank's rows).
        quant_out: return the [L_loc, H*D] result block-quantised for the o projection instead — (int8 [L_loc, H*D], f32
        [ceil(L_loc/128), H]); ``out`` then only supplies the 16-bit dtype (a tensor or a torch.dtype)."""
        ops, W = self.ops, self.world
        H, L_loc, D = q.shape
        L, per = self.L, self.per
        kb_tot = _cdiv(L, 64)
        sage = attention_type in ("sage", "sagesla")
        dense = attention_type in ("original", "sage")
        linear = not dense
        dt = q.dtype
        dev = q.device
        if not dense and min(kb_tot, int(topk_ratio * kb_tot)) < 1:
            # same condition on every rank, raised BEFORE any collective (sla.py mirrors SLA/utils.py:61-62 the same way)
            raise ValueError(f"block-sparse attention with topk ratio {topk_ratio} selects no block of {kb_tot} (L = {L} tokens)")

        from . import graph as _graph
        on_streams = q.is_cuda and _graph._ACTIVE is None     # side streams need real streams and no segmented capture in progress
        lay = PackLayout(H, per, D, self.groups_for(H, per), sage, dense, dt)
        # ---- (1) the EARLY exchange: this rank's per-head K column sums (-> the global smooth-K mean: chunk
        # partials + one in-order pass) and, for the linear branch, its fp32 partials — neither needs anything global, both are
        # small, latency-bound all-gathers.  The column sums go first (the K quantiser waits for them); the Q side of this
        # rank is prepared while they are in flight.  The mean itself is formed inside the K quantiser from the gathered sums.
        early = torch.empty((lay.early_sum + lay.early_lin,), dtype=torch.float32, device=dev)
        allp = km_work = alll = lin_work = None
        merged = self.merge_early and linear and (sage or not dense)
        if sage or not dense:
            part = ops.seq_sum(k, out=early[:lay.early_sum].view(H, D))
            if not merged:
                allp, km_work = self.all_gather(part, async_op=True)                # [W, H, D]
        lin_ks = early[lay.early_sum:lay.early_sum + H * D].view(H, D) if linear else None
        lin_kv = early[lay.early_sum + H * D:].view(H, D, D) if linear else None
        pack = ops.sp_pack_begin(k, v_src, v_strides, L_loc, lay, lin_kv, lin_ks)      # V^T tiles; linear partials -> early
        if merged:
            # ONE exchange: [sums | ks | kv] of every rank; both consumers read rank-major views of the same gathered buffer
            alle, km_work = self.all_gather(early, async_op=True)                   # [W, early_sum + early_lin]
            lin_work = km_work
            allp = alle[:, :lay.early_sum].view(W, H, D)
            alll = alle[:, lay.early_sum:]
        # ---- the linear branch's reduction over the ranks and its second pass (-> o_l [all heads], which the attention
        # epilogue adds): on a
        # side stream beside the Q / K quantisers and the pack's exchange below (a graph branch under capture), joined before attention
        o_l = e_ol = e_early = None
        pq = q_q = q_s = None
        q_side_done = False

        def q_side():
            nonlocal pq, q_q, q_s, q_side_done
            if sage:
                pq, q_q, q_s = ops.sage_quant_pool(q, None, 128, want_pool=not dense)
            elif not dense:
                pq, _, _ = ops.sage_quant_pool(q, None, 128, want_quant=False)
            q_side_done = True
        if linear and not on_streams:
            # ONE stream (a segmented capture, the CPU rig): this rank's Q quantiser goes BETWEEN the issue of the early exchange
            # and its wait, where the side-stream form puts it by construction (issue and wait back to back were two
            # eager points around an empty graph segment, the exchange fully exposed)
            q_side()
        if linear:
            if not merged:
                alll, lin_work = self.all_gather(early[lay.early_sum:], async_op=True)      # [W, H*D + H*D*D]
            main = torch.cuda.current_stream() if on_streams else None
            st_l = self._stream(100) if on_streams else None
            if st_l is not None:
                e_f = torch.cuda.Event()
                e_f.record(main)
                st_l.wait_event(e_f)
            with (torch.cuda.stream(st_l) if st_l is not None else _NullCtx()):
                lin_work.wait()
                if merged and st_l is not None:
                    # the ONE early gather has two consumers on two streams; its handle is awaited once, here — the main stream
                    # joins through this event (recorded right behind the wait: nothing else of this stream is in front of it)
                    e_early = torch.cuda.Event()
                    e_early.record(st_l)
                ks_parts = alll[:, :H * D].view(W, H, D)                       # [W, H, D] rank-major views of the early gather
                kv_parts = alll[:, H * D:].view(W, H, D, D)
                kv_t, ksum = ops.sla_linear_kv_final(kv_parts, ks_parts, W, D * D, kv_parts.stride(0), D, ks_parts.stride(0), H, D, dt)
                o_l = ops.sla_linear_out_t(q, kv_t, ksum, proj_w, proj_b)
                if st_l is not None:
                    e_ol = torch.cuda.Event()
                    e_ol.record(st_l)
        if not q_side_done:
            q_side()
        if allp is not None:
            if merged and linear and e_early is not None:
                torch.cuda.current_stream().wait_event(e_early)
            else:
                km_work.wait()      # (merged on ONE stream: already awaited above, in order — a no-op)

        # ---- (2) the part of the K-side state that needs the mean, written straight into the send buffer: ONE launch ----
        ops.sp_pack_finish(pack, k, None if allp is None else (allp, L), L_loc, lay)

        # ---- (2) + (3), pipelined over head groups: the pieces' all-gathers are issued back to back (asynchronously,
        # RCCL's own stream executes them in order); behind the first one the block map (ONE launch over all heads); then,
        # group by group, the gather is awaited and that group's attention runs while the later groups' bytes are still on
        # the links.  The bytes on the wire are the same as with one gather; what changes is that attention — about a third
        # of a layer's compute — overlaps most of the exchange instead of waiting for all of it.
        handles = [_Gather(self.group, torch.empty((W, lay.pieces[g][1]), dtype=torch.uint8, device=dev), lay.piece(pack, g), True)
                   for g in range(lay.G)]

        seg = _graph._ACTIVE is not None

        def issue_all():   # ONE eager point (graph.py): all pieces' gathers
            for h in handles:
                h.issue()
            if seg:        # segmented capture: the first piece's wait rides in the same eager point (no empty segment between them)
                handles[0]._wait()
        eager_point(issue_all)
        outs = [h.out for h in handles]

        if not seg:
            handles[0].wait()
        # The attention / block-map kernels read the K side STRAIGHT from the all-gathers' rank-major outputs (the *_sp
        # entry points: block j = block j % kbp of rank j // kbp): no re-layout of the gathered K / V^T / scales / pooled K.
        topk = min(kb_tot, int(topk_ratio * kb_tot)) if not dense else 0
        lut = None
        if not dense:
            lut = ops.sla_topk_sp(pq, lay.gathered(outs, None, "pk"), topk, kb_tot)                       # all heads
        if e_ol is not None:
            torch.cuda.current_stream().wait_event(e_ol)
        qo = None
        if quant_out:
            odt = out if isinstance(out, torch.dtype) else out.dtype
            qo = (torch.empty((L_loc, H * D), dtype=torch.int8, device=dev),
                  torch.empty((_cdiv(L_loc, 128), H), dtype=torch.float32, device=dev))
            out = odt
        par = on_streams and self.branches_in_parallel(H, per, lay.G)   # (a segment cannot end with forked streams)
        main = torch.cuda.current_stream() if par else None
        joins = []
        if par:
            e_fork = torch.cuda.Event()
            e_fork.record(main)
        for g in range(lay.G):
            h0, h1 = g * lay.hg, (g + 1) * lay.hg
            st = None
            if par and g > 0:
                st = self._stream(g)
                st.wait_event(e_fork)
            with (torch.cuda.stream(st) if st is not None else _NullCtx()):
                if g > 0:
                    handles[g].wait()      # (RCCL: the CURRENT stream waits for that group's gather)
                self._group(ops, lay, outs, g, q, q_q, q_s, lut, o_l, h0, h1, out, qo, o_stride_h, o_stride_l, L, sage, H)
                if st is not None:
                    e = torch.cuda.Event()
                    e.record(st)
                    joins.append(e)
        for e in joins:
            main.wait_event(e)
        # allocation safety without record_stream: tensors made inside a branch live on that branch's stream and are reused
        # there only; what the branches READ (q side, LUT, o_l, the gathered buffers) and WRITE (out / the quantised output)
        # belongs to the main stream and is released by Python after the join above has been enqueued
        return qo if quant_out else out

    def _stream(self, gi):
        key = (torch.cuda.current_device(), gi)
        st = self._group_streams.get(key)
        if st is None:
            st = self._group_streams[key] = torch.cuda.Stream()
        return st

    def _group(self, ops, lay, outs, g, q, q_q, q_s, lut, o_l, h0, h1, out, qo, o_stride_h, o_stride_l, L, sage, H):
        """Attention of the heads [h0, h1) against head group g's gathered K side (block map rows, o_l: slices of the all-head
        results made behind the first piece)."""
        vt_g = lay.gathered(outs, g, "vt")                                      # [W, hg, kbp, D, 64] view
        lut_g = None if lut is None else lut[h0:h1]
        add_g = None if o_l is None else o_l[h0:h1]
        quant = None if qo is None else (qo[0], qo[1], h0, H)
        out_g = out if qo is not None else out.view(-1)[h0 * o_stride_h:]   # same buffer, origin moved to the group's first head
        if sage:
            ops.attn_i8_sp(q_q[h0:h1], q_s[h0:h1], lay.gathered(outs, g, "k"), lay.gathered(outs, g, "ks"), vt_g, lut_g, out_g,
                           o_stride_h, o_stride_l, L, add_t=add_g, quant_out=quant)
        else:
            ops.attn_16_sp(q[h0:h1], lay.gathered(outs, g, "k"), vt_g, lut_g, out_g, o_stride_h, o_stride_l, L, add_t=add_g,
                           quant_out=quant)


class _ModelAdapter:
    """What ``WanModel._self_attention`` calls when ``model.seq_parallel`` is set."""

    def __init__(self, sp: SeqParallel, broadcast_inputs: bool = False):
        self.sp = sp
        self.broadcast_inputs = broadcast_inputs

    def broadcast(self, *tensors):
        """Inputs of one forward from the group's first rank (only when enabled through the reference's hook,
        ``WanModel.enable_context_parallel``; ``seqpar.enable`` callers hand identical inputs to every rank)."""
        # (x, t, text, y) as WanModel.forward passes them: the text (index 2) is constant over the steps of a video
        return tuple(self.sp.broadcast(t, slot=i, cached=(i == 2 and len(tensors) == 4)) for i, t in enumerate(tensors))

    def shard_tokens(self, x, cos, sin):
        return self.sp.shard_tokens(x, cos, sin)

    def gather_tokens(self, out, L):
        return self.sp.gather_tokens(out, L)

    def shard_range(self, L):
        """This rank's token range [start, stop) of L tokens (the fused patch-embedding kernel reads only those)."""
        return self.sp.plan(L)

    def self_attention(self, model, fused, q, k, qkv, out, quant_out=False):
        dim, D = model.dim, 128
        return self.sp.self_attention(q, k, qkv[:, 2 * dim:], (D, 3 * dim), out, D, dim, model.attention_type,
                                      model.sla_topk, fused.get("proj_w"), fused.get("proj_b"), quant_out=quant_out)


def enable(model, group=None, ops=None, broadcast_inputs=False):
    """What ``WanModel.enable_context_parallel`` (wan2pt1.py:786-792; same name on this repo's WanModel) does.
    broadcast_inputs: replicate x / t / text / y from the group's first rank at the top of every forward, as the
    reference's forward does (wan2pt1.py:627-636); off when the caller already holds identical inputs on every rank."""
    model.seq_parallel = _ModelAdapter(SeqParallel(group, ops), broadcast_inputs)
    return model


def disable(model):
    model.seq_parallel = None
    return model
