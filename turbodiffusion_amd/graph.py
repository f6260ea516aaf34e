"""hipGraph capture of one DiT forward ("a hipGraph-captured step").

A Wan-DiT step on the MI355X is ~1250 kernel launches of 5-1000 us each; enqueued one by one through
ctypes the host falls behind the GPU on the short ones.  ``GraphedModel`` wraps a ``WanModel`` (same call
surface), captures the whole forward into a HIP graph the first time a given input signature is seen
(after one eager warm-up call, which also runs every one-time ``hipFuncSetAttribute`` and builds the
RoPE / fused-weight caches) and afterwards replays it: inputs are copied into the graph's static
buffers, the output is the graph's static output tensor (cloned, so the caller owns it).

The captured work is exactly the eager work: every C-ABI entry point launches on torch's *current*
stream, which during capture is the capturing stream; outputs are allocated from the graph's private pool.

Sequence parallelism (``seqpar``): a forward then contains collectives (RCCL all-gathers and the waits for them).  They are
NOT captured; the forward becomes a chain of graph SEGMENTS with the collectives re-issued eagerly between them
(``SegmentRecorder``): ``seqpar`` wraps every collective / wait in ``eager_point(fn)``; while a recorder is capturing, that
ends the current segment, runs ``fn`` for real, remembers it and opens the next segment; a replay walks the recorded chain
(graph.replay() / fn()) in the same order.  All segments share one memory pool (legal because they are always replayed in
capture order), so a tensor produced in one segment and consumed after the collective stays where it was.  Without this a
rank of an 8-way split spends longer enqueueing its ~1100 launches from Python than the GPU spends executing them.
"""
from __future__ import annotations

from typing import Optional

import os

import torch


_ACTIVE = None   # the SegmentRecorder that is capturing on this thread, if any


def eager_point(fn):
    """Run ``fn()`` now; if a segmented capture is in progress, outside of it (and again at this point of every replay).
    ``fn`` must only touch tensors that stay alive (closure) — it is called again with the same objects."""
    rec = _ACTIVE
    if rec is None:
        return fn()
    return rec._eager(fn)


class SegmentRecorder:
    """A forward captured as graph segments separated by eager points (see module docstring)."""

    def __init__(self):
        self.chain = []          # ("graph", CUDAGraph) | ("eager", fn)
        self.pool = torch.cuda.graph_pool_handle()
        self.stream = torch.cuda.Stream()
        self._g = None

    def _begin(self):
        self._g = torch.cuda.CUDAGraph()
        self._g.capture_begin(pool=self.pool, capture_error_mode="thread_local")

    def _end(self):
        g, self._g = self._g, None
        g.capture_end()
        self.chain.append(("graph", g))

    def _eager(self, fn):
        self._end()
        out = fn()
        self.chain.append(("eager", fn))
        self._begin()
        return out

    def capture(self, fn):
        """Run ``fn()`` once in capture mode (kernels are recorded, not executed; eager points execute) -> fn's result
        (static tensors: valid after every ``replay``)."""
        global _ACTIVE
        assert _ACTIVE is None, "nested segmented capture"
        torch.cuda.synchronize()
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            self._begin()
            _ACTIVE = self
            try:
                out = fn()
            except BaseException:
                _ACTIVE = None
                if self._g is not None:      # close the open capture without masking the original error
                    try:
                        self._end()
                    except Exception:
                        pass
                raise
            _ACTIVE = None
            self._end()
        torch.cuda.current_stream().wait_stream(self.stream)
        return out

    def replay(self):
        for kind, item in self.chain:
            if kind == "graph":
                item.replay()
            else:
                item()

    @property
    def n_segments(self):
        return sum(1 for k, _ in self.chain if k == "graph")


def quiesce_collective_watchdog(group=None):
    """Call right before a hipGraph capture that will contain RCCL collectives.  torch's ProcessGroupNCCL keeps every EAGER
    collective's Work in a list that its watchdog thread polls every 100 ms (``hipEventQuery`` on the Work's end event, which
    was recorded on the communicator's own stream) until the Work is complete, then drops it.  A capture pulls that same
    communicator stream into capture mode at its first collective; if the watchdog then polls a leftover eager Work — one
    that finished milliseconds before the capture began, e.g. the warm-up forward's — HIP answers ``hipErrorCapturedEvent``
    ("operation not permitted on an event last recorded in a capturing stream"), the watchdog thread throws and the process
    aborts.  Seen once in the 1-rank RCCL test on the MI355X box (round 4; a toy forward whose capture lasts tens of ms: a
    narrow window) — with a full-size forward the capture lasts about a second and the poll would land in it every time.
    This torch build has no guard of its own (no "pending event queries" wait in ``capture_begin``) and exposes no way to ask
    whether the list is empty, so: finish all device work, then give the watchdog a few of its periods to retire it
    (``TD_SP_WATCHDOG_DRAIN_S``, default 0.4 s; once per captured input signature).  Works issued DURING a capture are not
    put on that list by torch (it checks the current stream's capture status), so nothing new appears until the capture ends."""
    import time
    try:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return 0.0
        g = group if group is not None and not type(group).__name__ == "EmulatedGroup" else None
        if "nccl" not in str(dist.get_backend(g)):
            return 0.0
    except Exception:
        return 0.0
    torch.cuda.synchronize()
    t = float(os.environ.get("TD_SP_WATCHDOG_DRAIN_S", "0.4"))
    if t > 0:
        time.sleep(t)
    return t


_AGREE_SEQ = {}   # per group: how many capture outcomes have been exchanged (identical on every rank: same code path)


def agree_on_capture_outcome(group, failed: bool, eager_points: int, timeout_s: float = 120.0):
    """Every rank of ``group`` reports how its segmented capture pass ended — (failed?, collectives issued before the end) —
    through the process group's key-value STORE, i.e. outside the communicator whose call sequence is in question, and
    learns the others' reports.  Returns ``failed`` if all ranks agree (a deterministic failure — same code, same shapes,
    same point — leaves every rank's collective sequence aligned, so a common eager fallback is safe).  A rank-local
    failure (out of memory, pool exhaustion on one GPU) leaves the ranks' sequences misaligned: the failing rank issued k
    collectives, the others N, and any further collective would hang or gather mismatched buffers — every rank raises
    instead, so the job dies with a message rather than hanging."""
    import datetime
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1:
        return failed
    try:
        store = dist.distributed_c10d._get_default_store()    # (private accessor: there is no public one)
    except Exception as e:
        raise RuntimeError("the sequence-parallel capture needs the default process group's key-value store to agree on the "
                           f"capture outcome across ranks, and torch does not expose it here ({e!r})") from e
    ranks = dist.get_process_group_ranks(group)
    me = dist.get_rank()
    gkey = tuple(ranks)           # the FULL rank tuple: a strided and a contiguous group with the same first rank and size differ
    seq = _AGREE_SEQ[gkey] = _AGREE_SEQ.get(gkey, 0) + 1
    import hashlib
    pre = f"td_sp_capture/{hashlib.sha1(repr(gkey).encode()).hexdigest()[:16]}/{seq}/"
    store.set(pre + str(me), f"{int(failed)},{eager_points}")
    keys = [pre + str(r) for r in ranks]
    store.wait(keys, datetime.timedelta(seconds=timeout_s))
    reports = {r: store.get(pre + str(r)).decode() for r in ranks}
    # housekeeping: everyone has read exchange seq-1 by the time anyone writes seq+1's keys AFTER passing this wait, so the
    # PREVIOUS exchange's own key can go (deleting the current one could race a slower rank's get)
    if seq > 1:
        try:
            store.delete_key(f"td_sp_capture/{hashlib.sha1(repr(gkey).encode()).hexdigest()[:16]}/{seq - 1}/{me}")
        except Exception:   # a store without delete support (FileStore): the keys are a few bytes each
            pass
    if len(set(reports.values())) != 1:
        raise RuntimeError("segmented hipGraph capture of the sequence-parallel forward ended differently on the ranks of the "
                           f"group (rank -> failed,collectives issued: {reports}); their collective sequences are misaligned and "
                           "cannot continue — fix the rank-local failure (memory?) or run with eager enqueue")
    return failed


class GraphedModel(torch.nn.Module):
    def __init__(self, net: torch.nn.Module):
        super().__init__()
        self.net = net
        self._graphs = {}
        self._text_src = {}   # per graph: (the text tensor last copied into its static buffer — the OBJECT, held so that its
        #                       address cannot be recycled for another prompt — , its version at that time)
        self._epoch = getattr(net, "_weights_epoch", 0)
        self._sp_eager = False        # set when a segmented capture failed: eager from then on (sp_capture_error says why)
        self.sp_capture_error = None
        self.sp_whole_graph_error = None   # why the collectives could not be captured inside ONE graph (None: they were / not tried)
        self.sp_graph_mode = None          # how the sequence-parallel forward is replayed (set at capture)
        import os
        # Collectives INSIDE one hipGraph: always for an emulated group (no communicator, no watchdog); for a real RCCL group
        # OPT-IN (TD_SP_WHOLE_GRAPH=1) until it has passed on more than one physical GPU — torch's ProcessGroupNCCL watchdog
        # aborts the PROCESS (no exception, no fallback) if it polls a leftover eager Work while the communicator stream is
        # in capture mode, and the only guard available from Python is a drain by time (quiesce_collective_watchdog).  The
        # default for real groups is the segmented form: graph segments around eagerly re-issued collectives, which needs
        # no such guard.  TD_SP_WHOLE_GRAPH=0 forces segments everywhere (A/B).
        self._whole_graph_env = os.environ.get("TD_SP_WHOLE_GRAPH", "")

    def _key(self, x, t, ctx, y):
        return (tuple(x.shape), x.dtype, tuple(t.shape), t.dtype, tuple(ctx.shape), ctx.dtype,
                None if y is None else (tuple(y.shape), y.dtype))

    def _refresh_text(self, key, sc, crossattn_emb):
        """Copy the caller's text embedding into the graph's static buffer unless it is the SAME tensor object at the same
        version as last time (the other steps of one video).  The object is held: a per-request embedding that is freed and
        re-created gets the same address and version 0 from the caching allocator — comparing (data_ptr, version) alone would
        skip the copy and render the previous prompt."""
        src = self._text_src.get(key)
        if src is None or src[0] is not crossattn_emb or src[1] != crossattn_emb._version:
            sc.copy_(crossattn_emb)
            self._text_src[key] = (crossattn_emb, crossattn_emb._version)

    @torch.no_grad()
    def forward(self, x_B_C_T_H_W, timesteps_B_T, crossattn_emb, frame_cond_crossattn_emb_B_L_D=None,
                y_B_C_T_H_W: Optional[torch.Tensor] = None, **kwargs):
        if frame_cond_crossattn_emb_B_L_D is not None or self._sp_eager:
            return self.net(x_B_C_T_H_W, timesteps_B_T, crossattn_emb,
                            frame_cond_crossattn_emb_B_L_D=frame_cond_crossattn_emb_B_L_D,
                            y_B_C_T_H_W=y_B_C_T_H_W, **kwargs)
        epoch = getattr(self.net, "_weights_epoch", 0)
        if epoch != self._epoch:   # derived weight copies were dropped: captured graphs hold pointers into freed tensors
            self._graphs.clear()
            self._text_src.clear()
            self._epoch = epoch
        key = self._key(x_B_C_T_H_W, timesteps_B_T, crossattn_emb, y_B_C_T_H_W)
        ent = self._graphs.get(key)
        cache_text = getattr(self.net, "cache_text_kv", False)
        if ent is not None and cache_text:
            # the text-only work (text MLP, cross-attention K / V^T of every block) lives OUTSIDE the captured graph: done
            # eagerly here when the text changed, a no-op (cache hit) for the other steps of a video.  The graph reads the
            # model's persistent text buffers; should the model have re-allocated them (cache eviction), re-capture.
            sc = ent[3]
            self._refresh_text(key, sc, crossattn_emb)
            if self.net.prepare_text(sc)[4] != ent[6]:
                del self._graphs[key]
                ent = None
        if ent is None:
            sx, st, sc = x_B_C_T_H_W.clone(), timesteps_B_T.clone(), crossattn_emb.clone()
            self._text_src[key] = (crossattn_emb, crossattn_emb._version)
            sy = None if y_B_C_T_H_W is None else y_B_C_T_H_W.clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):  # eager warm-up on the side stream (lazy one-time work happens here)
                self.net(sx, st, sc, y_B_C_T_H_W=sy)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            if getattr(self.net, "seq_parallel", None) is not None:
                sp_obj = getattr(self.net.seq_parallel, "sp", None)
                sp_group = getattr(sp_obj, "group", None)
                real_group = sp_group is not None and not type(sp_group).__name__ == "EmulatedGroup"
                g = None
                whole = (self._whole_graph_env == "1") if real_group else (self._whole_graph_env != "0")
                if getattr(sp_obj, "capturable", False) and whole:
                    # (1) ONE graph for the whole sharded forward, the collectives inside it: the nccl (= RCCL) backend issues
                    # them on its communicator stream, which forks off the capturing stream and joins it again at
                    # ``work.wait()`` — stream-ordered work like any kernel.  No host work between the ~1100 launches of a
                    # rank's forward (the segmented form below re-issues ~6 collectives per layer from Python).
                    err = None
                    quiesce_collective_watchdog(sp_group if real_group else None)   # no eager Work left for the watchdog to poll
                    try:
                        wg = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(wg, capture_error_mode="thread_local"):
                            so = self.net(sx, st, sc, y_B_C_T_H_W=sy)
                        g = wg
                    except Exception as e:   # e.g. a communicator that refuses capture on this stack
                        err = e
                        torch.cuda.synchronize()
                    if real_group:
                        agree_on_capture_outcome(sp_group, err is not None, 0)
                    if err is not None:
                        import warnings
                        self.sp_whole_graph_error = repr(err)
                        warnings.warn(f"whole-graph capture of the sequence-parallel forward (collectives inside the hipGraph) "
                                      f"failed ({err!r}); falling back to graph segments around eager collectives")
                    else:
                        self.sp_graph_mode = "one hipGraph, collectives captured"
                if g is None:
                    # (2) segments between the collectives (identical chain on every rank: the eager points are collective calls)
                    g = SegmentRecorder()
                    err = None
                    try:
                        so = g.capture(lambda: self.net(sx, st, sc, y_B_C_T_H_W=sy))
                    except Exception as e:
                        err = e
                    # the ranks compare outcomes through the store BEFORE anyone issues another collective (a rank-local
                    # failure raises on every rank; a common one falls back to eager enqueue everywhere)
                    if real_group:
                        agree_on_capture_outcome(sp_group, err is not None, sum(1 for k_, _ in g.chain if k_ == "eager"))
                    if err is not None:
                        e = err
                        import warnings
                        warnings.warn(f"segmented hipGraph capture of the sequence-parallel forward failed ({e!r}); "
                                      f"this model now enqueues eagerly (slower, same results)")
                        torch.cuda.synchronize()
                        self.sp_capture_error = repr(e)
                        self._sp_eager = True
                        self.sp_graph_mode = "eager enqueue"
                        return self.net(x_B_C_T_H_W, timesteps_B_T, crossattn_emb, y_B_C_T_H_W=y_B_C_T_H_W, **kwargs)
                    self.sp_graph_mode = f"{g.n_segments} hipGraph segments, collectives eager between them"
            else:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    so = self.net(sx, st, sc, y_B_C_T_H_W=sy)
            text_gen = self.net.prepare_text(sc)[4] if cache_text else 0   # generation of the persistent text buffers the
            ent = (g, sx, st, sc, sy, so, text_gen)                         # graph points into (context, every block's K / V^T)
            self._graphs[key] = ent
        g, sx, st, sc, sy, so, _ = ent
        sx.copy_(x_B_C_T_H_W)
        st.copy_(timesteps_B_T)
        self._refresh_text(key, sc, crossattn_emb)
        if sy is not None:
            sy.copy_(y_B_C_T_H_W)
        g.replay()
        return so.clone()
