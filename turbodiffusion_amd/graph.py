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
        self._g.capture_end()
        self.chain.append(("graph", self._g))
        self._g = None

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
            finally:
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


class GraphedModel(torch.nn.Module):
    def __init__(self, net: torch.nn.Module):
        super().__init__()
        self.net = net
        self._graphs = {}
        self._text_src = {}   # per graph: identity/version of the text tensor last copied into its static buffer
        self._epoch = getattr(net, "_weights_epoch", 0)
        self._sp_eager = False        # set when a segmented capture failed: eager from then on (sp_capture_error says why)
        self.sp_capture_error = None

    def _key(self, x, t, ctx, y):
        return (tuple(x.shape), x.dtype, tuple(t.shape), t.dtype, tuple(ctx.shape), ctx.dtype,
                None if y is None else (tuple(y.shape), y.dtype))

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
            src = (crossattn_emb.data_ptr(), crossattn_emb._version)
            if self._text_src.get(key) != src:
                sc.copy_(crossattn_emb)
                self._text_src[key] = src
            if self.net.prepare_text(sc)[2].data_ptr() != ent[6]:
                del self._graphs[key]
                ent = None
        if ent is None:
            sx, st, sc = x_B_C_T_H_W.clone(), timesteps_B_T.clone(), crossattn_emb.clone()
            self._text_src[key] = (crossattn_emb.data_ptr(), crossattn_emb._version)
            sy = None if y_B_C_T_H_W is None else y_B_C_T_H_W.clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):  # eager warm-up on the side stream (lazy one-time work happens here)
                self.net(sx, st, sc, y_B_C_T_H_W=sy)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            if getattr(self.net, "seq_parallel", None) is not None:
                # segments between the collectives (identical chain on every rank: the eager points are collective calls)
                g = SegmentRecorder()
                try:
                    so = g.capture(lambda: self.net(sx, st, sc, y_B_C_T_H_W=sy))
                except Exception as e:   # deterministic across ranks (same code, same shapes): every rank lands here
                    import warnings
                    warnings.warn(f"segmented hipGraph capture of the sequence-parallel forward failed ({e!r}); "
                                  f"this model now enqueues eagerly (slower, same results)")
                    torch.cuda.synchronize()
                    self.sp_capture_error = repr(e)
                    self._sp_eager = True
                    return self.net(x_B_C_T_H_W, timesteps_B_T, crossattn_emb, y_B_C_T_H_W=y_B_C_T_H_W, **kwargs)
            else:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    so = self.net(sx, st, sc, y_B_C_T_H_W=sy)
            text_ptr = self.net.prepare_text(sc)[2].data_ptr() if cache_text else 0
            ent = (g, sx, st, sc, sy, so, text_ptr)
            self._graphs[key] = ent
        g, sx, st, sc, sy, so, _ = ent
        sx.copy_(x_B_C_T_H_W)
        st.copy_(timesteps_B_T)
        src = (crossattn_emb.data_ptr(), crossattn_emb._version)
        if self._text_src.get(key) != src:     # a new text (or the same tensor modified in place): refresh the static copy
            sc.copy_(crossattn_emb)
            self._text_src[key] = src
        if sy is not None:
            sy.copy_(y_B_C_T_H_W)
        g.replay()
        return so.clone()
