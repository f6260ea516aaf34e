"""Host-side mirror of the reference attention-module API ``turbodiffusion.SLA``
(``/root/reference/turbodiffusion/SLA/__init__.py:16-24``, ``SLA/core.py``):
``SparseLinearAttention`` and ``SageSparseLinearAttention`` with the same constructor
arguments, the same ``forward(q, k, v, return_sparsity=False)`` on ``[B, L, H, D]`` tensors
and the same trainable ``proj_l`` parameter (checkpoint key ``...local_attn.proj_l.{weight,bias}``).

Everything below the module boundary is MI355X-native: block map, INT8 quantisation, the
block-sparse attention and the linear branch are hand-written HIP kernels.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import kernels as K

__all__ = ["SparseLinearAttention", "SageSparseLinearAttention", "sparse_linear_attention_hld"]


SUPPORTED = "head_dim = 128, BLKQ in (64, 128), BLKK = 64, feature_map in ('softmax', 'elu', 'relu')"


def _check_feature_map(feature_map):
    if feature_map not in ("softmax", "elu", "relu"):
        raise NotImplementedError(f"Not supported feature map {feature_map}.")   # the reference's own error, SLA/core.py:75


def _check_geometry(head_dim, blkq, blkk):
    """What the HIP attention / block-map kernels are instantiated for: the reference's class defaults (BLKQ = 64, BLKK = 64,
    SLA/core.py:39) and what its inference scripts construct (BLKQ = 128, BLKK = 64, inference/modify_model.py:50).  The
    reference additionally allows head_dim 64 (SLA/core.py:207); Wan's heads are 128 wide."""
    if head_dim != 128:
        raise ValueError(f"head_dim={head_dim}: the MI355X attention kernels are built for {SUPPORTED}")
    if blkq not in (64, 128) or blkk != 64:
        raise ValueError(f"BLKQ={blkq}, BLKK={blkk}: the MI355X attention kernels are built for {SUPPORTED}")


def _general_sla_hld(q, k, vt_src, proj_w, proj_b, topk_ratio, out, o_stride_h, o_stride_l, v_strides, blkq, feature_map):
    """``SparseLinearAttention`` (16-bit QK) outside the inference scripts' configuration: the class default BLKQ = 64
    and / or the elementwise feature maps of the linear branch (SLA/core.py:39,57-64).  Same kernels as the fast path, the
    fusions undone: the linear branch's second pass adds into the attention output (read-modify-write) instead of riding in
    the attention epilogue, and for BLKQ = 64 every 64-row Q block is run as the first half of its own 128-row slot of the
    attention kernel (one LUT row per slot; the other 64 rows are zero queries whose outputs are dropped) — twice the
    attention work, which only the reference's training-side default pays; the block map pools Q over 64 rows."""
    H, L_, D = k.shape
    kb = K.cdiv(L_, 64)
    topk = min(kb, int(topk_ratio * kb))
    if topk < 1:
        raise ValueError(f"block-sparse attention with topk ratio {topk_ratio} selects no block of {kb} (L = {L_} tokens)")
    vt = K.v_transpose(vt_src, v_strides[0], v_strides[1], L_, H, D, q.dtype)
    km = K.seq_mean(k)
    pq, _, _ = K.sage_quant_pool(q, None, blkq, want_quant=False)
    pk, _, _ = K.sage_quant_pool(k, km, 64, want_quant=False)
    lut = K.sla_topk(pq, pk, topk)
    if blkq == 128:
        K.attn_16(q, k, vt, lut, out, o_stride_h, o_stride_l)
    else:
        qb = K.cdiv(L_, 64)
        full, tail = L_ // 64, L_ % 64
        qpad = torch.zeros((H, qb, 128, D), dtype=q.dtype, device=q.device)
        qpad[:, :full, :64] = q[:, :full * 64].view(H, full, 64, D)
        if tail:
            qpad[:, full, :tail] = q[:, full * 64:]
        opad = torch.empty((qb * 128, H, D), dtype=q.dtype, device=q.device)
        K.attn_16(qpad.view(H, qb * 128, D), k, vt, lut, opad, D, H * D, lk=L_)
        rows = opad.view(qb, 128, H, D)[:, :64].reshape(qb * 64, H, D)[:L_]                  # [L, H, D]
        dst = torch.as_strided(out, (L_, H, D), (o_stride_l, o_stride_h, 1))
        dst.copy_(rows)
    if proj_w is not None:
        kv_t, ksum = K.sla_linear_kv(k, vt, feature_map=feature_map)
        K.sla_linear_out_(q, kv_t, ksum, proj_w, proj_b, out, o_stride_h, o_stride_l, feature_map=feature_map)
    return out, topk, kb


def sparse_linear_attention_hld(q, k, vt_src, proj_w, proj_b, topk_ratio, sage, out, o_stride_h, o_stride_l,
                                v_strides, blkq=128, blkk=64, dense=False, quant_out=False, km=None, pv="fp16", vt=None,
                                side=None, q_fn=None, feature_map="softmax"):
    """Core of both modules on head-major tensors.

    q, k: [H, L, D] 16-bit (after RoPE); vt_src: tensor holding V with element (h,l,d) at
    data_ptr + h*v_strides[0] + l*v_strides[1] + d; out: preallocated, element (h,l,d) at
    out_ptr + h*o_stride_h + l*o_stride_l + d.  Returns (out, real_topk, Kb).
    km: the per-head sequence mean of k [H, D] when the caller already has it (K.qk_norm_rope_pair), else computed here.
    quant_out: return the [L, H*D] result block-quantised for the o projection ((int8, scales) in place of ``out``,
    which then only supplies the dtype).
    side, q_fn: two-stream schedule (SageSLA, FP16 PV): ``q`` is None and ``q_fn()`` produces it; the Q-side chain
    (q_fn -> Sage quant/pool of q -> pass 2 of the linear branch) is enqueued on the stream ``side`` while the K-side chain
    (linear-branch pass 1 + smooth-K mean -> Sage quant/pool of k) runs on the current stream; they meet at the block map and
    at the attention kernel.  The Q-side kernels are HBM-bound, pass 1 / pass 2 of the linear branch instruction-bound: run
    side by side they share the CUs instead of queueing.  Same kernels, same arguments: bit-identical.
    vt: the V^T tiles when the caller already has them (K.gemm_w8a8_vt: the q|k|v GEMM's epilogue wrote them) — vt_src is
    then not read (pv = "fp16" only).
    pv: "fp16" (the reference's sm80 branch, SLA/core.py:211-216) or "fp8" (its sm89+ branch, :217-239: V as per-channel
    scaled e4m3, P rounded to e4m3, P.V on the fp8 MFMA) — Sage only.

    The linear branch (needs only q, k, v) runs FIRST and leaves o_l in a lane-private layout; the attention kernel adds
    it in its epilogue (o = o_s + o_l, the 16-bit add of SLA/core.py:253) — no read-modify-write pass over the output.
    """
    H, L_, D = k.shape
    _check_geometry(D, blkq, blkk)
    if blkq != 128 or feature_map != "softmax":
        if sage and blkq != 128:
            raise ValueError("SageAttention quantises Q per 128 rows (SLA/core.py:202): BLKQ = 128")
        if not sage and not dense and not quant_out and vt is None:
            return _general_sla_hld(q if q is not None else q_fn(), k, vt_src, proj_w, proj_b, topk_ratio, out, o_stride_h,
                                    o_stride_l, v_strides, blkq, feature_map)
        if blkq != 128:
            # every remaining combination would land on the BLKQ = 128 fast path with a LUT of one row per 64 queries
            raise ValueError("BLKQ = 64 (the class default, SLA/core.py:39) runs through the general 16-bit SLA path only: no dense "
                             "attention, no fused INT8 output (quant_out), no caller-supplied V^T tiles (vt)")
        return _other_feature_map(q if q is not None else q_fn(), k, vt_src, proj_w, proj_b, topk_ratio, sage, out, o_stride_h,
                                  o_stride_l, v_strides, dense, quant_out, km, pv, vt, feature_map)
    kb = K.cdiv(L_, blkk)
    topk = min(kb, int(topk_ratio * kb))
    if not dense and topk < 1:
        # SLA/utils.py:61-62 would select zero blocks here (and divide 0 by 0 downstream): refuse instead
        raise ValueError(f"block-sparse attention with topk ratio {topk_ratio} selects no block of {kb} "
                         f"(L = {L_} tokens): use a longer sequence or a larger ratio")
    if side is not None and q_fn is not None:
        if sage and not dense and proj_w is not None and pv == "fp16" and vt is not None and km is None:
            return _sagesla_two_streams(q_fn, k, vt, proj_w, proj_b, topk, kb, out, o_stride_h, o_stride_l, blkq, blkk,
                                        quant_out, side)
        q = q_fn()
    pdt = torch.float16 if sage else q.dtype
    if vt is None:
        vt = K.v_transpose(vt_src, v_strides[0], v_strides[1], L_, H, D, pdt)
    else:
        assert vt.dtype == pdt and tuple(vt.shape) == (H, kb, D, 64) and not (sage and pv == "fp8")
    o_l = None
    if proj_w is not None:
        # the linear branch's pass over K also accumulates the smooth-K mean (k.mean(dim=-2), SLA/core.py:197): +6 us on a 68-us
        # pass, where a pass of its own costs 20 (td_seq_mean) and column sums riding on k's norm + RoPE pass cost 23 — and the K
        # quantiser that waits for it is not on the critical path (norm + RoPE of k -> this pass -> pass 2 -> attention is);
        # round 5 tried both and a three-branch schedule around them: measured, no gain over this form
        # (profiles/r05_vs_r04_same_box_*.txt, profiles/NOTES_r05.md)
        if km is None:
            kv_t, ksum, km = K.sla_linear_kv(k, vt, want_kmean=True)
        else:
            kv_t, ksum = K.sla_linear_kv(k, vt)
        o_l = K.sla_linear_out_t(q, kv_t, ksum, proj_w, proj_b)
    elif km is None and (sage or not dense):
        km = K.seq_mean(k)
    if dense:
        lut = None
        pq = None
    if sage:
        pq, q_i8, q_s = K.sage_quant_pool(q, None, blkq, want_pool=not dense)
        pk, k_i8, k_s = K.sage_quant_pool(k, km, blkk, want_pool=not dense)
        if not dense:
            lut = K.sla_topk(pq, pk, topk)
        if pv == "fp8":
            vt8, v_scale = K.v_fp8_tiles(vt_src, v_strides[0], v_strides[1], L_, H, D, 2.25)
            res = K.attn_i8(q_i8, q_s, k_i8, k_s, vt8, lut, out, o_stride_h, o_stride_l, add_t=o_l, quant_out=quant_out,
                            v_scale=v_scale)
        else:
            res = K.attn_i8(q_i8, q_s, k_i8, k_s, vt, lut, out, o_stride_h, o_stride_l, add_t=o_l, quant_out=quant_out)
    else:
        if not dense:
            pq, _, _ = K.sage_quant_pool(q, None, blkq, want_quant=False)
            pk, _, _ = K.sage_quant_pool(k, km, blkk, want_quant=False)
            lut = K.sla_topk(pq, pk, topk)
        res = K.attn_16(q, k, vt, lut, out, o_stride_h, o_stride_l, add_t=o_l, quant_out=quant_out)
    return res, topk, kb


def _other_feature_map(q, k, vt_src, proj_w, proj_b, topk_ratio, sage, out, o_stride_h, o_stride_l, v_strides, dense, quant_out,
                       km, pv, vt, feature_map):
    """BLKQ = 128 with the elu / relu feature map (SLA/core.py:139-147) outside the general path (SageSLA, or 16-bit SLA with
    caller-supplied V^T tiles): the sparse branch as always — in the CALLER's arithmetic (``sage``) — the linear branch's two
    passes with the other map, added into the output afterwards (no epilogue fusion, no fused quantiser)."""
    if quant_out:
        raise ValueError("the attention epilogue's fused INT8 output needs the softmax feature map's fused linear branch")
    res, topk, kb = sparse_linear_attention_hld(q, k, vt_src, None, None, topk_ratio, sage, out, o_stride_h, o_stride_l,
                                                v_strides, dense=dense, km=km, pv=pv, vt=vt)
    if proj_w is not None and not dense:
        H, L_, D = k.shape
        vtl = vt if (vt is not None and vt.dtype == q.dtype) else K.v_transpose(vt_src, v_strides[0], v_strides[1], L_, H, D, q.dtype)
        kv_t, ksum = K.sla_linear_kv(k, vtl, feature_map=feature_map)
        K.sla_linear_out_(q, kv_t, ksum, proj_w, proj_b, out, o_stride_h, o_stride_l, feature_map=feature_map)
    return res, topk, kb


def _sagesla_two_streams(q_fn, k, vt, proj_w, proj_b, topk, kb, out, o_stride_h, o_stride_l, blkq, blkk, quant_out, side):
    """The SageSLA kernel sequence of ``sparse_linear_attention_hld`` with the Q-side chain on ``side`` (see there).
    Allocation safety without record_stream: tensors made on ``side`` are consumed on the current stream before this
    function's last kernel, and ``side`` only starts new work after waiting for an event of the current stream that is
    recorded later than that kernel (the next call's fork) — and vice versa for kv_t / ksum through ``e_ol``."""
    main = torch.cuda.current_stream()
    e_fork = torch.cuda.Event()
    e_fork.record(main)                      # q's source (the q|k|v projection) is complete
    side.wait_event(e_fork)
    with torch.cuda.stream(side):
        q = q_fn()
        pq, q_i8, q_s = K.sage_quant_pool(q, None, blkq, want_pool=True)
        e_pq = torch.cuda.Event()
        e_pq.record(side)
    kv_t, ksum, km = K.sla_linear_kv(k, vt, want_kmean=True)
    e_kv = torch.cuda.Event()
    e_kv.record(main)
    with torch.cuda.stream(side):
        side.wait_event(e_kv)
        o_l = K.sla_linear_out_t(q, kv_t, ksum, proj_w, proj_b)
        e_ol = torch.cuda.Event()
        e_ol.record(side)
    pk, k_i8, k_s = K.sage_quant_pool(k, km, blkk, want_pool=True)
    main.wait_event(e_pq)
    lut = K.sla_topk(pq, pk, topk)
    main.wait_event(e_ol)
    res = K.attn_i8(q_i8, q_s, k_i8, k_s, vt, lut, out, o_stride_h, o_stride_l, add_t=o_l, quant_out=quant_out)
    return res, topk, kb


def sagesla_split_projection(kv_proj, k_fn, q_proj, q_fn, proj_w, proj_b, L_, topk_ratio, out, o_stride_h, o_stride_l, quant_out,
                             side, blkq=128, blkk=64):
    """SageSLA self-attention of a block whose q|k|v projection is launched as TWO GEMMs — K|V first, Q second — so that the
    K-side chain (norm + RoPE of k, the linear branch's pass over K and V^T with the smooth-K mean, the K quantiser) runs on
    ``side`` UNDER the Q projection instead of after the whole projection (round 5; the chain norm(k) -> linear pass 1 ->
    linear pass 2 in front of the attention kernel was ~270 us of a 2.9-ms block at C1 with nothing beside it but other
    bandwidth-bound passes).  Same kernels on the same data as ``_sagesla_two_streams``: bit-identical.  Measured gain is
    small — +0.7 % on a quiet box, nothing on a noisy one (profiles/r05_split_qkv_ab.txt, r05_split_variants.txt): the Q GEMM
    beside the chain takes 183 us instead of 96 (it is issue-bound; the chain's waves take issue slots on the same SIMDs) and
    the chain 190 instead of 110 (profiles/r05_timeline_n1_split.txt).  Two other orders (Q first with the Q-side chain under
    K|V; only norm / mean / quantiser under the Q GEMM) measured the same to the noise.

    kv_proj() -> vt (and k's source columns written), on the calling stream;  k_fn() -> k [H, L, D] (side);
    q_proj() writes q's source columns (calling stream);  q_fn() -> q [H, L, D] (calling stream).

    Allocation safety without record_stream, as in ``_sagesla_two_streams``: every tensor made on ``side`` is last read by a
    kernel of the calling stream that is enqueued before this function returns (the attention kernel, behind e_kv / e_lut), and
    ``side`` starts its next work only behind the NEXT call's fork event, recorded on the calling stream after that kernel;
    tensors of the calling stream read on ``side`` (vt, q, the projection buffer) are read before e_kv / e_lut are recorded."""
    kb = K.cdiv(L_, blkk)
    topk = min(kb, int(topk_ratio * kb))
    if topk < 1:
        raise ValueError(f"topk ratio {topk_ratio} keeps no key block out of {kb}")
    main = torch.cuda.current_stream()
    vt = kv_proj()
    e_fork = torch.cuda.Event()
    e_fork.record(main)
    side.wait_event(e_fork)
    with torch.cuda.stream(side):
        k = k_fn()
        kv_t, ksum, km = K.sla_linear_kv(k, vt, want_kmean=True)
        e_kv = torch.cuda.Event()
        e_kv.record(side)
        pk, k_i8, k_s = K.sage_quant_pool(k, km, blkk, want_pool=True)
    q_proj()
    q = q_fn()
    e_q = torch.cuda.Event()
    e_q.record(main)
    with torch.cuda.stream(side):
        side.wait_event(e_q)
        pq, q_i8, q_s = K.sage_quant_pool(q, None, blkq, want_pool=True)
        lut = K.sla_topk(pq, pk, topk)
        e_lut = torch.cuda.Event()
        e_lut.record(side)
    main.wait_event(e_kv)
    o_l = K.sla_linear_out_t(q, kv_t, ksum, proj_w, proj_b)
    main.wait_event(e_lut)
    res = K.attn_i8(q_i8, q_s, k_i8, k_s, vt, lut, out, o_stride_h, o_stride_l, add_t=o_l, quant_out=quant_out)
    return res, topk, kb


class _SLABase(nn.Module):
    def __init__(self, head_dim, topk, feature_map, use_bf16, tie_feature_map_qk):
        super().__init__()
        _check_feature_map(feature_map)
        _check_geometry(head_dim, 128, 64)
        self.feature_map = feature_map      # (tie_feature_map_qk: q and k share the map either way, SLA/core.py:77-78)
        self.dtype = torch.bfloat16 if use_bf16 else torch.float16
        self.topk = topk
        self.head_dim = head_dim
        self.proj_l = nn.Linear(head_dim, head_dim, dtype=torch.float32)
        self.init_weights_()

    def init_weights_(self):
        with torch.no_grad():
            nn.init.zeros_(self.proj_l.weight)
            nn.init.zeros_(self.proj_l.bias)

    pv_dtype = "fp16"   # SageSparseLinearAttention: "fp8" selects the reference's sm89+ FP8-PV branch (SLA/core.py:217-239)

    def _forward(self, q, k, v, return_sparsity, sage, blkq, blkk):
        dtype = q.dtype
        B, L_, H, D = q.shape
        outs = []
        real = None
        for b in range(B):  # num_samples is 1 in the reference scripts; batch = outer loop
            qb = q[b].to(self.dtype).transpose(0, 1).contiguous()  # [H, L, D]
            kb_ = k[b].to(self.dtype).transpose(0, 1).contiguous()
            vb = v[b].to(self.dtype).contiguous()                  # [L, H, D]
            out = torch.empty((L_, H, D), dtype=self.dtype, device=q.device)
            _, real, kb_n = sparse_linear_attention_hld(
                qb, kb_, vb, self.proj_l.weight.float().contiguous(), self.proj_l.bias.float().contiguous(),
                self.topk, sage, out, D, H * D, (D, H * D), blkq, blkk, pv=self.pv_dtype if sage else "fp16",
                feature_map=self.feature_map)
            outs.append(out)
        o = torch.stack(outs, dim=0).to(dtype)  # [B, L, H, D]
        if return_sparsity:
            return o, real / kb_n
        return o


class SparseLinearAttention(_SLABase):
    """SLA/core.py:38-119 — block-sparse softmax attention (16-bit QK) + linear branch."""

    def __init__(self, head_dim, topk, feature_map="softmax", BLKQ=64, BLKK=64, use_bf16=True,
                 tie_feature_map_qk=True):
        super().__init__(head_dim, topk, feature_map, use_bf16, tie_feature_map_qk)
        _check_geometry(head_dim, BLKQ, BLKK)
        self.BLKQ = BLKQ
        self.BLKK = BLKK

    def forward(self, q, k, v, return_sparsity=False):
        return self._forward(q, k, v, return_sparsity, sage=False, blkq=self.BLKQ, blkk=self.BLKK)


class SageSparseLinearAttention(_SLABase):
    """SLA/core.py:122-258 — SageAttention INT8-QK / FP16-PV sparse branch + linear branch."""

    def __init__(self, head_dim, topk, feature_map="softmax", use_bf16=True, tie_feature_map_qk=True):
        super().__init__(head_dim, topk, feature_map, use_bf16, tie_feature_map_qk)

    def forward(self, q, k, v, return_sparsity=False):
        return self._forward(q, k, v, return_sparsity, sage=True, blkq=128, blkk=64)
