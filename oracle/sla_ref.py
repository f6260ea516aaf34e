"""CPU restatement of the SLA / SageSLA attention path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  torch-CPU.

Layout convention inside this file: tensors are [B, H, L, D] ("HND") unless
stated otherwise, i.e. *after* the transpose at SLA/core.py:93-95 / :181-183.
"""
from __future__ import annotations

import math
import torch
import torch.nn.functional as F

LOG2E = 1.4426950408889634  # SLA/kernel.py:55


def _cdiv(a, b):
    return (a + b - 1) // b


# --------------------------------------------------------------------------- #
# a11  block map
# --------------------------------------------------------------------------- #
def mean_pool(x, blk):
    """compress_kernel / mean_pool (SLA/utils.py:21-52): per block of ``blk`` rows,
    fp32 sum over the valid rows / valid count, cast back to the input dtype."""
    b, h, l, d = x.shape
    nb = _cdiv(l, blk)
    xf = torch.zeros(b, h, nb * blk, d, dtype=torch.float32)
    xf[:, :, :l] = x.float()
    s = xf.view(b, h, nb, blk, d).sum(dim=3)
    cnt = torch.full((nb,), float(blk))
    cnt[-1] = float(l - (nb - 1) * blk)
    return (s / cnt[None, None, :, None]).to(x.dtype)


def seq_mean(k):
    """torch.mean(k, dim=-2, keepdim=True) in k's dtype (SLA/utils.py:56,
    SLA/core.py:197): fp32 accumulate, one rounding to the dtype."""
    return k.float().mean(dim=-2, keepdim=True).to(k.dtype)


def pooled_scores(q, k, blkq, blkk):
    """SLA/utils.py:56-59: smooth-K (in the input dtype), block means (input dtype),
    pooled score matmul in the input dtype (fp32 accumulate, one rounding)."""
    arg_k = k - seq_mean(k)  # bf16 - bf16 -> bf16
    pq = mean_pool(q, blkq)
    pk = mean_pool(arg_k, blkk)
    score = (pq.float() @ pk.float().transpose(-1, -2)).to(q.dtype)
    return score, pq, pk


def select_topk(score, topk):
    """Deterministic top-k used by both the oracle and the HIP path: the ``topk``
    largest scores per row, ties broken towards the LOWER block index, returned in
    ASCENDING index order.  torch.topk(sorted=False) (SLA/utils.py:63) leaves order
    and tie-breaking unspecified, so this is one legal outcome of it."""
    k_blocks = score.shape[-1]
    s = score.float()
    # stable sort descending: equal scores keep ascending index order
    order = torch.sort(s, dim=-1, descending=True, stable=True).indices[..., :topk]
    return torch.sort(order, dim=-1).values


def get_block_map(q, k, topk_ratio, blkq=128, blkk=64):
    """get_block_map (SLA/utils.py:55-67).  Returns (sparse_map int8 [B,H,Qb,Kb],
    lut int64 [B,H,Qb,topk] ascending, topk)."""
    score, _, _ = pooled_scores(q, k, blkq, blkk)
    kb = score.shape[-1]
    topk = min(kb, int(topk_ratio * kb))
    lut = select_topk(score, topk)
    sparse_map = torch.zeros_like(score, dtype=torch.int8)
    sparse_map.scatter_(-1, lut, 1)
    return sparse_map, lut, topk


# --------------------------------------------------------------------------- #
# a12  block-sparse online-softmax attention, SLA (Triton) arithmetic
# --------------------------------------------------------------------------- #
def sla_sparse_attn(q, k, v, lut, blkq=128, blkk=64, qk_scale=None, p_dtype=None):
    """_attn_fwd (SLA/kernel.py:21-82): per Q block iterate the LUT's K blocks in LUT
    order; qk = (q @ k^T) * (qk_scale * log2e) in fp32 from bf16 operands, tail keys
    masked to -inf (:57-58), online softmax in the exp2 domain (:60-72), P cast to
    V's dtype before P@V (:68), O / l at the end (:74).  Returns O in V's dtype."""
    b, h, l, d = q.shape
    lk = k.shape[2]  # keys may outnumber queries (sequence-parallel shard / cross attention)
    if qk_scale is None:
        qk_scale = d ** -0.5
    p_dtype = p_dtype or v.dtype
    qb_n = _cdiv(l, blkq)
    out = torch.empty(b, h, l, d, dtype=v.dtype)
    sc = qk_scale * LOG2E
    for bi in range(b):
        for hi in range(h):
            for qb in range(qb_n):
                q0, q1 = qb * blkq, min(l, (qb + 1) * blkq)
                qt = q[bi, hi, q0:q1].float()
                m_i = torch.full((q1 - q0,), -float("inf"))
                l_i = torch.zeros(q1 - q0)
                o = torch.zeros(q1 - q0, d)
                for kb in lut[bi, hi, qb].tolist():
                    k0, k1 = kb * blkk, min(lk, (kb + 1) * blkk)
                    s = (qt @ k[bi, hi, k0:k1].float().t()) * sc
                    new_m = torch.maximum(m_i, s.max(dim=1).values)
                    p = torch.exp2(s - new_m[:, None])
                    alpha = torch.exp2(m_i - new_m)
                    o = o * alpha[:, None] + p.to(p_dtype).float() @ v[bi, hi, k0:k1].float()
                    l_i = l_i * alpha + p.sum(dim=1)
                    m_i = new_m
                out[bi, hi, q0:q1] = (o / l_i[:, None]).to(v.dtype)
    return out


# --------------------------------------------------------------------------- #
# a13  SageAttention arithmetic (SpargeAttn; PARITY UNPINNED — see oracle/__init__.py)
# --------------------------------------------------------------------------- #
def quant_per_block_int8(x, blk, km=None):
    """Per-block INT8 quantisation of Q (blk=128) or K (blk=64, after subtracting the
    per-head sequence mean ``km``) as called at SLA/core.py:197-203
    (spas_sage_attn.utils.get_vanilla_qk_quant — not in the tree).  Rule stated by
    this oracle (the SageAttention per-block rule):

        xf    = float(x) - float(km)            (fp32)
        scale = max|xf| / 127 + 1e-7            (fp32, over the block's valid rows)
        q     = trunc(xf / scale + 0.5*sign(xf/scale))   (round half away from zero)

    x: [B,H,L,D] bf16/fp16; returns (int8 [B,H,L,D], scale fp32 [B,H,ceil(L/blk)])."""
    b, h, l, d = x.shape
    nb = _cdiv(l, blk)
    xf = x.float()
    if km is not None:
        xf = xf - km.float()
    pad = torch.zeros(b, h, nb * blk, d, dtype=torch.float32)
    pad[:, :, :l] = xf
    blkv = pad.view(b, h, nb, blk, d)
    scale = blkv.abs().amax(dim=(3, 4)) / torch.tensor(127.0) + torch.tensor(1e-7)
    y = blkv / scale[..., None, None]
    y = y + 0.5 * torch.where(y >= 0, 1.0, -1.0)
    qi = torch.trunc(y).clamp_(-128, 127).to(torch.int8)
    return qi.view(b, h, nb * blk, d)[:, :, :l].contiguous(), scale.contiguous()


def sage_sparse_attn(q_i8, q_s, k_i8, k_s, v, lut, blkq=128, blkk=64, sm_scale=None,
                     out_dtype=torch.bfloat16, pv_dtype=torch.float16, nvalid=None):
    """INT8-QK / FP16-PV block-sparse attention as invoked at SLA/core.py:211-216
    (qk_int8_sv_f16_..._block_sparse_attn, tensor_layout=HND, qk_quant_gran=per-block,
    pv threshold disabled).  Rule stated by this oracle:

        S   = int32(Q_i8 @ K_i8^T)                       exact
        s   = float(S) * (q_scale*k_scale*sm_scale*log2e)  one fp32 multiplier per tile
        online softmax in the exp2 domain over the selected K blocks in ASCENDING
        order, tail keys masked; P rounded to ``pv_dtype`` (fp16), V in fp16,
        P@V accumulated in fp32 (the SpargeAttn sm80 kernel accumulates in fp16 with
        an fp32 flush buffer; fp32 accumulation is the MI355X-native choice — MFMA
        accumulates in fp32 natively); O / l, cast to out_dtype.

    lut: [B,H,Qb,topk] ascending block ids (nvalid: optional [B,H,Qb] counts)."""
    b, h, l, d = q_i8.shape
    lk = k_i8.shape[2]  # keys may outnumber queries (sequence-parallel shard)
    if sm_scale is None:
        sm_scale = 1.0 / math.sqrt(d)
    qb_n = _cdiv(l, blkq)
    out = torch.empty(b, h, l, d, dtype=out_dtype)
    vf = v.to(pv_dtype).float()
    qf, kf = q_i8.float(), k_i8.float()
    c = torch.tensor(sm_scale * LOG2E, dtype=torch.float32)
    for bi in range(b):
        for hi in range(h):
            for qb in range(qb_n):
                q0, q1 = qb * blkq, min(l, (qb + 1) * blkq)
                m_i = torch.full((q1 - q0,), -float("inf"))
                l_i = torch.zeros(q1 - q0)
                o = torch.zeros(q1 - q0, d)
                sel = lut[bi, hi, qb].tolist()
                if nvalid is not None:
                    sel = sel[: int(nvalid[bi, hi, qb])]
                for kb in sel:
                    k0, k1 = kb * blkk, min(lk, (kb + 1) * blkk)
                    mult = (q_s[bi, hi, qb] * k_s[bi, hi, kb]) * c  # fp32
                    s = (qf[bi, hi, q0:q1] @ kf[bi, hi, k0:k1].t()) * mult
                    new_m = torch.maximum(m_i, s.max(dim=1).values)
                    p = torch.exp2(s - new_m[:, None])
                    alpha = torch.exp2(m_i - new_m)
                    o = o * alpha[:, None] + p.to(pv_dtype).float() @ vf[bi, hi, k0:k1]
                    l_i = l_i * alpha + p.sum(dim=1)
                    m_i = new_m
                out[bi, hi, q0:q1] = (o / l_i[:, None]).to(out_dtype)
    return out


def dense_lut(b, h, l, blkq, blkk):
    qb, kb = _cdiv(l, blkq), _cdiv(l, blkk)
    return torch.arange(kb).expand(b, h, qb, kb).contiguous()


# --------------------------------------------------------------------------- #
# a14  linear-attention branch
# --------------------------------------------------------------------------- #
def linear_branch(q, k, v, proj_w, proj_b, dtype=torch.bfloat16):
    """SLA/core.py:104-113 / :243-252, feature_map='softmax':
        cq = softmax_D(q).to(dtype); ck = softmax_D(k).to(dtype)
        kvsum = ck^T @ v ; ksum = sum_L ck           (dtype matmul / dtype sum)
        o_l = (cq @ kvsum) / (1e-5 + sum_D(cq * ksum))
        o_l = proj_l(o_l)   under autocast(dtype): fp32 Linear evaluated in dtype
    q,k,v: [B,H,L,D] in ``dtype``.  Returns o_l in ``dtype``."""
    cq = F.softmax(q, dim=-1).contiguous().to(dtype)
    ck = F.softmax(k, dim=-1).contiguous().to(dtype)
    kvsum = (ck.float().transpose(-1, -2) @ v.float()).to(dtype)
    ksum = ck.float().sum(dim=-2, keepdim=True).to(dtype)
    num = (cq.float() @ kvsum.float()).to(dtype)
    den = (1e-5 + (cq * ksum).float().sum(dim=-1, keepdim=True).to(dtype))
    o_l = num / den  # dtype / dtype -> dtype
    o_l = (o_l.float() @ proj_w.to(dtype).float().t()).to(dtype) + proj_b.to(dtype)
    # F.linear under autocast computes addmm in dtype: one rounding after bias add
    return o_l


def linear_branch_exact_autocast(q, k, v, proj_w, proj_b, dtype=torch.bfloat16):
    """Same as linear_branch but letting torch do the dtype arithmetic itself (used to
    cross-check the restatement above on CPU)."""
    cq = F.softmax(q, dim=-1).contiguous().to(dtype)
    ck = F.softmax(k, dim=-1).contiguous().to(dtype)
    kvsum = ck.transpose(-1, -2) @ v
    ksum = torch.sum(ck, dim=-2, keepdim=True)
    o_l = (cq @ kvsum) / (1e-5 + (cq * ksum).sum(dim=-1, keepdim=True))
    with torch.amp.autocast("cpu", dtype=dtype):
        o_l = F.linear(o_l, proj_w, proj_b)
    return o_l


# --------------------------------------------------------------------------- #
# module-level compositions (inputs/outputs [B, L, H, D] like the reference modules)
# --------------------------------------------------------------------------- #
def sla_forward(q, k, v, proj_w, proj_b, topk, blkq=128, blkk=64, dtype=torch.bfloat16, lut=None):
    """SparseLinearAttention.forward (SLA/core.py:83-119).  ``lut``: visit the selected blocks in this order instead of
    ascending (the reference hands the Triton kernel torch.topk's unsorted order, SLA/utils.py:63; the online softmax
    is order-dependent at rounding level — SURVEY §8c caveat iv)."""
    in_dtype = q.dtype
    q = q.transpose(1, 2).contiguous()
    k = k.transpose(1, 2).contiguous()
    v = v.transpose(1, 2).contiguous()
    if lut is None:
        _, lut, _ = get_block_map(q, k, topk, blkq, blkk)
    q, k, v = q.to(dtype), k.to(dtype), v.to(dtype)
    o_s = sla_sparse_attn(q, k, v, lut, blkq, blkk)
    o_l = linear_branch_exact_autocast(q, k, v, proj_w, proj_b, dtype)
    return (o_s + o_l).to(in_dtype).transpose(1, 2)


def sagesla_forward(q, k, v, proj_w, proj_b, topk, dtype=torch.bfloat16, blkq=128, blkk=64):
    """SageSparseLinearAttention.forward (SLA/core.py:168-258), sm80 (FP16-PV) branch."""
    in_dtype = q.dtype
    q = q.transpose(1, 2).contiguous()
    k = k.transpose(1, 2).contiguous()
    v = v.transpose(1, 2).contiguous()
    _, lut, _ = get_block_map(q, k, topk, blkq, blkk)
    q, k, v = q.to(dtype), k.to(dtype), v.to(dtype)
    km = seq_mean(k)
    q_i8, q_s = quant_per_block_int8(q, blkq)
    k_i8, k_s = quant_per_block_int8(k, blkk, km)
    o_s = sage_sparse_attn(q_i8, q_s, k_i8, k_s, v, lut, blkq, blkk, out_dtype=dtype)
    o_l = linear_branch_exact_autocast(q, k, v, proj_w, proj_b, dtype)
    return (o_s + o_l).to(in_dtype).transpose(1, 2)


def sdpa_ref(q, k, v, scale=None):
    """fp32 softmax(QK^T/sqrt(D))V  — the 'original' dense path's maths
    (rcm/utils/attention.py:120-167 -> F.scaled_dot_product_attention)."""
    qf, kf, vf = q.float(), k.float(), v.float()
    d = q.shape[-1]
    s = (qf @ kf.transpose(-1, -2)) * (scale if scale is not None else d ** -0.5)
    return torch.softmax(s, dim=-1) @ vf


# --------------------------------------------------------------------------- #
# SpargeAttn host-side helpers as called by SageSparseLinearAttention.forward (PARITY UNPINNED: the package is not in
# the tree; stated from its call sites SLA/core.py:201-204,221-227)
# --------------------------------------------------------------------------- #
def block_map_lut(sparse_map):
    """block_map_lut_triton(sparse_map) (SLA/core.py:204): per Q block the selected K-block ids in ascending order,
    stored as DELTAS (lut[0] = first id, lut[i] = id_i - id_{i-1}; the kernel walks ``blk += lut[i]``), padded with
    zeros, plus the number of valid entries.  sparse_map [B,H,Qb,Kb] int8 -> (lut int32 [B,H,Qb,Kb], valid int32 [B,H,Qb])."""
    b, h, qb, kb = sparse_map.shape
    valid = sparse_map.to(torch.int32).sum(-1)
    # ascending ids of the selected blocks, unselected ones pushed to the end
    key = torch.where(sparse_map > 0, torch.arange(kb).expand_as(sparse_map), torch.full_like(sparse_map, kb, dtype=torch.long))
    ids = torch.sort(key, dim=-1).values
    prev = torch.cat([torch.zeros_like(ids[..., :1]), ids[..., :-1]], -1)
    delta = ids - prev
    pos = torch.arange(kb).expand_as(ids)
    delta = torch.where(pos < valid[..., None], delta, torch.zeros_like(delta))
    return delta.to(torch.int32), valid


def lut_from_delta(lut_delta):
    """Inverse of the delta encoding: absolute ascending block ids (entries past ``valid`` repeat the last id)."""
    return torch.cumsum(lut_delta.to(torch.int64), dim=-1)


# --------------------------------------------------------------------------- #
# a13, FP8-PV variant (the reference's sm89+ path, SLA/core.py:217-239; SpargeAttn kernels — PARITY UNPINNED)
# --------------------------------------------------------------------------- #
FP8_MAX = 448.0           # largest finite e4m3fn
P_FP8_OFFSET = 448.0      # P in [0,1] is scaled by 448 before the e4m3 conversion (SageAttention2's exp2 offset log2(448))


def fp8_e4m3(x):
    """fp32 -> OCP e4m3fn (round to nearest even, saturate to +-448) -> fp32 value."""
    return x.clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).float()


def transpose_pad_permute(v, vt):
    """fused.transpose_pad_permute_cuda(v, vt, 1) (SLA/core.py:221): vt[b,h,d,:kv_len] = v[b,h,:,d]^T, zero padded to a
    multiple of 128 keys.  (The CUDA routine also permutes keys inside 16-groups for its MMA operand layout; that is a
    storage detail of that kernel, not arithmetic, and is not modelled.)"""
    kv = v.shape[2]
    vt.zero_()
    vt[..., :kv] = v.transpose(-1, -2)


def v_fp8_quant(vt, kv_len, scale_max=2.25):
    """fused.scale_fuse_quant_cuda(vt, v_fp8, v_scale, kv_len, 2.25, 1) (SLA/core.py:224): per (b,h,d) channel
    scale = max_{keys < kv_len} |v| / scale_max (fp32), v_fp8 = e4m3(v / scale).  scale_max = 2.25 keeps
    448 (P) * 2.25 (V) * 64 keys = 64512 inside an fp16 accumulator.  Returns (v_fp8 [b,h,d,Lpad] float8_e4m3fn,
    v_scale fp32 [b,h,d])."""
    vf = vt.float()
    amax = vf[..., :kv_len].abs().amax(dim=-1)
    scale = amax / torch.tensor(scale_max)
    q = (vf / scale.clamp_min(1e-30)[..., None]).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn)
    return q, scale


def sage_sparse_attn_fp8(q_i8, q_s, k_i8, k_s, v_fp8_t, v_scale, lut, blkq=128, blkk=64, sm_scale=None,
                         out_dtype=torch.bfloat16, nvalid=None):
    """INT8-QK / FP8-PV block-sparse attention as invoked at SLA/core.py:227-239
    (qk_int8_sv_f8_accum_f32|f16_block_sparse_attn_inst_buf_fuse_v_scale_with_pv_threshold).  Rule stated by this oracle:

        s   as in ``sage_sparse_attn``; online softmax in the exp2 domain, ascending selected blocks, tail keys masked;
        P8  = e4m3(448 * exp2(s - m))          (the row sum l accumulates the un-rounded fp32 P)
        O  += P8 @ V8^T                        (fp32 accumulate; V8 = the per-channel-scaled e4m3 values)
        out = O * v_scale[d] / (448 * l)       (v scale fused in the epilogue), cast to out_dtype

    v_fp8_t: [B,H,D,Lpad] float8_e4m3fn, v_scale fp32 [B,H,D]."""
    b, h, l, d = q_i8.shape
    lk = k_i8.shape[2]
    if sm_scale is None:
        sm_scale = 1.0 / math.sqrt(d)
    qb_n = _cdiv(l, blkq)
    out = torch.empty(b, h, l, d, dtype=out_dtype)
    vf = v_fp8_t.float().transpose(-1, -2)  # [B,H,Lpad,D]
    qf, kf = q_i8.float(), k_i8.float()
    c = torch.tensor(sm_scale * LOG2E, dtype=torch.float32)
    for bi in range(b):
        for hi in range(h):
            for qb in range(qb_n):
                q0, q1 = qb * blkq, min(l, (qb + 1) * blkq)
                m_i = torch.full((q1 - q0,), -float("inf"))
                l_i = torch.zeros(q1 - q0)
                o = torch.zeros(q1 - q0, d)
                sel = lut[bi, hi, qb].tolist()
                if nvalid is not None:
                    sel = sel[: int(nvalid[bi, hi, qb])]
                for kb in sel:
                    k0, k1 = kb * blkk, min(lk, (kb + 1) * blkk)
                    mult = (q_s[bi, hi, qb] * k_s[bi, hi, kb]) * c
                    s = (qf[bi, hi, q0:q1] @ kf[bi, hi, k0:k1].t()) * mult
                    new_m = torch.maximum(m_i, s.max(dim=1).values)
                    p = torch.exp2(s - new_m[:, None])
                    alpha = torch.exp2(m_i - new_m)
                    o = o * alpha[:, None] + fp8_e4m3(p * P_FP8_OFFSET) @ vf[bi, hi, k0:k1]
                    l_i = l_i * alpha + p.sum(dim=1)
                    m_i = new_m
                out[bi, hi, q0:q1] = (o * v_scale[bi, hi][None, :] / (P_FP8_OFFSET * l_i[:, None])).to(out_dtype)
    return out


def sagesla_forward_fp8(q, k, v, proj_w, proj_b, topk, dtype=torch.bfloat16, blkq=128, blkk=64):
    """SageSparseLinearAttention.forward (SLA/core.py:168-258), sm89+ (FP8-PV) branch."""
    in_dtype = q.dtype
    q = q.transpose(1, 2).contiguous()
    k = k.transpose(1, 2).contiguous()
    v = v.transpose(1, 2).contiguous()
    _, lut, _ = get_block_map(q, k, topk, blkq, blkk)
    q, k, v = q.to(dtype), k.to(dtype), v.to(dtype)
    km = seq_mean(k)
    q_i8, q_s = quant_per_block_int8(q, blkq)
    k_i8, k_s = quant_per_block_int8(k, blkk, km)
    b, h, l, d = v.shape
    lpad = _cdiv(l, 128) * 128
    vt = torch.empty(b, h, d, lpad, dtype=v.dtype)
    transpose_pad_permute(v, vt)
    v8, vs = v_fp8_quant(vt, l, 2.25)
    o_s = sage_sparse_attn_fp8(q_i8, q_s, k_i8, k_s, v8, vs, lut, blkq, blkk, out_dtype=dtype)
    o_l = linear_branch_exact_autocast(q, k, v, proj_w, proj_b, dtype)
    return (o_s + o_l).to(in_dtype).transpose(1, 2)
