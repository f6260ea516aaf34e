"""CPU stand-in for ``turbodiffusion_amd.kernels`` built from the oracle — TEST ONLY.

Lets the world_size-2 gloo tests drive ``turbodiffusion_amd.seqpar.SeqParallel`` (sharding, packing,
all-gathers, rank-padded layouts) on CPU.  Same call signatures and memory layouts as the HIP wrappers.
"""
import math

import torch
import torch.nn.functional as F

from oracle import sla_ref as S

_PERM = torch.tensor([0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15])
_POS2KEY = (torch.arange(4)[:, None] * 16 + _PERM[None, :]).reshape(-1)  # position -> key in a 64-block


def _strided(t, shape, strides):
    return torch.as_strided(t, shape, strides, t.storage_offset())


def seq_sum_partial(k, out=None):
    H, L, D = k.shape
    ws = torch.zeros(H, 64, D)
    rows = math.ceil(L / 64)
    for c in range(64):
        ws[:, c] = k[:, c * rows:(c + 1) * rows].float().sum(1)
    return ws


def seq_sum(k, out=None):
    """kernels.seq_sum: this rank's per-head column sums f32 [H, D] (chunk partials added in order)."""
    ws = seq_sum_partial(k)
    s = torch.zeros(k.shape[0], k.shape[2])
    for c in range(64):
        s = s + ws[:, c]
    if out is not None:
        out.copy_(s)
        return out
    return s


def seq_mean_final(ws, nch, stride_h, stride_c, L_total, H, D, dtype):
    v = _strided(ws, (H, nch, D), (stride_h, stride_c, 1))
    s = torch.zeros(H, D)
    for c in range(nch):
        s = s + v[:, c]
    return (s / float(L_total)).to(dtype)


def v_transpose(v, stride_h, stride_l, L, H, D, out_dtype):
    vv = _strided(v, (H, L, D), (stride_h, stride_l, 1))
    kb = (L + 63) // 64
    pad = torch.zeros(H, kb * 64, D, dtype=out_dtype)
    pad[:, :L] = vv.to(out_dtype)
    return pad.view(H, kb, 64, D)[:, :, _POS2KEY, :].permute(0, 1, 3, 2).contiguous()


def _v_from_tiles(vt, L):
    H, kb, D, _ = vt.shape
    inv = torch.empty(64, dtype=torch.long)
    inv[_POS2KEY] = torch.arange(64)
    v = vt.permute(0, 1, 3, 2)[:, :, inv, :].reshape(H, kb * 64, D)
    return v[:, :L]


def sage_quant_pool(x, km, blk, want_pool=True, want_quant=True):
    xb = x[None]
    kmb = None if km is None else km[None, :, None, :]
    pooled = xq = xs = None
    if want_pool:
        pooled = S.mean_pool(xb if kmb is None else xb - kmb, blk)[0]
    if want_quant:
        q, s = S.quant_per_block_int8(xb, blk, kmb)
        xq, xs = q[0], s[0]
    return pooled, xq, xs


def sla_topk(pq, pk, topk, kb=None):
    kb = pk.shape[1] if kb is None else kb
    score = (pq.float() @ pk[:, :kb].float().transpose(-1, -2)).to(pq.dtype)
    return S.select_topk(score, topk).int()


def _write(out, o_stride_h, o_stride_l, val):
    H, L, D = val.shape
    _strided(out, (H, L, D), (o_stride_h, o_stride_l, 1)).copy_(val)


def _finish(o, out, o_stride_h, o_stride_l, add_t, quant_out):
    """o [H, L, D] 16-bit (+ o_l, the 16-bit add of SLA/core.py:253) -> strided ``out`` or its block-quantised form."""
    from oracle import ops_ref as O
    if add_t is not None:
        o = o + add_t
    if quant_out:
        H, L, D = o.shape
        return O.quant_block128(o.permute(1, 0, 2).reshape(L, H * D).contiguous())
    _write(out, o_stride_h, o_stride_l, o)
    return out


def attn_i8(q_i8, q_s, k_i8, k_s, vt, lut, out, o_stride_h, o_stride_l, sm_scale=None, lk=None, add_t=None,
            quant_out=False):
    H, L, D = q_i8.shape
    lk = k_i8.shape[1] if lk is None else lk
    kb = (lk + 63) // 64
    v = _v_from_tiles(vt, lk)
    lutb = S.dense_lut(1, H, L, 128, 64)[..., :kb] if lut is None else lut[None].long()
    if lut is None:
        lutb = torch.arange(kb).expand(1, H, (L + 127) // 128, kb)
    odt = out if isinstance(out, torch.dtype) else out.dtype
    o = S.sage_sparse_attn(q_i8[None], q_s[None], k_i8[None, :, :lk], k_s[None, :, :kb], v[None].float(), lutb,
                           out_dtype=odt)[0]
    return _finish(o, out, o_stride_h, o_stride_l, add_t, quant_out)


def attn_16(q, k, vt, lut, out, o_stride_h, o_stride_l, sm_scale=None, lk=None, add_t=None, quant_out=False):
    H, L, D = q.shape
    lk = k.shape[1] if lk is None else lk
    kb = (lk + 63) // 64
    v = _v_from_tiles(vt, lk)
    lutb = torch.arange(kb).expand(1, H, (L + 127) // 128, kb) if lut is None else lut[None].long()
    o = S.sla_sparse_attn(q[None], k[None, :, :lk], v[None], lutb, 128, 64)[0]
    return _finish(o, out, o_stride_h, o_stride_l, add_t, quant_out)


def sla_linear_kv_partial_f32(k, vt, kv_out=None, ks_out=None):
    H, L, D = k.shape
    v = _v_from_tiles(vt, L).float()
    ck = F.softmax(k, dim=-1).to(k.dtype).float()
    return ck.transpose(-1, -2) @ v, ck.sum(-2)


def sla_linear_kv_final(kv_parts, ks_parts, nch, kv_sh, kv_sc, ks_sh, ks_sc, H, D, dtype):
    kv = _strided(kv_parts, (H, nch, D, D), (kv_sh, kv_sc, D, 1))
    ks = _strided(ks_parts, (H, nch, D), (ks_sh, ks_sc, 1))
    kvs, kss = torch.zeros(H, D, D), torch.zeros(H, D)
    for c in range(nch):
        kvs, kss = kvs + kv[:, c], kss + ks[:, c]
    return kvs.to(dtype).transpose(-1, -2).contiguous(), kss.to(dtype)


def sla_linear_kv(k, vt, want_kmean=False):
    kv, ks = sla_linear_kv_partial_f32(k, vt)
    out = (kv.to(k.dtype).transpose(-1, -2).contiguous(), ks.to(k.dtype))
    return out + (seq_mean(k),) if want_kmean else out


def sla_linear_out_t(q, kv_t, ksum, wp, bp):
    """o_l [H, L, D] in q's dtype (the HIP wrapper returns it in a lane-private layout; here the plain tensor —
    only ``attn_*(add_t=...)`` of this module consumes it)."""
    dt = q.dtype
    cq = F.softmax(q, dim=-1).contiguous().to(dt)
    kvsum = kv_t.transpose(-1, -2).contiguous()
    o_l = (cq @ kvsum) / (1e-5 + (cq * ksum[:, None, :]).sum(dim=-1, keepdim=True))
    with torch.amp.autocast("cpu", dtype=dt):
        o_l = F.linear(o_l, wp, bp)
    return o_l.to(dt)


def sla_linear_out_(q, kv_t, ksum, wp, bp, out, o_stride_h, o_stride_l):
    H, L, D = q.shape
    o_l = sla_linear_out_t(q, kv_t, ksum, wp, bp)
    view = _strided(out, (H, L, D), (o_stride_h, o_stride_l, 1))
    view.copy_(view + o_l)
    return out


def seq_mean(k):
    return S.seq_mean(k[None])[0, :, 0]


# ---- rank-major gathered K side (the *_sp entry points): the CPU stand-in re-lays it out and calls the flat versions
def _seq_major(t):  # [W, H, n, ...] -> [H, W*n, ...]
    W = t.shape[0]
    return t.permute(1, 0, 2, *range(3, t.dim())).reshape(t.shape[1], W * t.shape[2], *t.shape[3:]).contiguous()


def sla_topk_sp(pq, pk_g, topk, kb):
    return sla_topk(pq, _seq_major(pk_g), topk, kb=kb)


def _group_quant(o, quant_out):
    """the head group's [H, L, D] result block-quantised into its columns / scale entries of the whole-row outputs."""
    from oracle import ops_ref as O
    oq, os_, h0, Ht = quant_out
    H, L, D = o.shape
    q, sc = O.quant_block128(o.permute(1, 0, 2).reshape(L, H * D).contiguous())
    oq[:, h0 * D:(h0 + H) * D] = q
    os_[:, h0:h0 + H] = sc


def attn_i8_sp(q_i8, q_s, k_g, ks_g, vt_g, lut, out, o_stride_h, o_stride_l, lk, sm_scale=None, add_t=None, quant_out=None):
    if quant_out is None:
        return attn_i8(q_i8, q_s, _seq_major(k_g), _seq_major(ks_g), _seq_major(vt_g), lut, out, o_stride_h, o_stride_l,
                       sm_scale=sm_scale, lk=lk, add_t=add_t)
    H, L, D = q_i8.shape
    tmp = torch.empty((H, L, D), dtype=out if isinstance(out, torch.dtype) else out.dtype)
    attn_i8(q_i8, q_s, _seq_major(k_g), _seq_major(ks_g), _seq_major(vt_g), lut, tmp, L * D, D, sm_scale=sm_scale, lk=lk, add_t=add_t)
    _group_quant(tmp, quant_out)
    return out


def attn_16_sp(q, k_g, vt_g, lut, out, o_stride_h, o_stride_l, lk, sm_scale=None, add_t=None, quant_out=None):
    if quant_out is None:
        return attn_16(q, _seq_major(k_g), _seq_major(vt_g), lut, out, o_stride_h, o_stride_l, sm_scale=sm_scale, lk=lk,
                       add_t=add_t)
    H, L, D = q.shape
    tmp = torch.empty((H, L, D), dtype=q.dtype)
    attn_16(q, _seq_major(k_g), _seq_major(vt_g), lut, tmp, L * D, D, sm_scale=sm_scale, lk=lk, add_t=add_t)
    _group_quant(tmp, quant_out)
    return out


def sp_pack_begin(k, v_src, v_strides, L_loc, lay, lin_kv=None, lin_ks=None):
    """CPU stand-in of kernels.sp_pack_begin: V^T tiles into the send buffer, the linear-branch partials into the early buffer."""
    H, _, D = k.shape
    kb_loc = -(-L_loc // 64)
    pack = torch.zeros((lay.total,), dtype=torch.uint8)
    vt = v_transpose(v_src, v_strides[0], v_strides[1], L_loc, H, D, lay.pdt)
    lay.group_section(pack, "vt")[:, :, :kb_loc] = vt.reshape((lay.G, lay.hg) + tuple(vt.shape[1:]))
    if lay.linear:
        kv32, ks32 = sla_linear_kv_partial_f32(k, vt)
        lin_kv.copy_(kv32)
        lin_ks.copy_(ks32)
    return pack


def sp_pack_finish(pack, k, km, L_loc, lay):
    """CPU stand-in of kernels.sp_pack_finish.  km: [H, D] or (gathered column sums f32 [W, H, D], L_total)."""
    H, _, D = k.shape
    if isinstance(km, tuple):
        allp, L_tot = km
        km = seq_mean_final(allp, allp.shape[0], D, allp.stride(0), L_tot, H, D, k.dtype)
    kb_loc = -(-L_loc // 64)

    def put(name, t, n):   # t [H, n_valid, ...] -> group section [G, hg, n_alloc, ...][:, :, :n]
        lay.group_section(pack, name)[:, :, :n] = t.reshape((lay.G, lay.hg) + tuple(t.shape[1:]))

    if lay.sage:
        pk, k_q, k_s = sage_quant_pool(k, km, 64, want_pool=not lay.dense)
        put("k", k_q, L_loc)
        put("ks", k_s, kb_loc)
    else:
        put("k", k, L_loc)
        pk = sage_quant_pool(k, km, 64, want_quant=False)[0] if not lay.dense else None
    if not lay.dense:
        lay.all_section(pack, "pk")[:, :kb_loc] = pk
    return pack
