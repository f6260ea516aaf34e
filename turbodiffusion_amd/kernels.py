"""Functional wrappers: one Python function per C-ABI entry point.

They allocate outputs with torch (device memory plumbing), pass raw pointers + the current
HIP stream to libturbodiffusion_amd.so and return torch tensors.  Shapes/dtypes are checked
here so that errors read like the reference's asserts; the library re-checks and returns a
status for everything it cannot run.
"""
from __future__ import annotations

import math
from typing import Tuple

import torch

from . import _lib as L
from ._lib import call as _call, dt_code, ptr, require_gpu, stream_ptr


def cdiv(a: int, b: int) -> int:
    return (a + b - 1) // b


def _f32c(t, what):
    """The kernels read these as contiguous fp32 (norm weights, RoPE tables, block scales): a buffer that a stray
    ``net.to(torch.bfloat16)`` converted would be read as garbage past its end — refuse instead."""
    if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
        raise TypeError(f"{what} must be a contiguous float32 tensor (got {t.dtype}, contiguous={t.is_contiguous()})")
    return t


class KernelTimer:
    """Times selected entry points with HIP events recorded on the stream the kernel is launched on
    (torch's current stream).  Used by bench.py for the roofline numbers; off by default."""

    def __init__(self, names, entries=()):
        self.names = set(names)
        self.entries = set(entries)   # C-ABI entry points timed by name, whatever wrapper calls them (round 6: the HBM-bound family)
        self.records = {}  # name -> list of (start_event, end_event, meta)

    def summary(self):
        out = {}
        for name, recs in self.records.items():
            ms = [a.elapsed_time(b) for a, b, _ in recs]
            out[name] = {"launches": len(ms), "avg_ms": sum(ms) / max(1, len(ms)), "total_ms": sum(ms),
                         "metas": [m for _, _, m in recs]}
        return out


_timer = None


def set_timer(t):
    global _timer
    _timer = t


def call(name, *args):
    """_lib.call; with a KernelTimer installed whose ``entries`` name this entry point, between two HIP events on the launch stream."""
    t = _timer
    if t is None or name not in t.entries:
        return _call(name, *args)
    st = torch.cuda.current_stream()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    r = _call(name, *args)
    b.record(st)
    t.records.setdefault(name, []).append((a, b, None))
    return r


def _timed(name, meta, fn):
    if _timer is None or name not in _timer.names:
        return fn()
    st = torch.cuda.current_stream()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    r = fn()
    b.record(st)
    _timer.records.setdefault(name, []).append((a, b, meta))
    return r


TUNE_GEMM_VARIANT = 0
TUNE_GEMM_ABLATE = 1
TUNE_GEMM_GROUP_M = 3  # m-tiles per raster group of the 256x256 GEMM kernels (0 = default 4)
TUNE_GEMM_FAST = 6   # 0 = library default, 1 = exact dequant, G in {2, 4, 8} = one-VALU dequant re-centred every G K blocks
TUNE_LIN_QB = 7      # linear branch pass 2: Q blocks per workgroup (0 = library default)
TUNE_ATTN_OCC = 8    # 2 = INT8/FP16-PV attention built for two workgroups per CU with explicit fragment prefetch, 3 = the Q64 build, 4 = 2 + row sum on the matrix pipe
TUNE_VAE_CONV = 9    # 1 = the first td_vae_conv kernel (cross-check of the default)
TUNE_GELU_TABLE = 11  # 1 = the fused FFN GEMM's GELU evaluated inline instead of looked up in the device-built table (bit-identical; A/B)
TUNE_GEMM_COTENANT = 12  # 1 while W8A8 GEMMs are launched beside another GEMM on a second stream (256-row tiles only)
TUNE_GEMM_W4 = 13    # bit mask of W8A8 launch kinds on the four-wave form (1 fused quantiser, 2 V^T tiles, 4 residual / row statistics, 8 plain; 16 none)
TUNE_GEMM16 = 10     # td_gemm_bf16: 2 = the four-wave 128x128 experiment (bit-identical, measured equal); default eight waves of 128x64


def set_tuning(key: int, value: int):
    """Kernel-selection knob (include/turbodiffusion_amd.h TD_TUNE_*).  Every variant of an operator is bit-identical
    except TUNE_GEMM_FAST >= 2 (one-VALU dequant: bounded difference, see csrc/gemm_w8a8_fi.hip)."""
    call("td_set_tuning", key, value)


# ----------------------------------------------------------------------------- f3: embeddings and head (csrc/embed_head.hip)
def patch_embed(x, y, w, bias, row0=0, rows=None):
    """patchify (1, 2, 2) + patch_embedding Linear: x [B, C1, T, Hin, Win] (+ y [B, C2, T, Hin, Win] concatenated on channels,
    or None), w [dim, (C1+C2)*4], bias [dim], all the same 16-bit dtype -> tokens [B, rows, dim] (rows [row0, row0+rows) of
    every batch entry; default all)."""
    require_gpu(x, y, w, bias)
    assert x.dim() == 5 and x.is_contiguous() and w.is_contiguous() and bias.is_contiguous()
    assert x.dtype == w.dtype == bias.dtype and x.dtype in (torch.bfloat16, torch.float16)
    B, c1, T, Hin, Win = x.shape
    c2 = 0
    if y is not None:
        assert y.is_contiguous() and y.dtype == x.dtype and y.shape[0] == B and tuple(y.shape[2:]) == (T, Hin, Win)
        c2 = y.shape[1]
    dim = w.shape[0]
    assert w.shape[1] == (c1 + c2) * 4 and bias.shape == (dim,)
    L_ = T * (Hin // 2) * (Win // 2)
    rows = L_ - row0 if rows is None else rows
    out = torch.empty((B, rows, dim), dtype=x.dtype, device=x.device)
    call("td_patch_embed", ptr(x), c1, ptr(y), c2, dt_code(x.dtype), B, T, Hin, Win, ptr(w), ptr(bias), ptr(out), dim, row0, rows,
         stream_ptr())
    return out


def head(x, scale, shift, w, bias, eps, out_dim, T, Hh, Ww, unpatchify=True, row0=0):
    """Head.forward + unpatchify: x [B, rows, dim] 16-bit, scale / shift f32 [B, dim] (e[1], e[0]), w f32 [out_dim*4, dim],
    bias f32 [out_dim*4] -> f32 [B, out_dim, T, 2*Hh, 2*Ww] (unpatchify; needs all tokens) or [B, rows, out_dim*4]."""
    require_gpu(x, scale, shift, w, bias)
    assert x.dim() == 3 and x.is_contiguous()
    B, rows, dim = x.shape
    scale, shift = _f32c(scale, "scale"), _f32c(shift, "shift")
    w, bias = _f32c(w, "head weight"), _f32c(bias, "head bias")
    assert scale.shape == (B, dim) and shift.shape == (B, dim) and w.shape == (out_dim * 4, dim) and bias.shape == (out_dim * 4,)
    if unpatchify:
        out = torch.empty((B, out_dim, T, 2 * Hh, 2 * Ww), dtype=torch.float32, device=x.device)
    else:
        out = torch.empty((B, rows, out_dim * 4), dtype=torch.float32, device=x.device)
    call("td_head", ptr(x), dt_code(x.dtype), ptr(scale), ptr(shift), ptr(w), ptr(bias), float(eps), ptr(out), 1 if unpatchify else 0,
         B, rows, dim, out_dim, T, Hh, Ww, row0, stream_ptr())
    return out


def time_sinusoid(t, freq_dim):
    """t [B] 16-bit -> f32 [B, freq_dim] (sinusoidal_embedding_1d in fp64, wan2pt1.py:144-153)."""
    require_gpu(t)
    assert t.dim() == 1 and t.is_contiguous() and t.dtype in (torch.bfloat16, torch.float16)
    out = torch.empty((t.shape[0], freq_dim), dtype=torch.float32, device=t.device)
    call("td_time_sinusoid", ptr(t), dt_code(t.dtype), ptr(out), t.shape[0], freq_dim, stream_ptr())
    return out


def gemv_f32(x, w, bias, silu_input=False):
    """f32 [B, N] = act(x f32 [B, K]) @ float(w [N, K])^T + float(bias); 16-bit w / bias; act = SiLU when silu_input."""
    require_gpu(x, w, bias)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous() and w.is_contiguous() and bias.is_contiguous()
    assert w.dtype == bias.dtype and w.dtype in (torch.bfloat16, torch.float16) and w.shape[1] == x.shape[1]
    out = torch.empty((x.shape[0], w.shape[0]), dtype=torch.float32, device=x.device)
    call("td_gemv_f32", ptr(x), ptr(w), ptr(bias), dt_code(w.dtype), 1 if silu_input else 0, ptr(out), x.shape[0], w.shape[0],
         x.shape[1], stream_ptr())
    return out


def bcast_add(m, e):
    """m f32 [A, R, D] + e f32 [B, RE, D] (RE == R or 1) -> f32 [A, B, R, D]."""
    require_gpu(m, e)
    m, e = _f32c(m, "m"), _f32c(e, "e")
    A, R, D = m.shape
    B, RE, D2 = e.shape
    assert D2 == D and RE in (R, 1)
    out = torch.empty((A, B, R, D), dtype=torch.float32, device=m.device)
    call("td_bcast_add", ptr(m), ptr(e), ptr(out), A, B, R, RE, D, stream_ptr())
    return out


def rcm_step_(x, v, eps, t_cur, t_next, dtype16=None):
    """In place on the fp64 state x: the SDE update (1 - t_next) (x - t_cur v) + t_next eps, or the ODE update x - (t_cur -
    t_next) v when eps is None; returns x cast to ``dtype16`` (the next step's network input) when asked, from the same pass."""
    require_gpu(x, v, eps)
    assert x.dtype == torch.float64 and v.dtype == torch.float32 and x.is_contiguous() and v.is_contiguous() and v.numel() == x.numel()
    if eps is not None:
        assert eps.dtype == torch.float32 and eps.is_contiguous() and eps.numel() == x.numel()
    x16 = torch.empty(x.shape, dtype=dtype16, device=x.device) if dtype16 is not None else None
    call("td_rcm_step", ptr(x), ptr(v), ptr(eps), ptr(x16), 0 if dtype16 is None else L.dt_code(dtype16), float(t_cur), float(t_next),
         x.numel(), stream_ptr())
    return x16


# ----------------------------------------------------------------------------- measurement support (csrc/calib.hip)
def box_calibration(gemm_fn=None, device=None):
    """What this box sustains right now (bench.py's "box" record): the dense INT8 matrix-pipe rate, the streaming HBM read
    bandwidth, and — when ``gemm_fn`` (a callable that enqueues one production GEMM on the current stream) is given — the
    shader clock while that GEMM runs, from a one-wave probe on a second stream reading s_memtime against the 100 MHz
    s_memrealtime.  ~100 ms of GPU time; every number is the best / median of a few repetitions."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    sink = torch.zeros(16, dtype=torch.float32, device=dev)
    ev = lambda: torch.cuda.Event(enable_timing=True)   # noqa: E731
    out = {}
    # ---- INT8 MFMA: 256 CUs x 4 workgroups x 4 waves, 4 chains each
    blocks, iters = 1024, 4096
    call("td_calib_mfma_i8", 64, blocks, ptr(sink), stream_ptr())       # warm-up (code object load)
    best = 0.0
    for _ in range(3):
        a, b = ev(), ev()
        a.record()
        call("td_calib_mfma_i8", iters, blocks, ptr(sink), stream_ptr())
        b.record()
        b.synchronize()
        ops = blocks * 4 * iters * 4 * 2.0 * 32 * 32 * 32
        best = max(best, ops / (a.elapsed_time(b) * 1e-3))
    out["i8_pops"] = best / 1e15
    # ---- HBM read: 4 GiB (16x the Infinity Cache), non-temporal
    buf = torch.empty(1 << 32, dtype=torch.uint8, device=dev)
    buf.view(torch.int32).fill_(1)
    call("td_calib_hbm_read", ptr(buf), buf.numel(), ptr(sink), stream_ptr())
    best = 0.0
    for _ in range(3):
        a, b = ev(), ev()
        a.record()
        call("td_calib_hbm_read", ptr(buf), buf.numel(), ptr(sink), stream_ptr())
        b.record()
        b.synchronize()
        best = max(best, buf.numel() / (a.elapsed_time(b) * 1e-3))
    out["hbm_read_tbps"] = best / 1e12
    del buf
    # ---- shader clock: idle-ish (probe alone) and under the production GEMM
    stamps = torch.zeros(4, dtype=torch.int64, device=dev)

    def probe(ticks):
        call("td_calib_clock_probe", ticks, ptr(stamps), stream_ptr())

    def mhz():
        c0, c1, r0, r1 = stamps.tolist()
        return (c1 - c0) / max(1, r1 - r0) * 100.0

    probe(20000)                       # 200 us
    torch.cuda.synchronize()
    out["sclk_mhz_idle"] = mhz()
    if gemm_fn is not None:
        for _ in range(3):
            gemm_fn()                  # warm (and ramp the power state)
        torch.cuda.synchronize()
        a, b = ev(), ev()
        a.record()
        for _ in range(4):
            gemm_fn()
        b.record()
        b.synchronize()
        t_gemm_us = a.elapsed_time(b) * 1e3 / 4
        side = torch.cuda.Stream()
        vals = []
        for _ in range(3):
            main = torch.cuda.current_stream()
            for _ in range(2):
                gemm_fn()              # already running when the probe starts
            side.wait_stream(main)     # (orders the probe behind the ENQUEUE of those two, not behind the six below)
            with torch.cuda.stream(side):
                probe(max(2000, int(t_gemm_us * 100 * 3)))      # spans ~3 launches
            for _ in range(6):
                gemm_fn()
            torch.cuda.synchronize()
            vals.append(mhz())
        out["sclk_mhz_gemm"] = sorted(vals)[1]
        out["gemm_probe_us"] = t_gemm_us
    return out


# ----------------------------------------------------------------------------- a16
def quant_i8_block128(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """x [m,n] f16|bf16 -> (q int8 [m,n], s f32 [ceil(m/128), ceil(n/128)])."""
    require_gpu(x)
    assert x.dim() == 2 and x.is_contiguous(), "quant: need a contiguous 2-D tensor"
    m, n = x.shape
    q = torch.empty((m, n), dtype=torch.int8, device=x.device)
    s = torch.empty((cdiv(m, 128), cdiv(n, 128)), dtype=torch.float32, device=x.device)
    call("td_quant_i8_block128", ptr(x), dt_code(x.dtype), ptr(q), ptr(s), m, n, stream_ptr())
    return q, s


# ----------------------------------------------------------------------------- a17
def gemm_w8a8(a_q, a_s, b_q, b_s, out_dtype=torch.bfloat16, bias=None, gelu_tanh=False, out=None):
    require_gpu(a_q, a_s, b_q, b_s, bias)
    assert a_q.dtype == torch.int8 and b_q.dtype == torch.int8
    assert a_q.is_contiguous() and b_q.is_contiguous() and a_s.is_contiguous() and b_s.is_contiguous()
    m, k = a_q.shape
    n, k2 = b_q.shape
    assert k == k2, "gemm_w8a8: K mismatch"
    assert a_s.shape == (cdiv(m, 128), k // 128) and b_s.shape == (cdiv(n, 128), k // 128), "scale shapes"
    _f32c(a_s, "a_s"), _f32c(b_s, "b_s")
    if out is None:
        out = torch.empty((m, n), dtype=out_dtype, device=a_q.device)
    else:
        assert out.shape == (m, n) and out.stride(1) == 1 and out.dtype == out_dtype
    if bias is not None:
        assert bias.dtype == out_dtype and bias.shape == (n,) and bias.is_contiguous()
    _timed("td_gemm_w8a8", (m, n, k), lambda: call(
        "td_gemm_w8a8", ptr(a_q), ptr(a_s), ptr(b_q), ptr(b_s), ptr(bias), ptr(out), dt_code(out_dtype),
        L.TD_EPI_GELU_TANH if gelu_tanh else L.TD_EPI_NONE, m, n, k, out.stride(0), stream_ptr()))
    return out


def gemm_w8a8_quant(a_q, a_s, b_q, b_s, act_dtype=torch.bfloat16, bias=None, gelu_tanh=False):
    """W8A8 GEMM whose epilogue block-quantises its own result for the next Int8Linear:
    == quant_i8_block128(gemm_w8a8(...)) bit for bit, without the 16-bit round trip through HBM."""
    require_gpu(a_q, a_s, b_q, b_s, bias)
    assert a_q.dtype == torch.int8 and b_q.dtype == torch.int8
    assert a_q.is_contiguous() and b_q.is_contiguous() and a_s.is_contiguous() and b_s.is_contiguous()
    m, k = a_q.shape
    n, k2 = b_q.shape
    assert k == k2, "gemm_w8a8_quant: K mismatch"
    assert a_s.shape == (cdiv(m, 128), k // 128) and b_s.shape == (cdiv(n, 128), k // 128), "scale shapes"
    _f32c(a_s, "a_s"), _f32c(b_s, "b_s")
    if bias is not None:
        assert bias.dtype == act_dtype and bias.shape == (n,) and bias.is_contiguous()
    q = torch.empty((m, n), dtype=torch.int8, device=a_q.device)
    s = torch.empty((cdiv(m, 128), cdiv(n, 128)), dtype=torch.float32, device=a_q.device)
    _timed("td_gemm_w8a8", (m, n, k), lambda: call(
        "td_gemm_w8a8_quant", ptr(a_q), ptr(a_s), ptr(b_q), ptr(b_s), ptr(bias), ptr(q), ptr(s), dt_code(act_dtype),
        L.TD_EPI_GELU_TANH if gelu_tanh else L.TD_EPI_NONE, m, n, k, stream_ptr()))
    return q, s


# ----------------------------------------------------------------------------- a5 / a6 / a7
def rmsnorm(x, w, eps, out_dtype=None):
    require_gpu(x, w)
    assert x.is_contiguous(), "Input must be contiguous"
    n = x.shape[-1]
    x2 = x.reshape(-1, n)
    out_dtype = out_dtype or x.dtype
    y = torch.empty(x2.shape, dtype=out_dtype, device=x.device)
    w = w.float().contiguous()
    call("td_rmsnorm", ptr(x2), dt_code(x.dtype), ptr(w), ptr(y), dt_code(out_dtype), float(eps),
         x2.shape[0], n, stream_ptr())
    return y.reshape(x.shape)


def triton_ln_pad_cols(n: int) -> int:
    """next_power_of_2(n) - n: the phantom columns the reference's Triton LayerNorm sums (x - mean)^2 over (they load as 0,
    ops/core.py:213-224, 313-324): 512 at n = 1536, 3072 at n = 5120, 0 for a power of two.  Pass it as ``pad_cols`` to
    reproduce FastLayerNorm / ops.layernorm exactly; 0 is the textbook variance (eager WanLayerNorm, layer_norm_cuda)."""
    return (1 << max(0, (int(n) - 1).bit_length())) - int(n)


def layernorm(x, w, b, eps, scale=None, shift=None, rows_per_batch=0, out_dtype=None, pad_cols=0):
    """LayerNorm over the last dim; optional fused AdaLN modulate (scale/shift f32 [B, n]).  pad_cols: triton_ln_pad_cols."""
    require_gpu(x, w, b, scale, shift)
    assert x.is_contiguous(), "Input must be contiguous"
    n = x.shape[-1]
    x2 = x.reshape(-1, n)
    out_dtype = out_dtype or x.dtype
    y = torch.empty(x2.shape, dtype=out_dtype, device=x.device)
    if w is not None:
        w = w.float().contiguous()
        b = b.float().contiguous() if b is not None else torch.zeros_like(w)
    if scale is not None:
        scale = scale.float().contiguous().reshape(-1, n)
        shift = shift.float().contiguous().reshape(-1, n)
        if rows_per_batch == 0:
            assert x2.shape[0] % scale.shape[0] == 0
            rows_per_batch = x2.shape[0] // scale.shape[0]
    call("td_layernorm", ptr(x2), dt_code(x.dtype), ptr(w), ptr(b), ptr(scale), ptr(shift), rows_per_batch,
         ptr(y), dt_code(out_dtype), float(eps), int(pad_cols), x2.shape[0], n, stream_ptr())
    return y.reshape(x.shape)


def gemm_w8a8_residual_(x, a_q, a_s, b_q, b_s, bias=None, gate=None):
    """In place: x = x + (W8A8 GEMM + bias) * gate.type_as(x)  (gate f32 [n] or None for a plain add) ==
    gated_residual_(x, gemm_w8a8(...), gate) bit for bit, without the 16-bit round trip of the GEMM result."""
    require_gpu(x, a_q, a_s, b_q, b_s, bias, gate)
    assert a_q.dtype == torch.int8 and b_q.dtype == torch.int8 and x.dim() == 2 and x.stride(1) == 1
    assert a_q.is_contiguous() and b_q.is_contiguous() and a_s.is_contiguous() and b_s.is_contiguous()
    m, k = a_q.shape
    n, k2 = b_q.shape
    assert k == k2 and x.shape == (m, n), "gemm_w8a8_residual_: shape mismatch"
    assert a_s.shape == (cdiv(m, 128), k // 128) and b_s.shape == (cdiv(n, 128), k // 128), "scale shapes"
    _f32c(a_s, "a_s"), _f32c(b_s, "b_s")
    if bias is not None:
        assert bias.dtype == x.dtype and bias.shape == (n,) and bias.is_contiguous()
    if gate is not None:
        gate = gate.float().contiguous().reshape(-1)
        assert gate.numel() == n, "gemm_w8a8_residual_: one gate row (call per batch entry)"
    _timed("td_gemm_w8a8", (m, n, k), lambda: call(
        "td_gemm_w8a8_residual", ptr(a_q), ptr(a_s), ptr(b_q), ptr(b_s), ptr(bias), ptr(x), ptr(gate), dt_code(x.dtype),
        m, n, k, x.stride(0), stream_ptr()))
    return x


def gemm_w8a8_stats(a_q, a_s, b_q, b_s, bias, x=None, gate=None, out_dtype=torch.bfloat16, ws=None):
    """gemm_w8a8 (x is None: returns (y, ws)) or gemm_w8a8_residual_ (in place on x: returns (x, ws)) whose epilogue also
    writes the row-statistics partials ws f32 [m, n/64, 2] = per 64-column piece (mean, M2) of the stored values
    — same output bits as the plain calls.  ws: a preallocated (row slice of a) partials tensor to write into."""
    require_gpu(a_q, a_s, b_q, b_s, bias, x, gate)
    m, k = a_q.shape
    n = b_q.shape[0]
    assert a_q.dtype == torch.int8 and b_q.dtype == torch.int8 and b_q.shape[1] == k and n % 64 == 0 and bias is not None
    assert a_q.is_contiguous() and b_q.is_contiguous() and bias.dtype == torch.bfloat16 and bias.is_contiguous()
    _f32c(a_s, "a_s"), _f32c(b_s, "b_s")
    if ws is None:
        ws = torch.empty((m, n // 64, 2), dtype=torch.float32, device=a_q.device)
    assert ws.shape == (m, n // 64, 2) and ws.dtype == torch.float32 and ws.is_contiguous()
    if x is None:
        y = torch.empty((m, n), dtype=out_dtype, device=a_q.device)
        tgt, ld, res = y, n, 0
    else:
        assert x.shape == (m, n) and x.stride(1) == 1 and x.dtype == torch.bfloat16
        tgt, ld, res = x, x.stride(0), 1
    if gate is not None:
        gate = gate.float().contiguous().reshape(-1)
        assert gate.numel() == n
    _timed("td_gemm_w8a8", (m, n, k), lambda: call(
        "td_gemm_w8a8_stats", ptr(a_q), ptr(a_s), ptr(b_q), ptr(b_s), ptr(bias), ptr(tgt), ptr(gate), res, dt_code(tgt.dtype),
        m, n, k, ld, ptr(ws), stream_ptr()))
    return tgt, ws


def gemm_w8a8_vt(a_q, a_s, b_q, b_s, bias, v_col0, vt_dtype, out_dtype=torch.bfloat16, out=None):
    """Fused q|k|v projection: -> (d [m, n] whose columns >= v_col0 are NOT written, vt [heads, ceil(m/64), 128, 64] = the
    V^T tiles ``v_transpose`` would make of those columns).  ``out``: d as a column range of a wider row-major buffer."""
    require_gpu(a_q, a_s, b_q, b_s, bias)
    m, k = a_q.shape
    n = b_q.shape[0]
    assert b_q.shape[1] == k and a_q.is_contiguous() and b_q.is_contiguous() and bias.dtype == out_dtype
    assert bias.shape == (n,) and bias.is_contiguous() and a_s.is_contiguous() and b_s.is_contiguous()
    assert a_s.shape == (cdiv(m, 128), k // 128) and b_s.shape == (cdiv(n, 128), k // 128), "scale shapes"
    a_s, b_s = _f32c(a_s, "a_s"), _f32c(b_s, "b_s")
    if out is None:
        d = torch.empty((m, n), dtype=out_dtype, device=a_q.device)
    else:
        d = out
        assert d.shape == (m, n) and d.stride(1) == 1 and d.dtype == out_dtype and d.stride(0) % 8 == 0 and d.data_ptr() % 16 == 0
    vt = torch.empty(((n - v_col0) // 128, cdiv(m, 64), 128, 64), dtype=vt_dtype, device=a_q.device)
    _timed("td_gemm_w8a8", (m, n, k), lambda: call(
        "td_gemm_w8a8_vt", ptr(a_q), ptr(a_s), ptr(b_q), ptr(b_s), ptr(bias), ptr(d), dt_code(out_dtype), m, n, k, d.stride(0),
        v_col0, ptr(vt), dt_code(vt_dtype), stream_ptr()))
    return d, vt


def row_stats_finalize(ws, n, eps, rms=False, pad_cols=0):
    """ws f32 [m, pieces, 2] (per 64-column piece: mean, M2) -> LayerNorm (mean, rstd) f32 [m, 2], or (rms=True) the
    RMSNorm rstd f32 [m].  pad_cols: triton_ln_pad_cols (LayerNorm only)."""
    require_gpu(ws)
    m, pieces, _ = ws.shape
    out = torch.empty((m,) if rms else (m, 2), dtype=torch.float32, device=ws.device)
    call("td_row_stats_finalize", ptr(ws), pieces, n, float(eps), 0 if rms else int(pad_cols), 1 if rms else 0, ptr(out), m,
         stream_ptr())
    return out


LNQ_MAX_N = 8192


def layernorm_quant(x, w, b, eps, scale=None, shift=None, rows_per_batch=0, stats=None, pad_cols=0):
    """LayerNorm (+ affine, + AdaLN modulate) fused with the per-128x128-block INT8 quantiser of the consuming
    Int8Linear: returns (q int8 [m,n], s f32 [ceil(m/128), ceil(n/128)]) == quant_i8_block128(layernorm(...)).
    stats: the rows' (mean, rstd) f32 [m, 2] when the producer of x already supplied them (row_stats_finalize — pad_cols
    then went into THAT call).  pad_cols: triton_ln_pad_cols."""
    require_gpu(x, w, b, scale, shift)
    assert x.is_contiguous() and x.dim() == 2, "Input must be a contiguous 2-D tensor"
    m, n = x.shape
    if scale is not None and rows_per_batch == 0:
        nb = scale.numel() // n
        assert m % nb == 0
        rows_per_batch = m // nb
    if n > LNQ_MAX_N or n % 8 or (scale is not None and rows_per_batch < 128):
        assert stats is None
        return quant_i8_block128(layernorm(x, w, b, eps, scale, shift, rows_per_batch, pad_cols=pad_cols))
    if w is not None:
        w = w.float().contiguous()
        b = b.float().contiguous() if b is not None else None
    if scale is not None:
        scale = scale.float().contiguous().reshape(-1, n)
        shift = shift.float().contiguous().reshape(-1, n)
    q = torch.empty((m, n), dtype=torch.int8, device=x.device)
    s = torch.empty((cdiv(m, 128), cdiv(n, 128)), dtype=torch.float32, device=x.device)
    if stats is not None:
        assert stats.shape == (m, 2) and stats.dtype == torch.float32 and stats.is_contiguous()
        call("td_layernorm_quant_stats", ptr(x), dt_code(x.dtype), ptr(w), ptr(b), ptr(scale), ptr(shift), rows_per_batch,
             ptr(q), ptr(s), ptr(stats), m, n, stream_ptr())
        return q, s
    ws = torch.empty((m, 2), dtype=torch.float32, device=x.device)   # rows' (mean, rstd) between the two passes
    call("td_layernorm_quant", ptr(x), dt_code(x.dtype), ptr(w), ptr(b), ptr(scale), ptr(shift), rows_per_batch,
         ptr(q), ptr(s), ptr(ws), float(eps), int(pad_cols), m, n, stream_ptr())
    return q, s


def gated_residual_(x, y, gate=None):
    """In place: x = x + y * gate.type_as(x)   (gate f32 [B, n] or None for a plain add)."""
    require_gpu(x, y, gate)
    assert x.is_contiguous() and y.is_contiguous() and x.shape == y.shape and x.dtype == y.dtype
    n = x.shape[-1]
    m = x.numel() // n
    rpb = 0
    if gate is not None:
        gate = gate.float().contiguous().reshape(-1, n)
        assert m % gate.shape[0] == 0
        rpb = m // gate.shape[0]
    call("td_gated_residual", ptr(x), ptr(y), ptr(gate), rpb, dt_code(x.dtype), m, n, stream_ptr())
    return x


# ----------------------------------------------------------------------------- a3 / a8
def qk_norm_rope(src, col0, H, D, w, cos, sin, eps):
    """src [L, ld] (a GEMM output); columns [col0, col0+H*D) -> [H, L, D] with RMSNorm over the
    H*D columns (w f32 [H*D] or None) and interleaved RoPE (cos/sin f32 [L, D/2] or None)."""
    require_gpu(src, w, cos, sin)
    assert src.dim() == 2 and src.stride(1) == 1
    _f32c(w, "norm weight"), _f32c(cos, "cos"), _f32c(sin, "sin")
    Lr = src.shape[0]
    dst = torch.empty((H, Lr, D), dtype=src.dtype, device=src.device)
    view = src[:, col0:col0 + H * D]
    call("td_qk_norm_rope", ptr(view), src.stride(0), ptr(w), ptr(cos), ptr(sin), ptr(dst), dt_code(src.dtype),
         float(eps), Lr, H, D, stream_ptr())
    return dst


def qk_norm_rope_pair(src, col_q, col_k, H, D, wq, wk, cos, sin, eps):
    """qk_norm_rope for the q columns [col_q, col_q + H*D) AND the k columns [col_k, ...) of one GEMM output in ONE launch
    -> (q [H, L, D], k [H, L, D]); bit-identical to the two single calls."""
    require_gpu(src, wq, wk, cos, sin)
    assert src.dim() == 2 and src.stride(1) == 1
    _f32c(wq, "norm weight"), _f32c(wk, "norm weight"), _f32c(cos, "cos"), _f32c(sin, "sin")
    Lr = src.shape[0]
    both = torch.empty((2, H, Lr, D), dtype=src.dtype, device=src.device)
    call("td_qk_norm_rope_pair", ptr(src[:, col_q:col_q + H * D]), ptr(src[:, col_k:col_k + H * D]), src.stride(0), ptr(wq), ptr(wk),
         ptr(cos), ptr(sin), ptr(both[0]), ptr(both[1]), dt_code(src.dtype), float(eps), Lr, H, D, stream_ptr())
    return both[0], both[1]


def seq_sum(k, out=None):
    """k [H, L, D] -> f32 [H, D] column sums over this tensor's rows (td_seq_sum: 64 chunk partials per head + a pass that adds
    them in order: two launches, no library reduction, no copy).  out: a contiguous f32 [H, D]
    destination (e.g. a slice of a send buffer)."""
    require_gpu(k, out)
    assert k.is_contiguous()
    H, L_, D = k.shape
    ws = torch.empty((H, 64, D), dtype=torch.float32, device=k.device)
    if out is None:
        out = torch.empty((H, D), dtype=torch.float32, device=k.device)
    assert out.dtype == torch.float32 and tuple(out.shape) == (H, D) and out.is_contiguous()
    call("td_seq_sum", ptr(k), ptr(ws), ptr(out), dt_code(k.dtype), L_, H, D, stream_ptr())
    return out


def v_transpose(v, stride_h, stride_l, L_, H, D, out_dtype):
    """v: tensor whose element (h,l,d) is at data_ptr + h*stride_h + l*stride_l + d -> vt tiles."""
    require_gpu(v)
    kb = cdiv(L_, 64)
    vt = torch.empty((H, kb, D, 64), dtype=out_dtype, device=v.device)
    call("td_v_transpose", ptr(v), dt_code(v.dtype), stride_h, stride_l, ptr(vt), dt_code(out_dtype), L_, H, D,
         stream_ptr())
    return vt


# ----------------------------------------------------------------------------- a11 / a13
def seq_mean(k):
    """k [H, L, D] -> km [H, D] (k.mean(dim=-2) in k's dtype)."""
    require_gpu(k)
    assert k.is_contiguous()
    H, L_, D = k.shape
    km = torch.empty((H, D), dtype=k.dtype, device=k.device)
    ws = torch.empty((H, 64, D), dtype=torch.float32, device=k.device)
    call("td_seq_mean", ptr(k), ptr(km), ptr(ws), dt_code(k.dtype), L_, H, D, stream_ptr())
    return km


def sage_quant_pool(x, km, blk, want_pool=True, want_quant=True):
    """x [H, L, D] -> (pooled [H, nb, D] | None, xq int8 [H, L, D] | None, xs f32 [H, nb] | None)."""
    require_gpu(x, km)
    assert x.is_contiguous()
    H, L_, D = x.shape
    nb = cdiv(L_, blk)
    pooled = torch.empty((H, nb, D), dtype=x.dtype, device=x.device) if want_pool else None
    xq = torch.empty((H, L_, D), dtype=torch.int8, device=x.device) if want_quant else None
    xs = torch.empty((H, nb), dtype=torch.float32, device=x.device) if want_quant else None
    call("td_sage_quant_pool", ptr(x), ptr(km), dt_code(x.dtype), blk, ptr(pooled), ptr(xq), ptr(xs), L_, H, D,
         stream_ptr())
    return pooled, xq, xs


def sla_topk(pq, pk, topk, kb=None):
    """pq [H, Qb, D], pk [H, Kb_alloc, D] (first ``kb`` blocks valid) -> lut int32 [H, Qb, topk]
    (ascending block ids)."""
    require_gpu(pq, pk)
    H, Qb, D = pq.shape
    kb_alloc = pk.shape[1]
    kb = kb_alloc if kb is None else kb
    lut = torch.empty((H, Qb, topk), dtype=torch.int32, device=pq.device)
    call("td_sla_topk", ptr(pq), ptr(pk), dt_code(pq.dtype), ptr(lut), H, Qb, kb, kb_alloc, D, topk, stream_ptr())
    return lut


# ----------------------------------------------------------------------------- a9 / a12 / a13
def _attn_quant_outputs(L_, H, device):
    return (torch.empty((L_, H * 128), dtype=torch.int8, device=device),
            torch.empty((cdiv(L_, 128), H), dtype=torch.float32, device=device))


def attn_i8(q_i8, q_s, k_i8, k_s, vt, lut, out, o_stride_h, o_stride_l, sm_scale=None, lk=None, add_t=None,
            quant_out=False, v_scale=None):
    """SageAttention INT8-QK/FP16-PV. q_i8 [H,L,128], k_i8 [H,Lk,128], vt f16 tiles; lut or None (dense).
    out: preallocated 16-bit tensor addressed as out_ptr + h*o_stride_h + l*o_stride_l + d.
    add_t: o_l from sla_linear_out_t (the output becomes o_s + o_l).  quant_out: instead of ``out`` return the
    [L, H*128] output block-quantised for the o projection: (int8 [L, H*128], f32 [ceil(L/128), H]); ``out`` then
    only supplies the 16-bit dtype (a tensor or a torch.dtype)."""
    require_gpu(q_i8, k_i8, vt, lut, add_t, v_scale)
    _f32c(q_s, "q_s"), _f32c(k_s, "k_s"), _f32c(v_scale, "v_scale")
    H, L_, D = q_i8.shape
    lk_alloc = k_i8.shape[1]
    Lk = lk_alloc if lk is None else lk  # lk < allocation: rank-padded gathered layout
    fp8 = v_scale is not None   # vt: e4m3 tiles of v_fp8_tiles (uint8) + per-channel scale; else fp16 tiles of v_transpose
    assert D == 128 and vt.dtype == (torch.uint8 if fp8 else torch.float16) and vt.shape[1] * 64 >= lk_alloc
    if sm_scale is None:
        sm_scale = 1.0 / math.sqrt(D)
    nsel = 0 if lut is None else lut.shape[-1]
    odt = out if isinstance(out, torch.dtype) else out.dtype
    oq, os_ = _attn_quant_outputs(L_, H, q_i8.device) if quant_out else (None, None)
    if fp8:
        _timed("td_attn_i8", (H, L_, Lk, nsel), lambda: call(
            "td_attn_i8_fp8pv", ptr(q_i8), ptr(q_s), ptr(k_i8), ptr(k_s), ptr(vt), ptr(v_scale), ptr(lut), nsel,
            None if quant_out else ptr(out), dt_code(odt), o_stride_h, o_stride_l, float(sm_scale), L_, Lk, lk_alloc, H,
            ptr(add_t), ptr(oq), ptr(os_), stream_ptr()))
        return (oq, os_) if quant_out else out
    _timed("td_attn_i8", (H, L_, Lk, nsel), lambda: call(
        "td_attn_i8_ex", ptr(q_i8), ptr(q_s), ptr(k_i8), ptr(k_s), ptr(vt), ptr(lut), nsel,
        None if quant_out else ptr(out), dt_code(odt), o_stride_h, o_stride_l, float(sm_scale), L_, Lk, lk_alloc, H,
        ptr(add_t), ptr(oq), ptr(os_), stream_ptr()))
    return (oq, os_) if quant_out else out


def v_fp8_tiles(v, stride_h, stride_l, L_, H, D, scale_max=2.25):
    """V (element (h,l,d) at data_ptr + h*stride_h + l*stride_l + d) -> (vt8 e4m3 tiles [H, ceil(L/64), 128, 64] as uint8,
    v_scale f32 [H, 128] = max_l |v| / scale_max) for attn_i8(..., v_scale=...) — the FP8-PV SageAttention variant
    (SLA/core.py:217-224)."""
    require_gpu(v)
    kb = cdiv(L_, 64)
    vt8 = torch.empty((H, kb, D, 64), dtype=torch.uint8, device=v.device)
    vs = torch.empty((H, D), dtype=torch.float32, device=v.device)
    ws = torch.empty((H, 64, D), dtype=torch.float32, device=v.device)
    call("td_v_fp8_tiles", ptr(v), dt_code(v.dtype), stride_h, stride_l, ptr(vt8), ptr(vs), ptr(ws), float(scale_max), L_, H, D,
         stream_ptr())
    return vt8, vs


def attn_16(q, k, vt, lut, out, o_stride_h, o_stride_l, sm_scale=None, lk=None, add_t=None, quant_out=False):
    """16-bit QK attention (q,k [H,L,128] bf16|f16, vt tiles same dtype); add_t / quant_out as attn_i8."""
    require_gpu(q, k, vt, lut, add_t)
    H, L_, D = q.shape
    lk_alloc = k.shape[1]
    Lk = lk_alloc if lk is None else lk
    assert D == 128 and vt.dtype == q.dtype and (quant_out or out.dtype == q.dtype)
    if sm_scale is None:
        sm_scale = 1.0 / math.sqrt(D)
    nsel = 0 if lut is None else lut.shape[-1]
    oq, os_ = _attn_quant_outputs(L_, H, q.device) if quant_out else (None, None)
    call("td_attn_16_ex", ptr(q), ptr(k), ptr(vt), ptr(lut), nsel, None if quant_out else ptr(out), dt_code(q.dtype),
         o_stride_h, o_stride_l, float(sm_scale), L_, Lk, lk_alloc, H, ptr(add_t), ptr(oq), ptr(os_), stream_ptr())
    return (oq, os_) if quant_out else out


def rms_stats(src, n, eps):
    """src [L, ld] 16-bit -> rstd f32 [L] = 1/sqrt(mean(src[:, :n]^2) + eps) (the row statistic of td_qk_norm_rope)."""
    require_gpu(src)
    assert src.dim() == 2 and src.stride(1) == 1
    rstd = torch.empty((src.shape[0],), dtype=torch.float32, device=src.device)
    call("td_rms_stats", ptr(src), src.stride(0), dt_code(src.dtype), ptr(rstd), float(eps), src.shape[0], n, stream_ptr())
    return rstd


def attn_16_qnorm(q_src, rstd, w, k, vt, lut, out, o_stride_h, o_stride_l, sm_scale=None, lk=None, quant_out=False):
    """attn_16 whose Q is a [L, H*128] linear output normalised on load (== attn_16(qk_norm_rope(q_src, w, no RoPE), ...)).
    rstd: f32 [L] (rms_stats / row_stats_finalize), or a pair (ws, eps): the STATS partials f32 [L, H*128/64, 2] of the GEMM that
    produced q_src — the statistic is then formed inside the kernel (td_attn_16_qnorm_pieces: no finaliser launch; same bits)."""
    require_gpu(q_src, w, k, vt, lut)
    H, lk_alloc, D = k.shape
    L_ = q_src.shape[0]
    Lk = lk_alloc if lk is None else lk
    assert D == 128 and vt.dtype == q_src.dtype and q_src.stride(1) == 1
    _f32c(w, "norm weight")
    if sm_scale is None:
        sm_scale = 1.0 / math.sqrt(D)
    nsel = 0 if lut is None else lut.shape[-1]
    oq, os_ = _attn_quant_outputs(L_, H, q_src.device) if quant_out else (None, None)
    if isinstance(rstd, tuple):
        ws, eps = rstd
        require_gpu(ws)
        assert ws.dtype == torch.float32 and ws.is_contiguous() and tuple(ws.shape) == (L_, H * D // 64, 2)
        call("td_attn_16_qnorm_pieces", ptr(q_src), q_src.stride(0), ptr(ws), ws.shape[1], float(eps), ptr(w), ptr(k), ptr(vt), ptr(lut),
             nsel, None if quant_out else ptr(out), dt_code(q_src.dtype), o_stride_h, o_stride_l, float(sm_scale), L_, Lk, lk_alloc,
             H, None, ptr(oq), ptr(os_), stream_ptr())
        return (oq, os_) if quant_out else out
    require_gpu(rstd)
    _f32c(rstd, "rstd")
    call("td_attn_16_qnorm", ptr(q_src), q_src.stride(0), ptr(rstd), ptr(w), ptr(k), ptr(vt), ptr(lut), nsel,
         None if quant_out else ptr(out), dt_code(q_src.dtype), o_stride_h, o_stride_l, float(sm_scale), L_, Lk, lk_alloc,
         H, None, ptr(oq), ptr(os_), stream_ptr())
    return (oq, os_) if quant_out else out


# ----------------------------------------------------------------------------- a14
SLA_NCH = 64  # TD_SLA_NCH in include/turbodiffusion_amd.h


FEATURE_MAPS = {"softmax": 0, "elu": 1, "relu": 2}


def sla_linear_kv(k, vt, want_kmean=False, feature_map="softmax"):
    """-> (kvsum_t, ksum) of the linear branch [, km = seq_mean(k) accumulated during the same pass over K (softmax map only)]."""
    require_gpu(k, vt)
    H, L_, D = k.shape
    if feature_map != "softmax":
        assert not want_kmean
        ws_kv = torch.empty((H, SLA_NCH, D, D), dtype=torch.float32, device=k.device)
        ws_ks = torch.empty((H, SLA_NCH, D), dtype=torch.float32, device=k.device)
        kv_t = torch.empty((H, D, D), dtype=k.dtype, device=k.device)
        ksum = torch.empty((H, D), dtype=k.dtype, device=k.device)
        call("td_sla_linear_kv_fm", ptr(k), dt_code(k.dtype), ptr(vt), dt_code(vt.dtype), ptr(ws_kv), ptr(ws_ks), ptr(kv_t),
             ptr(ksum), FEATURE_MAPS[feature_map], L_, H, D, stream_ptr())
        return kv_t, ksum
    ws_km = torch.empty((H, SLA_NCH, D), dtype=torch.float32, device=k.device) if want_kmean else None
    km = torch.empty((H, D), dtype=k.dtype, device=k.device) if want_kmean else None
    ws_kv = torch.empty((H, SLA_NCH, D, D), dtype=torch.float32, device=k.device)
    ws_ks = torch.empty((H, SLA_NCH, D), dtype=torch.float32, device=k.device)
    kv_t = torch.empty((H, D, D), dtype=k.dtype, device=k.device)
    ksum = torch.empty((H, D), dtype=k.dtype, device=k.device)
    call("td_sla_linear_kv", ptr(k), dt_code(k.dtype), ptr(vt), dt_code(vt.dtype), ptr(ws_kv), ptr(ws_ks),
         ptr(kv_t), ptr(ksum), ptr(ws_km), ptr(km), L_, H, D, stream_ptr())
    return (kv_t, ksum, km) if want_kmean else (kv_t, ksum)


def seq_sum_partial(k, out=None):
    """k [H, L, D] -> f32 partial sums [H, 64, D] of this rank's tokens."""
    require_gpu(k)
    H, L_, D = k.shape
    ws = out if out is not None else torch.empty((H, 64, D), dtype=torch.float32, device=k.device)
    call("td_seq_sum_partial", ptr(k), ptr(ws), dt_code(k.dtype), L_, H, D, stream_ptr())
    return ws


def seq_mean_final(ws, nch, stride_h, stride_c, L_total, H, D, dtype):
    km = torch.empty((H, D), dtype=dtype, device=ws.device)
    call("td_seq_mean_final", ptr(ws), nch, stride_h, stride_c, ptr(km), dt_code(dtype), L_total, H, D, stream_ptr())
    return km


def sla_linear_kv_partial_f32(k, vt, kv_out=None, ks_out=None):
    """This rank's un-rounded contribution: (kv f32 [H, D, D] (d1,d2), ks f32 [H, D])."""
    require_gpu(k, vt)
    H, L_, D = k.shape
    ws_kv = torch.empty((H, SLA_NCH, D, D), dtype=torch.float32, device=k.device)
    ws_ks = torch.empty((H, SLA_NCH, D), dtype=torch.float32, device=k.device)
    call("td_sla_linear_kv_partial", ptr(k), dt_code(k.dtype), ptr(vt), dt_code(vt.dtype), ptr(ws_kv), ptr(ws_ks),
         L_, H, D, stream_ptr())
    kv = kv_out if kv_out is not None else torch.empty((H, D, D), dtype=torch.float32, device=k.device)
    ks = ks_out if ks_out is not None else torch.empty((H, D), dtype=torch.float32, device=k.device)
    call("td_sla_linear_kv_final", ptr(ws_kv), ptr(ws_ks), SLA_NCH, SLA_NCH * D * D, D * D, SLA_NCH * D, D, ptr(kv), ptr(ks),
         L.TD_F32, H, D, stream_ptr())
    return kv, ks


def sla_linear_kv_final(kv_parts, ks_parts, nch, kv_sh, kv_sc, ks_sh, ks_sc, H, D, dtype):
    """Sum ``nch`` fp32 partials (strided) and round: -> (kvsum_t [H, D, D] dtype, ksum [H, D] dtype)."""
    kv_t = torch.empty((H, D, D), dtype=dtype, device=kv_parts.device)
    ksum = torch.empty((H, D), dtype=dtype, device=kv_parts.device)
    call("td_sla_linear_kv_final", ptr(kv_parts), ptr(ks_parts), nch, kv_sh, kv_sc, ks_sh, ks_sc, ptr(kv_t),
         ptr(ksum), dt_code(dtype), H, D, stream_ptr())
    return kv_t, ksum


def sla_linear_out_t(q, kv_t, ksum, wp, bp):
    """o_l = cast(proj_l((cq @ kvsum) / (1e-5 + cq.ksum))) in the lane-private layout the attention kernels add in their
    epilogue (``add_t``): 16-bit [H, ceil(L/128), 4, 16, 64, 4]."""
    require_gpu(q, kv_t, ksum, wp, bp)
    H, L_, D = q.shape
    assert wp.dtype == torch.float32 and bp.dtype == torch.float32 and wp.is_contiguous()
    t = torch.empty((H, cdiv(L_, 128), 4, 16, 64, 4), dtype=q.dtype, device=q.device)
    call("td_sla_linear_out_t", ptr(q), dt_code(q.dtype), ptr(kv_t), ptr(ksum), ptr(wp), ptr(bp), ptr(t), L_, H, D,
         stream_ptr())
    return t


def sla_linear_out_(q, kv_t, ksum, wp, bp, out, o_stride_h, o_stride_l, feature_map="softmax"):
    """out[h, l, :] += cast(proj_l((cq @ kvsum) / (1e-5 + cq.ksum)))  (read-modify-write of the attention output)"""
    require_gpu(q, kv_t, ksum, wp, bp, out)
    H, L_, D = q.shape
    assert wp.dtype == torch.float32 and bp.dtype == torch.float32 and wp.is_contiguous()
    if feature_map != "softmax":
        call("td_sla_linear_out_fm", ptr(q), dt_code(q.dtype), ptr(kv_t), ptr(ksum), ptr(wp), ptr(bp), ptr(out),
             o_stride_h, o_stride_l, FEATURE_MAPS[feature_map], L_, H, D, stream_ptr())
        return out
    call("td_sla_linear_out", ptr(q), dt_code(q.dtype), ptr(kv_t), ptr(ksum), ptr(wp), ptr(bp), ptr(out),
         o_stride_h, o_stride_l, L_, H, D, stream_ptr())
    return out


# ----------------------------------------------------------------------------- sequence parallelism: gathered (rank-major) K side
def _rank_major(t, inner_shape):
    """t: a [W, *inner_shape] view of an all-gather output whose only non-contiguous dim is the rank dim -> byte stride."""
    assert tuple(t.shape[1:]) == tuple(inner_shape) and t[0].is_contiguous(), "gathered part must be contiguous per rank"
    return t.stride(0) * t.element_size()


def sla_topk_sp(pq, pk_g, topk, kb):
    """pq [H, Qb, D]; pk_g [W, H, kbp, D] rank-major view of the gathered pooled K (first ``kb`` global blocks valid)."""
    require_gpu(pq, pk_g)
    H, Qb, D = pq.shape
    W, H2, kbp, _ = pk_g.shape
    assert H2 == H and pq.is_contiguous()
    rs = _rank_major(pk_g, (H, kbp, D)) // pk_g.element_size()
    lut = torch.empty((H, Qb, topk), dtype=torch.int32, device=pq.device)
    call("td_sla_topk_sp", ptr(pq), ptr(pk_g), dt_code(pq.dtype), ptr(lut), H, Qb, kb, kbp, rs, D, topk, stream_ptr())
    return lut


def _sp_quant_out(quant_out, H):
    """quant_out of the *_sp attention wrappers: None, or (oq int8 [L, Ht*128], os f32 [ceil(L/128), Ht], h0, Ht) — the launch's H
    heads are heads [h0, h0 + H) of a row of Ht heads -> (q_out pointer, q_scale pointer, Ht)."""
    if quant_out is None:
        return None, None, 0
    oq, os_, h0, Ht = quant_out
    assert oq.dtype == torch.int8 and oq.is_contiguous() and oq.shape[1] == Ht * 128 and os_.dtype == torch.float32
    assert os_.is_contiguous() and os_.shape[1] == Ht and 0 <= h0 and h0 + H <= Ht
    return L.ctypes.c_void_p(oq.data_ptr() + h0 * 128), L.ctypes.c_void_p(os_.data_ptr() + 4 * h0), Ht


def attn_i8_sp(q_i8, q_s, k_g, ks_g, vt_g, lut, out, o_stride_h, o_stride_l, lk, sm_scale=None, add_t=None, quant_out=None):
    """attn_i8 with the K side read from the all-gather's rank-major output: k_g int8 [W, H, per, 128],
    ks_g f32 [W, H, per/64], vt_g f16 [W, H, per/64, 128, 64] (views; only the rank dim may be strided).
    quant_out: see _sp_quant_out (``out`` then only supplies the 16-bit dtype: a tensor or a torch.dtype)."""
    require_gpu(q_i8, k_g, ks_g, vt_g, lut, add_t)
    H, L_, D = q_i8.shape
    W, H2, per, _ = k_g.shape
    kbp = per // 64
    assert H2 == H and D == 128 and per % 64 == 0 and vt_g.dtype == torch.float16 and q_i8.is_contiguous() and q_s.is_contiguous()
    k_rs = _rank_major(k_g, (H, per, D))
    ks_rs = _rank_major(ks_g, (H, kbp)) // 4
    v_rs = _rank_major(vt_g, (H, kbp, D, 64))
    if sm_scale is None:
        sm_scale = 1.0 / math.sqrt(D)
    nsel = 0 if lut is None else lut.shape[-1]
    oq, os_, ht = _sp_quant_out(quant_out, H)
    odt = out if isinstance(out, torch.dtype) else out.dtype
    _timed("td_attn_i8", (H, L_, lk, nsel), lambda: call(
        "td_attn_i8_sp", ptr(q_i8), ptr(q_s), ptr(k_g), ptr(ks_g), ptr(vt_g), ptr(lut), nsel, None if oq is not None else ptr(out),
        dt_code(odt), o_stride_h, o_stride_l, float(sm_scale), L_, lk, H, kbp, k_rs, ks_rs, v_rs, ptr(add_t), oq, os_, ht, stream_ptr()))
    return out


def attn_16_sp(q, k_g, vt_g, lut, out, o_stride_h, o_stride_l, lk, sm_scale=None, add_t=None, quant_out=None):
    """attn_16 with rank-major gathered K [W, H, per, 128] / V^T tiles [W, H, per/64, 128, 64] (same 16-bit dtype as q)."""
    require_gpu(q, k_g, vt_g, lut, add_t)
    H, L_, D = q.shape
    W, H2, per, _ = k_g.shape
    kbp = per // 64
    assert H2 == H and D == 128 and per % 64 == 0 and vt_g.dtype == q.dtype and k_g.dtype == q.dtype and q.is_contiguous()
    k_rs = _rank_major(k_g, (H, per, D))
    v_rs = _rank_major(vt_g, (H, kbp, D, 64))
    if sm_scale is None:
        sm_scale = 1.0 / math.sqrt(D)
    nsel = 0 if lut is None else lut.shape[-1]
    oq, os_, ht = _sp_quant_out(quant_out, H)
    call("td_attn_16_sp", ptr(q), ptr(k_g), ptr(vt_g), ptr(lut), nsel, None if oq is not None else ptr(out), dt_code(q.dtype), o_stride_h,
         o_stride_l, float(sm_scale), L_, lk, H, kbp, k_rs, v_rs, ptr(add_t), oq, os_, ht, stream_ptr())
    return out


def sp_pack_begin(k, v_src, v_strides, L_loc, lay, lin_kv=None, lin_ks=None):
    """First half of a rank's K-side pack (``lay``: a ``seqpar.PackLayout``; flat uint8 send buffer: the pooled K of all heads,
    then per head group g = heads [g*hg, (g+1)*hg) the sections k | vt | ks) — everything that does NOT depend on the global
    smooth-K mean: the V^T MFMA tiles of V (element (h,l,d) at v_src + h*v_strides[0] + l*v_strides[1] + d) straight into the vt
    sections, and — linear branch — this rank's fp32 partials ck^T v -> lin_kv f32 [H, D, D], sum ck -> lin_ks f32 [H, D] (slices
    of the EARLY send buffer: they travel with the K column sums, ahead of the pack, so that the branch's reduction and second
    pass run beside the K quantiser and the exchange instead of behind them).  Returns the pack (k / ks / pk still unwritten)."""
    require_gpu(k, v_src, lin_kv, lin_ks)
    H, L_, D = k.shape
    assert L_ == L_loc and k.is_contiguous() and H == lay.G * lay.hg and D == lay.D
    pack = (torch.zeros if L_loc < lay.per else torch.empty)((lay.total,), dtype=torch.uint8, device=k.device)
    vt0 = L.ctypes.c_void_p(pack.data_ptr() + lay.ab + lay.offs["vt"])
    call("td_v_transpose_packed", ptr(v_src), dt_code(v_src.dtype), v_strides[0], v_strides[1], vt0, dt_code(lay.pdt),
         L_loc, lay.per, lay.hg, lay.gb, H, D, stream_ptr())
    if lay.linear:
        assert lin_kv.dtype == torch.float32 and tuple(lin_kv.shape) == (H, D, D) and lin_kv.is_contiguous()
        assert lin_ks.dtype == torch.float32 and tuple(lin_ks.shape) == (H, D) and lin_ks.is_contiguous()
        ws_kv = torch.empty((H, SLA_NCH, D, D), dtype=torch.float32, device=k.device)
        ws_ks = torch.empty((H, SLA_NCH, D), dtype=torch.float32, device=k.device)
        call("td_sla_linear_kv_partial_packed", ptr(k), dt_code(k.dtype), vt0, dt_code(lay.pdt), ptr(ws_kv), ptr(ws_ks),
             L_loc, lay.per, lay.hg, lay.gb, H, D, stream_ptr())
        call("td_sla_linear_kv_final_packed", ptr(ws_kv), ptr(ws_ks), SLA_NCH, SLA_NCH * D * D, D * D, SLA_NCH * D, D,
             ptr(lin_kv), ptr(lin_ks), L.TD_F32, 0, 0, H, D, stream_ptr())
    return pack


def sp_pack_finish(pack, k, km, L_loc, lay):
    """Second half: what needs the smooth-K mean — ONE launch: the Sage INT8 codes + scales of ``k`` into the group sections
    and the pooled block means (of ALL heads) into the flat all-head section (16-bit K: pooled K only + one strided copy of K).
    km: the mean [H, D] in k's dtype, or (parts f32 [W, H, D], L_total): the gathered per-rank column sums of ``seq_sum`` — the
    mean is then formed inside the kernel (no td_seq_mean_final launch)."""
    require_gpu(pack, k)
    H, L_, D = k.shape
    base = pack.data_ptr()

    def grp(name):     # group section of group 0 (the kernels step lay.gb per group)
        return L.ctypes.c_void_p(base + lay.ab + lay.offs[name])

    if lay.sage or not lay.dense:
        kmp = allp = None
        n_parts = stride = rows = 0
        if isinstance(km, tuple):
            allp, rows = km
            require_gpu(allp)
            assert allp.dtype == torch.float32 and allp.dim() == 3 and tuple(allp.shape[1:]) == (H, D) and allp[0].is_contiguous()
            if allp.shape[0] <= 8:       # the kernel forms the mean itself from up to 8 per-rank partials (one node of MI355X)
                n_parts, stride = allp.shape[0], allp.stride(0)
            else:                        # wider groups (two nodes and more): one td_seq_mean_final launch in rank order, as before round 5
                kmp = seq_mean_final(allp, allp.shape[0], D, allp.stride(0), int(rows), H, D, k.dtype)
                allp, rows = None, 0
        else:
            require_gpu(km)
            kmp = km
        call("td_sage_quant_pool_packed_kmsum", ptr(k), ptr(kmp), ptr(allp), n_parts, stride, int(rows), dt_code(k.dtype), 64,
             L.ctypes.c_void_p(base + lay.aoffs["pk"]) if not lay.dense else None, 0, 0, grp("k") if lay.sage else None,
             grp("ks") if lay.sage else None, L_loc, lay.per, lay.hg, lay.gb, H, D, stream_ptr())
    if not lay.sage:   # 16-bit K travels as it is: one strided copy into the k section
        lay.group_section(pack, "k")[:, :, :L_loc].copy_(k.view(lay.G, lay.hg, L_loc, D))
    return pack


def sp_pack_k_side(k, km, v_src, v_strides, L_loc, lay, lin_kv=None, lin_ks=None):
    """sp_pack_begin + sp_pack_finish (tests; the layer itself issues the early exchange between the two)."""
    H, _, D = k.shape
    if lay.linear and lin_kv is None:
        lin_kv = torch.empty((H, D, D), dtype=torch.float32, device=k.device)
        lin_ks = torch.empty((H, D), dtype=torch.float32, device=k.device)
    return sp_pack_finish(sp_pack_begin(k, v_src, v_strides, L_loc, lay, lin_kv, lin_ks), k, km, L_loc, lay)


# ----------------------------------------------------------------------------- f4: VAE decoder convolutions (vae_conv.hip)
def vae_conv_strided(x, w2d, bias, kt, kh, kw, stride_t=1, stride_hw=1, pad_t=None, pad_hw=None):
    """td_vae_conv_ex: the encoder's down-sampling convolutions.  x [B, T, H, W, Ci] channels-last bf16; LEFT zero padding
    ``pad_t`` frames (default kt - 1: causal) and ``pad_hw`` rows / columns (default k // 2); the output grid is what the
    reference's padded convolutions produce: T' = (T + pad_t - kt) // stride_t + 1, H' = (H + 2 (k // 2) - k) // stride_hw + 1
    for the centred case and H // 2 for ZeroPad2d((0, 1, 0, 1)) + stride 2 (``pad_hw = 0``)."""
    require_gpu(x, w2d, bias)
    B, T, H, W, Ci = x.shape
    Co = w2d.shape[0]
    assert x.dtype == torch.bfloat16 and w2d.dtype == torch.bfloat16 and w2d.shape[1] == kt * kh * kw * Ci and w2d.is_contiguous()
    assert x.stride()[1:] == (H * W * Ci, W * Ci, Ci, 1)
    pt = kt - 1 if pad_t is None else pad_t
    ph, pw = (kh // 2, kw // 2) if pad_hw is None else (pad_hw, pad_hw)
    To = (T + pt - kt) // stride_t + 1
    # right padding: centred -> k // 2; the down-sampler's (0, 1) -> 1 when the left pad is 0 and k = 3
    rh, rw = (kh // 2, kw // 2) if pad_hw is None else ((1, 1) if (pad_hw == 0 and kh == 3) else (pad_hw, pad_hw))
    Ho, Wo = (H + ph + rh - kh) // stride_hw + 1, (W + pw + rw - kw) // stride_hw + 1
    out = torch.empty((B, To, Ho, Wo, Co), dtype=torch.bfloat16, device=x.device)
    call("td_vae_conv_ex", ptr(x), x.stride(0), ptr(w2d), ptr(bias), None, ptr(out), out.stride(0), B, T, H, W, Ci, Co, kt, kh, kw,
         0, 0, To, Ho, Wo, stride_t, stride_hw, pt, ph, pw, stream_ptr())
    return out


def vae_conv(x, w2d, bias, kt, kh, kw, res=None, up2=False, interleave=False, out=None):
    """Channels-last causal convolution on the bf16 matrix pipe.  x [B, T, H, W, Ci] bf16 (only the batch dim may be
    strided), w2d [Co, kt*kh*kw*Ci] bf16 with K ordered (dt, dh, dw, c), bias [Co] bf16 or None, res like the result or
    None.  Returns [B, T, H', W', Co] (H' = 2H with ``up2``), or with ``interleave`` writes [B, 2T, H, W, Co/2] — into
    ``out`` when given (a view whose batch dim may be strided: the time up-sampler writes behind the first frame)."""
    require_gpu(x, w2d, bias, res, out)
    B, T, H, W, Ci = x.shape
    Co = w2d.shape[0]
    assert x.dtype == torch.bfloat16 and w2d.dtype == torch.bfloat16 and w2d.shape[1] == kt * kh * kw * Ci and w2d.is_contiguous()
    assert x.stride()[1:] == (H * W * Ci, W * Ci, Ci, 1), "x must be channels-last contiguous within a batch entry"
    Ho, Wo = (2 * H, 2 * W) if up2 else (H, W)
    shape = (B, 2 * T, H, W, Co // 2) if interleave else (B, T, Ho, Wo, Co)
    if out is None:
        out = torch.empty(shape, dtype=torch.bfloat16, device=x.device)
    assert tuple(out.shape) == shape and out.dtype == torch.bfloat16
    assert out.stride()[1:] == (shape[2] * shape[3] * shape[4], shape[3] * shape[4], shape[4], 1)
    if res is not None:
        assert tuple(res.shape) == shape and res.stride() == out.stride() and res.dtype == torch.bfloat16
    call("td_vae_conv", ptr(x), x.stride(0), ptr(w2d), ptr(bias), ptr(res), ptr(out), out.stride(0), B, T, H, W, Ci, Co,
         kt, kh, kw, int(up2), int(interleave), stream_ptr())
    return out


def vae_chan_rms(x, gamma, silu=True):
    """RMS_norm over the last (channel) axis of a contiguous channels-last bf16 tensor, optional SiLU (wan2pt1.py:69-70 with
    the bf16 path's rounding points)."""
    require_gpu(x, gamma)
    assert x.dtype == torch.bfloat16 and gamma.dtype == torch.bfloat16 and x.is_contiguous() and gamma.numel() == x.shape[-1]
    y = torch.empty_like(x)
    call("td_vae_chan_rms", ptr(x), ptr(gamma), ptr(y), x.numel() // x.shape[-1], x.shape[-1], int(silu), stream_ptr())
    return y


# ----------------------------------------------------------------------------- 16-bit GEMM / softmax / T5 norm (gemm_bf16.hip)
GEMM16_EPI = {"none": 0, "gelu_tanh": 1, "geglu": 2, "gelu_erf": 3}


def _unit_inner(t):
    """last dim contiguous (a size-1 dim may carry any stride)"""
    return t.shape[-1] == 1 or t.stride(-1) == 1


def _pad_k64(t):
    """[.., k] -> [.., ceil64(k)] zero-padded copy when k is not a multiple of 64 or the rows are not 16-byte aligned
    (toy shapes only: every production width here is a multiple of 64)."""
    k = t.shape[-1]
    kp = cdiv(k, 64) * 64
    if kp == k and t.stride(-1) == 1 and t.stride(-2) % 8 == 0 and t.data_ptr() % 16 == 0 and (t.dim() < 3 or t.stride(0) % 8 == 0):
        return t
    out = torch.zeros(t.shape[:-1] + (kp,), dtype=t.dtype, device=t.device)
    out[..., :k] = t
    return out


def gemm_bf16(a, w, bias=None, res=None, epilogue="none", out_dtype=None, out=None):
    """a [m, k] @ w [n, k]^T (+ bias [n]) -> [m, n] (``epilogue="geglu"``: [m, n / 2], w = interleaved gate|fc1 rows,
    ``geglu_interleave``) in the operands' 16-bit dtype or fp32.  Row-strided 2-D views are taken as they are (last dim
    contiguous, 16-byte aligned rows); res [m, n]: x + Linear(a)."""
    require_gpu(a, w, bias, res, out)
    assert a.dim() == 2 and w.dim() == 2 and a.dtype == w.dtype and a.dtype in (torch.bfloat16, torch.float16)
    assert a.shape[1] == w.shape[1], (a.shape, w.shape)
    a, w = _pad_k64(a), _pad_k64(w)
    m, k = a.shape
    n = w.shape[0]
    odt = out_dtype or a.dtype
    n_out = n // 2 if epilogue == "geglu" else n
    if out is None:
        out = torch.empty((m, n_out), dtype=odt, device=a.device)
    assert tuple(out.shape) == (m, n_out) and out.dtype == odt and _unit_inner(out)
    if bias is not None:
        assert bias.dtype == a.dtype and bias.numel() == n and bias.is_contiguous()
    if res is not None:
        assert res.dtype == a.dtype and tuple(res.shape) == (m, n) and _unit_inner(res)
    splits = _splitk(m, n, k) if odt != torch.float32 else 1
    if splits > 1:
        # small m: a batch of K-slices with fp32 partials, then the reduce + epilogue pass (csrc/gemm_bf16.hip "split-K")
        ws = torch.empty((splits, m, n), dtype=torch.float32, device=a.device)
        ks = k // splits
        call("td_gemm_bf16", ptr(a), ptr(w), None, None, ptr(ws), L.dt_code(a.dtype), L.TD_F32, 0, m, n, ks, a.stride(0),
             w.stride(0), n, 0, splits, ks, ks, m * n, 0, stream_ptr())
        call("td_gemm_bf16_splitk_reduce", ptr(ws), splits, ptr(bias), ptr(res), ptr(out), L.dt_code(a.dtype), GEMM16_EPI[epilogue],
             m, n, out.stride(0), 0 if res is None else res.stride(0), stream_ptr())
        return out
    _timed("td_gemm_bf16", (m, n, k), lambda: call(
        "td_gemm_bf16", ptr(a), ptr(w), ptr(bias), ptr(res), ptr(out), L.dt_code(a.dtype), L.dt_code(odt), GEMM16_EPI[epilogue],
        m, n, k, a.stride(0), w.stride(0), out.stride(0), 0 if res is None else res.stride(0), 1, 0, 0, 0, 0, stream_ptr()))
    return out


def _splitk(m, n, k):
    """K-slices for a problem whose 256x256 tiles cannot fill the chip: enough slices for >= ~256 workgroups, each slice at
    least 512 deep and a multiple of 64; 1 = no split."""
    tiles = cdiv(m, 256) * cdiv(n, 256)
    if tiles >= 128 or k < 1024:
        return 1
    s = min(16, max(1, 256 // tiles), k // 512)
    while s > 1 and k % (64 * s):
        s -= 1
    return s


def gemm_bf16_batched(a, b, out_dtype=None, out=None, bias=None):
    """a [B, m, k] x b [B, n, k]^T (+ bias [n]) -> [B, m, n] per batch entry; a, b may be strided views (batch and row
    strides free — batch stride 0 = one operand shared by all entries — last dim contiguous)."""
    require_gpu(a, b, out, bias)
    assert a.dim() == 3 and b.dim() == 3 and a.shape[0] == b.shape[0] and a.shape[2] == b.shape[2] and a.dtype == b.dtype
    a, b = _pad_k64(a), _pad_k64(b)
    B, m, k = a.shape
    n = b.shape[1]
    odt = out_dtype or a.dtype
    if out is None:
        npad = n if odt == torch.float32 else cdiv(n, 8) * 8
        out = torch.empty((B, m, npad), dtype=odt, device=a.device)[:, :, :n]
    assert tuple(out.shape) == (B, m, n) and out.dtype == odt and _unit_inner(out)
    if bias is not None:
        assert bias.dtype == a.dtype and bias.numel() == n and bias.is_contiguous()
    call("td_gemm_bf16", ptr(a), ptr(b), ptr(bias), None, ptr(out), L.dt_code(a.dtype), L.dt_code(odt), 0, m, n, k, a.stride(1),
         b.stride(1), out.stride(1), 0, B, a.stride(0), b.stride(0), out.stride(0), 0, stream_ptr())
    return out


def geglu_interleave(gate_w, fc1_w):
    """[f, k], [f, k] -> [2 f, k] in the row order td_gemm_bf16's gated-GELU epilogue reads: blocks of 32 gate rows followed
    by the 32 fc1 rows of the same output columns (f % 32 == 0)."""
    f, k = gate_w.shape
    assert fc1_w.shape == gate_w.shape and f % 32 == 0
    return torch.stack([gate_w.view(f // 32, 32, k), fc1_w.view(f // 32, 32, k)], dim=1).reshape(2 * f, k).contiguous()


def softmax_rows(s, scale=1.0, bias=None, out=None, out_dtype=None, padded=False):
    """softmax over the last dim of s [rows, cols] (row stride free) -> out [rows, cols] 16-bit whose PADDED row (out's row
    stride) is zero-filled behind ``cols``; bias [bias_rows, cols] 16-bit is added first (rows cycle).  padded: return the
    [rows, ceil64(cols)] storage (what a following GEMM contracts over) instead of the [rows, cols] view."""
    require_gpu(s, bias, out)
    assert s.dim() == 2 and _unit_inner(s)
    rows, cols = s.shape
    pdt = out_dtype or (torch.bfloat16 if s.dtype == torch.float32 else s.dtype)
    full = None
    if out is None:
        ldp = cdiv(cols, 64) * 64
        full = torch.empty((rows, ldp), dtype=pdt, device=s.device)
        out = full[:, :cols]
    assert tuple(out.shape) == (rows, cols) and out.dtype == pdt and _unit_inner(out)
    assert not padded or full is not None
    if bias is not None:
        assert bias.dim() == 2 and bias.dtype == pdt and bias.shape[1] == cols and _unit_inner(bias)
    call("td_softmax_rows", ptr(s), L.dt_code(s.dtype), ptr(out), L.dt_code(pdt), ptr(bias), rows, cols, s.stride(0), out.stride(0),
         0 if bias is None else bias.shape[0], 0 if bias is None else bias.stride(0), float(scale), stream_ptr())
    return full if padded else out


def t5_norm(x, w, eps=1e-6):
    """T5LayerNorm on [rows, n] 16-bit (two roundings, umt5.py:130-142)."""
    require_gpu(x, w)
    assert x.dim() == 2 and _unit_inner(x) and w.dtype == x.dtype and w.numel() == x.shape[1] and w.is_contiguous()
    y = torch.empty((x.shape[0], x.shape[1]), dtype=x.dtype, device=x.device)
    call("td_t5_norm", ptr(x), ptr(w), ptr(y), L.dt_code(x.dtype), float(eps), x.shape[0], x.shape[1], x.stride(0), y.stride(0),
         stream_ptr())
    return y
