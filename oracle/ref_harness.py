"""Import the REAL reference (``/root/reference``) on the CPU, unmodified, through shims.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Only usable in the build container
(``/root/reference`` does not exist on the GPU box); used by ``oracle/make_golden.py``
to generate ``tests/golden/*.pt`` and by the ``not gpu`` tests that pin the oracle
restatements against the reference.

Shims (none of them edits a reference source file):
  1. ``imaginaire.utils.log`` needs loguru, ``imaginaire.utils.distributed`` needs
     pynvml -> stub modules pre-seeded in ``sys.modules`` (only ``log.*`` and
     ``distributed.get_rank`` are touched by the hot path: context_parallel.py:22).
  2. ``VideoRopePosition3DEmb.cache_parameters`` calls ``.cuda()`` three times
     (wan2pt1.py:81-83) -> replaced by the same code without ``.cuda()``.
  3. flash-attn is absent so ``flash_apply_rotary_emb`` is None (wan2pt1.py:26-30) ->
     supplied with the interleaved-pair rotation flash-attn implements
     (``interleaved=True``: (x0,x1) -> (x0*c - x1*s, x0*s + x1*c), wan2pt1.py:176).
  4. ``rcm.utils.attention._get_sdpa_config`` (attention.py:91-117) selects CUDA-only
     SDPA backends -> overridden to the CPU flash/math backends.
"""
from __future__ import annotations

import os
import sys
import types
import importlib

REF_ROOT = os.environ.get("TD_REFERENCE_ROOT", "/root/reference")
REF_PKG = os.path.join(REF_ROOT, "turbodiffusion")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_PKG, "rcm", "networks"))


def _stub_modules():
    if "imaginaire.utils.log" in sys.modules and getattr(sys.modules["imaginaire.utils.log"], "_td_stub", False):
        return
    imaginaire = types.ModuleType("imaginaire")
    imaginaire.__path__ = []  # mark as package
    utils = types.ModuleType("imaginaire.utils")
    utils.__path__ = []
    log = types.ModuleType("imaginaire.utils.log")
    log._td_stub = True
    for name in ("info", "debug", "warning", "error", "critical", "success", "trace"):
        setattr(log, name, lambda *a, **k: None)
    dist = types.ModuleType("imaginaire.utils.distributed")
    dist.get_rank = lambda *a, **k: 0
    dist.get_world_size = lambda *a, **k: 1
    imaginaire.utils = utils
    utils.log = log
    utils.distributed = dist
    sys.modules.setdefault("imaginaire", imaginaire)
    sys.modules.setdefault("imaginaire.utils", utils)
    sys.modules["imaginaire.utils.log"] = log
    sys.modules["imaginaire.utils.distributed"] = dist


_loaded = {}


def load(which: str = "wan2pt1"):
    """Returns the reference module ``rcm.networks.<which>`` with the shims applied."""
    if which in _loaded:
        return _loaded[which]
    if not available():
        raise RuntimeError(f"reference not present at {REF_ROOT}")
    import torch

    _stub_modules()
    if REF_PKG not in sys.path:
        sys.path.insert(0, REF_PKG)
    mod = importlib.import_module(f"rcm.networks.{which}")

    # shim 2
    def cache_parameters(self):
        if self._is_initialized:
            return
        dim_h, dim_t = self._dim_h, self._dim_t
        self.seq = torch.arange(max(self.max_h, self.max_w, self.max_t)).float()
        self.dim_spatial_range = torch.arange(0, dim_h, 2)[: (dim_h // 2)].float() / dim_h
        self.dim_temporal_range = torch.arange(0, dim_t, 2)[: (dim_t // 2)].float() / dim_t
        self._is_initialized = True

    mod.VideoRopePosition3DEmb.cache_parameters = cache_parameters

    # shim 3
    def apply_rotary_emb(x, cos, sin, interleaved=True, inplace=False):
        assert interleaved and not inplace
        b, l, h, d = x.shape
        xr = x.reshape(b, l, h, d // 2, 2)
        x0, x1 = xr[..., 0], xr[..., 1]
        c = cos[None, :, None, :]
        s = sin[None, :, None, :]
        return torch.stack([x0 * c - x1 * s, x0 * s + x1 * c], dim=-1).reshape(b, l, h, d)

    mod.flash_apply_rotary_emb = apply_rotary_emb

    # shim 4
    attn = importlib.import_module("rcm.utils.attention")
    from torch.nn.attention import SDPBackend, sdpa_kernel
    from functools import partial

    def _get_sdpa_config(device, is_half):
        backends = [SDPBackend.FLASH_ATTENTION, SDPBackend.MATH]
        try:
            sdpa_kernel(backends=backends, set_priority_order=True)
            kern = partial(sdpa_kernel, set_priority_order=True)
        except TypeError:
            kern = sdpa_kernel
        return (0, backends, kern)

    attn._get_sdpa_config = _get_sdpa_config
    _loaded[which] = mod
    return mod


def load_aux(which: str):
    """The reference's VAE (``"tokenizers.wan2pt1"``: rcm/tokenizers/wan2pt1.py) or umT5 encoder (``"utils.umt5"``:
    rcm/utils/umt5.py), unmodified, for the f4 pins.  Extra shims, none touching a reference file: ``sync_model_states``,
    ``easy_io`` and ``misc`` of the absent imaginaire package and the absent ``ftfy`` (text cleaning only) as stubs;
    ``torch.cuda.current_device`` answers 0 while umt5.py is imported (a default ARGUMENT calls it, umt5.py:484)."""
    key = "aux:" + which
    if key in _loaded:
        return _loaded[key]
    if not available():
        raise RuntimeError(f"reference not present at {REF_ROOT}")
    import torch

    _stub_modules()
    dist = sys.modules["imaginaire.utils.distributed"]
    dist.sync_model_states = lambda *a, **k: None
    dist.is_rank0 = lambda *a, **k: True
    for name in ("easy_io", "misc"):
        full = "imaginaire.utils." + name
        if full not in sys.modules:
            m = types.ModuleType(full)
            m.__path__ = []
            sys.modules[full] = m
            setattr(sys.modules["imaginaire.utils"], name, m)
    eio = types.ModuleType("imaginaire.utils.easy_io.easy_io")
    eio.load = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("no checkpoints in the pins"))
    sys.modules["imaginaire.utils.easy_io"].easy_io = eio
    sys.modules["imaginaire.utils.easy_io.easy_io"] = eio
    if "ftfy" not in sys.modules:
        ft = types.ModuleType("ftfy")
        ft.fix_text = lambda t: t
        sys.modules["ftfy"] = ft
    if REF_PKG not in sys.path:
        sys.path.insert(0, REF_PKG)
    cur = torch.cuda.current_device
    torch.cuda.current_device = lambda: 0
    try:
        mod = importlib.import_module("rcm." + which)
    finally:
        torch.cuda.current_device = cur
    _loaded[key] = mod
    return mod


def load_sla():
    """The reference SLA package (imports Triton; only the pure-torch parts run on CPU)."""
    if not available():
        raise RuntimeError(f"reference not present at {REF_ROOT}")
    if REF_PKG not in sys.path:
        sys.path.insert(0, REF_PKG)
    return importlib.import_module("SLA")


def make_reference_wan(cfg: dict, seed: int = 0, dtype=None):
    """Construct the reference WanModel (wan2pt1) on CPU with seeded weights following
    BASELINE.md §3: the reference's own init_weights, then head.head.weight ~ N(0,0.02)
    (it is zero-initialised, wan2pt1.py:762-764) so the output is not identically 0."""
    import torch

    mod = load("wan2pt1")
    torch.manual_seed(seed)
    net = mod.WanModel(**cfg)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        net.head.head.weight.normal_(0, 0.02, generator=g)
        net.head.head.bias.normal_(0, 0.02, generator=g)
        # init_weights zeroes every bias; give them small seeded values so bias paths count
        for name, p in net.named_parameters():
            if name.endswith(".bias") and "head.head" not in name:
                p.normal_(0, 0.02, generator=g)
    net.eval()
    if dtype is not None:
        net = net.to(dtype)
    return net


FP32_ISLANDS = ("time_embedding", "time_projection", "head.head", "norm3")


def reference_wan_from_sd(cfg: dict, sd: dict, act_dtype=None, which: str = "wan2pt1"):
    """Reference WanModel (wan2pt1) holding exactly the weights of ``sd`` (reference key names).

    ``act_dtype=torch.bfloat16`` emulates the CUDA run of a bf16 checkpoint on the CPU: CUDA's
    ``amp.autocast("cuda", dtype=float32)`` islands (wan2pt1.py:211,399,405,412,451,671) run
    Linear / LayerNorm in fp32 with the bf16 weights up-cast; CPU autocast cannot target fp32, so
    the parameters those islands touch (time_embedding, time_projection, head.head, norm3.{weight,
    bias}) are KEPT in fp32 (their values are bf16-representable) while everything else is cast to
    bf16 — the arithmetic is then identical.  act_dtype=None keeps a plain fp32 model."""
    import torch

    mod = load(which)      # "wan2pt2": Wan2.2 (I2V conditions through y only, plain text cross-attention, wan2pt2.py:282,581-645)
    ref_cfg = {k: v for k, v in cfg.items() if k in (
        "model_type", "patch_size", "text_len", "in_dim", "dim", "ffn_dim", "freq_dim", "text_dim", "out_dim",
        "num_heads", "num_layers", "qk_norm", "cross_attn_norm", "eps")}
    net = mod.WanModel(**ref_cfg)
    own = net.state_dict()
    load_sd = {k: v for k, v in sd.items() if k in own}
    missing = [k for k in own if k not in load_sd]
    assert not missing, f"missing {missing[:4]}"
    net.load_state_dict(load_sd)
    net.eval()
    if act_dtype is not None:
        for name, p in net.named_parameters():
            clean = name.replace("_checkpoint_wrapped_module.", "")
            if not any(isl in clean for isl in FP32_ISLANDS):
                p.data = p.data.to(act_dtype)
    return net


# --------------------------------------------------------------------------- #
# the reference SLA / SageSLA modules on the CPU, with only their GPU leaves patched
# --------------------------------------------------------------------------- #
class _PatchedSLA:
    """Context manager: the reference ``SLA`` package (SLA/core.py, SLA/utils.py) runnable on the CPU.

    The module ``forward``s and ``get_block_map`` are pure torch except for leaf calls into Triton / SpargeAttn /
    CUDA; exactly those leaves are replaced (nothing in /root/reference is edited):
      * ``SLA.utils.mean_pool`` (Triton ``compress_kernel``, SLA/utils.py:21-52) -> a torch block mean with the same
        arithmetic (fp32 sum over the valid rows / valid count, cast to the input dtype);
      * ``SLA.core._attention.apply`` (Triton ``_attn_fwd``, SLA/kernel.py:21-82,240-274) -> ``sla_ref.sla_sparse_attn``;
      * ``SLA.core.get_cuda_arch`` (``torch.cuda.get_device_capability``, SLA/utils.py:70-72) -> ``"sm80"`` (the
        FP16-PV branch, SLA/core.py:211-216) or ``"sm89"`` (the FP8-PV branch, :217-239);
      * the un-vendored SpargeAttn entry points the Sage module calls (SLA/core.py:22-24,201-204,214,221-237):
        ``get_vanilla_qk_quant``, ``block_map_lut_triton``, ``qattn.qk_int8_sv_f16_*``, ``fused.*`` and the FP8 kernels
        -> the oracle's statement of that arithmetic (``sla_ref``; PARITY UNPINNED for these leaves, oracle/__init__.py);
      * ``torch.amp.autocast('cuda', ...)`` (SLA/core.py:116,255) -> the same autocast on ``'cpu'`` (the tensors live
        there), so ``proj_l`` (fp32 nn.Linear) runs in the module dtype exactly as under CUDA autocast.
    Everything else — the [B,L,H,D] transposes, dtype casts, block-map call, smooth-K mean, linear branch, the 16-bit
    ``o_s + o_l`` and ``return_sparsity`` — is the reference's own code."""

    def __init__(self, arch: str = "sm80"):
        self.arch = arch
        self._saved = []
        self.recorded = {}   # what the reference module handed its sparse-attention leaf on the last call

    def _set(self, obj, name, val):
        self._saved.append((obj, name, getattr(obj, name, _MISSING)))
        setattr(obj, name, val)

    def __enter__(self):
        import torch
        from . import sla_ref as S

        sla = load_sla()
        core = importlib.import_module("SLA.core")
        utils = importlib.import_module("SLA.utils")
        arch = self.arch

        self._set(utils, "mean_pool", lambda x, BLK: S.mean_pool(x, BLK))

        rec = self.recorded

        class _Attn:
            @staticmethod
            def apply(q, k, v, sparse_map, lut, real_topk, BLKQ, BLKK, qk_scale=None):
                rec.update(sparse_map=sparse_map.clone(), lut=lut.clone(), real_topk=real_topk)
                return S.sla_sparse_attn(q, k, v, lut, BLKQ, BLKK, qk_scale)

        self._set(core, "_attention", _Attn)
        self._set(core, "get_cuda_arch", lambda idx: arch)
        self._set(core, "SAGESLA_ENABLED", True)
        self._set(core, "SAGE2PP_ENABLED", False)

        def get_vanilla_qk_quant(q, k, km, blkq, blkk):
            q8, qs = S.quant_per_block_int8(q, blkq)
            k8, ks = S.quant_per_block_int8(k, blkk, km)
            return q8, qs, k8, ks

        def block_map_lut_triton(sparse_map):
            return S.block_map_lut(sparse_map)

        class _QAttn:
            @staticmethod
            def qk_int8_sv_f16_accum_f16_block_sparse_attn_inst_buf_with_pv_threshold(
                    q8, k8, v16, o, lut, nvalid, pvthr, qs, ks, layout, causal, gran, scale, ret_lse):
                assert layout == 1 and not causal and gran == 1 and ret_lse == 0 and float(pvthr.min()) >= 1e6
                o.copy_(S.sage_sparse_attn(q8, qs, k8, ks, v16, S.lut_from_delta(lut), 128, 64, sm_scale=scale,
                                           out_dtype=o.dtype, pv_dtype=torch.float16, nvalid=nvalid))

            @staticmethod
            def qk_int8_sv_f8_accum_f32_block_sparse_attn_inst_buf_fuse_v_scale_with_pv_threshold(
                    q8, k8, v8, o, lut, nvalid, pvthr, qs, ks, vs, layout, causal, gran, scale, ret_lse):
                assert layout == 1 and not causal and gran == 1 and ret_lse == 0 and float(pvthr.min()) >= 1e6
                o.copy_(S.sage_sparse_attn_fp8(q8, qs, k8, ks, v8, vs, S.lut_from_delta(lut), 128, 64, sm_scale=scale,
                                               out_dtype=o.dtype, nvalid=nvalid))

        class _Fused:
            @staticmethod
            def transpose_pad_permute_cuda(v, vt, layout):
                assert layout == 1
                S.transpose_pad_permute(v, vt)

            @staticmethod
            def scale_fuse_quant_cuda(vt, v8, vs, kv_len, scale_max, layout):
                assert layout == 1
                q, s = S.v_fp8_quant(vt, kv_len, scale_max)
                v8.copy_(q)
                vs.copy_(s)

        self._set(core, "get_vanilla_qk_quant", get_vanilla_qk_quant)
        self._set(core, "block_map_lut_triton", block_map_lut_triton)
        self._set(core, "qattn", _QAttn)
        self._set(core, "fused", _Fused)

        real_autocast = torch.amp.autocast

        def cpu_autocast(device_type, *a, **k):
            return real_autocast("cpu" if device_type == "cuda" else device_type, *a, **k)

        self._set(torch.amp, "autocast", cpu_autocast)
        sla._td_recorded = self.recorded
        return sla

    def __exit__(self, *exc):
        for obj, name, old in reversed(self._saved):
            if old is _MISSING:
                delattr(obj, name)
            else:
                setattr(obj, name, old)
        self._saved.clear()
        return False


_MISSING = object()


def patched_sla(arch: str = "sm80"):
    return _PatchedSLA(arch)
