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


def reference_wan_from_sd(cfg: dict, sd: dict, act_dtype=None):
    """Reference WanModel (wan2pt1) holding exactly the weights of ``sd`` (reference key names).

    ``act_dtype=torch.bfloat16`` emulates the CUDA run of a bf16 checkpoint on the CPU: CUDA's
    ``amp.autocast("cuda", dtype=float32)`` islands (wan2pt1.py:211,399,405,412,451,671) run
    Linear / LayerNorm in fp32 with the bf16 weights up-cast; CPU autocast cannot target fp32, so
    the parameters those islands touch (time_embedding, time_projection, head.head, norm3.{weight,
    bias}) are KEPT in fp32 (their values are bf16-representable) while everything else is cast to
    bf16 — the arithmetic is then identical.  act_dtype=None keeps a plain fp32 model."""
    import torch

    mod = load("wan2pt1")
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
