"""Offline weight quantiser / checkpoint converter and loader — the MI355X-native
counterpart of ``turbodiffusion/inference/modify_model.py:156-183`` (``__main__``) and ``:128-139`` (``create_model``).

    python -m turbodiffusion_amd.convert --model Wan2.1-1.3B --input_path rcm_sla.pth --output_path TurboWan-quant.pth \\
        --attention_type sagesla --quant_linear

* ``normalize_checkpoint``  : what the reference script does to an rCM training checkpoint before loading it —
  take ``["state_dict"]`` if present, drop the ``net.`` prefix, reshape ``patch_embedding.{weight,bias}``
  (``modify_model.py:161-171``).
* ``quantize_state_dict``   : float state dict -> the frozen ``*-quant.pth`` layout (``Int8Linear.from_linear``,
  ``ops/core.py:415-432``): every ``nn.Linear`` inside ``blocks`` except ``proj_l`` becomes ``int8_weight [out,in]`` +
  ``scale [ceil(out/128), ceil(in/128)]`` (+ ``bias``), quantised per 128x128 block by the HIP quantiser
  (td_quant_i8_block128: amax/128 scale, RNE, saturate — ``ops/quant/quant.hpp:91-98``); norms keep their keys with
  fp32 weights (``FastRMSNorm/FastLayerNorm.from_*``, ``ops/core.py:444-492``).  Needs the MI355X (no CPU fallback).
* ``load_checkpoint``       : load either layout into a ``turbodiffusion_amd.wan.WanModel`` (quantised checkpoints
  with ``load_state_dict``; float ones through the quantiser) — the state-dict keys are the reference's, so published
  TurboWan ``*-quant.pth`` files load unchanged.
"""
from __future__ import annotations

import argparse
from typing import Dict

import torch

from . import kernels as K
from .wan import MODEL_CONFIGS, WanModel, select_model

QUANT_SUFFIXES = (".int8_weight", ".scale")


def normalize_checkpoint(ckpt: Dict, patch_weight_shape=None, patch_bias_shape=None, prefix: str = "net.") -> Dict:
    sd = ckpt["state_dict"] if "state_dict" in ckpt and isinstance(ckpt["state_dict"], dict) else ckpt
    out = {}
    for k, v in sd.items():
        nk = k[len(prefix):] if k.startswith(prefix) else k
        if patch_weight_shape is not None and k.endswith("patch_embedding.weight"):
            v = v.reshape(patch_weight_shape)
        if patch_bias_shape is not None and k.endswith("patch_embedding.bias"):
            v = v.reshape(patch_bias_shape)
        out[nk] = v
    return out


def is_quantized(sd: Dict) -> bool:
    return any(k.endswith(".int8_weight") for k in sd)


def _is_block_linear_weight(key: str, v: torch.Tensor) -> bool:
    """modify_model.replace_linear_norm (:56-81): Linear modules under ``blocks`` whose name does not contain
    ``proj_l`` (skip_layer) are replaced by Int8Linear."""
    return key.startswith("blocks.") and key.endswith(".weight") and v.dim() == 2 and "proj_l" not in key


@torch.no_grad()
def quantize_state_dict(sd: Dict, device="cuda", dtype=torch.bfloat16) -> Dict:
    """Float state dict -> quantised-checkpoint layout.  Non-Linear tensors pass through (norm weights as fp32)."""
    out = {}
    for k, v in sd.items():
        if _is_block_linear_weight(k, v):
            w = v.to(device=device, dtype=dtype).contiguous()
            q, s = K.quant_i8_block128(w)
            out[k[:-7] + ".int8_weight"] = q.cpu()
            out[k[:-7] + ".scale"] = s.cpu()
        elif k.startswith("blocks.") and (".norm" in k) and v.dim() == 1:
            out[k] = v.float()  # FastRMSNorm / FastLayerNorm keep fp32 parameters (ops/core.py:447,470-471)
        else:
            out[k] = v
    return out


@torch.no_grad()
def load_checkpoint(net: WanModel, ckpt: Dict) -> WanModel:
    """Load a float or a quantised (``*-quant.pth``) state dict into ``net`` (already on the GPU)."""
    sd = normalize_checkpoint(ckpt, net.patch_embedding.weight.shape, net.patch_embedding.bias.shape)
    own = net.state_dict()
    if is_quantized(sd) or not net.quant_linear:
        unexpected = [k for k in sd if k not in own]
        missing = [k for k in own if k not in sd]
        if unexpected or missing:
            raise KeyError(f"checkpoint/model mismatch: unexpected {unexpected[:4]}, missing {missing[:4]}")
        net.load_state_dict({k: v.to(own[k].device) for k, v in sd.items()}, assign=False)
        net.invalidate_caches()
    else:
        dev = next(net.parameters()).device
        net.load_from_float_state_dict({k: (v.to(dev).to(own[k].dtype) if k in own else v.to(dev)) for k, v in sd.items()})
    return net


def main():
    ap = argparse.ArgumentParser(description="quantise a Wan DiT checkpoint into the TurboDiffusion *-quant.pth layout")
    ap.add_argument("--model", choices=sorted(MODEL_CONFIGS), default="Wan2.1-1.3B")
    ap.add_argument("--input_path", required=True)
    ap.add_argument("--output_path", required=True)
    ap.add_argument("--attention_type", choices=["sla", "sagesla", "original"], default="original")
    ap.add_argument("--sla_topk", type=float, default=0.2)
    ap.add_argument("--quant_linear", action="store_true")
    args = ap.parse_args()
    with torch.device("meta"):
        net = select_model(args.model, attention_type=args.attention_type, sla_topk=args.sla_topk,
                           quant_linear=args.quant_linear)
    ckpt = torch.load(args.input_path, map_location="cpu", weights_only=True)
    sd = normalize_checkpoint(ckpt, net.patch_embedding.weight.shape, net.patch_embedding.bias.shape)
    if args.quant_linear:
        sd = quantize_state_dict(sd)
    own = set(net.state_dict())
    extra, missing = sorted(set(sd) - own), sorted(own - set(sd))
    if extra or missing:
        raise SystemExit(f"key mismatch after conversion: unexpected {extra[:4]}, missing {missing[:4]}")
    torch.save(sd, args.output_path)
    print(f"wrote {args.output_path}: {len(sd)} tensors")


if __name__ == "__main__":
    main()
