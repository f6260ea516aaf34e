"""Shared helpers for the parity tests (oracle = checker, never the thing under test)."""
import torch


def ulp_diff_bf16(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Distance in bf16 ulps between two bf16 tensors (monotone integer mapping of the bit patterns)."""
    def key(t):
        i = t.contiguous().view(torch.int16).to(torch.int32) & 0xFFFF
        return torch.where(i >= 0x8000, 0x8000 - i, i)
    return (key(a.cpu()) - key(b.cpu())).abs()


def ulp_diff_f16(a, b):
    return ulp_diff_bf16(a.view(torch.int16).view(torch.bfloat16), b.view(torch.int16).view(torch.bfloat16))


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


def cosine(a, b):
    # fp64: an fp32 dot / norm over the 5e7 elements of a full-size token tensor is good to two digits only
    a, b = a.double().cpu().flatten(), b.double().cpu().flatten()
    return (a @ b / (a.norm() * b.norm()).clamp_min(1e-20)).item()


def act_like(m, n, dtype, seed, outliers=True):
    """N(0,1) activations with a few x20 outlier channels (SURVEY §8d microbench inputs)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(m, n, generator=g)
    if outliers and n >= 16:
        idx = torch.randperm(n, generator=g)[: max(1, n // 1000 + 1)]
        x[:, idx] *= 20.0
    return x.to(dtype)
