import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "shipping_gemm: run with the library's DEFAULT W8A8 dequant (round 6: the one-VALU form) — "
                                       "model-level parity tests; every other GPU test pins the exact arithmetic of the reference")


@pytest.fixture(autouse=True)
def _gemm_dequant_mode(request):
    """Round 6: the library's default W8A8 dequant is the one-VALU form (bounded difference, csrc/capi.hip).  The
    operator-level tests state bit-exactness against the oracle's restatement of the reference arithmetic and bit-identity
    between kernel variants (the 128x128 kernel and small problems are exact-only): they run with the EXACT form selected.
    Tests marked ``shipping_gemm`` (the full-size model parity tests, the sampler, smoke-level checks) run what ships."""
    if "gpu" not in request.keywords:
        yield
        return
    import torch
    if not torch.cuda.is_available():
        yield
        return
    from turbodiffusion_amd import kernels as K
    K.set_tuning(K.TUNE_GEMM_FAST, 0 if request.node.get_closest_marker("shipping_gemm") else 1)
    try:
        yield
    finally:
        K.set_tuning(K.TUNE_GEMM_FAST, 0)


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
