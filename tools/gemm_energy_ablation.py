"""Where do a W8A8 GEMM launch's time and joules go?  The ffn.2 GEMM (M 32760, N 1536, K 8960) timed and metered
(socket energy counter) for the production library and for ablated builds of csrc/gemm_w8a8_fi.hip in which one
ingredient of the main loop is compiled out (results are garbage; only time and energy are read):
    nodeq = no dequant VALU (v_add + v_fmac), nomfma = no MFMA, nolds = no LDS fragment reads (ds_read_b128),
    nodeq_nolds = MFMA + LDS-DMA staging + barriers only, nomfma_nodeq = staging + fragment reads only.
The ablated libraries are built by tools/gpu/gemm_ablation.sh (tools/build_variant.sh on patched copies of the source).
One line per library: called as  python tools/gemm_energy_ablation.py  with TD_LIB_PATH set per run."""
import math
import os
import re
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from turbodiffusion_amd import kernels as K  # noqa: E402


def energy_uj():
    out = subprocess.run(["rocm-smi", "--showenergycounter"], capture_output=True, text=True).stdout
    m = re.search(r"Accumulated Energy \(uJ\):\s*([0-9.]+)", out)
    return float(m.group(1)) if m else float("nan")


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    L, dim, ffn = 32760, 1536, 8960
    a = torch.randn(L, ffn, device=dev).bfloat16()
    aq, as_ = K.quant_i8_block128(a)
    wq, ws = K.quant_i8_block128((torch.randn(dim, ffn, device=dev) / math.sqrt(ffn)).bfloat16())
    b = torch.zeros(dim, device=dev).bfloat16()
    fn = lambda: K.gemm_w8a8(aq, as_, wq, ws, torch.bfloat16, bias=b)   # noqa: E731
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    reps = 3000
    time.sleep(0.5)
    e0 = energy_uj()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    e1 = energy_uj()
    tag = os.path.basename(os.environ.get("TD_LIB_PATH", "production")).replace("libtd_abl_", "").replace(".so", "")
    print(f"{tag:16s} {dt / reps * 1e6:8.1f} us  {(e1 - e0) * 1e-6 / dt:6.0f} W  {(e1 - e0) * 1e-6 / reps:7.4f} J/launch", flush=True)


if __name__ == "__main__":
    main()
