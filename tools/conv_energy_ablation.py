"""Time and joules of one 3x3x3 convolution of the VAE decoder's 96-channel level (480p clip) for the library selected by
TD_LIB_PATH: the production build or ablated builds of csrc/vae_conv.hip (no MFMA / no LDS fragment reads / no global
gather; patched copies built with tools/build_variant.sh — results are garbage, only time and energy are read)."""
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
    return float(re.search(r"Accumulated Energy \(uJ\):\s*([0-9.]+)", out).group(1))


g = torch.Generator(device="cuda").manual_seed(1)
T, H, W, Ci, Co = 81, 480, 832, 96, 96
x = torch.randn(1, T, H, W, Ci, device="cuda", generator=g, dtype=torch.bfloat16)
w = (torch.randn(Co, 27 * Ci, device="cuda", generator=g) / (27 * Ci) ** 0.5).bfloat16()
b = torch.zeros(Co, device="cuda", dtype=torch.bfloat16)
fn = lambda: K.vae_conv(x, w, b, 3, 3, 3)   # noqa: E731
for _ in range(3):
    fn()
torch.cuda.synchronize()
time.sleep(0.5)
reps = 60
e0, t0 = energy_uj(), time.perf_counter()
for _ in range(reps):
    fn()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
e1 = energy_uj()
tag = os.path.basename(os.environ.get("TD_LIB_PATH", "production")).replace("libtd_cabl_", "").replace(".so", "")
print(f"{tag:16s} {dt / reps * 1e3:7.2f} ms  {(e1 - e0) * 1e-6 / dt:6.0f} W  {(e1 - e0) * 1e-6 / reps:7.3f} J/launch", flush=True)
