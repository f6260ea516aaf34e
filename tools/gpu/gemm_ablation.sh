#!/bin/bash
# time and joules of the ffn.2 GEMM for the production library and the ablated builds turbodiffusion_amd/libtd_abl_*.so
# (built beforehand, on the build host, by patching macro definitions of csrc/gemm_w8a8_fi.hip and tools/build_variant.sh)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
O=gpurun_out/gemm_energy_ablation.txt; : > $O
for r in 1 2; do
  for lib in "" nodeq nomfma nolds nodeq_nolds nomfma_nodeq; do
    if [ -z "$lib" ]; then unset TD_LIB_PATH; else export TD_LIB_PATH=$PWD/turbodiffusion_amd/libtd_abl_$lib.so; fi
    timeout 120 python tools/gemm_energy_ablation.py 2>&1 | grep -v amdgpu >> $O
  done
done
cat $O
