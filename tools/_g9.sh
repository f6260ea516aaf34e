#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-g9}
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --timeout=600 --no-header -p no:cacheprovider -k "gemm" 2>&1 | tail -5
timeout 300 python tools/gemm_phases.py 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/phases_$T.log
timeout 600 python tools/kbench.py --iters 10 --only gemm 2>&1 | grep -v amdgpu.ids | tee gpurun_out/kbench_$T.log
