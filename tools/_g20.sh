#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-lnq}
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "layernorm" --no-header -p no:cacheprovider -x 2>&1 | tail -4
timeout 300 python tools/kbench.py --only norm,quant --iters 20 2>&1 | grep -v amdgpu.ids | cut -c1-220
