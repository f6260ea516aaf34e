#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sla.py -m gpu -q -x --no-header -p no:cacheprovider -k "fp8" 2>&1 | tail -15
timeout 600 python tools/kbench.py --only attn --iters 20 2>&1 | grep -v amdgpu.ids
