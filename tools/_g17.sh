#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-m32}
timeout 300 python tools/m32_check.py > gpurun_out/m32_check_$T.log 2>&1; echo "exit $?" >> gpurun_out/m32_check_$T.log; cat gpurun_out/m32_check_$T.log | cut -c1-700
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k gemm --no-header -p no:cacheprovider -x 2>&1 | tail -5
timeout 200 tools/ubench/peaks > gpurun_out/peaks_$T.log 2>&1; tail -6 gpurun_out/peaks_$T.log
