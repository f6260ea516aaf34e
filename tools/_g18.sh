#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-m32}
timeout 200 python tools/gemm_trace.py 5 > gpurun_out/trace5_$T.log 2>&1; cat gpurun_out/trace5_$T.log
timeout 200 python tools/gemm_trace.py 4 > gpurun_out/trace4_$T.log 2>&1; head -4 gpurun_out/trace4_$T.log
