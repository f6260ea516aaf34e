#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-g4}
timeout 600 python tools/gemm_exp.py > gpurun_out/gemm_exp_$T.log 2>&1; grep -v amdgpu.ids gpurun_out/gemm_exp_$T.log
timeout 300 python tools/gemm_trace.py 4 > gpurun_out/trace_$T.log 2>&1; grep -v amdgpu.ids gpurun_out/trace_$T.log
