#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-sm}
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "small_problem or gemm" --no-header -p no:cacheprovider 2>&1 | grep -v amdgpu | tail -8
timeout 600 python tools/gemm_small_m.py > gpurun_out/gemm_small_m_$T.jsonl 2> gpurun_out/gemm_small_m_$T.err; grep -v amdgpu gpurun_out/gemm_small_m_$T.err | tail -3; cat gpurun_out/gemm_small_m_$T.jsonl
