#!/bin/bash
# round 6, call 1: the four-wave GEMM form + run-time one-VALU dequant — bit-identity tests, then us + J per launch of
# the block's six GEMMs on every form (tools/gemm_forms.py), at C1's rows and at an eighth of them
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r06a}
timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --timeout=900 --no-header -p no:cacheprovider -k "gemm" > gpurun_out/pytest_gemm_$T.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gemm_$T.log
grep -v amdgpu gpurun_out/pytest_gemm_$T.log | tail -15
timeout 900 python tools/gemm_forms.py --forms 4,8 --fast 1,8 > gpurun_out/gemm_forms_$T.txt 2>&1; echo "exit $?" >> gpurun_out/gemm_forms_$T.txt
tail -40 gpurun_out/gemm_forms_$T.txt
timeout 600 python tools/gemm_forms.py --rows 4096 --forms 0,8 --fast 1,8 --seconds 0.5 > gpurun_out/gemm_forms_m4096_$T.txt 2>&1; echo "exit $?" >> gpurun_out/gemm_forms_m4096_$T.txt
tail -30 gpurun_out/gemm_forms_m4096_$T.txt
