#!/bin/bash
# round 2, call 1: test suite at HEAD + fast-dequant experiment + baseline bench on this round's box
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x --timeout=600 --no-header -p no:cacheprovider > gpurun_out/r02c1_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r02c1_pytest.log; tail -15 gpurun_out/r02c1_pytest.log
timeout 900 python tools/gemm_fast_exp.py --iters 10 > gpurun_out/r02c1_gemm_fast.log 2>&1; echo "exit $?" >> gpurun_out/r02c1_gemm_fast.log; tail -5 gpurun_out/r02c1_gemm_fast.log | cut -c1-600
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-two-in-flight > gpurun_out/r02c1_bench.log 2>&1; tail -1 gpurun_out/r02c1_bench.log | cut -c1-800
