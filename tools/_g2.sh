#!/bin/bash
# GPU visit: parity tests, GEMM variant microbench, bench (graph / eager / GEMM v4)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-g2}
timeout 1200 python -m pytest tests -m gpu -q --timeout=600 --no-header -p no:cacheprovider > gpurun_out/pytest_$T.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_$T.log; tail -8 gpurun_out/pytest_$T.log
timeout 600 python tools/kbench.py --iters 10 --only gemm > gpurun_out/kbench_$T.log 2>&1
echo "kbench exit $?" >> gpurun_out/kbench_$T.log; cat gpurun_out/kbench_$T.log | grep -v amdgpu.ids
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_$T.log 2>&1
echo "bench exit $?" >> gpurun_out/bench_$T.log; tail -3 gpurun_out/bench_$T.log
timeout 600 python bench.py --steps 3 --warmup 1 --gemm-variant 4 --no-cpu-baseline > gpurun_out/bench_${T}_v4.log 2>&1
echo "bench exit $?" >> gpurun_out/bench_${T}_v4.log; tail -3 gpurun_out/bench_${T}_v4.log
