#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-g7}
timeout 1200 python -m pytest tests -m gpu -q -x --timeout=600 --no-header -p no:cacheprovider > gpurun_out/pytest_$T.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_$T.log; tail -12 gpurun_out/pytest_$T.log
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_$T.log 2>&1
echo "bench exit $?" >> gpurun_out/bench_$T.log; tail -2 gpurun_out/bench_$T.log | cut -c1-900
