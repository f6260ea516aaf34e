#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-g12}
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_wan.py -m gpu -q -x --timeout=600 --no-header -p no:cacheprovider 2>&1 | tail -4
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_$T.log 2>&1
echo "bench exit $?" >> gpurun_out/bench_$T.log; tail -2 gpurun_out/bench_$T.log | cut -c1-330
