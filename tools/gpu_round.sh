#!/bin/bash
# One GPU-box visit: parity tests, kernel microbench, (optional) rocprof. Everything is bounded by `timeout`.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r}
timeout 1200 python -m pytest tests -m gpu -q --timeout=600 --no-header -p no:cacheprovider > gpurun_out/pytest_$TAG.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_$TAG.log
tail -40 gpurun_out/pytest_$TAG.log
timeout 900 python tools/kbench.py --iters 5 > gpurun_out/kbench_$TAG.log 2>&1
echo "kbench exit $?" >> gpurun_out/kbench_$TAG.log
tail -30 gpurun_out/kbench_$TAG.log
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/bench_$TAG.log 2>&1
echo "bench exit $?" >> gpurun_out/bench_$TAG.log
tail -5 gpurun_out/bench_$TAG.log
