#!/bin/bash
# One GPU-box visit: parity tests, kernel microbench, bench, rocprofv3 kernel trace. Everything is bounded by `timeout`.
# usage: tools/gpu_round.sh TAG [tests|notests] [prof|noprof]
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-r}
if [ "${2:-tests}" = "tests" ]; then
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 --no-header -p no:cacheprovider > gpurun_out/pytest_$TAG.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_$TAG.log
tail -15 gpurun_out/pytest_$TAG.log
fi
timeout 900 python tools/kbench.py --iters 5 > gpurun_out/kbench_$TAG.log 2>&1
echo "kbench exit $?" >> gpurun_out/kbench_$TAG.log
tail -40 gpurun_out/kbench_$TAG.log
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/bench_$TAG.log 2>&1
echo "bench exit $?" >> gpurun_out/bench_$TAG.log
tail -5 gpurun_out/bench_$TAG.log
if [ "${3:-prof}" = "prof" ]; then
R=$PWD
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_$TAG.log 2>&1)
echo "rocprof exit $?" >> gpurun_out/prof_$TAG.log
tail -3 gpurun_out/prof_$TAG.log
find gpurun_out/prof_$TAG -name '*kernel_stats*' | head
find gpurun_out/prof_$TAG -name '*kernel_trace*' -size +20M -delete
f=$(find gpurun_out/prof_$TAG -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -40 "$f"
fi
