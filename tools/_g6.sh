#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-g6}; R=$PWD
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -3
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$T -o bench --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graph > $R/gpurun_out/prof_$T.log 2>&1)
echo "rocprof exit $?"; tail -2 gpurun_out/prof_$T.log
f=$(find gpurun_out/prof_$T -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_$T.csv && head -45 "$f"
find gpurun_out/prof_$T -name '*kernel_trace*' -size +20M -delete
