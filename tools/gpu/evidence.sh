#!/bin/bash
# end-of-round evidence: full GPU test suite, default bench (with CPU baseline), kbench (+ CPU oracle timings), rocprofv3
# kernel-trace stats and PMC HBM traffic of bench.py, the other BASELINE.json configurations and the opt-in variants
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-final}; R=$PWD
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 --no-header -p no:cacheprovider > gpurun_out/pytest_$T.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_$T.log; tail -3 gpurun_out/pytest_$T.log
timeout 1200 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_$T.log 2>&1; echo "bench exit $?" >> gpurun_out/bench_$T.log; tail -2 gpurun_out/bench_$T.log | cut -c1-1800
timeout 900 python tools/kbench.py --iters 10 --cpu > gpurun_out/kbench_$T.log 2>&1
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$T -o bench --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graph --no-two-in-flight > $R/gpurun_out/prof_$T.log 2>&1)
f=$(find gpurun_out/prof_$T -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_$T.csv
find gpurun_out/prof_$T -name '*kernel_trace*' -size +20M -delete
bash tools/gpu/pmc_hbm_traffic.sh $T > gpurun_out/pmc_$T.log 2>&1; tail -4 gpurun_out/pmc_$T.log
bash tools/gpu/other_configs.sh $T 2>&1 | tail -12
