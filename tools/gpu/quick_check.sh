#!/bin/bash
# a short confidence run after a small change: the model-level / round-3 GPU tests, one headline bench, kernel times of the f3 kernels
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-quick}; R=$PWD
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 --timeout=600 --no-header -p no:cacheprovider > gpurun_out/pytest_$T.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_$T.log; grep -v amdgpu gpurun_out/pytest_$T.log | tail -12
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$T.log 2>&1; echo "bench exit $?" >> gpurun_out/bench_$T.log
grep '^\[bench' gpurun_out/bench_$T.log | head -3; grep '^{' gpurun_out/bench_$T.log | cut -c1-330
(cd /tmp && TD_BENCH_MODEL_FLAGS=two_streams=0,split_tokens=0 timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$T -o k --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graph --no-box-calibration > $R/gpurun_out/prof_$T.log 2>&1)
f=$(find gpurun_out/prof_$T -name '*kernel_stats.csv' | head -1); grep -E "head_kernel|patch_embed|row_stats_finalize|gemv_rows|bcast_add|time_sinusoid" $f | cut -c1-60,100-200 | head
find gpurun_out/prof_$T -name '*kernel_trace*' -size +20M -delete
