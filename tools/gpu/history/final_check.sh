#!/bin/bash
# the short end-of-round pass (evidence.sh without the PMC passes, the other configurations and kbench): GPU test suite,
# smoke, the default bench line (with CPU baseline and box calibration), the two rocprofv3 kernel-trace summaries, the
# multi-rank rig
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-final}; R=$PWD
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 --no-header -p no:cacheprovider > gpurun_out/pytest_$T.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_$T.log; grep -v "amdgpu\|MIOpen" gpurun_out/pytest_$T.log | tail -6
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke_$T.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke_$T.log; tail -2 gpurun_out/smoke_$T.log
timeout 1200 python bench.py > gpurun_out/bench_$T.log 2>&1; echo "bench exit $?" >> gpurun_out/bench_$T.log; grep '^\[bench' gpurun_out/bench_$T.log | head -4; grep '^{' gpurun_out/bench_$T.log | cut -c1-600
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$T -o bench --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graph --no-box-calibration > $R/gpurun_out/prof_$T.log 2>&1)
f=$(find gpurun_out/prof_$T -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_$T.csv
find gpurun_out/prof_$T -name '*kernel_trace*' -size +20M -delete
(cd /tmp && TD_BENCH_MODEL_FLAGS=two_streams=0,split_tokens=0 timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof1_$T -o bench --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graph --no-box-calibration > $R/gpurun_out/prof1_$T.log 2>&1)
f=$(find gpurun_out/prof1_$T -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_single_stream_$T.csv
find gpurun_out/prof1_$T -name '*kernel_trace*' -size +20M -delete
head -8 gpurun_out/kernel_stats_single_stream_$T.csv | cut -c1-70,100-160
bash tools/gpu/multirank_rig.sh $T 2>&1 | grep -o '"value": [0-9.]*, "unit": "videos/s", "n_gpus": [0-9]*\|exit [0-9]*'
