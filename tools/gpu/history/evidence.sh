#!/bin/bash
# end-of-round evidence: full GPU test suite, smoke, default bench (with CPU baseline), kbench (+ CPU oracle timings), rocprofv3
# kernel-trace stats of bench.py (production schedule AND single-stream), PMC HBM traffic, the other BASELINE.json
# configurations and the opt-in variants, the (opt-in) two-videos-in-flight extra
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-final}; R=$PWD
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 --no-header -p no:cacheprovider > gpurun_out/pytest_$T.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_$T.log; tail -4 gpurun_out/pytest_$T.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke_$T.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke_$T.log; tail -2 gpurun_out/smoke_$T.log
timeout 1200 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_$T.log 2>&1; echo "bench exit $?" >> gpurun_out/bench_$T.log; grep '^\[bench' gpurun_out/bench_$T.log; grep '^{' gpurun_out/bench_$T.log | cut -c1-1800
timeout 900 python tools/kbench.py --iters 10 --cpu > gpurun_out/kbench_$T.log 2>&1
# kernel trace of the production schedule (two streams, token split; eager enqueue so that every launch is a record) ...
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$T -o bench --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graph --no-box-calibration > $R/gpurun_out/prof_$T.log 2>&1)
f=$(find gpurun_out/prof_$T -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_$T.csv
find gpurun_out/prof_$T -name '*kernel_trace*' -size +20M -delete
# ... and single-stream (every kernel's own duration: no second stream, no token split)
(cd /tmp && TD_BENCH_MODEL_FLAGS=two_streams=0,split_tokens=0 timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof1_$T -o bench --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graph --no-box-calibration > $R/gpurun_out/prof1_$T.log 2>&1)
f=$(find gpurun_out/prof1_$T -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_single_stream_$T.csv
find gpurun_out/prof1_$T -name '*kernel_trace*' -size +20M -delete
bash tools/gpu/pmc_hbm_traffic.sh $T > gpurun_out/pmc_$T.log 2>&1; tail -4 gpurun_out/pmc_$T.log
bash tools/gpu/other_configs.sh $T 2>&1 | tail -14
# the serving-style extra (opt-in, single host thread): headline model, then the 14B size that used to hang with two threads
timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --two-in-flight > gpurun_out/two_in_flight_$T.log 2>&1; echo "exit $?" >> gpurun_out/two_in_flight_$T.log
grep -o '"value": [0-9.]*\|"two_videos_in_flight_videos_per_s": [0-9.a-zA-Z"]*' gpurun_out/two_in_flight_$T.log
timeout 400 python bench.py --model Wan2.1-14B --res 720p --steps 1 --warmup 1 --no-cpu-baseline --two-in-flight > gpurun_out/two_in_flight_14b_$T.log 2>&1; echo "exit $?" >> gpurun_out/two_in_flight_14b_$T.log
grep '^\[bench' gpurun_out/two_in_flight_14b_$T.log | tail -3; grep -o '"value": [0-9.]*\|"two_videos_in_flight_videos_per_s": [0-9.a-zA-Z"]*' gpurun_out/two_in_flight_14b_$T.log; tail -1 gpurun_out/two_in_flight_14b_$T.log
