#!/bin/bash
# round 3, call 2: the whole GPU suite at the new tree, the two-in-flight hang experiment, the headline bench with the box
# calibration + a fresh prompt per video
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r03b}
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --timeout=900 --no-header -p no:cacheprovider > gpurun_out/pytest_$T.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_$T.log; tail -25 gpurun_out/pytest_$T.log
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$T.log 2>&1; echo "bench exit $?" >> gpurun_out/bench_$T.log
grep '^\[bench' gpurun_out/bench_$T.log; grep '^{' gpurun_out/bench_$T.log | cut -c1-2500
TD_BENCH_SAME_TEXT=1 timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-box-calibration > gpurun_out/bench_sametext_$T.log 2>&1
grep '^{' gpurun_out/bench_sametext_$T.log | cut -c1-300
bash tools/gpu/ab_flag.sh split_tokens 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_split_tokens_$T.txt
for ts in 0 1; do
  timeout 400 python tools/two_in_flight_hang.py --two-streams $ts --trials 3 --limit 60 > gpurun_out/hang_${T}_$ts.log 2>&1; echo "two_streams=$ts exit $?" >> gpurun_out/hang_${T}_$ts.log
  grep -v amdgpu.ids gpurun_out/hang_${T}_$ts.log | tail -5
done
