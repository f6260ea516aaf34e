#!/bin/bash
# round 3, first call: (1) the reference's Triton leaves on the MI355X (oracle/triton_leaves.py), (2) the 1-rank RCCL
# sequence-parallel tests, (3) the two 720p configurations that hit the 600-s limit at the end of round 2, now with phase
# stamps + a Python stack watchdog on stderr, (4) the headline bench on the same box
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r03a}
timeout 300 python -m oracle.triton_leaves run > gpurun_out/triton_leaves_$T.log 2>&1; echo "triton leaves exit $?" | tee -a gpurun_out/triton_leaves_$T.log
tail -3 gpurun_out/triton_leaves_$T.log
timeout 400 python -m pytest tests/test_gpu_seqpar.py -q -k rccl --no-header -p no:cacheprovider > gpurun_out/pytest_rccl_$T.log 2>&1; echo "pytest rccl exit $?" | tee -a gpurun_out/pytest_rccl_$T.log
tail -15 gpurun_out/pytest_rccl_$T.log
export TD_BENCH_WATCHDOG_S=240
i=0
for a in "--model Wan2.1-14B --res 720p" "--model Wan2.2-A14B --res 720p --two-experts"; do
  i=$((i+1)); log=gpurun_out/cfg720_${T}_$i.log; t0=$(date +%s)
  timeout 540 python bench.py $a --steps 1 --warmup 1 --no-cpu-baseline > $log 2>&1; rc=$?
  echo "[$a] exit $rc after $(( $(date +%s) - t0 )) s" | tee -a $log
  grep '^\[bench' $log | tail -12
  grep '^{' $log | cut -c1-400
done
unset TD_BENCH_WATCHDOG_S
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_$T.log 2>&1; echo "bench exit $?" >> gpurun_out/bench_$T.log
grep '^\[bench' gpurun_out/bench_$T.log; grep '^{' gpurun_out/bench_$T.log | cut -c1-900
