#!/bin/bash
# other BASELINE.json configs as sanity / coverage runs (not the headline line)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python bench.py --model Wan2.1-14B --res 720p --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_c4.log 2>&1; echo "c4 exit $?"; tail -1 gpurun_out/bench_c4.log | cut -c1-700
timeout 600 python bench.py --workload c2 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_c2.log 2>&1; echo "c2 exit $?"; tail -1 gpurun_out/bench_c2.log | cut -c1-500
timeout 600 python bench.py --workload c3 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_c3.log 2>&1; echo "c3 exit $?"; tail -1 gpurun_out/bench_c3.log | cut -c1-500
timeout 900 python bench.py --model Wan2.2-A14B --res 720p --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_c5_1gpu.log 2>&1; echo "c5 exit $?"; tail -1 gpurun_out/bench_c5_1gpu.log | cut -c1-500
