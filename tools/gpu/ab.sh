#!/bin/bash
# round 6: interleaved end-to-end A/B at N = 1 of any number of arms (each argument = one arm's bench.py flags), REPS repetitions
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
OUT=gpurun_out/${TAG:-ab}.txt; : > $OUT
for i in $(seq 1 ${REPS:-3}); do for fl in "$@"; do
timeout 600 python bench.py --steps ${STEPS:-6} --warmup 1 --no-cpu-baseline --no-box-calibration $fl 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('[$fl]', round(r['value'],4), 'videos/s', round(r['dit_step_ms'],2), 'ms per DiT step; gemm avg', round(r['roofline']['avg_launch_ms']*1e3,1), 'us, frac', round(r['roofline']['frac'],4))" | tee -a $OUT
done; done
