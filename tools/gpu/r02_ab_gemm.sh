#!/bin/bash
# interleaved end-to-end A/B of the W8A8 GEMM modes (variant / dequant / schedule) through bench.py
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { # tag variant fast sched
  timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-two-in-flight --gemm-variant $2 --tune 6=$3 --tune 4=$4 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$1', 'videos/s', round(r['value'],4), 'dit_ms', round(r['dit_step_ms'],2), 'gemm_frac', round(r['roofline']['frac'],4), 'gemm_avg_ms', round(r['roofline']['avg_launch_ms'],4))"
}
for rep in 1 2; do
run A_fi_exact 0 0 0
run B_m32_exact_s3 5 0 3
run C_m32_fast4_s1 5 4 1
run D_m32_fast4_s3 5 4 3
run E_fi_fast4 4 4 0
run F_m32_fast4_s0 5 4 0
done 2>&1 | tee gpurun_out/r02_ab_gemm.log
