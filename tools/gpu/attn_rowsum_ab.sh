#!/bin/bash
# interleaved end-to-end A/B (round 4): production attention (3 workgroups per CU) vs the two-per-CU prefetch build (8=2) vs
# that build with the row sum on the matrix pipe (8=4); per-launch attention / GEMM averages from the bench's event timer
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-ab}
OUT=gpurun_out/attn_rowsum_ab_$T.txt; : > $OUT
for rep in 1 2; do
  for occ in 0 2 4; do
    timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-box-calibration --tune 8=$occ 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('attn_occ $occ rep $rep', 'videos/s %.4f' % r['value'], 'dit_step_ms %.2f' % r['dit_step_ms'], 'attn avg ms %.4f' % r['roofline_attention']['avg_launch_ms'], 'gemm avg ms %.4f' % r['roofline']['avg_launch_ms'])
" | tee -a $OUT
  done
done
