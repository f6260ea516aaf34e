cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
OUT=gpurun_out/attn_occ2_ab_c8.txt; : > $OUT
for rep in 1 2 3 4; do
  for occ in 0 2; do
    timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-box-calibration --tune 8=$occ 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('attn_occ $occ rep $rep', 'videos/s %.4f' % r['value'], 'dit_step_ms %.2f' % r['dit_step_ms'], 'attn avg ms %.4f' % r['roofline_attention']['avg_launch_ms'], 'gemm avg ms %.4f' % r['roofline']['avg_launch_ms'])
" | tee -a $OUT
  done
done
