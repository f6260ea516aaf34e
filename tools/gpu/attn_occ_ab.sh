#!/bin/bash
# interleaved end-to-end A/B: INT8/FP16-PV attention built for 3 workgroups per CU (168 VGPRs, 2 tile buffers) vs 2 per CU
# (252 VGPRs, 3 tile buffers, explicit K / V fragment prefetch; TD_TUNE_ATTN_OCC = 8 -> 2).  Bit-identical kernels.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-ab}
OUT=gpurun_out/attn_occ_ab_$T.txt; : > $OUT
for rep in 1 2 3; do
  for occ in 0 2; do
    timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-two-in-flight --tune 8=$occ 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('attn_occ $occ rep $rep', 'videos/s %.4f' % r['value'], 'dit_step_ms %.2f' % r['dit_step_ms'], 'attn avg ms %.4f' % r['roofline_attention']['avg_launch_ms'], 'gemm avg ms %.4f' % r['roofline']['avg_launch_ms'])
" | tee -a $OUT
  done
done
