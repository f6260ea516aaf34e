#!/bin/bash
# interleaved end-to-end A/B of the attention builds (TD_TUNE_ATTN_OCC: 0 = default three-per-CU LDS-DMA build, 2 = two-per-CU with
# explicit prefetch, 3 = two-per-CU with VGPR-staged tiles): bash tools/gpu/attn_occ_ab.sh TAG "0 3" [rounds]
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-occ}; OCCS=${2:-"0 2"}; R=${3:-3}
timeout 600 python -m pytest tests/test_gpu_sla.py -m gpu -q -k "two_per_cu" --no-header -p no:cacheprovider 2>&1 | grep -v amdgpu | tail -3
: > gpurun_out/attn_occ_ab_$T.txt
for r in $(seq 1 $R); do
  for occ in $OCCS; do
    timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-box-calibration --tune 8=$occ 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('ATTN_OCC = $occ run $r:', round(d['value'], 4), 'videos/s', round(d['dit_step_ms'], 2), 'ms per step; attention', round(d['roofline_attention']['avg_launch_ms'] * 1e3, 1), 'us, frac', round(d['roofline_attention']['frac'], 3))
" | tee -a gpurun_out/attn_occ_ab_$T.txt
  done
done
