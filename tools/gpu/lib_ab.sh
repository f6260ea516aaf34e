#!/bin/bash
# interleaved end-to-end A/B of two builds of the library with the SAME ABI: this tree's vs the one passed in $2 (TD_LIB_PATH)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-ab}; PREV=$PWD/${2:-turbodiffusion_amd/libturbodiffusion_amd_prev.so}
OUT=gpurun_out/lib_ab_$T.txt; : > $OUT
for rep in 1 2 3; do
  for cfg in prev this; do
    if [ $cfg = prev ]; then export TD_LIB_PATH=$PREV; else unset TD_LIB_PATH; fi
    timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-two-in-flight 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$cfg rep $rep', 'videos/s %.4f' % r['value'], 'dit_step_ms %.2f' % r['dit_step_ms'], 'gemm avg ms %.4f' % r['roofline']['avg_launch_ms'], 'frac %.4f' % r['roofline']['frac'])
" | tee -a $OUT
  done
done
unset TD_LIB_PATH
