#!/bin/bash
# interleaved end-to-end A/B: GEMM epilogue with LDS-staged full-line stores (this tree) vs the previous direct 16-byte-piece
# stores (libturbodiffusion_amd_directstore.so, built from the commit before; TD_LIB_PATH) and vs staged + V^T tiles from
# the q|k|v epilogue (WanModel.fuse_vt)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-ab}
OUT=gpurun_out/store_ab_$T.txt; : > $OUT
PREV=$PWD/turbodiffusion_amd/libturbodiffusion_amd_directstore.so
for rep in 1 2; do
  for cfg in direct staged staged_vt; do
    case $cfg in
      direct) export TD_LIB_PATH=$PREV; export TD_BENCH_MODEL_FLAGS=fuse_vt=0;;
      staged) unset TD_LIB_PATH; export TD_BENCH_MODEL_FLAGS=fuse_vt=0;;
      staged_vt) unset TD_LIB_PATH; export TD_BENCH_MODEL_FLAGS=fuse_vt=1;;
    esac
    timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-two-in-flight 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('$cfg rep $rep', 'videos/s %.4f' % r['value'], 'dit_step_ms %.2f' % r['dit_step_ms'], 'gemm avg ms %.4f' % r['roofline']['avg_launch_ms'], 'frac %.4f' % r['roofline']['frac'])
" | tee -a $OUT
  done
done
unset TD_LIB_PATH TD_BENCH_MODEL_FLAGS
