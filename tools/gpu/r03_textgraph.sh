#!/bin/bash
# A/B of the captured text refresh (GraphedModel._prepare_text): model-level tests, then interleaved bench pairs
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r03t}
timeout 900 python -m pytest tests/test_gpu_wan.py tests/test_gpu_seqpar.py tests/test_gpu_bench.py -m gpu -q --maxfail=10 --timeout=600 --no-header -p no:cacheprovider > gpurun_out/pytest_$T.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_$T.log; grep -v amdgpu gpurun_out/pytest_$T.log | tail -12
for r in 1 2 3; do
  for v in 1 0; do
    TD_BENCH_GRAPH_TEXT=$v timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-box-calibration 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('graph_text = $v run $r:', round(d['value'], 4), 'videos/s', round(d['ms_per_step'], 2), 'ms per video')
" | tee -a gpurun_out/ab_graph_text_$T.txt
  done
done
