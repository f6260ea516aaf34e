#!/bin/bash
# interleaved end-to-end A/B of WanModel.two_streams (Q-side chain of the SageSLA self-attention on a second stream)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-ab}
OUT=gpurun_out/two_streams_ab_$T.txt; : > $OUT
for rep in 1 2 3; do
  for v in 0 1; do
    TD_BENCH_MODEL_FLAGS=two_streams=$v timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-two-in-flight 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('two_streams $v rep $rep', 'videos/s %.4f' % r['value'], 'dit_step_ms %.2f' % r['dit_step_ms'], 'eager %.3f' % r['eager_videos_per_s'])
" | tee -a $OUT
  done
done
