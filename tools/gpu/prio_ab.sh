#!/bin/bash
# does a high-priority capture stream (main chain: K-side quantiser, block map, attention) beside default-priority side streams
# shorten the step?  interleaved
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
for rep in 1 2 3; do for p in 0 1; do
  TD_TOPK_HI=$p timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-box-calibration 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('topk on a high-priority stream = $p, rep $rep: %.2f ms per DiT step' % r['dit_step_ms'])"
done; done 2>&1 | tee gpurun_out/prio_ab.txt
