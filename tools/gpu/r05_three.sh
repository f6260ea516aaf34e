#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_wan.py tests/test_gpu_sla.py tests/test_gpu_c1.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -3
for i in 1 2 3; do for fl in "three_branches=1" "three_branches=0"; do
TD_BENCH_MODEL_FLAGS=$fl timeout 600 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-box-calibration 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('$fl:', round(r['value'],4), 'videos/s', round(r['dit_step_ms'],2), 'ms per DiT step')"
done; done
