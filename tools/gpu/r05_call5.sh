#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
timeout 900 python -m pytest tests/test_gpu_r05.py tests/test_gpu_wan.py tests/test_gpu_c1.py tests/test_gpu_r04.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -4
one() { d=$1; shift; (cd $d && timeout 600 python bench.py "$@" --no-cpu-baseline --no-box-calibration 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print(round(r['value'],4), 'videos/s', round(r['dit_step_ms'],2), 'ms per DiT step')"); }
for i in 1 2 3; do
  echo "N=1 round 5: $(one $R --steps 6 --warmup 1)"
  echo "N=1 round 4: $(one $R/.r04tree --steps 6 --warmup 1)"
done
