#!/bin/bash
# interleaved end-to-end A/B at N = 1: $1 = label A flags, $2 = label B flags (bench.py arguments), 3 repetitions
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2 3; do for fl in "$1" "$2"; do
timeout 600 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-box-calibration $fl 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('[$fl]', round(r['value'],4), 'videos/s', round(r['dit_step_ms'],2), 'ms per DiT step; gemm avg', round(r['roofline']['avg_launch_ms']*1e3,1), 'us')"
done; done
