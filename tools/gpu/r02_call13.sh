#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_wan.py tests/test_gpu_c1.py tests/test_oracle_cpu.py -q -x --no-header -p no:cacheprovider -k "row_stats or one_block or export" 2>&1 | tail -12
for rep in 1 2; do
for f in 1 0; do
TD_BENCH_MODEL_FLAGS=fuse_row_stats=$f timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-two-in-flight 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('fuse_row_stats=$f', 'videos/s', round(r['value'],4), 'dit_ms', round(r['dit_step_ms'],2), 'gemm_ms', round(r['roofline']['avg_launch_ms'],4))"
done; done
