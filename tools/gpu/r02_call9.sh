#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sla.py tests/test_oracle_cpu.py -q -x --no-header -p no:cacheprovider -k "fp8 or export" 2>&1 | tail -5
for rep in 1 2; do
for f in fp16 fp8; do
TD_BENCH_MODEL_FLAGS=sage_pv=$f timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-two-in-flight 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); a=r['roofline_attention']; print('sage_pv=$f', 'videos/s', round(r['value'],4), 'dit_ms', round(r['dit_step_ms'],2), 'attn_ms', round(a['avg_launch_ms'],4))"
done; done
