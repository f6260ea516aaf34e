#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_wan.py tests/test_gpu_seqpar.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -8
for rep in 1 2; do
for f in 1 0; do
TD_BENCH_MODEL_FLAGS=cache_text_kv=$f timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-two-in-flight 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('cache_text_kv=$f', 'videos/s', round(r['value'],4), 'dit_ms', round(r['dit_step_ms'],2), 'attn_frac', round(r['roofline_attention']['frac'],3))"
done; done
