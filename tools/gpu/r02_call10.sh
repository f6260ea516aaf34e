#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x --timeout=900 --no-header -p no:cacheprovider 2>&1 | tail -4
for i in 1 2; do timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-two-in-flight 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('videos/s', round(r['value'],4), 'dit_ms', round(r['dit_step_ms'],2), 'eager', round(r['eager_videos_per_s'],4))"; done
