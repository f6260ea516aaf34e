#!/bin/bash
# interleaved end-to-end A/B of one td_set_tuning knob: bash tools/gpu/ab_tune.sh KEY VAL_A VAL_B [rounds]
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
KEY=$1; A=$2; B=$3; R=${4:-2}
for r in $(seq 1 $R); do
  for v in $A $B; do
    timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --tune $KEY=$v > gpurun_out/ab_${KEY}_${v}_$r.log 2>&1
    python - <<PY
import json
l=[x for x in open("gpurun_out/ab_${KEY}_${v}_$r.log") if x.startswith("{")]
d=json.loads(l[-1]); print("key $KEY = $v run $r:", round(d["value"],4), "videos/s; attn avg ms", round(d["roofline_attention"]["avg_launch_ms"],4), "gemm avg ms", round(d["roofline"]["avg_launch_ms"],4))
PY
  done
done
