#!/bin/bash
# interleaved end-to-end A/B of a boolean WanModel attribute: bash tools/gpu/ab_flag.sh ATTR [rounds]
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
ATTR=$1; R=${2:-2}
for r in $(seq 1 $R); do
  for v in 1 0; do
    TD_BENCH_MODEL_FLAGS="$ATTR=$v" timeout 300 python bench.py --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline --no-box-calibration > gpurun_out/abf_${ATTR}_${v}_$r.log 2>&1
    python - <<PY
import json
l=[x for x in open("gpurun_out/abf_${ATTR}_${v}_$r.log") if x.startswith("{")]
d=json.loads(l[-1]); print("$ATTR = $v run $r:", round(d["value"],4), "videos/s", round(d["dit_step_ms"],2), "ms per DiT step")
PY
  done
done
