#!/bin/bash
# interleaved end-to-end A/B of a td_set_tuning knob: bash tools/gpu/ab_tune.sh KEY VALUE [rounds]   (baseline: the knob at 0)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
KEY=$1; VAL=$2; R=${3:-2}
for r in $(seq 1 $R); do
  for v in $VAL 0; do
    timeout 300 python bench.py --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline --no-box-calibration --tune $KEY=$v > gpurun_out/abt_${KEY}_${v}_$r.log 2>&1
    python - <<PY
import json
l=[x for x in open("gpurun_out/abt_${KEY}_${v}_$r.log") if x.startswith("{")]
d=json.loads(l[-1]); a=d.get("roofline_attention",{})
print("tune $KEY = $v run $r:", round(d["value"],4), "videos/s", round(d["dit_step_ms"],2), "ms per DiT step; attention", a.get("avg_launch_ms"), "ms, frac", a.get("frac"))
PY
  done
done
