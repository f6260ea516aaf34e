#!/bin/bash
# A/B of GEMM variant 4 (16x16x64) vs 5 (32x32x32) end to end, interleaved; shader clock / power sampled during one run
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-ab}
( for i in $(seq 1 60); do /opt/rocm/bin/rocm-smi -c -P -t 2>/dev/null | grep -i "sclk\|power\|Temperature (Sensor junction)\|mclk" | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/smi_$T.log 2>&1 &
SMI=$!
for r in 1 2; do
  for v in 4 5; do
    timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --gemm-variant $v > gpurun_out/bench_v${v}_r${r}_$T.log 2>&1
    python - <<PY
import json
l=[x for x in open("gpurun_out/bench_v${v}_r${r}_$T.log") if x.startswith("{")]
d=json.loads(l[-1]); print("variant $v run $r:", round(d["value"],4), "videos/s; gemm avg ms", round(d["roofline"]["avg_launch_ms"],4), "frac", round(d["roofline"]["frac"],4))
PY
  done
done
kill $SMI 2>/dev/null
head -40 gpurun_out/smi_$T.log | cut -c1-300
