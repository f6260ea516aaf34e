#!/bin/bash
# the other BASELINE.json configurations, one video each (same code path as the headline run)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-cfg}
: > gpurun_out/other_configs_$T.jsonl
i=0
for a in "--workload c2" "--workload c3" "--model Wan2.1-14B --res 720p" "--model Wan2.2-A14B --res 720p --two-experts" "--model Wan2.1-14B --res 480p" "--sage-pv fp8" "--gemm-fast 4" "--gemm-fast 4 --sage-pv fp8"; do
  i=$((i+1)); log=gpurun_out/cfg_${T}_$i.log     # one log per configuration (a timeout must stay diagnosable)
  t0=$(date +%s)
  TD_BENCH_WATCHDOG_S=200 timeout 600 python bench.py $a --steps 2 --warmup 1 --no-cpu-baseline > $log 2>&1; rc=$?
  echo "[$a] exit $rc after $(( $(date +%s) - t0 )) s" >> $log
  grep '^{' $log >> gpurun_out/other_configs_$T.jsonl || { echo "FAILED (exit $rc): $a"; tail -5 $log; }
done
python - <<PY
import json
for l in open("gpurun_out/other_configs_$T.jsonl"):
    d=json.loads(l); print(d["config"]["workload"][:70], "|", d["config"]["model"], d["config"]["resolution"], "|", d["dtype"][:60], "| ms/step", round(d["dit_step_ms"],1), "| videos/s", round(d["value"],4), "| vs_baseline", d["vs_baseline"])
PY
