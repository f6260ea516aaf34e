#!/bin/bash
# kernel-trace summary of rank 0 of an emulated 8-way split of C1 (shipped group rule), eager enqueue so that every launch is attributed
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/emu8 -o e --output-format csv -- python $R/bench.py --emulate-rank 0/8 --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-box-calibration > $R/gpurun_out/emu8.log 2>&1)
f=$(find gpurun_out/emu8 -name '*kernel_stats.csv' | head -1); cp $f gpurun_out/kernel_stats_emu8.csv
find gpurun_out/emu8 -name '*kernel_trace*' -size +20M -delete
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/kernel_stats_emu8.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("kernel time total %.1f ms over 8 forwards -> %.2f ms per forward" % (tot/1e6, tot/8e6))
for r in rows[:28]:
    print("%-84s n=%5s avg=%7.1f us %5.1f%%"%(r['Name'][:84], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
