#!/bin/bash
# kernel timeline (start / end / queue) of the production schedule (hipGraph replay, two streams, token split), 4 layers
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/tl -o tl --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --layers 4 --no-cpu-baseline --no-box-calibration > $R/gpurun_out/tl.log 2>&1)
f=$(find gpurun_out/tl -name '*kernel_trace.csv' | head -1); ls -la $f; head -2 $f | cut -c1-600
python - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
print(len(rows), "dispatches; columns:", list(rows[0].keys()))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last forward: find the last 'patch_embed' or first kernel of the last graph replay: take the last 260 dispatches
tail=rows[-300:]
t0=int(tail[0]["Start_Timestamp"])
out=open("gpurun_out/timeline_tail.txt","w")
for r in tail:
    s=(int(r["Start_Timestamp"])-t0)/1e3; e=(int(r["End_Timestamp"])-t0)/1e3
    out.write("%9.1f %9.1f %7.1f q=%s %s\n"%(s,e,e-s,r.get("Queue_Id","?"),r["Kernel_Name"][:70]))
out.close()
print(open("gpurun_out/timeline_tail.txt").read()[:200])
PY
find gpurun_out/tl -name '*.csv' -size +4M -delete
