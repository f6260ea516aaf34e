#!/bin/bash
# per-launch timeline of an emulated rank's layer under hipGraph replay (3 layers); $1 = tag, $2 = r/N
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; T=${1:-tl}; RN=${2:-0/8}
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/tl_$T -o tl --output-format csv -- python $R/bench.py --emulate-rank $RN --steps 2 --warmup 1 --layers 3 --no-cpu-baseline --no-box-calibration > $R/gpurun_out/tl_$T.log 2>&1)
f=$(find gpurun_out/tl_$T -name '*kernel_trace.csv' | head -1)
python - "$f" gpurun_out/timeline_$T.txt <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the last eager video is after the timed region: take the dispatches of the LAST graph replay = before the final eager video.
# simpler: print the last 700 dispatches; the reader finds the replayed forward by its dense spacing
tail=rows[-900:]
t0=int(tail[0]["Start_Timestamp"])
out=open(sys.argv[2],"w")
for r in tail:
    s=(int(r["Start_Timestamp"])-t0)/1e3; e=(int(r["End_Timestamp"])-t0)/1e3
    out.write("%9.1f %9.1f %7.1f q=%s %s\n"%(s,e,e-s,r.get("Queue_Id","?"),r["Kernel_Name"][:80]))
out.close()
print(len(rows), "dispatches")
PY
find gpurun_out/tl_$T -name '*.csv' -size +4M -delete
