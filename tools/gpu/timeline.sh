#!/bin/bash
# kernel timeline (start / end / queue) of the production schedule under hipGraph replay (two streams, token split, two-launch
# projection), 4 layers: the dispatches of the LAST graph-replayed DiT forward of the timed region -> gpurun_out/timeline_graph_$T.txt
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; T=${1:-tl}
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/tl_$T -o tl --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --layers 4 --no-cpu-baseline --no-box-calibration > $R/gpurun_out/tl_$T.log 2>&1)
f=$(find gpurun_out/tl_$T -name '*kernel_trace.csv' | head -1)
python - "$f" gpurun_out/timeline_graph_$T.txt <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# one DiT forward ends with head_kernel; videos: warm-up (eager + capture), 2 timed (graph replay: 8 forwards), then the eager videos.
heads=[i for i,r in enumerate(rows) if "head_kernel" in r["Kernel_Name"]]
# replayed forwards are the densest: pick the forward (between consecutive head kernels) with the smallest wall time
best=None
for a,b in zip(heads[:-1],heads[1:]):
    n=b-a
    if n<50: continue
    dur=int(rows[b]["End_Timestamp"])-int(rows[a]["End_Timestamp"])
    if best is None or dur<best[0]: best=(dur,a,b)
dur,a,b=best
t0=int(rows[a+1]["Start_Timestamp"])
out=open(sys.argv[2],"w")
out.write("# one graph-replayed DiT forward (4 layers): %d dispatches, %.1f us\n"%(b-a,dur/1e3))
for r in rows[a+1:b+1]:
    s=(int(r["Start_Timestamp"])-t0)/1e3; e=(int(r["End_Timestamp"])-t0)/1e3
    out.write("%9.1f %9.1f %7.1f q=%s %s\n"%(s,e,e-s,r.get("Queue_Id","?"),r["Kernel_Name"][:90]))
out.close()
print(len(rows),"dispatches; replayed forward:",b-a,"dispatches,",dur/1e3,"us")
PY
find gpurun_out/tl_$T -name '*.csv' -size +4M -delete
