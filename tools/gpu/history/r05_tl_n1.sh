#!/bin/bash
# per-launch timeline of the production schedule at N = 1 under hipGraph replay (4 layers): one forward out of the timed region
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; T=${1:-n1}
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/tl_$T -o tl --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --layers 4 --no-cpu-baseline --no-box-calibration > $R/gpurun_out/tl_$T.log 2>&1)
f=$(find gpurun_out/tl_$T -name '*kernel_trace.csv' | head -1)
python - "$f" gpurun_out/timeline_$T.txt <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# forwards start at patch_embed_kernel; take the forward before the last 5 (the last 4 belong to the eager video)
starts=[i for i,r in enumerate(rows) if r["Kernel_Name"].startswith("void patch_embed_kernel")]
a,b=starts[-6],starts[-5]
t0=int(rows[a]["Start_Timestamp"])
out=open(sys.argv[2],"w")
for r in rows[a:b]:
    s=(int(r["Start_Timestamp"])-t0)/1e3; e=(int(r["End_Timestamp"])-t0)/1e3
    out.write("%9.1f %9.1f %7.1f q=%s %s\n"%(s,e,e-s,r.get("Queue_Id","?"),r["Kernel_Name"][:80]))
out.close()
print(len(rows), "dispatches;", b-a, "in the chosen forward")
PY
find gpurun_out/tl_$T -name '*.csv' -size +4M -delete
