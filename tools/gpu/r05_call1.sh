#!/bin/bash
# round-5 call 1: baseline of the box (N = 1 line), emulated ranks 0/8 and 0/2 (graph replay), and the per-launch timeline of one
# emulated-rank layer (eager, 3 layers) — what the fused K/Q-side passes and the small-M GEMM work are measured against
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD; T=${1:-r05a}
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_$T.log 2>&1; grep '^{' gpurun_out/bench_$T.log | cut -c1-300
for rn in 0/8 0/2; do
  timeout 600 python bench.py --emulate-rank $rn --steps 4 --warmup 1 --no-cpu-baseline --no-box-calibration > gpurun_out/emu_${T}_${rn/\//of}.log 2>&1
  grep '^{' gpurun_out/emu_${T}_${rn/\//of}.log | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('emu $rn', r['ms_per_step']/4, 'ms per DiT step', {k:v for k,v in r.items() if 'emul' in k})" | cut -c1-400
done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/tl_$T -o tl --output-format csv -- python $R/bench.py --emulate-rank 0/8 --steps 1 --warmup 1 --layers 3 --no-graph --no-cpu-baseline --no-box-calibration > $R/gpurun_out/tl_$T.log 2>&1)
f=$(find gpurun_out/tl_$T -name '*kernel_trace.csv' | head -1)
python - "$f" gpurun_out/timeline_$T.txt <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
tail=rows[-420:]
t0=int(tail[0]["Start_Timestamp"])
out=open(sys.argv[2],"w")
for r in tail:
    s=(int(r["Start_Timestamp"])-t0)/1e3; e=(int(r["End_Timestamp"])-t0)/1e3
    out.write("%9.1f %9.1f %7.1f q=%s g=%s %s\n"%(s,e,e-s,r.get("Queue_Id","?"),r.get("Grid_Size","?"),r["Kernel_Name"][:90]))
out.close()
print(len(rows), "dispatches")
PY
find gpurun_out/tl_$T -name '*.csv' -size +4M -delete
