#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_lk2 -o k --output-format csv -- python $R/tools/kbench.py --only attn --iters 10 > $R/gpurun_out/prof_lk2.log 2>&1)
grep -v amdgpu gpurun_out/prof_lk2.log | cut -c1-200 | tail -8
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_lk2/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if not r['Name'].startswith('void at::'): print(r['Name'][:90], r['Calls'], r['AverageNs'], r['MinNs'])
PY
