#!/bin/bash
# rocprofv3 kernel-trace stats of one tools/kbench.py section: bash tools/gpu/kernel_times.sh SECTION [iters]
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; S=${1:-attn}; I=${2:-10}
rm -rf gpurun_out/prof_kt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -o k --output-format csv -- python $R/tools/kbench.py --only $S --iters $I > $R/gpurun_out/prof_kt.log 2>&1)
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_kt/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if not r['Name'].startswith('void at::'): print(r['Name'][:90], r['Calls'], r['AverageNs'], r['MinNs'])
PY
