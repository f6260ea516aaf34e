#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_gpu_sla.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -4
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_sqp -o k --output-format csv -- python $R/tools/kbench.py --only prep --iters 20 > $R/gpurun_out/prof_sqp.log 2>&1)
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_sqp/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if not r['Name'].startswith('void at::'): print(r['Name'][:70], r['Calls'], r['AverageNs'], r['MinNs'])
PY
