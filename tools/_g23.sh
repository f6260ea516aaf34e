#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sla.py -m gpu -q -k "rope or seq_mean" --no-header -p no:cacheprovider -x 2>&1 | tail -4
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_qkp -o k --output-format csv -- python $R/tools/kbench.py --only prep --iters 20 > $R/gpurun_out/prof_qkp.log 2>&1)
grep "qk_norm\|seq_mean" gpurun_out/prof_qkp.log | cut -c1-200
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_qkp/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if any(t in r['Name'] for t in ('qk_norm','seq_mean')): print(r['Name'][:70], r['Calls'], r['AverageNs'], r['MinNs'])
PY
