#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_gpu_sla.py tests/test_gpu_wan.py tests/test_gpu_seqpar.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -4
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_lkv -o k --output-format csv -- python $R/tools/kbench.py --only prep,linear,sla --iters 20 > $R/gpurun_out/prof_lkv.log 2>&1)
grep "linear_kv\|seq_mean" gpurun_out/prof_lkv.log | cut -c1-200
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_lkv/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if any(t in r['Name'] for t in ('linear_kv','seq_mean')): print(r['Name'][:70], r['Calls'], r['AverageNs'], r['MinNs'])
PY
for i in 1 2; do timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_lkv$i.log 2>&1; grep '^{' gpurun_out/bench_lkv$i.log | cut -c1-330; done
