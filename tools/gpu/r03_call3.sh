#!/bin/bash
# round 3, call 3: GPU suite with the f3 kernels, headline bench, fuse_embed_head A/B, single-stream kernel trace
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r03c}; R=$PWD
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --timeout=900 --no-header -p no:cacheprovider > gpurun_out/pytest_$T.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_$T.log; tail -25 gpurun_out/pytest_$T.log
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$T.log 2>&1; echo "bench exit $?" >> gpurun_out/bench_$T.log
grep '^\[bench' gpurun_out/bench_$T.log; grep '^{' gpurun_out/bench_$T.log | cut -c1-600
STEPS=6 bash tools/gpu/ab_flag.sh fuse_embed_head 2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ab_fuse_embed_head_$T.txt
# single-stream kernel trace (no graph, no second stream, no token split): every kernel's own duration
rm -rf gpurun_out/prof_$T
(cd /tmp && TD_BENCH_MODEL_FLAGS=two_streams=0,split_tokens=0 timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$T -o k --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graph --no-box-calibration > $R/gpurun_out/prof_$T.log 2>&1)
f=$(find gpurun_out/prof_$T -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_single_stream_$T.csv
find gpurun_out/prof_$T -name '*kernel_trace*' -size +20M -delete
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/kernel_stats_single_stream_$T.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:30]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(8), ('%.2f%%'%(100*float(r['TotalDurationNs'])/tot)).rjust(7))
PY
