#!/bin/bash
# second part of the round-5 evidence: the fixture tests of test_gpu_r05.py, the default bench line, the FETCH / WRITE passes
# (projection as one launch, like the bench's own per-kernel timing video)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp; T=${1:-r05g}; R=$PWD; NOCPU=$2
timeout 900 python -m pytest tests/test_gpu_r05.py tests/test_gpu_bench.py -m gpu -q --timeout=900 --no-header -p no:cacheprovider -s 2>&1 | grep -E "passed|failed|FAILED|rel-L2|Error" | tail -12
timeout 1500 python bench.py --steps 10 --warmup 2 $NOCPU > gpurun_out/bench_$T.log 2>&1; echo "bench exit $?" >> gpurun_out/bench_$T.log; grep '^{' gpurun_out/bench_$T.log | cut -c1-400
B="python $R/bench.py --steps 1 --warmup 0 --num-steps 1 --no-graph --no-cpu-baseline --no-box-calibration"
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  (cd /tmp && TD_BENCH_MODEL_FLAGS=split_tokens=0,split_qkv=0 timeout 900 rocprofv3 --pmc $ctrs --kernel-trace -d $R/gpurun_out/pmcb_${T}_$name -o b --output-format csv -- $B > $R/gpurun_out/pmcb_${T}_$name.log 2>&1)
  echo "pmc $name exit $?"
done
F=$(find gpurun_out/pmcb_${T}_fetch -name '*counter_collection.csv' | head -1)
W=$(find gpurun_out/pmcb_${T}_write -name '*counter_collection.csv' | head -1)
python tools/pmc_traffic.py $F $W gpurun_out/pmc_hbm_traffic_$T.json | tail -8 | cut -c1-200
find gpurun_out -name '*.csv' -size +4M -delete
