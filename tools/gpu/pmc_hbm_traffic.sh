#!/bin/bash
# PMC HBM-traffic passes over bench.py (FETCH_SIZE and WRITE_SIZE separately; full-size launches: token split off) + kernel trace
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-g10}; R=$PWD
cd /tmp
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  TD_BENCH_MODEL_FLAGS=split_tokens=0 timeout 900 rocprofv3 --pmc $ctrs --kernel-trace -d $R/gpurun_out/pmcb_${T}_$name -o b --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline --no-box-calibration > $R/gpurun_out/pmcb_${T}_$name.log 2>&1
  echo "pmc $name exit $?"
done
cd $R
F=$(find gpurun_out/pmcb_${T}_fetch -name '*counter_collection.csv' | head -1)
W=$(find gpurun_out/pmcb_${T}_write -name '*counter_collection.csv' | head -1)
python tools/pmc_traffic.py $F $W gpurun_out/pmc_hbm_traffic_$T.json
find gpurun_out/pmcb_${T}_fetch gpurun_out/pmcb_${T}_write -name '*.csv' -size +4M -delete
