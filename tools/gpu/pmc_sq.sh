#!/bin/bash
# round-4 counter refresh: SQ sets (+ GRBM_GUI_ACTIVE) over one DiT forward of bench.py (eager enqueue, full-size launches)
# and over one 480p VAE decode; FETCH / WRITE over the same forward -> gpurun_out/pmc_sq_$T.json, pmc_hbm_traffic_$T.json
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r04}; R=$PWD
B="python $R/bench.py --steps 1 --warmup 0 --num-steps 1 --no-graph --no-cpu-baseline --no-box-calibration"
cd /tmp
csvs=""
for pass in "sq1:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "sq2:SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" \
            "sq3:SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  TD_BENCH_MODEL_FLAGS=split_tokens=0,two_streams=0 timeout 500 rocprofv3 --pmc $ctrs --kernel-trace -d $R/gpurun_out/pmcs_${T}_$name -o b --output-format csv -- $B --layers 6 > $R/gpurun_out/pmcs_${T}_$name.log 2>&1
  echo "pmc $name (DiT) exit $?"
  f=$(find $R/gpurun_out/pmcs_${T}_$name -name '*counter_collection.csv' | head -1); [ -n "$f" ] && csvs="$csvs $f"
  if [ "$name" != "sq3" ]; then
    timeout 500 rocprofv3 --pmc $ctrs --kernel-trace -d $R/gpurun_out/pmcv_${T}_$name -o v --output-format csv -- python $R/tools/f4_time.py vae480 > $R/gpurun_out/pmcv_${T}_$name.log 2>&1
    echo "pmc $name (VAE decode) exit $?"
    f=$(find $R/gpurun_out/pmcv_${T}_$name -name '*counter_collection.csv' | head -1); [ -n "$f" ] && csvs="$csvs $f"
  fi
done
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  TD_BENCH_MODEL_FLAGS=split_tokens=0 timeout 900 rocprofv3 --pmc $ctrs --kernel-trace -d $R/gpurun_out/pmcb_${T}_$name -o b --output-format csv -- $B > $R/gpurun_out/pmcb_${T}_$name.log 2>&1
  echo "pmc $name exit $?"
done
cd $R
python tools/pmc_sq.py gpurun_out/pmc_sq_$T.json $csvs
F=$(find gpurun_out/pmcb_${T}_fetch -name '*counter_collection.csv' | head -1)
W=$(find gpurun_out/pmcb_${T}_write -name '*counter_collection.csv' | head -1)
python tools/pmc_traffic.py $F $W gpurun_out/pmc_hbm_traffic_$T.json
find gpurun_out -name '*.csv' -size +4M -delete
