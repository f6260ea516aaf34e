#!/bin/bash
# GPU visit: instruction-mix microbenchmark, fi-kernel segment trace, PMC passes over kbench
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-g3}; R=$PWD
timeout 120 tools/ubench/mix_rate > gpurun_out/mix_rate_$T.log 2>&1; cat gpurun_out/mix_rate_$T.log
timeout 300 python tools/gemm_trace.py 4 > gpurun_out/trace_$T.log 2>&1; grep -v amdgpu.ids gpurun_out/trace_$T.log
cd /tmp
for pass in "sq1:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "sq2:SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" \
            "tcc1:FETCH_SIZE TCC_HIT_sum" "tcc2:WRITE_SIZE TCC_MISS_sum"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  timeout 400 rocprofv3 --pmc $ctrs --kernel-trace -d $R/gpurun_out/pmc_${T}_$name -o k --output-format csv -- python $R/tools/kbench.py --iters 2 --only gemm,attn > $R/gpurun_out/pmc_${T}_$name.log 2>&1
  echo "pmc $name exit $?"
  f=$(find $R/gpurun_out/pmc_${T}_$name -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python $R/tools/pmc_summary.py "$f" > $R/gpurun_out/pmc_${T}_$name.txt 2>&1 && grep -E "gemm_w8a8_(fi|pp)|attn_kernel" $R/gpurun_out/pmc_${T}_$name.txt | head -12
  find $R/gpurun_out/pmc_${T}_$name -name '*.csv' -size +8M -delete
done
