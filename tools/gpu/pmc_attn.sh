#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-g14}; R=$PWD
cd /tmp
for pass in "sq1:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "sq2:SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  timeout 400 rocprofv3 --pmc $ctrs --kernel-trace -d $R/gpurun_out/pmc_${T}_$name -o k --output-format csv -- python $R/tools/kbench.py --iters 2 --only attn > $R/gpurun_out/pmc_${T}_$name.log 2>&1
  f=$(find $R/gpurun_out/pmc_${T}_$name -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python $R/tools/pmc_summary.py "$f" > $R/gpurun_out/pmc_${T}_$name.txt 2>&1 && grep -E "attn_kernel|linear_" $R/gpurun_out/pmc_${T}_$name.txt | head -8
  find $R/gpurun_out/pmc_${T}_$name -name '*.csv' -size +8M -delete
done
