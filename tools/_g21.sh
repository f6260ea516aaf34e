#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_lnq -o k --output-format csv -- python $R/tools/kbench.py --only norm --iters 20 > $R/gpurun_out/prof_lnq.log 2>&1)
f=$(find gpurun_out/prof_lnq -name '*kernel_stats.csv' | head -1); cut -d, -f1-4 "$f" | cut -c1-200 | head -12
