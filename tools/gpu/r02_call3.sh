#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/gemm_sched_exp.py > gpurun_out/r02c3_sched.log 2>&1; cat gpurun_out/r02c3_sched.log | cut -c1-400
