#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/gemm_trace.py 5 > gpurun_out/r02c2_trace.log 2>&1; cat gpurun_out/r02c2_trace.log | cut -c1-1500
