#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/kbench.py --only none --cpu 2>&1 | grep -v amdgpu | tee gpurun_out/r02_kbench_cpu.jsonl | cut -c1-200
timeout 1200 python tools/drift.py 2>&1 | grep -v amdgpu | tee gpurun_out/r02_drift.jsonl
