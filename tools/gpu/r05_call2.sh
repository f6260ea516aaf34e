#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/gelu_table_ab.py > gpurun_out/gelu_table_ab.txt 2>&1; cat gpurun_out/gelu_table_ab.txt | tail -26
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --no-header -p no:cacheprovider -k "gemm" 2>&1 | tail -5
