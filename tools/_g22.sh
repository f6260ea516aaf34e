#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_wan.py tests/test_gpu_seqpar.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -4
for i in 1 2; do timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_lnq$i.log 2>&1; grep '^{' gpurun_out/bench_lnq$i.log | cut -c1-330; done
