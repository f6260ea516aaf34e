#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sla.py tests/test_gpu_seqpar.py tests/test_oracle_cpu.py -q -x --no-header -p no:cacheprovider -k "gathered or seqpar or export" 2>&1 | tail -12
bash tools/gpu/multirank_rig.sh r02b 2>&1 | grep -E "^exit|value" | cut -c1-200
