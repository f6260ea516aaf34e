#!/bin/bash
# full GPU suite + smoke + default bench line (round 5)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp; T=${1:-r05}
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 --no-header -p no:cacheprovider -s > gpurun_out/pytest_$T.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_$T.log; grep -E "passed|failed|FAILED|ERROR" gpurun_out/pytest_$T.log | tail -15
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke_$T.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke_$T.log; tail -2 gpurun_out/smoke_$T.log
