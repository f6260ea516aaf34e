#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --timeout=900 --no-header -p no:cacheprovider --durations=8 > gpurun_out/r02c5_pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/r02c5_pytest.log; tail -40 gpurun_out/r02c5_pytest.log | cut -c1-300
