#!/bin/bash
# round 3: the multi-rank bench paths on the one-GPU box over gloo (functional check at HEAD: fused patch-embed row ranges,
# token-major head + gather, capture-outcome agreement over the store), then the 30-block drift tool
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r03}
bash tools/gpu/multirank_rig.sh $T 2>&1 | grep -v amdgpu.ids | cut -c1-900
timeout 600 python tools/drift.py > gpurun_out/drift_$T.log 2>&1; echo "drift exit $?" >> gpurun_out/drift_$T.log; grep -v amdgpu gpurun_out/drift_$T.log | tail -12
