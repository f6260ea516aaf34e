#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_wan.py tests/test_gpu_ops.py -m gpu -q -x --no-header -p no:cacheprovider -k "sampler or compat or fast" 2>&1 | tail -5
bash tools/gpu/multirank_rig.sh r02
# C5 as the reference runs it: two experts resident, the switch in the timed region (1 GPU)
timeout 900 python bench.py --model Wan2.2-A14B --res 720p --two-experts --steps 1 --warmup 1 --no-cpu-baseline --no-two-in-flight > gpurun_out/r02_c5_two_experts.log 2>&1; echo "exit $?" >> gpurun_out/r02_c5_two_experts.log; tail -2 gpurun_out/r02_c5_two_experts.log | cut -c1-1500
