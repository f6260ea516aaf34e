#!/bin/bash
# the sequence-parallel path over a real 1-rank RCCL communicator: the new tests, then C1 at full depth (whole-graph capture of
# ~1 s with the collectives inside, replayed for 12 DiT forwards) beside the plain single-GPU line on the same box
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r1}
timeout 900 python -m pytest tests/test_gpu_seqpar.py tests/test_gpu_bench.py -m gpu -q --no-header -p no:cacheprovider -s -k "rccl or watchdog" > gpurun_out/pytest_rccl_$T.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_rccl_$T.log; grep -E "passed|failed|RCCL|rccl" gpurun_out/pytest_rccl_$T.log | tail -8
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-box-calibration > gpurun_out/n1_$T.log 2>&1
timeout 600 python bench.py --rccl-one-rank --steps 3 --warmup 1 --no-cpu-baseline --no-box-calibration > gpurun_out/rccl1_$T.log 2>&1; echo "exit $?" >> gpurun_out/rccl1_$T.log
python - <<PY
import json
def line(f):
    for l in open(f):
        if l.startswith("{"): return json.loads(l)
a=line("gpurun_out/n1_$T.log"); b=line("gpurun_out/rccl1_$T.log")
print("single GPU: %.2f ms per DiT step" % a["dit_step_ms"])
print("one-rank RCCL rig: %.2f ms per DiT step; %s; whole-graph error %s" % (b["dit_step_ms"], b["rccl_one_rank"]["graph_mode"], b["rccl_one_rank"]["whole_graph_error"]))
PY
tail -3 gpurun_out/rccl1_$T.log | cut -c1-300
