#!/bin/bash
# multi-rank bench rig: several ranks on the ONE GPU of the test box, gloo collectives (TD_BENCH_BACKEND=gloo) — a
# functional check of every line of bench.py's N > 1 paths, not a performance number
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-rig}
export TD_BENCH_BACKEND=gloo
# default N > 1 mode: ONE video sharded by sequence over all ranks (value), then the N-replica leg
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 1 --warmup 1 --layers 4 > gpurun_out/bench2_$T.log 2>&1; echo "exit $?" >> gpurun_out/bench2_$T.log; tail -3 gpurun_out/bench2_$T.log | cut -c1-2500
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 4 --steps 1 --warmup 1 --layers 2 > gpurun_out/bench4_$T.log 2>&1; echo "exit $?" >> gpurun_out/bench4_$T.log; tail -3 gpurun_out/bench4_$T.log | cut -c1-1500
# hybrid: 2 groups of 2
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 4 --steps 1 --warmup 1 --layers 2 --sp 2 > gpurun_out/bench4h_$T.log 2>&1; echo "exit $?" >> gpurun_out/bench4h_$T.log; tail -3 gpurun_out/bench4h_$T.log | cut -c1-800
unset TD_BENCH_BACKEND
