#!/bin/bash
# peaks microbenchmark + multi-rank bench rig (several ranks on the one GPU of the test box, gloo collectives)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-rig}
timeout 300 tools/ubench/peaks > gpurun_out/peaks_$T.log 2>&1; cat gpurun_out/peaks_$T.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench1_$T.log 2>&1; echo "exit $?" >> gpurun_out/bench1_$T.log; tail -2 gpurun_out/bench1_$T.log | cut -c1-400
export TD_BENCH_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 1 --warmup 1 --layers 4 > gpurun_out/bench2_$T.log 2>&1; echo "exit $?" >> gpurun_out/bench2_$T.log; tail -3 gpurun_out/bench2_$T.log | cut -c1-2500
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 4 --steps 1 --warmup 1 --layers 2 --sp 2 > gpurun_out/bench4_$T.log 2>&1; echo "exit $?" >> gpurun_out/bench4_$T.log; tail -3 gpurun_out/bench4_$T.log | cut -c1-1500
unset TD_BENCH_BACKEND
timeout 600 python -m pytest tests/test_gpu_seqpar.py -m gpu -q --no-header -p no:cacheprovider 2>&1 | tail -3
