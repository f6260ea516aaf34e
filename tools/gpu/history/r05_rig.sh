#!/bin/bash
# round 5: the N > 1 bench paths at HEAD with several ranks on the ONE GPU of the box over gloo (functional check of the driver's
# command forms and of the multi-rank record: rccl / per_rank_dit_step_ms / exposed_wait), then prompt ids -> pixels
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp; T=${1:-r05rig}
export TD_BENCH_BACKEND=gloo
show() { grep '^{' $1 | python -c "
import sys,json
r=json.loads(sys.stdin.readline())
print('$2: n_gpus', r['n_gpus'], '|', r['config']['parallelism'], '| value', round(r['value'],4), r['unit'], '| ms per DiT step', round(r['dit_step_ms'],2), '| scaling', r['scaling'])
print('   rccl:', json.dumps(r.get('rccl'))[:400])
print('   per_rank_dit_step_ms:', json.dumps(r.get('per_rank_dit_step_ms'))[:300])
print('   exposed_wait:', json.dumps(r.get('exposed_wait'))[:300])
print('   replicas:', json.dumps(r.get('replicas'))[:300])
"; }
OUT=gpurun_out/multirank_rig_$T.txt; : > $OUT
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 1 --warmup 1 --layers 4 --no-cpu-baseline > gpurun_out/bench2_$T.log 2>&1; echo "exit $?" >> gpurun_out/bench2_$T.log
show gpurun_out/bench2_$T.log "torchrun, 2 ranks, 4 layers" | tee -a $OUT
timeout 900 python bench.py --gpus 8 --steps 1 --warmup 1 --layers 2 --no-cpu-baseline > gpurun_out/bench8_$T.log 2>&1; echo "exit $?" >> gpurun_out/bench8_$T.log
show gpurun_out/bench8_$T.log "self-spawned, 8 ranks, 2 layers" | tee -a $OUT
timeout 600 python bench.py --gpus 4 --sp 2 --steps 1 --warmup 1 --layers 2 --no-cpu-baseline > gpurun_out/bench4h_$T.log 2>&1; echo "exit $?" >> gpurun_out/bench4h_$T.log
show gpurun_out/bench4h_$T.log "self-spawned, 4 ranks as 2 groups of 2, 2 layers" | tee -a $OUT
unset TD_BENCH_BACKEND
timeout 600 python bench.py --prompt-to-pixels --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/p2p_$T.log 2>&1; grep '^{' gpurun_out/p2p_$T.log | python -c "
import sys,json
r=json.loads(sys.stdin.readline()); print('prompt ids -> pixels:', json.dumps(r.get('prompt_to_pixels'))[:700])" | tee -a $OUT
