#!/bin/bash
# round 4, first GPU call: the whole GPU suite (new: whole-graph capture over RCCL, self-spawning bench, rank emulation,
# td_gemm_bf16 / softmax / t5 norm, the VAE and umT5 on them), headline bench, the rank-emulation table, the counter refresh
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-c1}; R=$PWD
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 --no-header -p no:cacheprovider -s > gpurun_out/pytest_$T.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_$T.log
grep -E "passed|failed|FAILED|Error|whole-graph|rel-L2" gpurun_out/pytest_$T.log | tail -40
timeout 300 python tools/gemm16_bench.py > gpurun_out/gemm16_$T.jsonl 2>gpurun_out/gemm16_$T.err; cat gpurun_out/gemm16_$T.jsonl | cut -c1-260
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$T.log 2>&1; echo "bench exit $?" >> gpurun_out/bench_$T.log
grep '^{' gpurun_out/bench_$T.log | cut -c1-400
for spec in "0/2" "0/4" "0/8" "7/8"; do
  tag=$(echo $spec | tr / _)
  timeout 400 python bench.py --emulate-rank $spec --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/emu_c1_${tag}_$T.log 2>&1; echo "exit $?" >> gpurun_out/emu_c1_${tag}_$T.log
  grep '^{' gpurun_out/emu_c1_${tag}_$T.log | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('C1 emu $spec', r['dit_step_ms'], r['launch_mode'][:60], r['emulated_rank']['modelled_wire_ms_per_dit_step'], r['roofline']['avg_launch_ms'])" 2>&1 | tail -1
done
timeout 900 python bench.py --emulate-rank 0/8 --model Wan2.2-A14B --res 720p --two-experts --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/emu_c5_0_8_$T.log 2>&1; echo "exit $?" >> gpurun_out/emu_c5_0_8_$T.log
grep '^{' gpurun_out/emu_c5_0_8_$T.log | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('C5 emu 0/8', r['dit_step_ms'], r['launch_mode'][:60], r['emulated_rank']['modelled_wire_ms_per_dit_step'], r['roofline']['avg_launch_ms'])" 2>&1 | tail -1
grep '^\[bench' gpurun_out/emu_c5_0_8_$T.log | tail -4
bash tools/gpu/pmc_sq.sh $T > gpurun_out/pmc_sq_$T.log 2>&1; tail -30 gpurun_out/pmc_sq_$T.log
