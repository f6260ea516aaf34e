#!/bin/bash
# round 4, fourth GPU call: GEMM tests (gated GELU rounding, split-K), umT5 timing with split-K, the group-count / branch sweep
# of the emulated ranks (C1 N = 2, 4, 8; C5 N = 8) that the rule in seqpar.groups_for is read off
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-c4}; R=$PWD
timeout 900 python -m pytest tests/test_gpu_f4.py tests/test_gpu_r04.py -m gpu -q --timeout=900 --no-header -p no:cacheprovider -s > gpurun_out/pytest_$T.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_$T.log
grep -E "passed|failed|FAILED|Error|rel-L2|outside the bound" gpurun_out/pytest_$T.log | tail -20
timeout 300 python tools/f4_time.py umt5 > gpurun_out/f4_umt5_$T.jsonl 2>/dev/null; cut -c1-200 gpurun_out/f4_umt5_$T.jsonl
timeout 200 python tools/gemm16_bench.py > gpurun_out/gemm16_$T.jsonl 2>/dev/null; cut -c1-230 gpurun_out/gemm16_$T.jsonl | grep -E "umt5|text"
OUT=gpurun_out/emu_group_sweep_$T.txt; : > $OUT
for n in 2 4 8; do
  for cfg in "1:0" "2:1" "4:1" "4:0"; do
    G=${cfg%%:*}; P=${cfg#*:}
    TD_SP_HEAD_GROUPS=$G TD_SP_PARALLEL_GROUPS=$P timeout 300 python bench.py --emulate-rank 0/$n --steps 4 --warmup 2 --no-cpu-baseline --no-box-calibration 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.readline()); e=r['emulated_rank']; w=e['modelled_wire_ms_per_dit_step']; print('C1 N=$n groups $G parallel $P: compute %.2f ms, wire exposed %.2f, sum %.2f' % (r['dit_step_ms'], w['first_head_group_exposed'], r['dit_step_ms'] + w['first_head_group_exposed']))" | tee -a $OUT
  done
done
timeout 200 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-box-calibration 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('C1 N=1 on this box: %.2f ms per DiT step' % r['dit_step_ms'])" | tee -a $OUT
for cfg in "4:0" "4:1" "2:0"; do
  G=${cfg%%:*}; P=${cfg#*:}
  TD_SP_HEAD_GROUPS=$G TD_SP_PARALLEL_GROUPS=$P timeout 600 python bench.py --emulate-rank 0/8 --model Wan2.2-A14B --res 720p --two-experts --steps 2 --warmup 1 --no-cpu-baseline --no-box-calibration 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.readline()); e=r['emulated_rank']; w=e['modelled_wire_ms_per_dit_step']; print('C5 N=8 groups $G parallel $P: compute %.1f ms, wire exposed %.1f, sum %.1f' % (r['dit_step_ms'], w['first_head_group_exposed'], r['dit_step_ms'] + w['first_head_group_exposed']))" | tee -a $OUT
done
timeout 600 python bench.py --model Wan2.2-A14B --res 720p --two-experts --steps 2 --warmup 1 --no-cpu-baseline --no-box-calibration 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('C5 N=1 on this box: %.1f ms per DiT step' % r['dit_step_ms'])" | tee -a $OUT
