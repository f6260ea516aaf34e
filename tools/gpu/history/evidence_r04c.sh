#!/bin/bash
# at HEAD: the other BASELINE configurations and the emulated-rank table (the shipped group rule), as in evidence_r04.sh
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r04c}
bash tools/gpu/other_configs.sh $T 2>&1 | tail -10
OUT=gpurun_out/emu_table_$T.txt; : > $OUT
timeout 200 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-box-calibration 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('C1 N=1 on this box: %.2f ms per DiT step' % r['dit_step_ms'])" | tee -a $OUT
for n in 2 4 8; do
  timeout 300 python bench.py --emulate-rank 0/$n --steps 4 --warmup 2 --no-cpu-baseline --no-box-calibration 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.readline()); e=r['emulated_rank']; w=e['modelled_wire_ms_per_dit_step']; print('C1 N=$n groups %d parallel %s: compute %.2f ms (of which emulation copies %.2f), wire exposed %.2f, sum %.2f' % (e['head_groups'], e['branches_in_parallel'], r['dit_step_ms'], e['of_which_emulation_gather_copies_ms'], w['first_head_group_exposed'], r['dit_step_ms'] + w['first_head_group_exposed']))" | tee -a $OUT
done
timeout 600 python bench.py --model Wan2.2-A14B --res 720p --two-experts --steps 2 --warmup 1 --no-cpu-baseline --no-box-calibration 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('C5 N=1 on this box: %.1f ms per DiT step' % r['dit_step_ms'])" | tee -a $OUT
timeout 600 python bench.py --emulate-rank 0/8 --model Wan2.2-A14B --res 720p --two-experts --steps 2 --warmup 1 --no-cpu-baseline --no-box-calibration 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.readline()); e=r['emulated_rank']; w=e['modelled_wire_ms_per_dit_step']; print('C5 N=8 groups %d parallel %s: compute %.1f ms (of which emulation copies %.1f), wire exposed %.1f, sum %.1f' % (e['head_groups'], e['branches_in_parallel'], r['dit_step_ms'], e['of_which_emulation_gather_copies_ms'], w['first_head_group_exposed'], r['dit_step_ms'] + w['first_head_group_exposed']))" | tee -a $OUT
