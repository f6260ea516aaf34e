#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp; T=${1:-r05b}
timeout 900 python -m pytest tests/test_gpu_r05.py tests/test_gpu_seqpar.py tests/test_gpu_sla.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -12
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-box-calibration 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('N=1:', r['value'], 'videos/s', r['dit_step_ms'], 'ms per DiT step')"
for rn in 0/8 0/4 0/2; do
  timeout 600 python bench.py --emulate-rank $rn --steps 4 --warmup 1 --no-cpu-baseline --no-box-calibration > gpurun_out/emu_${T}_${rn/\//of}.log 2>&1
  grep '^{' gpurun_out/emu_${T}_${rn/\//of}.log | python -c "import sys,json; r=json.loads(sys.stdin.readline()); e=r['emulated_rank']; print('emu $rn', round(r['ms_per_step']/4,2), 'ms per DiT step; copies', round(e['of_which_emulation_gather_copies_ms'],2), 'groups', e['head_groups'], e['modelled_wire_ms_per_dit_step'])" || tail -5 gpurun_out/emu_${T}_${rn/\//of}.log
done
