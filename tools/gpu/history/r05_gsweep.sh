#!/bin/bash
# head-group sweep of the emulated ranks with the round-5 layer (transfers on their own stream): compute per DiT step
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { # $1 = r/N, env passed through
  timeout 600 python bench.py --emulate-rank $1 --steps 4 --warmup 1 --no-cpu-baseline --no-box-calibration $2 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.readline()); e=r['emulated_rank']; print(round(r['ms_per_step']/4,2), 'ms; groups', e['head_groups'], 'parallel', e['branches_in_parallel'], 'piece0 wire', round(e['modelled_wire_ms_per_dit_step']['first_head_group_exposed'],2))"
}
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-box-calibration 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('N=1:', r['dit_step_ms'], 'ms per DiT step')"
echo "wire stream off (round-4 emulation), default groups:"; TD_EMU_WIRE_STREAM=0 run 0/8
for g in 1 2 3 4 6; do echo "0/8 groups $g:"; TD_SP_HEAD_GROUPS=$g run 0/8; done
for g in 2 4; do echo "0/4 groups $g:"; TD_SP_HEAD_GROUPS=$g run 0/4; done
for g in 2 4 6; do echo "0/2 groups $g:"; TD_SP_HEAD_GROUPS=$g run 0/2; done
echo "C5 0/8:"; for g in 2 4; do TD_SP_HEAD_GROUPS=$g run 0/8 "--config C5"; done
