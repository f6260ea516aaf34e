#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp; T=${1:-r05e}
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --no-header -p no:cacheprovider -k "gemm" 2>&1 | tail -4
timeout 900 python tools/gemm_small_m.py --M 4096 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('M =', r['M'])
    names = sorted({k.rsplit(' ', 2)[0] for k in r if k != 'M'})
    for n in names:
        print('  %-22s' % n, '  '.join('%s %7.1f' % (t, r.get(f'{n} {t} us', float('nan'))) for t in ('tile256', 'tile128x256', 'tile128', 'auto')))
"
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-box-calibration 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('N=1:', r['value'], 'videos/s', r['dit_step_ms'], 'ms per DiT step')"
for rn in 0/8 0/4 0/2; do
  timeout 600 python bench.py --emulate-rank $rn --steps 4 --warmup 1 --no-cpu-baseline --no-box-calibration > gpurun_out/emu_${T}_${rn/\//of}.log 2>&1
  grep '^{' gpurun_out/emu_${T}_${rn/\//of}.log | python -c "import sys,json; r=json.loads(sys.stdin.readline()); e=r['emulated_rank']; print('emu $rn', round(r['ms_per_step']/4,2), 'ms per DiT step; copies', round(e['of_which_emulation_gather_copies_ms'],2), 'groups', e['head_groups'])" || tail -5 gpurun_out/emu_${T}_${rn/\//of}.log
done
