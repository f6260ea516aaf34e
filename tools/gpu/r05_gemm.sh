#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x --no-header -p no:cacheprovider -k "gemm" 2>&1 | tail -3
timeout 900 python tools/gemm_small_m.py --M 4096,8192,16384 2>&1 | grep -v amdgpu.ids | tee gpurun_out/gemm_small_m.jsonl | python -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): print(l.rstrip()); continue
    r = json.loads(l); print('M =', r['M'])
    names = sorted({k.rsplit(' ', 2)[0] for k in r if k != 'M'})
    for n in names:
        print('  %-22s' % n, '  '.join('%s %7.1f' % (t, r.get(f'{n} {t} us', float('nan'))) for t in ('tile256', 'tile128x256', 'tile128', 'auto')))
"
timeout 600 python tools/gemm_small_m.py --M 9472 --dim 5120 --ffn 13824 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('M =', r['M'], 'dim 5120')
    names = sorted({k.rsplit(' ', 2)[0] for k in r if k != 'M'})
    for n in names:
        print('  %-22s' % n, '  '.join('%s %7.1f' % (t, r.get(f'{n} {t} us', float('nan'))) for t in ('tile256', 'tile128x256', 'tile128', 'auto')))
"
