#!/bin/bash
# emulated ranks of C1, this tree beside the round-4 tree (.r04tree), interleaved, same box
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
one() { d=$1; shift; (cd $d && timeout 600 python bench.py "$@" --no-cpu-baseline --no-box-calibration 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print(round(r['value'],4), 'videos/s', round(r['dit_step_ms'],2), 'ms per DiT step')"); }
for rn in 0/8 0/4 0/2; do for i in 1 2; do
  echo "emu $rn round 5: $(one $R --emulate-rank $rn --steps 4 --warmup 1)"
  echo "emu $rn round 4: $(one $R/.r04tree --emulate-rank $rn --steps 4 --warmup 1)"
done; done 2>&1 | tee gpurun_out/r05_emu_vs_r04.txt
