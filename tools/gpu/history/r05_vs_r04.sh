#!/bin/bash
# same-box comparison with the round-4 tree (.r04tree: `git worktree add .r04tree 15ece79` + its own build; not committed):
# default bench (N = 1), emulated ranks, interleaved
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
one() { # $1 = tree dir, rest = bench args
  d=$1; shift
  (cd $d && timeout 600 python bench.py "$@" --no-cpu-baseline --no-box-calibration 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print(round(r['value'],4), 'videos/s', round(r['dit_step_ms'],2), 'ms per DiT step')")
}
for i in 1 2 3; do
  echo "N=1 round 5: $(one $R --steps 6 --warmup 1)"
  echo "N=1 round 4: $(one $R/.r04tree --steps 6 --warmup 1)"
done
for rn in 0/8 0/4 0/2; do for i in 1 2; do
  echo "emu $rn round 5: $(one $R --emulate-rank $rn --steps 4 --warmup 1)"
  echo "emu $rn round 4: $(one $R/.r04tree --emulate-rank $rn --steps 4 --warmup 1)"
done; done
echo "C5 emu 0/8 round 5: $(one $R --emulate-rank 0/8 --model Wan2.2-A14B --res 720p --two-experts --steps 2 --warmup 1)"
echo "C5 emu 0/8 round 4: $(one $R/.r04tree --emulate-rank 0/8 --model Wan2.2-A14B --res 720p --two-experts --steps 2 --warmup 1)"
echo "C5 N=1 round 5: $(one $R --model Wan2.2-A14B --res 720p --two-experts --steps 2 --warmup 1)"
echo "C5 N=1 round 4: $(one $R/.r04tree --model Wan2.2-A14B --res 720p --two-experts --steps 2 --warmup 1)"
