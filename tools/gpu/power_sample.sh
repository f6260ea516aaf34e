#!/bin/bash
# socket power and clocks (rocm-smi / amd-smi, whichever answers) sampled while bench.py's timed region runs, beside the
# board's power cap: is the denoising loop power-limited?  -> gpurun_out/power_sample_$T.txt
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-pw}; O=gpurun_out/power_sample_$T.txt
{
echo "== caps / limits"; rocm-smi --showmaxpower 2>&1 | grep -i "max\|cap" | head -4
echo "== idle"; rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -i "power\|sclk\|mclk\|junction" | head -8
} > $O 2>&1
python bench.py --steps 40 --warmup 2 --no-cpu-baseline --no-box-calibration > gpurun_out/power_bench_$T.log 2>&1 &
BP=$!
sleep 6   # model build + warm-up + capture
echo "== during the timed region (one sample per ~0.25 s)" >> $O
for i in $(seq 1 28); do
  rocm-smi --showpower --showclocks 2>&1 | grep -i "power (w)\|socket power\|sclk" | tr '\n' ' ' >> $O; echo >> $O
  sleep 0.15
done
wait $BP
grep '^{' gpurun_out/power_bench_$T.log | cut -c1-200 >> $O
cat $O
