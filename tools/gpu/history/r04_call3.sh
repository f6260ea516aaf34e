#!/bin/bash
# round 4, third GPU call: remaining test fixes + new fixtures, the attention row-sum A/B, the raster energy sweep, the
# rank-emulation table with the final group rule, f4 timings on the new kernels
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-c3}; R=$PWD
timeout 1200 python -m pytest tests/test_gpu_f4.py tests/test_gpu_r04.py tests/test_gpu_seqpar.py tests/test_gpu_sla.py -m gpu -q --timeout=900 --no-header -p no:cacheprovider -s > gpurun_out/pytest_$T.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_$T.log
grep -E "passed|failed|FAILED|Error|rel-L2|whole-graph|outside the bound" gpurun_out/pytest_$T.log | tail -30
bash tools/gpu/attn_rowsum_ab.sh $T
timeout 300 python tools/gemm_raster_energy.py > gpurun_out/gemm_raster_energy_$T.txt 2>&1; cat gpurun_out/gemm_raster_energy_$T.txt | tail -12
for spec in "0/2" "0/4" "0/8"; do
  tag=$(echo $spec | tr / _)
  timeout 400 python bench.py --emulate-rank $spec --steps 5 --warmup 2 --no-cpu-baseline --no-box-calibration > gpurun_out/emu_c1_${tag}_$T.log 2>&1; echo "exit $?" >> gpurun_out/emu_c1_${tag}_$T.log
  grep '^{' gpurun_out/emu_c1_${tag}_$T.log | python -c "import sys,json; r=json.loads(sys.stdin.readline()); e=r['emulated_rank']; print('C1 emu $spec', round(r['dit_step_ms'],2), 'groups', e['head_groups'], e['modelled_wire_ms_per_dit_step'], 'gemm us', round(r['roofline']['avg_launch_ms']*1e3,1), 'attn us', round(r['roofline_attention']['avg_launch_ms']*1e3,1))" 2>&1 | tail -1
done
timeout 600 python bench.py --model Wan2.2-A14B --res 720p --two-experts --steps 2 --warmup 1 --no-cpu-baseline --no-box-calibration > gpurun_out/c5_n1_$T.log 2>&1; echo "exit $?" >> gpurun_out/c5_n1_$T.log
grep '^{' gpurun_out/c5_n1_$T.log | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('C5 N=1', round(r['dit_step_ms'],1), 'ms per DiT step')" 2>&1 | tail -1
timeout 900 python bench.py --emulate-rank 0/8 --model Wan2.2-A14B --res 720p --two-experts --steps 2 --warmup 1 --no-cpu-baseline --no-box-calibration > gpurun_out/emu_c5_0_8_$T.log 2>&1; echo "exit $?" >> gpurun_out/emu_c5_0_8_$T.log
grep '^{' gpurun_out/emu_c5_0_8_$T.log | python -c "import sys,json; r=json.loads(sys.stdin.readline()); e=r['emulated_rank']; print('C5 emu 0/8', round(r['dit_step_ms'],1), 'groups', e['head_groups'], e['modelled_wire_ms_per_dit_step'])" 2>&1 | tail -1
timeout 600 python tools/f4_time.py vae480 enc480 umt5 > gpurun_out/f4_$T.jsonl 2>gpurun_out/f4_$T.err; cut -c1-300 gpurun_out/f4_$T.jsonl; tail -3 gpurun_out/f4_$T.err
