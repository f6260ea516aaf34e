#!/bin/bash
# round-4 evidence at HEAD: full GPU suite, smoke, default bench (CPU baseline), bench with its own traffic counters, kernel-trace
# stats (production schedule + single stream), SQ / traffic counters (pmc_sq.sh), the other BASELINE configurations, the
# emulated-rank table with the shipped group rule, prompt-to-pixels
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r04}; R=$PWD
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 --no-header -p no:cacheprovider -s > gpurun_out/pytest_$T.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_$T.log; grep -E "passed|failed|FAILED|rel-L2" gpurun_out/pytest_$T.log | tail -30
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke_$T.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke_$T.log; tail -2 gpurun_out/smoke_$T.log
timeout 1200 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_$T.log 2>&1; echo "bench exit $?" >> gpurun_out/bench_$T.log; grep '^{' gpurun_out/bench_$T.log | cut -c1-600
timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --collect-traffic > gpurun_out/bench_traffic_$T.log 2>&1; echo "exit $?" >> gpurun_out/bench_traffic_$T.log
grep '^{' gpurun_out/bench_traffic_$T.log | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('live traffic:', r['roofline'].get('traffic'), r['roofline'].get('traffic_source','')[:40], r['roofline_attention'].get('traffic'), r['roofline_attention'].get('hbm_frac'))" 2>&1 | tail -1
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$T -o bench --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graph --no-box-calibration > $R/gpurun_out/prof_$T.log 2>&1)
f=$(find gpurun_out/prof_$T -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_$T.csv
find gpurun_out/prof_$T -name '*kernel_trace*' -size +20M -delete
(cd /tmp && TD_BENCH_MODEL_FLAGS=two_streams=0,split_tokens=0 timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof1_$T -o bench --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graph --no-box-calibration > $R/gpurun_out/prof1_$T.log 2>&1)
f=$(find gpurun_out/prof1_$T -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_single_stream_$T.csv
find gpurun_out/prof1_$T -name '*kernel_trace*' -size +20M -delete
bash tools/gpu/pmc_sq.sh $T > gpurun_out/pmc_sq_$T.log 2>&1; tail -22 gpurun_out/pmc_sq_$T.log | cut -c1-200
bash tools/gpu/other_configs.sh $T 2>&1 | tail -10
OUT=gpurun_out/emu_table_$T.txt; : > $OUT
timeout 200 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-box-calibration 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('C1 N=1 on this box: %.2f ms per DiT step' % r['dit_step_ms'])" | tee -a $OUT
for n in 2 4 8; do
  timeout 300 python bench.py --emulate-rank 0/$n --steps 4 --warmup 2 --no-cpu-baseline --no-box-calibration > gpurun_out/emu_c1_0_${n}_$T.log 2>&1
  grep '^{' gpurun_out/emu_c1_0_${n}_$T.log | python -c "import sys,json; r=json.loads(sys.stdin.readline()); e=r['emulated_rank']; w=e['modelled_wire_ms_per_dit_step']; print('C1 N=$n groups %d parallel %s: compute %.2f ms (of which emulation copies %.2f), wire exposed %.2f, sum %.2f' % (e['head_groups'], e['branches_in_parallel'], r['dit_step_ms'], e['of_which_emulation_gather_copies_ms'], w['first_head_group_exposed'], r['dit_step_ms'] + w['first_head_group_exposed']))" | tee -a $OUT
done
timeout 600 python bench.py --model Wan2.2-A14B --res 720p --two-experts --steps 2 --warmup 1 --no-cpu-baseline --no-box-calibration 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('C5 N=1 on this box: %.1f ms per DiT step' % r['dit_step_ms'])" | tee -a $OUT
timeout 600 python bench.py --emulate-rank 0/8 --model Wan2.2-A14B --res 720p --two-experts --steps 2 --warmup 1 --no-cpu-baseline --no-box-calibration > gpurun_out/emu_c5_0_8_$T.log 2>&1
grep '^{' gpurun_out/emu_c5_0_8_$T.log | python -c "import sys,json; r=json.loads(sys.stdin.readline()); e=r['emulated_rank']; w=e['modelled_wire_ms_per_dit_step']; print('C5 N=8 groups %d parallel %s: compute %.1f ms (of which emulation copies %.1f), wire exposed %.1f, sum %.1f' % (e['head_groups'], e['branches_in_parallel'], r['dit_step_ms'], e['of_which_emulation_gather_copies_ms'], w['first_head_group_exposed'], r['dit_step_ms'] + w['first_head_group_exposed']))" | tee -a $OUT
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-box-calibration --prompt-to-pixels > gpurun_out/p2p_$T.log 2>&1
grep '^{' gpurun_out/p2p_$T.log | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('prompt to pixels:', r['prompt_to_pixels'])" | cut -c1-400
