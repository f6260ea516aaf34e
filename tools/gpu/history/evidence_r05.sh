#!/bin/bash
# round-5 evidence at HEAD: full GPU suite, smoke, default bench (CPU baseline by BASELINE.md §3), kernel-trace stats (production
# schedule + single stream), SQ / traffic counters of the DiT kernels, every BASELINE configuration through `bench.py --config`,
# the emulated-rank table
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r05}; R=$PWD
timeout 1800 python -m pytest tests -m gpu -q --timeout=900 --no-header -p no:cacheprovider -s > gpurun_out/pytest_$T.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_$T.log; grep -E "passed|failed|FAILED|rel-L2" gpurun_out/pytest_$T.log | tail -30
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke_$T.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke_$T.log; tail -2 gpurun_out/smoke_$T.log
timeout 1500 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_$T.log 2>&1; echo "bench exit $?" >> gpurun_out/bench_$T.log; grep '^{' gpurun_out/bench_$T.log | cut -c1-600
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$T -o bench --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graph --no-box-calibration > $R/gpurun_out/prof_$T.log 2>&1)
f=$(find gpurun_out/prof_$T -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_$T.csv
find gpurun_out/prof_$T -name '*kernel_trace*' -size +20M -delete
(cd /tmp && TD_BENCH_MODEL_FLAGS=two_streams=0,split_tokens=0 timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof1_$T -o bench --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graph --no-box-calibration > $R/gpurun_out/prof1_$T.log 2>&1)
f=$(find gpurun_out/prof1_$T -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_single_stream_$T.csv
find gpurun_out/prof1_$T -name '*kernel_trace*' -size +20M -delete
# counters: SQ sets over 6 layers, FETCH / WRITE over a whole forward (separate passes each; no trace domains beside --pmc)
B="python $R/bench.py --steps 1 --warmup 0 --num-steps 1 --no-graph --no-cpu-baseline --no-box-calibration"
csvs=""
for pass in "sq1:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "sq2:SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" \
            "sq3:SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  (cd /tmp && TD_BENCH_MODEL_FLAGS=split_tokens=0,two_streams=0 timeout 500 rocprofv3 --pmc $ctrs --kernel-trace -d $R/gpurun_out/pmcs_${T}_$name -o b --output-format csv -- $B --layers 6 > $R/gpurun_out/pmcs_${T}_$name.log 2>&1)
  echo "pmc $name (DiT) exit $?"
  f=$(find $R/gpurun_out/pmcs_${T}_$name -name '*counter_collection.csv' | head -1); [ -n "$f" ] && csvs="$csvs $f"
done
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  (cd /tmp && TD_BENCH_MODEL_FLAGS=split_tokens=0,split_qkv=0 timeout 900 rocprofv3 --pmc $ctrs --kernel-trace -d $R/gpurun_out/pmcb_${T}_$name -o b --output-format csv -- $B > $R/gpurun_out/pmcb_${T}_$name.log 2>&1)
  echo "pmc $name exit $?"
done
python tools/pmc_sq.py gpurun_out/pmc_sq_$T.json $csvs | tail -14 | cut -c1-200
F=$(find gpurun_out/pmcb_${T}_fetch -name '*counter_collection.csv' | head -1)
W=$(find gpurun_out/pmcb_${T}_write -name '*counter_collection.csv' | head -1)
python tools/pmc_traffic.py $F $W gpurun_out/pmc_hbm_traffic_$T.json | tail -8 | cut -c1-200
find gpurun_out -name '*.csv' -size +4M -delete
# every BASELINE.json configuration on one GPU, one flag each
: > gpurun_out/other_configs_$T.jsonl
for c in C2 C3 C4 C5; do
  TD_BENCH_WATCHDOG_S=200 timeout 900 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/cfg_${T}_$c.log 2>&1; echo "[$c] exit $?" >> gpurun_out/cfg_${T}_$c.log
  grep '^{' gpurun_out/cfg_${T}_$c.log >> gpurun_out/other_configs_$T.jsonl || { echo "FAILED: $c"; tail -5 gpurun_out/cfg_${T}_$c.log; }
done
for a in "--model Wan2.1-14B --res 480p" "--workload c2w8a8" "--sage-pv fp8"; do
  timeout 600 python bench.py $a --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' >> gpurun_out/other_configs_$T.jsonl
done
python - <<PY
import json
for l in open("gpurun_out/other_configs_$T.jsonl"):
    d=json.loads(l); r=d.get("roofline") or {}
    print(d["config"]["workload"][:64], "|", d["config"]["model"], d["config"]["resolution"], "| ms/step", round(d["dit_step_ms"],1), "| videos/s", round(d["value"],4), "| roofline", r.get("kernel","")[:28], round(r.get("frac",0),3))
PY
OUT=gpurun_out/emu_table_$T.txt; : > $OUT
timeout 200 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-box-calibration 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('C1 N=1 on this box: %.2f ms per DiT step' % r['dit_step_ms'])" | tee -a $OUT
for n in 2 4 8; do
  timeout 300 python bench.py --emulate-rank 0/$n --steps 4 --warmup 2 --no-cpu-baseline --no-box-calibration > gpurun_out/emu_c1_0_${n}_$T.log 2>&1
  grep '^{' gpurun_out/emu_c1_0_${n}_$T.log | python -c "import sys,json; r=json.loads(sys.stdin.readline()); e=r['emulated_rank']; w=e['modelled_wire_ms_per_dit_step']; print('C1 N=$n groups %d parallel %s: compute %.2f ms (emulated transfers on their own stream; alone they take %.2f), wire exposed %.2f (model), sum %.2f' % (e['head_groups'], e['branches_in_parallel'], r['dit_step_ms'], e['of_which_emulation_gather_copies_ms'], w['first_head_group_exposed'], r['dit_step_ms'] + w['first_head_group_exposed']))" | tee -a $OUT
done
timeout 600 python bench.py --config C5 --steps 2 --warmup 1 --no-cpu-baseline --no-box-calibration 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('C5 N=1 on this box: %.1f ms per DiT step' % r['dit_step_ms'])" | tee -a $OUT
timeout 600 python bench.py --emulate-rank 0/8 --config C5 --steps 2 --warmup 1 --no-cpu-baseline --no-box-calibration > gpurun_out/emu_c5_0_8_$T.log 2>&1
grep '^{' gpurun_out/emu_c5_0_8_$T.log | python -c "import sys,json; r=json.loads(sys.stdin.readline()); e=r['emulated_rank']; w=e['modelled_wire_ms_per_dit_step']; print('C5 N=8 groups %d parallel %s: compute %.1f ms (emulated transfers alone: %.1f), wire exposed %.1f (model), sum %.1f' % (e['head_groups'], e['branches_in_parallel'], r['dit_step_ms'], e['of_which_emulation_gather_copies_ms'], w['first_head_group_exposed'], r['dit_step_ms'] + w['first_head_group_exposed']))" | tee -a $OUT
# the block's six GEMMs at the row counts of a sequence shard: 256-row tiles / 128-row tiles / the 128x128 kernel / the shipped plan
timeout 300 python tools/gemm_small_m.py --M 4096,8192,16384 > gpurun_out/gemm_small_m_$T.jsonl 2> gpurun_out/gemm_small_m_$T.err; tail -3 gpurun_out/gemm_small_m_$T.jsonl | cut -c1-200
# N = 1 beside the round-4 tree (a git worktree of the round-4 commit with its own build, when present) on THIS box
if [ -d .r04tree ]; then
  one() { d=$1; shift; (cd $d && timeout 600 python bench.py "$@" --no-cpu-baseline --no-box-calibration 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print(round(r['value'],4), 'videos/s', round(r['dit_step_ms'],2), 'ms per DiT step')"); }
  for i in 1 2 3; do echo "N=1 round 5: $(one $R --steps 6 --warmup 1)"; echo "N=1 round 4: $(one $R/.r04tree --steps 6 --warmup 1)"; done | tee gpurun_out/vs_r04_$T.txt
fi
