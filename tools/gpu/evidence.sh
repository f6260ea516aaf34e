#!/bin/bash
# THE evidence script (round 6): regenerates every profiles/r06_* file the bench line, DESIGN.md and the notebook cite.
#   bash tools/gpu/evidence.sh [TAG] [PARTS]     PARTS = comma list of: suite,bench,trace,pmc,configs,emu,gemm,ab (default: all)
# Everything lands under gpurun_out/ (scratch); copy what is quoted into profiles/r06_<name> (the mapping is printed at the end).
# The per-call scripts of earlier rounds live in tools/gpu/history/.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r06}; PARTS=${2:-suite,bench,trace,pmc,configs,emu,gemm,ab}; R=$PWD
has() { [[ ",$PARTS," == *",$1,"* ]]; }
if has suite; then
  timeout 2400 python -m pytest tests -m gpu -q --timeout=900 --no-header -p no:cacheprovider -s > gpurun_out/pytest_$T.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_$T.log
  grep -E "passed|failed|FAILED|rel-L2" gpurun_out/pytest_$T.log | cut -c1-300 | tail -40
  timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke_$T.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke_$T.log; tail -2 gpurun_out/smoke_$T.log
fi
if has bench; then
  timeout 1500 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_$T.log 2>&1; echo "bench exit $?" >> gpurun_out/bench_$T.log; grep '^{' gpurun_out/bench_$T.log | cut -c1-600
fi
if has trace; then   # rocprofv3 kernel-trace summaries: the production schedule, and one stream (each kernel's own duration)
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$T -o bench --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graph --no-box-calibration > $R/gpurun_out/prof_$T.log 2>&1)
  f=$(find gpurun_out/prof_$T -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_$T.csv
  find gpurun_out/prof_$T -name '*kernel_trace*' -size +20M -delete
  (cd /tmp && TD_BENCH_MODEL_FLAGS=two_streams=0,split_tokens=0,split_qkv=0 timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof1_$T -o bench --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-graph --no-box-calibration > $R/gpurun_out/prof1_$T.log 2>&1)
  f=$(find gpurun_out/prof1_$T -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_single_stream_$T.csv
  find gpurun_out/prof1_$T -name '*kernel_trace*' -size +20M -delete
  head -12 gpurun_out/kernel_stats_single_stream_$T.csv | cut -c1-160
fi
if has pmc; then     # counters: SQ sets over 6 layers, FETCH / WRITE over a whole forward (separate passes each; no trace domains beside --pmc)
  B="python $R/bench.py --steps 1 --warmup 0 --num-steps 1 --no-graph --no-cpu-baseline --no-box-calibration"
  csvs=""
  for pass in "sq1:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
              "sq2:SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" \
              "sq3:SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    (cd /tmp && TD_BENCH_MODEL_FLAGS=split_tokens=0,two_streams=0,split_qkv=0 timeout 500 rocprofv3 --pmc $ctrs --kernel-trace -d $R/gpurun_out/pmcs_${T}_$name -o b --output-format csv -- $B --layers 6 > $R/gpurun_out/pmcs_${T}_$name.log 2>&1)
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
fi
if has configs; then  # every BASELINE.json configuration on one GPU, one flag each
  : > gpurun_out/other_configs_$T.jsonl
  for c in C2 C3 C4 C5; do
    TD_BENCH_WATCHDOG_S=200 timeout 900 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/cfg_${T}_$c.log 2>&1; echo "[$c] exit $?" >> gpurun_out/cfg_${T}_$c.log
    grep '^{' gpurun_out/cfg_${T}_$c.log >> gpurun_out/other_configs_$T.jsonl || { echo "FAILED: $c"; tail -5 gpurun_out/cfg_${T}_$c.log; }
  done
  for a in "--model Wan2.1-14B --res 480p" "--workload c2w8a8" "--sage-pv fp8" "--gemm-exact"; do
    timeout 600 python bench.py $a --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | grep '^{' >> gpurun_out/other_configs_$T.jsonl
  done
  python - <<PY
import json
for l in open("gpurun_out/other_configs_$T.jsonl"):
    d=json.loads(l); r=d.get("roofline") or {}
    print(d["config"]["workload"][:64], "|", d["config"]["model"], d["config"]["resolution"], "|", d["dtype"][:48], "| ms/step", round(d["dit_step_ms"],1), "| videos/s", round(d["value"],4), "| roofline", r.get("kernel","")[:28], round(r.get("frac",0),3))
PY
fi
if has emu; then      # the emulated-rank table (per-rank compute measured, wire modelled with a per-collective latency term)
  OUT=gpurun_out/emu_table_$T.txt; : > $OUT
  timeout 200 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-box-calibration 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('C1 N=1 on this box: %.2f ms per DiT step' % r['dit_step_ms'])" | tee -a $OUT
  for n in 2 4 8; do
    timeout 300 python bench.py --emulate-rank 0/$n --steps 4 --warmup 2 --no-cpu-baseline --no-box-calibration > gpurun_out/emu_c1_0_${n}_$T.log 2>&1
    grep '^{' gpurun_out/emu_c1_0_${n}_$T.log | python -c "import sys,json; r=json.loads(sys.stdin.readline()); e=r['emulated_rank']; w=e['modelled_wire_ms_per_dit_step']; print('C1 N=$n groups %d parallel %s, %d collectives per layer: compute %.2f ms (emulated transfers alone %.2f), wire exposed %.2f (bytes only) | with 10 / 20 / 40 us per collective: %s' % (e['head_groups'], e['branches_in_parallel'], w['collectives_per_layer'], r['dit_step_ms'], e['of_which_emulation_gather_copies_ms'], w['first_head_group_exposed'], ' / '.join('%.2f' % (r['dit_step_ms'] + w['with_latency_us_per_collective'][k]) for k in ('10','20','40'))))" | tee -a $OUT
  done
  timeout 600 python bench.py --config C5 --steps 2 --warmup 1 --no-cpu-baseline --no-box-calibration 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('C5 N=1 on this box: %.1f ms per DiT step' % r['dit_step_ms'])" | tee -a $OUT
  timeout 600 python bench.py --emulate-rank 0/8 --config C5 --steps 2 --warmup 1 --no-cpu-baseline --no-box-calibration > gpurun_out/emu_c5_0_8_$T.log 2>&1
  grep '^{' gpurun_out/emu_c5_0_8_$T.log | python -c "import sys,json; r=json.loads(sys.stdin.readline()); e=r['emulated_rank']; w=e['modelled_wire_ms_per_dit_step']; print('C5 N=8 groups %d parallel %s, %d collectives per layer: compute %.1f ms (emulated transfers alone: %.1f), wire exposed %.1f (bytes only) | with 10 / 20 / 40 us per collective: %s' % (e['head_groups'], e['branches_in_parallel'], w['collectives_per_layer'], r['dit_step_ms'], e['of_which_emulation_gather_copies_ms'], w['first_head_group_exposed'], ' / '.join('%.1f' % (r['dit_step_ms'] + w['with_latency_us_per_collective'][k]) for k in ('10','20','40'))))" | tee -a $OUT
  # the real backend on one rank at full size: the `rccl` / `exposed_wait` records of the multi-rank bench line
  timeout 900 python bench.py --rccl-one-rank --steps 4 --warmup 1 --no-cpu-baseline --no-box-calibration > gpurun_out/rccl_one_rank_$T.log 2>&1
  grep '^{' gpurun_out/rccl_one_rank_$T.log | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('RCCL one rank: %.2f ms per DiT step; rccl record: %s; exposed_wait: %s' % (r['dit_step_ms'], json.dumps(r.get('rccl')), json.dumps(r.get('exposed_wait'))))" | cut -c1-900 | tee gpurun_out/rccl_one_rank_$T.txt
fi
if has gemm; then     # the block's six GEMMs on every tile form / dequant mode: us + J per launch, C1's rows and an eighth of them
  timeout 900 python tools/gemm_forms.py --forms 4,8 --fast 1,8 > gpurun_out/gemm_forms_$T.txt 2>&1; tail -26 gpurun_out/gemm_forms_$T.txt
  timeout 600 python tools/gemm_forms.py --rows 4096 --forms 0,4,6,8 --fast 8 --seconds 0.5 > gpurun_out/gemm_forms_m4096_$T.txt 2>&1; tail -24 gpurun_out/gemm_forms_m4096_$T.txt
fi
if has ab; then       # same-box A/B: exact dequant | one-VALU | + four-wave form for ffn.0 | + q|k|v (the default) | every GEMM
  TAG=fast_dequant_ab_$T REPS=3 bash tools/gpu/ab.sh "--gemm-exact --tune 13=16" "--tune 13=16" "--tune 13=1" "" "--tune 13=15"
fi
echo "copy: pytest_$T.log -> profiles/r06_pytest_gpu.txt, bench_$T.log's JSON line -> r06_bench.json, kernel_stats*_$T.csv -> r06_kernel_stats*.csv,"
echo "      pmc_sq_$T.json, pmc_hbm_traffic_$T.json, other_configs_$T.jsonl, emu_table_$T.txt, rccl_one_rank_$T.txt, gemm_forms*_$T.txt, fast_dequant_ab_$T.txt -> profiles/r06_*"
