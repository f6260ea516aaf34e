#!/bin/bash
# round-4 evidence after the 2-D-tile convolution kernel, the RCCL watchdog drain and the tokenizer: full GPU suite, smoke, default
# bench line, prompt-to-pixels, f4 timings, SQ counters over one 480p VAE decode (the DiT kernels' counters of evidence_r04.sh
# stay valid: their sources are unchanged — bench.py checks the digest)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r04b}; R=$PWD
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 --no-header -p no:cacheprovider -s > gpurun_out/pytest_$T.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_$T.log; grep -E "passed|failed|FAILED" gpurun_out/pytest_$T.log | tail -5
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke_$T.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke_$T.log; tail -2 gpurun_out/smoke_$T.log
timeout 1200 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_$T.log 2>&1; echo "bench exit $?" >> gpurun_out/bench_$T.log; grep '^{' gpurun_out/bench_$T.log | cut -c1-400
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-box-calibration --prompt-to-pixels > gpurun_out/p2p_$T.log 2>&1
grep '^{' gpurun_out/p2p_$T.log | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('prompt to pixels:', r['prompt_to_pixels'])" | cut -c1-400
timeout 600 python tools/f4_time.py > gpurun_out/f4_$T.jsonl 2> gpurun_out/f4_$T.err; cut -c1-160 gpurun_out/f4_$T.jsonl
cd /tmp; csvs=""
for pass in "sq1:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "sq2:SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" \
            "sq3:SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  timeout 500 rocprofv3 --pmc $ctrs --kernel-trace -d $R/gpurun_out/pmcv_${T}_$name -o v --output-format csv -- python $R/tools/f4_time.py vae480 > $R/gpurun_out/pmcv_${T}_$name.log 2>&1
  echo "pmc $name (VAE decode) exit $?"
  f=$(find $R/gpurun_out/pmcv_${T}_$name -name '*counter_collection.csv' | head -1); [ -n "$f" ] && csvs="$csvs $f"
done
cd $R
python tools/pmc_sq.py gpurun_out/pmc_sq_vae_$T.json $csvs | tail -12 | cut -c1-220
find gpurun_out -name '*.csv' -size +4M -delete
