#!/bin/bash
# round 4, second GPU call: the tests that failed / are new, the emulated rank's kernel trace (where do 33 ms go?), A/B of the
# parallel head-group branches and of the group count
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-c2}; R=$PWD
timeout 900 python -m pytest tests/test_gpu_f4.py tests/test_gpu_r03.py tests/test_gpu_seqpar.py tests/test_gpu_bench.py "tests/test_gpu_wan.py::test_graph_replay_sees_a_new_prompt_at_a_recycled_address" -m gpu -q --timeout=600 --no-header -p no:cacheprovider -s > gpurun_out/pytest_$T.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_$T.log
grep -E "passed|failed|FAILED|Error|rel-L2|whole-graph" gpurun_out/pytest_$T.log | tail -30
emu() {  # tag, env..., then bench args
  tag=$1; shift
  env "$@" timeout 400 python bench.py --emulate-rank 0/8 --steps 5 --warmup 2 --no-cpu-baseline --no-box-calibration > gpurun_out/emu_${tag}_$T.log 2>&1; echo "exit $?" >> gpurun_out/emu_${tag}_$T.log
  grep '^{' gpurun_out/emu_${tag}_$T.log | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('C1 emu 0/8 $tag', round(r['dit_step_ms'],2), 'gemm avg us', round(r['roofline']['avg_launch_ms']*1e3,1), 'attn avg us', round(r['roofline_attention']['avg_launch_ms']*1e3,1))" 2>&1 | tail -1
}
emu par4 TD_SP_PARALLEL_GROUPS=1
emu seq4 TD_SP_PARALLEL_GROUPS=0
emu seq1 TD_SP_PARALLEL_GROUPS=0 TD_SP_HEAD_GROUPS=1
emu par2 TD_SP_PARALLEL_GROUPS=1 TD_SP_HEAD_GROUPS=2
# kernel trace of one emulated rank (eager enqueue: one record per launch)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_emu_$T -o k --output-format csv -- python $R/bench.py --emulate-rank 0/8 --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-box-calibration > $R/gpurun_out/prof_emu_$T.log 2>&1)
f=$(find gpurun_out/prof_emu_$T -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_emu_0_8_$T.csv && head -45 "$f" | cut -c1-150
find gpurun_out/prof_emu_$T -name '*kernel_trace*' -size +20M -delete
