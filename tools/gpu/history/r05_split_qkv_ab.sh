#!/bin/bash
# q|k|v projection as K|V then Q with the K-side glue under the Q GEMM: parity tests, then same-box A/B against the one-launch form and round 4
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
timeout 600 python -m pytest tests/test_gpu_wan.py tests/test_gpu_c1.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -3
one() { d=$1; shift; (cd $d && timeout 600 python bench.py "$@" --no-cpu-baseline --no-box-calibration 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.readline()); ra=r['roofline_attention']; print(round(r['value'],4), 'videos/s', round(r['dit_step_ms'],2), 'ms per DiT step; gemm', round(r['roofline']['avg_launch_ms']*1e3,1), 'attn', round(ra['avg_launch_ms']*1e3,1))"); }
for i in 1 2 3; do
  echo "r05 split q|k|v:    $(one $R --steps 6 --warmup 1)"
  echo "r05 one launch:     $(TD_BENCH_MODEL_FLAGS=split_qkv=0 one $R --steps 6 --warmup 1)"
  echo "r04:                $(one $R/.r04tree --steps 6 --warmup 1)"
done 2>&1 | tee gpurun_out/r05_split_qkv_ab.txt
