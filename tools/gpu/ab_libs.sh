#!/bin/bash
# interleaved end-to-end comparison of several builds of the library: bash tools/gpu/ab_libs.sh ROUNDS lib1.so lib2.so ...
# ("-" = the in-tree library)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$1; shift
for r in $(seq 1 $R); do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset TD_LIB_PATH; else export TD_LIB_PATH=$PWD/$lib; fi
    tag=$(basename "$lib" .so)
    timeout 300 python bench.py --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline --no-box-calibration $EXTRA > gpurun_out/abl_${tag}_$r.log 2>&1
    python - <<PY
import json
l=[x for x in open("gpurun_out/abl_${tag}_$r.log") if x.startswith("{")]
d=json.loads(l[-1]); a=d.get("roofline_attention",{}); g=d["roofline"]
print("$tag run $r:", round(d["value"],4), "videos/s", round(d["dit_step_ms"],2), "ms per DiT step; attention", round(a.get("avg_launch_ms",0),4), "ms; GEMM avg", round(g.get("avg_launch_ms",0),4), "ms")
PY
  done
done
