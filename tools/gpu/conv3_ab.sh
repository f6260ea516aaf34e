#!/bin/bash
# the 2-D-tile LDS-DMA convolution kernel (TD_TUNE_VAE_CONV = 7 / 8) against the row-tile default: parity tests, then the 480p decode
# with every convolution timed, interleaved
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-c3}
timeout 900 python -m pytest tests/test_gpu_f4.py -m gpu -q -x --no-header -p no:cacheprovider -k "vae_conv" > gpurun_out/pytest_conv_$T.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_conv_$T.log; tail -15 gpurun_out/pytest_conv_$T.log
OUT=gpurun_out/conv3_ab_$T.txt; : > $OUT
for rep in 1 2; do for v in ${VARIANTS:-0 7 8}; do
  F4_DETAIL=1 F4_VAE_CONV=$v timeout 300 python tools/f4_time.py vae480 >> $OUT 2>&1
done; done
[ -n "$2" ] && for v in ${VARIANTS:-0 7}; do F4_VAE_CONV=$v timeout 300 python tools/f4_time.py vae720 >> $OUT 2>&1; done
grep -E "TD_TUNE|seconds" $OUT | cut -c1-260
