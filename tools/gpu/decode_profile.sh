#!/bin/bash
# kernel-trace summary of one 480p VAE decode + encode (where the non-convolution time goes), then the f4 timings
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-dp}; R=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_dec_$T -o dec --output-format csv -- python $R/tools/f4_time.py vae480 > $R/gpurun_out/prof_dec_$T.log 2>&1)
f=$(find gpurun_out/prof_dec_$T -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_decode_$T.csv
find gpurun_out/prof_dec_$T -name '*kernel_trace*' -size +20M -delete
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_enc_$T -o enc --output-format csv -- python $R/tools/f4_time.py enc480 > $R/gpurun_out/prof_enc_$T.log 2>&1)
f=$(find gpurun_out/prof_enc_$T -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/kernel_stats_encode_$T.csv
find gpurun_out/prof_enc_$T -name '*kernel_trace*' -size +20M -delete
timeout 600 python tools/f4_time.py vae480 vae720 enc480 > gpurun_out/f4_$T.jsonl 2> gpurun_out/f4_$T.err; cut -c1-200 gpurun_out/f4_$T.jsonl
head -25 gpurun_out/kernel_stats_decode_$T.csv | cut -c1-200
