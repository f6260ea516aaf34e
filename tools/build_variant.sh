#!/bin/bash
# Build a second copy of the library with ONE source file taken from another git revision (or another path), for
# interleaved A/B runs of compile-time changes on one box (tools/gpu/ab_lib.sh; TD_LIB_PATH selects the library):
#   bash tools/build_variant.sh attn.hip HEAD turbodiffusion_amd/libtd_base.so      (VFLAGS="-DX=1": extra compile flags; a file path instead of a revision also works)
set -e
cd "$(dirname "$0")/.."
F=$1; REV=$2; OUT=${3:-turbodiffusion_amd/libtd_base.so}
C=turbodiffusion_amd/csrc
python -m turbodiffusion_amd.build > /dev/null
if [ -f "$REV" ]; then cp "$REV" $C/_variant_$F; else git show "$REV:$C/$F" > $C/_variant_$F; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wno-unused-result $VFLAGS -c $C/_variant_$F -o $C/_variant.o
objs=$(ls $C/*.o | grep -v "_variant.o" | grep -v "/${F%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $objs $C/_variant.o
rm -f $C/_variant_$F $C/_variant.o
echo $OUT
