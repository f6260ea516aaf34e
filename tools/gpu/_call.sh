cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=m32_ab REPS=2 bash tools/gpu/ab.sh "" "--gemm-variant 5 --gemm-fast 4" "--tune 3=8" "--tune 3=16"
