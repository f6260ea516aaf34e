cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sla.py tests/test_gpu_wan.py -m gpu -q -x --no-header -p no:cacheprovider 2>&1 | tail -5
timeout 600 python tools/kbench.py --iters 10 --only attn 2>&1 | tail -6
