#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for v in ${VARIANTS:-no_finalize no_topk no_qpool no_linear}; do
  echo "== $v"; ABLATE_ONLY=$v timeout 300 python tools/ablate_glue.py 3 2>&1 | grep -E "ms per DiT|fault|Error|error" | tail -4
done 2>&1 | tee gpurun_out/ablate_glue_${TAG:-a}.txt
