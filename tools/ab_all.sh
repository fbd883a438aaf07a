#!/bin/bash
# Runs ON the GPU box: A/B every variant library given (names under tools/_variants/), each guarded by a timeout.
#   gpurun --timeout 600 -- 'bash tools/ab_all.sh aux0 aux2 aux16'
root="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$root"; mkdir -p gpurun_out
for v in "$@"; do
  if [ "$v" = product ]; then unset DAD3D_LIB_PATH; else export DAD3D_LIB_PATH="$root/tools/_variants/lib_$v.so"; fi
  timeout 240 python tools/ab_decode.py "$v" 2>&1 | grep -E "^AB|Error|error" | tail -3
done | tee -a gpurun_out/ab_all.log
