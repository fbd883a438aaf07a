#!/bin/bash
# Round 6: quick loop for the split kernel -- its test file, then timings (product + the variants named).
export TMPDIR=/tmp
root="${GRAFT_REPO_ROOT:-/root/repo}"; out="$root/gpurun_out/r06"; mkdir -p "$out"; cd "$root"
timeout 900 python -m pytest tests/test_gpu_decode_split.py -x -q > "$out/pytest_split.txt" 2>&1; tail -4 "$out/pytest_split.txt"
export DAD3D_DECODE_KERNEL="${SPLIT_FORM:-split}"
for v in product "$@"; do
  if [ "$v" = product ]; then unset DAD3D_LIB_PATH; else export DAD3D_LIB_PATH="$root/tools/_variants/lib_$v.so"; fi
  timeout 240 python tools/ab_sizes.py "$v" 64 256 1024 2048 2>&1 | grep -E "^ABS|Error|error" | tail -3
done | tee -a "$out/ab_split_quick.txt"
