#!/bin/bash
# Round 6: where the split kernel's phase goes -- the same call on builds without finishing (1), staging (2), MFMAs (4), stores.
export TMPDIR=/tmp
root="${GRAFT_REPO_ROOT:-/root/repo}"; out="$root/gpurun_out/r06"; mkdir -p "$out"; cd "$root"
export DAD3D_DECODE_KERNEL="${SPLIT_FORM:-split}"
for v in product "$@"; do
  if [ "$v" = product ]; then unset DAD3D_LIB_PATH; else export DAD3D_LIB_PATH="$root/tools/_variants/lib_$v.so"; fi
  timeout 240 python tools/ab_sizes.py "$v" 64 256 1024 2048 2>&1 | grep -E "^ABS|Error|error" | tail -3
done | tee "$out/ab_split_ablate.txt"
