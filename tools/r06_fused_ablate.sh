#!/bin/bash
# Round 6: anatomy of the split mode's one-launch form: builds without finishing (1), beta loads (32), beta split (64), constants (128).
export TMPDIR=/tmp
root="${GRAFT_REPO_ROOT:-/root/repo}"; out="$root/gpurun_out/r06"; mkdir -p "$out"; cd "$root"
export DAD3D_DECODE_KERNEL=split
for v in product "$@"; do
  if [ "$v" = product ]; then unset DAD3D_LIB_PATH; else export DAD3D_LIB_PATH="$root/tools/_variants/lib_$v.so"; fi
  timeout 240 python tools/ab_sizes.py "$v" 16 32 48 64 2>&1 | grep -E "^ABS|rror" | tail -3
done | tee "$out/ab_fused_ablate.txt"
