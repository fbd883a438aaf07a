#!/bin/bash
# Round 6: which part of the split kernel's work lowers the shader clock? The instrumented build (-DDAD3D_SPLIT_CLOCKS: s_memtime cycles per
# s_memrealtime tick over a launch, printed every 256th launch) with parts of the work removed, synthetic rows, 2048 images.
export TMPDIR=/tmp
root="${GRAFT_REPO_ROOT:-/root/repo}"; out="$root/gpurun_out/r06"; mkdir -p "$out"; cd "$root"
export DAD3D_DECODE_KERNEL="${SPLIT_FORM:-split}"
for v in "$@"; do
  export DAD3D_LIB_PATH="$root/tools/_variants/lib_$v.so"
  timeout 240 python tools/ab_sizes.py "$v" 2048 2>&1 | grep -E "^ABS|CLK" | awk -v v="$v" '/CLK/{n++; mhz+=$9; cyc+=$11} /ABS/{abs=$0} END{printf "%-12s launches sampled %d  mean MHz %.0f  cycles/phase %.0f  | %s\n", v, n, mhz/n, cyc/n, abs}'
done | tee "$out/clock_ablate.txt"
