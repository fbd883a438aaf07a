#!/bin/bash
# Round 6: does the split kernel's time depend on the DATA through the clock? Samples sclk / power while the B = 2048 loop runs on the
# product build and on the build whose pre-pass leaves the scratch zero (a2048).
export TMPDIR=/tmp
root="${GRAFT_REPO_ROOT:-/root/repo}"; out="$root/gpurun_out/r06"; mkdir -p "$out"; cd "$root"
export DAD3D_DECODE_KERNEL=split
for v in product "$@"; do
  if [ "$v" = product ]; then unset DAD3D_LIB_PATH; else export DAD3D_LIB_PATH="$root/tools/_variants/lib_$v.so"; fi
  ( timeout 240 python tools/ab_sizes.py "$v" 2048 2048 2048 2048 2048 2048 2>&1 | grep -E "^ABS|rror" ) &
  pid=$!
  sleep 6
  for i in 1 2 3 4 5 6; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)|Socket Power|Average Graphics" | tr -s ' ' | tr '\n' ';'; echo
    sleep 0.7
  done
  wait $pid
done 2>&1 | tee "$out/clock_probe.txt"
