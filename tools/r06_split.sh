#!/bin/bash
# Round 6: the bf16x3 split kernel -- error and parity first, then time at several batch sizes next to the fp32 kernel.
export TMPDIR=/tmp
root="${GRAFT_REPO_ROOT:-/root/repo}"; out="$root/gpurun_out/r06"; mkdir -p "$out"; cd "$root"
timeout 900 python -m pytest tests/test_gpu_decode_split.py -x -q > "$out/pytest_split.txt" 2>&1; tail -25 "$out/pytest_split.txt"
( timeout 300 python tools/ab_sizes.py fp32 32 64 128 256 512 1024 2048
  DAD3D_DECODE_KERNEL=split timeout 300 python tools/ab_sizes.py split 32 64 128 256 512 1024 2048 ) 2>&1 | grep ABS | tee "$out/ab_split.txt"
[ "$1" = more ] && { timeout 900 python -m pytest tests/test_gpu_decode_fuzz.py tests/test_gpu_parity_pixels.py tests/test_gpu_landmark_subset.py -x -q > "$out/pytest_split_more.txt" 2>&1; tail -5 "$out/pytest_split_more.txt"; }
true
