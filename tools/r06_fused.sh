#!/bin/bash
# Round 6: the split mode's one-launch form (<= 64 images) against its two-launch form and the fp32 kernel, same box.
export TMPDIR=/tmp
root="${GRAFT_REPO_ROOT:-/root/repo}"; out="$root/gpurun_out/r06"; mkdir -p "$out"; cd "$root"
timeout 900 python -m pytest tests/test_gpu_decode_split.py -x -q > "$out/pytest_split.txt" 2>&1; tail -4 "$out/pytest_split.txt"
{
DAD3D_DECODE_KERNEL=split timeout 240 python tools/ab_sizes.py split_one_launch 16 32 48 64 2>&1 | grep -E "^ABS|rror" | tail -3
DAD3D_DECODE_KERNEL=split DAD3D_SPLIT_FUSED_MAX_PHASES=0 timeout 240 python tools/ab_sizes.py split_two_launch 16 32 48 64 2>&1 | grep -E "^ABS|rror" | tail -3
timeout 240 python tools/ab_sizes.py fp32_default 16 32 48 64 2>&1 | grep -E "^ABS|rror" | tail -3
} | tee -a "$out/ab_split_fused.txt"
