#!/bin/bash
# Round 6, first GPU call: numerics of the bf16x3 split, co-issue behaviour of the bf16 MFMAs, and the suite as it stands.
export TMPDIR=/tmp
root="${GRAFT_REPO_ROOT:-/root/repo}"; out="$root/gpurun_out/r06"; mkdir -p "$out"; cd "$root"
timeout 120 tools/split_probe > "$out/split_probe.txt" 2>&1; tail -5 "$out/split_probe.txt"
timeout 300 tools/coissue_probe_bf16_1 > "$out/coissue_bf16_16x16x32.txt" 2>&1
timeout 300 tools/coissue_probe_bf16_2 > "$out/coissue_bf16_32x32x16.txt" 2>&1
tail -3 "$out/coissue_bf16_32x32x16.txt"
timeout 900 python -m pytest tests -m gpu -x -q > "$out/pytest_gpu_start.txt" 2>&1; tail -3 "$out/pytest_gpu_start.txt"
