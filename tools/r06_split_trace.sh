#!/bin/bash
# Round 6: kernel durations (rocprofv3 --kernel-trace --stats) of the split mode's two kernels at the batch sizes given, product and variants.
export TMPDIR=/tmp
root="${GRAFT_REPO_ROOT:-/root/repo}"; out="$root/gpurun_out/r06/split_trace"; mkdir -p "$out"; cd /tmp
export DAD3D_DECODE_KERNEL="${SPLIT_FORM:-split}"
sizes="$1"; shift
for v in product "$@"; do
  if [ "$v" = product ]; then unset DAD3D_LIB_PATH; else export DAD3D_LIB_PATH="$root/tools/_variants/lib_$v.so"; fi
  for b in $sizes; do
    rm -rf "/tmp/tr_$v_$b"
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "/tmp/tr_${v}_$b" -- python "$root/tools/ab_sizes.py" "$v" "$b" > /dev/null 2>&1
    f=$(find "/tmp/tr_${v}_$b" -name "*kernel_stats.csv" | head -1)
    echo "== $v B$b"; [ -n "$f" ] && grep -E "split|Name" "$f" | cut -d, -f1-6 | sed 's/flame_decode_split_kernel/tile/;s/split_params_kernel/prepass/'
  done
done | tee "$out/summary.txt"
