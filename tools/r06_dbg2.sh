#!/bin/bash
root="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$root"
for v in "$@"; do
  export DAD3D_LIB_PATH="$root/tools/_variants/lib_$v.so"
  echo "== $v"; timeout 200 python tools/split_debug.py ${SIZES:-2048} 2>&1 | grep -E "split debug|  tile .* phase|^B=|bad tiles" | head -${LINES:-8} | cut -c1-200
done
