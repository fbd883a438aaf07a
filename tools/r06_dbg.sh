#!/bin/bash
root="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$root"
for v in "$@"; do
  if [ "$v" = product ]; then unset DAD3D_LIB_PATH; else export DAD3D_LIB_PATH="$root/tools/_variants/lib_$v.so"; fi
  echo "== $v"; timeout 200 python tools/split_debug.py ${SIZES:-2048 2048} 2>&1 | grep -E "^B=|bad tiles|bad rows|Error|   tile|    bad|    fit" | cut -c1-330
done
