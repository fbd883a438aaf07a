#!/bin/bash
# Round 6: the whole GPU suite + smoke, as the driver runs them.
export TMPDIR=/tmp
root="${GRAFT_REPO_ROOT:-/root/repo}"; out="$root/gpurun_out/r06"; mkdir -p "$out"; cd "$root"
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > "$out/pytest_gpu_all.txt" 2>&1; tail -6 "$out/pytest_gpu_all.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | tail -2
