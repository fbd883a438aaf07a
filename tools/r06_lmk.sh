#!/bin/bash
# Round 6: the landmark sub-model on the pipelined kernel -- its tests, the pipelined kernel's tests, and the bench line.
export TMPDIR=/tmp
root="${GRAFT_REPO_ROOT:-/root/repo}"; out="$root/gpurun_out/r06"; mkdir -p "$out"; cd "$root"
timeout 900 python -m pytest tests/test_gpu_landmark_subset.py tests/test_gpu_decode_pipe.py tests/test_gpu_sharding.py -x -q > "$out/pytest_lmk.txt" 2>&1; tail -15 "$out/pytest_lmk.txt"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_lmk.json" 2> "$out/bench_lmk.err"; tail -2 "$out/bench_lmk.err"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/bench_lmk.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"])
print(json.dumps(d["secondary"]["decode_b256"]["landmarks_only"], indent=1))
PY
