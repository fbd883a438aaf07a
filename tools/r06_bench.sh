#!/bin/bash
# Round 6: the driver's bench command, timed, with the new legs printed.
export TMPDIR=/tmp
root="${GRAFT_REPO_ROOT:-/root/repo}"; out="$root/gpurun_out/r06"; mkdir -p "$out"; cd "$root"
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > "$out/bench_driver_command.json" 2> "$out/bench_driver_command.err"
tail -4 "$out/bench_driver_command.err"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06/bench_driver_command.json").read().strip().splitlines()[-1])
print("value", d["value"], "compute", d["value_compute"], "with_gather", d["value_with_gather"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"])
print("toolchain", d["config"]["toolchain"]); print("note", d["config"]["scaling_note"])
s = d["secondary"]
print("b256", s["decode_b256"]["ms_per_step"], s["decode_b256"]["frac"], s["decode_b256"]["outputs_verified"])
print("split", json.dumps(s["decode_b256_split"]))
print("split_f16", json.dumps(s["decode_b256_split_f16"]))
print("b64 on the split forms", s["decode_b256_split"]["contract_step_b64"], s["decode_b256_split_f16"]["contract_step_b64"])
print("lmk", json.dumps(s["decode_b256"]["landmarks_only"]))
print("e2e", json.dumps(s["e2e_b64"]))
print("verified", s["outputs_verified"], "render", s["render_b64"]["us_per_batch"])
PY
