#!/bin/bash
# Round-3 record run ON the GPU box; everything lands in gpurun_out/r03rec/ (copied into profiles/r03_* afterwards).
#   bash tools/r03_measure.sh [fast]      fast: skip the end-to-end CNN and the PMC passes
export TMPDIR=/tmp; root="${GRAFT_REPO_ROOT:-/root/repo}"; out="$root/gpurun_out/r03rec"; mkdir -p "$out"; cd "$root"
tr() { python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 1 "${@:3}" 2> "$out/$2.err" | grep '^{' > "$out/$2.json"; }
python bench.py --steps 20 --warmup 5 > "$out/bench_driver_args.json" 2> "$out/bench_driver_args.err"
python bench.py > "$out/bench.json" 2> "$out/bench.err"
tr 29531 bench_torchrun_n1 --steps 20 --warmup 5 --no-cpu-baseline
tr 29532 bench_torchrun_n1_2000 --no-cpu-baseline
python bench.py --workload render --steps 500 --warmup 50 > "$out/bench_render.json" 2> "$out/bench_render.err"
tr 29533 bench_render_torchrun_n1 --workload render --steps 500 --warmup 50 --no-cpu-baseline
(cd /tmp && rm -rf /tmp/prof_r03 && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r03 -- python $root/bench.py --no-cpu-baseline > /dev/null 2> "$out/rocprof.err"; f=$(find /tmp/prof_r03 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/bench_kernel_stats.csv")
(cd /tmp && rm -rf /tmp/prof_r03r && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r03r -- python $root/bench.py --workload render --steps 500 --warmup 50 --no-cpu-baseline > /dev/null 2>> "$out/rocprof.err"; f=$(find /tmp/prof_r03r -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/bench_render_kernel_stats.csv")
timeout 600 python tests/perf/bench_extra.py > "$out/bench_extra.json" 2> "$out/bench_extra.err"
timeout 300 bash tools/prof_sim3dr.sh product > "$out/sim3dr_kernels.txt" 2>&1
timeout 300 bash tools/ab_all.sh product > "$out/ab_decode.txt" 2>&1
if [ "$1" != fast ]; then
  timeout 900 python tools/bench_e2e.py 64 2> "$out/bench_e2e.err" | tail -1 > "$out/bench_e2e.json"
  timeout 600 bash tools/r03_pmc.sh > "$out/pmc.log" 2>&1
fi
for f in bench_driver_args bench bench_torchrun_n1 bench_torchrun_n1_2000 bench_render bench_render_torchrun_n1; do python - "$out/$f.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c=d["config"]
    print(sys.argv[1].split('/')[-1], round(d["value"]), d["unit"], "ms/step", round(d["ms_per_step"],5), "wall", round(d["wall_ms_per_step"],5), "frac", round(d["roofline"]["frac"],3), "verified", c.get("outputs_verified", c.get("gather_verified")), "gather_us", c.get("gather_us"), "in_region", c.get("gather_in_region_us"), "clock", d["roofline"].get("shader_clock_mhz"))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
head -4 "$out/bench_kernel_stats.csv" | cut -c1-160; head -5 "$out/bench_render_kernel_stats.csv" | cut -c1-160; cat "$out/sim3dr_kernels.txt" | tail -8; tail -2 "$out/ab_decode.txt"
