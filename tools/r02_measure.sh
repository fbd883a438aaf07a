#!/bin/bash
# Round-2 record run ON the GPU box: the driver's bench command, the default bench, the torchrun N=1 line, the render
# workload, rocprofv3 kernel stats of the default bench, secondary measurements. Everything lands in gpurun_out/r02/.
export TMPDIR=/tmp; root="${GRAFT_REPO_ROOT:-/root/repo}"; out="$root/gpurun_out/r02"; mkdir -p "$out"; cd "$root"
python bench.py --steps 20 --warmup 5 > "$out/bench_driver_args.json" 2> "$out/bench_driver_args.err"
python bench.py > "$out/bench.json" 2> "$out/bench.err"
python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline 2> "$out/bench_torchrun.err" | grep '^{' > "$out/bench_torchrun_n1.json"
python bench.py --workload render --steps 500 --warmup 50 > "$out/bench_render.json" 2> "$out/bench_render.err"
python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --workload render --steps 500 --warmup 50 --no-cpu-baseline 2>> "$out/bench_render.err" | grep '^{' > "$out/bench_render_torchrun_n1.json"
(cd /tmp && rm -rf /tmp/prof_r02 && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r02 -- python $root/bench.py --no-cpu-baseline > /dev/null 2> "$out/rocprof.err"; f=$(find /tmp/prof_r02 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/bench_kernel_stats.csv")
(cd /tmp && rm -rf /tmp/prof_r02r && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r02r -- python $root/bench.py --workload render --steps 500 --warmup 50 --no-cpu-baseline > /dev/null 2>> "$out/rocprof.err"; f=$(find /tmp/prof_r02r -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/bench_render_kernel_stats.csv")
timeout 600 python tests/perf/bench_extra.py > "$out/bench_extra.json" 2> "$out/bench_extra.err"
for f in bench_driver_args bench bench_torchrun_n1 bench_render bench_render_torchrun_n1; do python - "$out/$f.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], round(d["value"]), d["unit"], "ms/step", round(d["ms_per_step"],5), "wall", round(d["wall_ms_per_step"],5), "frac", round(d["roofline"]["frac"],3), "verified", d["config"].get("outputs_verified", d["config"].get("gather_verified")))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
head -5 "$out/bench_kernel_stats.csv"; head -6 "$out/bench_render_kernel_stats.csv"
