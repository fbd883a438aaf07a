#!/bin/bash
# Round-4 record run ON the GPU box; everything lands in gpurun_out/r04rec/ (copied into profiles/r04_* afterwards).
#   bash tools/r04_measure.sh [fast]      fast: skip the PMC passes
export TMPDIR=/tmp; root="${GRAFT_REPO_ROOT:-/root/repo}"; out="$root/gpurun_out/r04rec"; mkdir -p "$out"; cd "$root"
tr() { python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 1 "${@:3}" 2> "$out/$2.err" > "$out/$2.json"; }
python bench.py --steps 20 --warmup 5 > "$out/bench_driver_args.json" 2> "$out/bench_driver_args.err"
python bench.py > "$out/bench.json" 2> "$out/bench.err"
tr 29531 bench_torchrun_n1 --steps 20 --warmup 5 --no-cpu-baseline
tr 29532 bench_torchrun_n1_2000 --no-cpu-baseline
python bench.py --workload render --steps 500 --warmup 50 > "$out/bench_render.json" 2> "$out/bench_render.err"
python bench.py --workload render --streams 2 --steps 500 --warmup 50 > "$out/bench_render_s2.json" 2> "$out/bench_render_s2.err"
(cd /tmp && rm -rf /tmp/prof_r04 && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r04 -- python $root/bench.py --no-cpu-baseline > /dev/null 2> "$out/rocprof.err"; f=$(find /tmp/prof_r04 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/bench_kernel_stats.csv")
(cd /tmp && rm -rf /tmp/prof_r04r && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r04r -- python $root/bench.py --workload render --steps 500 --warmup 50 --no-cpu-baseline > /dev/null 2>> "$out/rocprof.err"; f=$(find /tmp/prof_r04r -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/bench_render_kernel_stats.csv")
S="1 16 17 32 33 48 64 65 96 128 192 256 512 1024 2048"
(python tools/ab_sizes.py pipelined $S; DAD3D_DECODE_KERNEL=v1 python tools/ab_sizes.py two_role $S; python tools/ab_decode.py pipelined; DAD3D_DECODE_KERNEL=v1 python tools/ab_decode.py two_role) 2>&1 | grep -E "^ABS|^AB " > "$out/ab_decode.txt"
timeout 300 bash tools/prof_sim3dr.sh product > "$out/sim3dr_kernels.txt" 2>&1
for b in 64 256; do timeout 100 python tools/trace_pipe.py $b 2>&1 | grep -v amdgpu.ids | cut -c1-160; done > "$out/trace_pipe.txt"
if [ "$1" != fast ]; then
  PMC_OUT="$out/pmc_decode" timeout 700 bash tools/pmc_decode.sh > "$out/pmc_decode.log" 2>&1
fi
for f in bench_driver_args bench bench_torchrun_n1 bench_torchrun_n1_2000 bench_render bench_render_s2; do python - "$out/$f.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c=d["config"]
    print(sys.argv[1].split('/')[-1], round(d["value"]), d["unit"], "ms/step", round(d["ms_per_step"],5), "compute", d.get("ms_per_step_compute"), "with_gather", d.get("ms_per_step_with_gather"), "frac", round(d["roofline"]["frac"],3), "verified", c.get("outputs_verified", c.get("gather_verified")), c.get("timed_images_match_reference_raster"), "gather_us", c.get("gather_us"), "in_region", c.get("gather_in_region_us"), "clock", d["roofline"].get("shader_clock_mhz"), "lines", len(open(sys.argv[1]).read().strip().splitlines()))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
head -4 "$out/bench_kernel_stats.csv" | cut -c1-170; head -5 "$out/bench_render_kernel_stats.csv" | cut -c1-170; tail -8 "$out/sim3dr_kernels.txt"; cat "$out/ab_decode.txt"; tail -25 "$out/pmc_decode.log" 2>/dev/null | head -40
