#!/bin/bash
# Round 5 evidence in one gpurun call: the driver's bench command plain and under rocprofv3 (kernel trace + stats), the end-to-end
# predictor with the CPU comparator leg. Output under gpurun_out/r05/.
export TMPDIR=/tmp
root="${GRAFT_REPO_ROOT:-/root/repo}"; out="$root/gpurun_out/r05"; mkdir -p "$out"; cd "$root"
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > "$out/bench_driver_command.json" 2> "$out/bench_driver_command.err"
tail -3 "$out/bench_driver_command.err"
cd /tmp; rm -rf /tmp/prof_bench
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python "$root/bench.py" --gpus 1 --steps 20 --warmup 5 > "$out/bench_profiled.json" 2> "$out/bench_profiled.err"
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$out/bench_kernel_stats.csv"
python "$root/tools/r05_kernel_summary.py" /tmp/prof_bench "$out/bench_profiled.json" > "$out/bench_kernel_summary.md"; cat "$out/bench_kernel_summary.md"
cd "$root"
# the end-to-end leg re-tunes MIOpen and times the CPU predictor at four thread counts: ~8 minutes -- only on request
[ "$RUN_E2E" = 1 ] && { timeout 900 python tools/bench_e2e.py 64 --cpu --quick > "$out/bench_e2e.json" 2> "$out/bench_e2e.err"; tail -c 1500 "$out/bench_e2e.json"; }
true
