#!/bin/bash
# Round 5: the other bench lines next to the driver's command (gpurun_out/r05/): plain 2000 steps, torchrun N = 1 at 20 and 2000 steps (aligned
# start + direct RCCL gather on a world of one), the render workload plain (gather to root is a no-op without a group) and under torchrun N = 1
# with both gather modes.
root="${GRAFT_REPO_ROOT:-/root/repo}"; out="$root/gpurun_out/r05"; mkdir -p "$out"; cd "$root"
tr() { python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 2000)) bench.py "$@"; }
python bench.py --no-cpu-baseline --no-secondary > "$out/bench_2000.json" 2>/dev/null
tr --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > "$out/bench_torchrun_n1.json" 2>/dev/null
tr --gpus 1 --steps 2000 --warmup 100 --no-cpu-baseline > "$out/bench_torchrun_n1_2000.json" 2>/dev/null
python bench.py --workload render --steps 500 --warmup 50 > "$out/bench_render.json" 2>/dev/null
tr --gpus 1 --workload render --steps 500 --warmup 50 --no-cpu-baseline > "$out/bench_render_torchrun_n1.json" 2>/dev/null
tr --gpus 1 --workload render --steps 500 --warmup 50 --no-cpu-baseline --gather all > "$out/bench_render_torchrun_n1_allgather.json" 2>/dev/null
python - "$out" <<'PY'
import json, sys, os
for f in ("bench_2000", "bench_torchrun_n1", "bench_torchrun_n1_2000", "bench_render", "bench_render_torchrun_n1", "bench_render_torchrun_n1_allgather"):
    try:
        d = json.loads([l for l in open(os.path.join(sys.argv[1], f + ".json")).read().strip().splitlines() if l.startswith("{")][-1])
        c = d["config"]
        print(f"{f:40s} value {d['value']:.0f} ms/step {d['ms_per_step']*1e3:.2f} us compute {d.get('ms_per_step_compute') and d['ms_per_step_compute']*1e3} "
              f"with_gather {d.get('ms_per_step_with_gather') and d['ms_per_step_with_gather']*1e3} skew {c.get('start_skew_us')} gather_us {c.get('gather_us')} {c.get('gather_mode','')}")
    except Exception as e:
        print(f, "FAILED", e)
PY
