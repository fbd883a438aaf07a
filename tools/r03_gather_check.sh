mkdir -p gpurun_out/r03
run() { python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $2 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03/bench_$1.json 2> gpurun_out/r03/bench_$1.err; }
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03/bench_plain20.json 2> gpurun_out/r03/bench_plain20.err
DAD3D_BENCH_REWARM=0 run torchrun20_rw0 29521
DAD3D_BENCH_REWARM=8 run torchrun20_rw8 29522
DAD3D_BENCH_REWARM=64 run torchrun20_rw8b 29523
for f in plain20 torchrun20_rw0 torchrun20_rw8 torchrun20_rw8b; do python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/r03/bench_$f.json") if l.startswith("{")][-1])
    print("$f", round(d["value"]), d["ms_per_step"]*1e3, d["wall_ms_per_step"]*1e3, d["config"].get("gather_us"), d["roofline"].get("shader_clock_mhz"), d["roofline"]["frac"], d["roofline"]["kernel_us"], d["config"].get("gather_in_region_us"), d["config"].get("gather_host_call_us"))
except Exception as e:
    print("$f failed", e); print(open("gpurun_out/r03/bench_$f.err").read()[-1500:])
PY
done
