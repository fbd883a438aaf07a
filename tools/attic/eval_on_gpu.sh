#!/bin/bash
# Runs ON the GPU box (through gpurun) against an alternate library built here with build_alt.sh:
#   gpurun --timeout 240 -- 'bash tools/attic/eval_on_gpu.sh tools/attic/alt.so'
# Guarded: a 25-second probe first; the tests, the phase trace and two 2000-step benches only if the probe came back.
root="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$root"
export DAD3D_LIB_PATH="$root/${1:-tools/attic/alt.so}"
[ -f "$DAD3D_LIB_PATH" ] || { echo "no $DAD3D_LIB_PATH (build_alt.sh first; *.so travels with the snapshot)"; exit 1; }
timeout 25 python tools/attic/probe_mw8.py 2>&1 | grep -v amdgpu.ids | tail -4 > /tmp/probe.txt; cat /tmp/probe.txt
grep -q "time-outs" /tmp/probe.txt || { echo "PROBE FAILED (hang or error): stopping"; exit 2; }
timeout 90 python -m pytest tests/test_gpu_decode.py tests/test_gpu_autograd.py -x -q -m gpu 2>&1 | tail -2
timeout 30 python tools/trace_decode.py 64 2>&1 | sed -n 3,11p
for i in 1 2; do
  timeout 60 python bench.py --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', round(d['value']), 'img/s', round(d['roofline']['kernel_us'],2), 'us', d['config']['outputs_verified'])"
done
unset DAD3D_LIB_PATH
timeout 60 python bench.py --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('product', round(d['value']), 'img/s', round(d['roofline']['kernel_us'],2), 'us')"
