#!/usr/bin/env python3
"""Diagnostics: N launches of the decode at one batch size on the kernel named (fp32 | split | split_f16), every output -- the workload of the PMC passes
of tools/r06_measure.sh (rocprofv3 --pmc ... -- python tools/split_pmc_driver.py split 256 200)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dad_3dheads_amd import _lib, landmarks, synthetic  # noqa: E402
from dad_3dheads_amd.head_mesh import HeadMesh  # noqa: E402

kernel, b, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
st = synthetic.load_static()
hm = HeadMesh(flame_model=synthetic.synthetic_flame_model(0, st), landmarks=landmarks.canonical("445", st), static=st, device=0)
hm.flame.select_kernel({"split": "split_bf16", "split_f16": "split_f16"}.get(kernel, "pipelined"))
lib = _lib.load()
p = torch.from_numpy(synthetic.synthetic_params(b, seed=b)).cuda()
v3 = torch.empty((b, 5023, 3), device="cuda"); pr = torch.empty((b, 5023, 3), device="cuda")
lp = torch.empty((b, 445, 2), dtype=torch.int32, device="cuda")
call = (hm.flame._handle, p.data_ptr(), b, _lib.MUTATE_PARAMS, v3.data_ptr(), pr.data_ptr(), None, lp.data_ptr(), None)
for _ in range(n):
    _lib.check(lib.dad3d_flame_decode(*call))
torch.cuda.synchronize()
