#!/usr/bin/env python3
"""Diagnostics: does a decode launch's time depend on the VALUES of params? us/launch at the batch sizes given with synthetic rows,
all-zero rows, and rows of 4x the usual magnitude -- kernel from DAD3D_DECODE_KERNEL (split | split_f16 | unset).

    [DAD3D_DECODE_KERNEL=split] python tools/data_dependence_ab.py 256 2048"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dad_3dheads_amd import _lib, landmarks, synthetic  # noqa: E402
from dad_3dheads_amd.head_mesh import HeadMesh  # noqa: E402

st = synthetic.load_static()
hm = HeadMesh(flame_model=synthetic.synthetic_flame_model(0, st), landmarks=landmarks.canonical("445", st), static=st, device=0)
lib = _lib.load()
for b in [int(x) for x in sys.argv[1:]]:
    base = torch.from_numpy(synthetic.synthetic_params(b, seed=b)).cuda()
    v3 = torch.empty((b, 5023, 3), device="cuda"); pr = torch.empty((b, 5023, 2), device="cuda")
    lp = torch.empty((b, 445, 2), dtype=torch.int32, device="cuda")
    out = []
    for name, p in (("synthetic", base.clone()), ("zeros", torch.zeros_like(base)), ("4x", base * 4), ("synthetic again", base.clone())):
        call = (hm.flame._handle, p.data_ptr(), b, _lib.TO_2D | _lib.MUTATE_PARAMS, v3.data_ptr(), pr.data_ptr(), None, lp.data_ptr(), None)
        iters = max(200, 200000 // b)
        for _ in range(200):
            _lib.check(lib.dad3d_flame_decode(*call))
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                lib.dad3d_flame_decode(*call)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / iters * 1e3)
        out.append(f"{name} {best:.2f}")
    print(f"DATA {os.environ.get('DAD3D_DECODE_KERNEL', 'default'):8s} B{b}: " + " | ".join(out), flush=True)
