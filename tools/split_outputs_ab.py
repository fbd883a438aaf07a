#!/usr/bin/env python3
"""Diagnostics: us/launch of the decode with SUBSETS of the outputs (which store costs what), kernel from DAD3D_DECODE_KERNEL.

    DAD3D_DECODE_KERNEL=split python tools/split_outputs_ab.py 1024 2048"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dad_3dheads_amd import _lib, landmarks, synthetic  # noqa: E402
from dad_3dheads_amd.head_mesh import HeadMesh  # noqa: E402

sizes = [int(x) for x in sys.argv[1:]]
st = synthetic.load_static()
hm = HeadMesh(flame_model=synthetic.synthetic_flame_model(0, st), landmarks=landmarks.canonical("445", st), static=st, device=0)
lib = _lib.load()
for b in sizes:
    p = torch.from_numpy(synthetic.synthetic_params(b, seed=b)).cuda()
    v3 = torch.empty((b, 5023, 3), device="cuda"); pr = torch.empty((b, 5023, 2), device="cuda"); pr3 = torch.empty((b, 5023, 3), device="cuda")
    lp = torch.empty((b, 445, 2), dtype=torch.int32, device="cuda"); lx = torch.empty((b, 445, 2), device="cuda")
    cases = {"all(3d+2d+lp)": (_lib.TO_2D, v3, pr, None, lp), "3d+2d": (_lib.TO_2D, v3, pr, None, None), "3d": (_lib.TO_2D, v3, None, None, None),
             "2d": (_lib.TO_2D, None, pr, None, None), "proj3": (0, None, pr3, None, None), "3d+proj3": (0, v3, pr3, None, None),
             "3d+lp": (_lib.TO_2D, v3, None, None, lp), "3d+2d+lx+lp": (_lib.TO_2D, v3, pr, lx, lp)}
    out = []
    for name, (fl, a3, ap, alx, alp) in cases.items():
        ptr = lambda t: None if t is None else t.data_ptr()
        call = (hm.flame._handle, p.data_ptr(), b, fl | _lib.MUTATE_PARAMS, ptr(a3), ptr(ap), ptr(alx), ptr(alp), None)
        iters = max(200, 200000 // b)
        for _ in range(100):
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
        out.append(f"{name} {best:.1f}")
    print(f"OUTS B{b}: " + " | ".join(out), flush=True)
