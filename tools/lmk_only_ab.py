#!/usr/bin/env python3
"""Diagnostics: us/launch of LANDMARK-ONLY decodes (445 int landmarks, nothing else) at the batch sizes given, kernel from DAD3D_DECODE_KERNEL.

    [DAD3D_DECODE_KERNEL=split|split_f16] python tools/lmk_only_ab.py tag 256 2048"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dad_3dheads_amd import _lib, landmarks, synthetic  # noqa: E402
from dad_3dheads_amd.head_mesh import HeadMesh  # noqa: E402

tag, sizes = sys.argv[1], [int(x) for x in sys.argv[2:]]
st = synthetic.load_static()
hm = HeadMesh(flame_model=synthetic.synthetic_flame_model(0, st), landmarks=landmarks.canonical("445", st), static=st, device=0)
lib = _lib.load()
out = []
for b in sizes:
    p = torch.from_numpy(synthetic.synthetic_params(b, seed=b)).cuda()
    lp = torch.empty((b, 445, 2), dtype=torch.int32, device="cuda")
    call = (hm.flame._handle, p.data_ptr(), b, _lib.TO_2D | _lib.MUTATE_PARAMS, None, None, None, lp.data_ptr(), None)
    iters = max(300, 200000 // b)
    for _ in range(300):
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
    out.append(f"B{b} {best:.2f}")
print(f"LMK {tag:12s} " + "  ".join(out), flush=True)
