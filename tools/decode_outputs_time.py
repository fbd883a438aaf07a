#!/usr/bin/env python3
"""Diagnostics: us per launch of the fused decode at one batch size for different output sets (which store stream costs what).

    python tools/decode_outputs_time.py [batch ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dad_3dheads_amd import _lib, landmarks, synthetic  # noqa: E402
from dad_3dheads_amd.head_mesh import HeadMesh  # noqa: E402

st = synthetic.load_static()
hm = HeadMesh(flame_model=synthetic.synthetic_flame_model(0, st), landmarks=landmarks.canonical("445", st), static=st, device=0)
lib = _lib.load()
for b in [int(x) for x in sys.argv[1:]] or [256]:
    p = torch.from_numpy(synthetic.synthetic_params(b, seed=b)).cuda()
    v3 = torch.empty((b, 5023, 3), device="cuda"); p2 = torch.empty((b, 5023, 2), device="cuda"); p3 = torch.empty((b, 5023, 3), device="cuda")
    lp = torch.empty((b, 445, 2), dtype=torch.int32, device="cuda")
    cases = {"2d: v3d+proj2+lmk": (_lib.TO_2D, v3, p2, lp), "3d: v3d+proj3+lmk": (0, v3, p3, lp), "3d: v3d+proj3": (0, v3, p3, None),
             "3d: proj3 only": (0, None, p3, None), "3d: v3d only": (0, v3, None, None), "2d: proj2 only": (_lib.TO_2D, None, p2, None),
             "2d: v3d+proj2": (_lib.TO_2D, v3, p2, None), "lmk_px only (sub-model unless DAD3D_LANDMARK_SUBSET=0)": (_lib.TO_2D, None, None, lp)}
    res = []
    for name, (fl, a, q, l) in cases.items():
        call = (hm.flame._handle, p.data_ptr(), b, fl | _lib.MUTATE_PARAMS, a.data_ptr() if a is not None else None,
                q.data_ptr() if q is not None else None, None, l.data_ptr() if l is not None else None, None)
        for _ in range(100):
            _lib.check(lib.dad3d_flame_decode(*call))
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(300):
                lib.dad3d_flame_decode(*call)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 300 * 1e3)
        res.append(f"{name} {best:.2f}")
    print(f"OUT B{b}: " + " | ".join(res), flush=True)
