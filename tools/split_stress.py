#!/usr/bin/env python3
"""Diagnostics: every vertex, projection and landmark of both split forms against the fp32 kernel, many launches at many batch sizes
(whole-array comparison: a rare per-lane fault shows as an outlier of the size of the result -- how the packed-f32 fault of section 4 of
profiles/r06_kernel_log.md was found). Prints one line per (form, batch) and a verdict; exit code 1 on any outlier.

    python tools/split_stress.py [repetitions per batch size = 6]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dad_3dheads_amd import landmarks, synthetic  # noqa: E402
from dad_3dheads_amd.head_mesh import HeadMesh  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
st = synthetic.load_static()
model = synthetic.synthetic_flame_model(0, st)
lm = landmarks.canonical("445", st)
ref = HeadMesh(flame_model=model, landmarks=lm, static=st, device=0)
ref.flame.select_kernel("pipelined")
forms = {}
for f in ("split_bf16", "split_f16"):
    forms[f] = HeadMesh(flame_model=model, landmarks=lm, static=st, device=0)
    forms[f].flame.select_kernel(f)
bad = 0
for batch in (1, 17, 64, 100, 256, 639, 640, 1000, 1024, 2048, 3000):
    for f, hm in forms.items():
        worst3, worstp, launches = 0.0, 0.0, 0
        for r in range(reps):
            to_2d = bool(r % 2)
            p = synthetic.synthetic_params(batch, seed=9000 + 17 * r + batch, profile="survey" if r % 3 == 2 else "crop")
            want = ref.decode(torch.from_numpy(p.copy()).cuda(), to_2d=to_2d, landmarks=True, landmarks_px=True)
            got = hm.decode(torch.from_numpy(p.copy()).cuda(), to_2d=to_2d, landmarks=True, landmarks_px=True)
            again = hm.decode(torch.from_numpy(p.copy()).cuda(), to_2d=to_2d, landmarks=True, landmarks_px=True)
            only = hm.decode(torch.from_numpy(p.copy()).cuda(), verts3d=False, proj=False, landmarks=True, landmarks_px=True)
            torch.cuda.synchronize()
            d3 = float((got["verts3d"] - want["verts3d"]).abs().max())
            dp = float((got["proj"] - want["proj"]).abs().max())
            same = all(torch.equal(got[k], again[k]) for k in got) and torch.equal(only["lmk_xy"], got["lmk_xy"]) and torch.equal(only["lmk_px"], got["lmk_px"])
            finite = bool(torch.isfinite(got["verts3d"]).all()) and bool(torch.isfinite(got["proj"]).all())
            worst3, worstp, launches = max(worst3, d3), max(worstp, dp), launches + 3
            if not (d3 < 1e-6 and dp < 5e-4 and same and finite):
                bad += 1
                print(f"OUTLIER {f} B={batch} rep={r} to_2d={to_2d}: 3-D {d3:.3e} px {dp:.3e} repeatable+landmark-only-equal {same} finite {finite}", flush=True)
        print(f"STRESS {f:10s} B={batch:5d}: {launches} launches, max |3-D - fp32 kernel| {worst3:.2e}, max |px| {worstp:.2e}", flush=True)
print("STRESS verdict:", "clean" if bad == 0 else f"{bad} outliers")
sys.exit(1 if bad else 0)
