#!/usr/bin/env python3
"""Diagnostics: fused-decode time per launch for small and ragged batches (back-to-back launches, one stream)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dad_3dheads_amd import _lib, landmarks, synthetic
from dad_3dheads_amd.head_mesh import HeadMesh
st = synthetic.load_static()
hm = HeadMesh(flame_model=synthetic.synthetic_flame_model(0, st), landmarks=landmarks.canonical("445", st), static=st, device=0)
out = {}
for b in (1, 8, 16, 17, 32, 48, 64, 65, 80, 96, 128):
    p = torch.from_numpy(synthetic.synthetic_params(b, seed=b)).cuda()
    buf = {}
    for _ in range(50): hm.decode(p, landmarks=False, landmarks_px=True, out=buf)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(1000): hm.decode(p, landmarks=False, landmarks_px=True, out=buf)
    torch.cuda.synchronize(); out[b] = round((time.perf_counter() - t0) / 1000 * 1e6, 2)
print("us per launch by batch:", out)
