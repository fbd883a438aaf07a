#!/usr/bin/env python3
"""Diagnostics (round 5): why is the decode 2.5 us slower inside the render chain than alone? Run under
`rocprofv3 --kernel-trace`; the decode launches of every phase are identified by launch order (each phase = N iterations).

  phase A  decode<false> (proj only, flip_z) alone, back to back
  phase B  decode<false> + geometry(+light, clear) + raster  (the render step of bench.py)
  phase C  decode<false> + a 192 MB device memset between launches (evicts the memory-side cache)
  phase D  decode<true> headline call (verts3d + proj2 + landmarks) alone
  phase E  decode<false> with verts3d too, alone
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dad_3dheads_amd import landmarks, synthetic  # noqa: E402
from dad_3dheads_amd.head_mesh import HeadMesh  # noqa: E402
from dad_3dheads_amd.Sim3DR import Mesh  # noqa: E402
from dad_3dheads_amd.sharding import ShardedRenderer  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
st = synthetic.load_static()
hm = HeadMesh(flame_model=synthetic.synthetic_flame_model(0, st), landmarks=landmarks.canonical("445", st), static=st, device=0)
p = torch.from_numpy(synthetic.synthetic_params(64, seed=2)).cuda()
mesh = Mesh(st["faces"], 5023, device=0)
r = ShardedRenderer(hm, mesh)
big = torch.empty(192 * 1024 * 1024, dtype=torch.uint8, device="cuda")
dec = {}
for _ in range(50):
    r.render_local(p)
torch.cuda.synchronize()
for _ in range(N):  # A
    hm.flame.decode(p, proj=True, to_2d=False, flip_z=True, out=dec)
torch.cuda.synchronize()
for _ in range(N):  # B
    r.render_local(p)
torch.cuda.synchronize()
for _ in range(N):  # C
    hm.flame.decode(p, proj=True, to_2d=False, flip_z=True, out=dec)
    big.zero_()
torch.cuda.synchronize()
d2 = {}
for _ in range(N):  # D
    hm.flame.decode(p, verts3d=True, proj=True, to_2d=True, landmarks_px=True, mutate=True, out=d2)
torch.cuda.synchronize()
d3 = {}
for _ in range(N):  # E
    hm.flame.decode(p, verts3d=True, proj=True, to_2d=False, flip_z=True, out=d3)
torch.cuda.synchronize()
print("done", N)
