#!/usr/bin/env python3
"""Diagnostics for the decode-kernel leads (profiles/r01_kernel_log.md): three batch-64 launches through an ALTERNATE
library (tools/attic/build_alt.sh -> tools/attic/alt.so, or $DAD3D_LIB_PATH), timed, and compared with the batch-16
path of the same library (which takes the unchanged small-batch instantiation). Run under `timeout 25`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from dad_3dheads_amd import _lib
_lib.LIB_PATH = os.environ.get("DAD3D_LIB_PATH") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "alt.so")
from dad_3dheads_amd import landmarks, synthetic
from dad_3dheads_amd.head_mesh import HeadMesh
import ctypes as C
st = synthetic.load_static()
hm = HeadMesh(flame_model=synthetic.synthetic_flame_model(0, st), landmarks=landmarks.canonical("445", st), static=st, device=0)
p = torch.from_numpy(synthetic.synthetic_params(64, seed=0)).cuda()
ref = hm.decode(p[:16].contiguous(), to_2d=True, landmarks_px=True)
torch.cuda.synchronize(); print("batch 16 (MW=4) ok", flush=True)
for k in range(3):
    t0 = time.perf_counter()
    out = hm.decode(p.clone(), to_2d=True, landmarks_px=True)
    torch.cuda.synchronize()
    print("batch 64 launch %d: %.3f ms" % (k, (time.perf_counter() - t0) * 1e3), flush=True)
n = C.c_uint(0); _lib.load().dad3d_flame_handoff_timeouts(hm.flame._handle, C.byref(n))
print("hand-off time-outs:", n.value, " max |diff| rows 0..15:", float((out["verts3d"][:16] - ref["verts3d"]).abs().max()),
      float((out["proj"][:16] - ref["proj"]).abs().max()), "rows 48..63 finite:", bool(torch.isfinite(out["verts3d"][48:]).all()), flush=True)
