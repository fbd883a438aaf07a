#!/usr/bin/env python3
"""Phase breakdown of the fused decode kernel from its in-kernel shader-clock stamps (diagnostics only).

    python tools/trace_decode.py [batch]

Stamps per wave: 0 start | 1 loads+DMAs issued | 2 operands landed (after barrier) | 3 GEMM done |
4 accumulator tile staged | 5 end. s_memtime counts at 100 MHz on gfx950 (10 ns ticks)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dad_3dheads_amd import _lib, landmarks, synthetic  # noqa: E402
from dad_3dheads_amd.head_mesh import HeadMesh  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dbg = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0
st = synthetic.load_static()
hm = HeadMesh(flame_model=synthetic.synthetic_flame_model(0, st), landmarks=landmarks.canonical("445", st), static=st, device=0)
p = torch.from_numpy(synthetic.synthetic_params(batch, seed=0)).cuda()
nbb = (batch + 63) // 64
grid = 240 * nbb  # decode-role workgroups (the pose role is not traced)
trace = torch.zeros((grid, 4, 32), dtype=torch.int64, device="cuda")
lib = _lib.load()
for _ in range(20):
    hm.decode(p, to_2d=True, landmarks_px=True)
_lib.check(lib.dad3d_flame_debug_trace(hm.flame._handle, trace.data_ptr()))
v3 = torch.empty((batch, 5023, 3), device="cuda"); pr = torch.empty((batch, 5023, 2), device="cuda"); lp = torch.empty((batch, 445, 2), dtype=torch.int32, device="cuda")
_lib.check(lib.dad3d_flame_decode(hm.flame._handle, p.data_ptr(), batch, _lib.TO_2D | dbg, v3.data_ptr(), pr.data_ptr(), None, lp.data_ptr(), None))
torch.cuda.synchronize()
_lib.check(lib.dad3d_flame_debug_trace(hm.flame._handle, None))
full = trace.cpu().numpy().astype(np.float64)
t = full[..., :6]
t0 = t[..., 0].min()
names = ["issue loads", "wait operands", "GEMM", "stage tile", "epilogue"]
d = np.diff(t, axis=-1)
print(f"batch {batch}: grid {grid} blocks; ticks are s_memtime units")
print("kernel span (first start -> last end):", t[..., 5].max() - t0)
print("block start spread:", t[..., 0].max() - t0)
for i, n in enumerate(names):
    print(f"  {n:14s} mean {d[..., i].mean():9.1f}  min {d[..., i].min():9.1f}  max {d[..., i].max():9.1f}")
print("per-wave total mean", (t[..., 5] - t[..., 0]).mean())
