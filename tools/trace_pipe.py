#!/usr/bin/env python3
"""Phase breakdown of the pipelined decode kernel (flame_decode_pipe.hip) from its in-kernel shader-clock stamps (diagnostics).

    python tools/trace_pipe.py [batch]

Per wave: 0 start | 6, 7 at / past the first barrier (stagers: 7 = first part of A(0) published) | 1 first seven MFMA groups done (mma) / first loads issued (stager) | 4 constants round 0 written | 3 last barrier
passed | 5 end | 12, 13 wall clock (100 MHz); first four half-blocks h: 16+4h before the GEMM / phase start, 17+4h accumulators
parked / A(h+1) written, 18+4h past the barrier / next loads issued, 19+4h (stagers) past the barrier."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dad_3dheads_amd import _lib, landmarks, synthetic  # noqa: E402
from dad_3dheads_amd.head_mesh import HeadMesh  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
st = synthetic.load_static()
hm = HeadMesh(flame_model=synthetic.synthetic_flame_model(0, st), landmarks=landmarks.canonical("445", st), static=st, device=0)
p = torch.from_numpy(synthetic.synthetic_params(batch, seed=0)).cuda()
tiles = 252
trace = torch.zeros((4096, 32), dtype=torch.int64, device="cuda")
lib = _lib.load()
v3 = torch.empty((batch, 5023, 3), device="cuda"); pr = torch.empty((batch, 5023, 2), device="cuda"); lp = torch.empty((batch, 445, 2), dtype=torch.int32, device="cuda")
call = (hm.flame._handle, p.data_ptr(), batch, _lib.TO_2D, v3.data_ptr(), pr.data_ptr(), None, lp.data_ptr(), None)
for _ in range(200):
    lib.dad3d_flame_decode(*call)
hm.flame.select_kernel("pipelined")  # these stamps are the pipelined kernel's layout
_lib.check(lib.dad3d_flame_debug_trace(hm.flame._handle, trace.data_ptr(), trace.numel()))
_lib.check(lib.dad3d_flame_decode(*call))
torch.cuda.synchronize()
_lib.check(lib.dad3d_flame_debug_trace(hm.flame._handle, None, 0))
t = trace.cpu().numpy().astype(np.float64)[: tiles * 8].reshape(tiles, 8, 32)
rel = t - t[:, :, 0:1]
wall = (t[:, :, 13] - t[:, :, 12]) / 100.0
clk = (t[:, 0, 5] - t[:, 0, 0]) / wall[:, 0]
print(f"batch {batch}: wave lifetime {np.median(wall[:, :4]):.2f} us (mma) {np.median(wall[:, 4:]):.2f} us (stagers); shader clock {np.median(clk):.0f} MHz; "
      f"workgroup start spread {(t[:, 0, 12].max() - t[:, 0, 12].min()) / 100.0:.2f} us")
def med(w, slot):
    return float(np.median(rel[:, w, slot]))
print("mma waves (cycles since wave start, median over tiles; wave 0 / wave 3):")
print(f"  at the first barrier {med(0, 6):.0f} / {med(3, 6):.0f}, past it {med(0, 7):.0f} / {med(3, 7):.0f}, basis requests issued {med(0, 16):.0f} / {med(3, 16):.0f}")
print(f"  first MFMA groups done {med(0, 1):.0f} / {med(3, 1):.0f}; constants round 0 written {med(0, 4):.0f} / {med(3, 4):.0f}; end {med(0, 5):.0f} / {med(3, 5):.0f}")
for h in range(min(4, (batch + 31) // 32)):
    a = [med(0, 16 + 4 * h + k) for k in range(4)]
    b = [med(3, 16 + 4 * h + k) for k in range(4)]
    print(f"  half-block {h}: GEMM start {a[0]:.0f}/{b[0]:.0f}  parked {a[1]:.0f}/{b[1]:.0f} (GEMM incl. the previous half-block's epilogue {a[1] - a[0]:.0f})  "
          f"past barrier {a[2]:.0f}/{b[2]:.0f} (wait {a[2] - a[1]:.0f})")
print(f"  last half-block finished by all eight waves: {med(0, 5) - med(0, 3):.0f} cycles")
print("stager waves (wave 4 / wave 7):")
print(f"  first loads issued {med(4, 1):.0f} / {med(7, 1):.0f}; past the first barrier {med(4, 6):.0f}; first part of A(0) published {med(4, 7):.0f}; A(0) published {med(4, 2):.0f}; end {med(4, 5):.0f}")
for h in range(min(4, (batch + 31) // 32)):
    a = [med(4, 16 + 4 * h + k) for k in range(4)]
    print(f"  phase {h}: start {a[0]:.0f}  A({h + 1}) written {a[1]:.0f}  loads issued {a[2]:.0f}  past barrier {a[3]:.0f}")
