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
hm.flame.select_kernel("two_role")  # this script parses the two-role kernel's stamp layout (tools/trace_pipe.py: the pipelined one)
p = torch.from_numpy(synthetic.synthetic_params(batch, seed=0)).cuda()
nbb = (batch + 63) // 64
grid = 240 * nbb  # decode-role workgroups (the pose role is not traced)
n_pose = (batch + 3) // 4
lib = _lib.load()
n_entries = int(lib.dad3d_flame_debug_trace_entries(hm.flame._handle, batch))  # the library's own bound; a launch that would overrun is refused
trace = torch.zeros((n_entries // 32, 32), dtype=torch.int64, device="cuda")
for _ in range(20):
    hm.decode(p, to_2d=True, landmarks_px=True)
_lib.check(lib.dad3d_flame_debug_trace(hm.flame._handle, trace.data_ptr(), trace.numel()))
v3 = torch.empty((batch, 5023, 3), device="cuda"); pr = torch.empty((batch, 5023, 2), device="cuda"); lp = torch.empty((batch, 445, 2), dtype=torch.int32, device="cuda")
drop = os.environ.get("DAD3D_TRACE_DROP", "")  # diagnostics: leave outputs out ("v" = 3d_vertices, "p" = projection, "l" = landmarks)
_lib.check(lib.dad3d_flame_decode(hm.flame._handle, p.data_ptr(), batch, _lib.TO_2D | dbg, None if "v" in drop else v3.data_ptr(),
                                  None if "p" in drop else pr.data_ptr(), None, None if "l" in drop else lp.data_ptr(), None))
torch.cuda.synchronize()
_lib.check(lib.dad3d_flame_debug_trace(hm.flame._handle, None, 0))
allrows = trace.cpu().numpy().astype(np.float64)
pose = allrows[grid * 8 : grid * 8 + n_pose * 4].reshape(n_pose, 4, 32)
full = allrows[: grid * 8].reshape(grid, 8, 32)
t = full[..., :6]
names = ["issue first loads", "first chunk lands", "GEMM (+staging)", "stage acc tile / hand-off", "epilogue"]
t[:, :4, 1] = t[:, :4, 0]  # mma waves have no "loads issued" stamp
d = np.diff(t, axis=-1)
print(f"batch {batch}: {grid} decode workgroups; ticks are s_memtime shader-clock units (one counter per XCD)")
print("phase                        mma waves (0-3)   feeder waves (4-7)")
for i, n in enumerate(names):
    print(f"  {n:26s} {d[:, :4, i].mean():9.1f}         {d[:, 4:, i].mean():9.1f}")
print("per-wave total mean", (t[..., 5] - t[..., 0]).mean())
if full[..., 8].max() > 0:  # fine stamps of a diagnostics build (DAD3D_ABLATE & 64)
    m = full[:, :4]
    print("mma fine stamps (ticks since GEMM start): ", [round(float((m[..., k] - m[..., 2]).mean())) for k in range(8, 15)], " GEMM end", round(float((m[..., 3] - m[..., 2]).mean())))
    if m[..., 16].max() > 0:
        g = np.stack([(m[..., 16 + k] - m[..., 2]).mean() for k in range(13)] + [(m[..., 3] - m[..., 2]).mean()])
        print("mma cycles per MFMA by pair of groups (0-1, 2-3, ...):", [round(float(x) / 32, 1) for x in np.diff(g)])
    f = full[:, 4:]
    print("feeder fine stamps (ticks since wave start): stamp1 %d, stamp2 %d, s8 %d, s9 %d, s10 %d, all parts published %d" % tuple(
        round(float((f[..., k] - f[..., 0]).mean())) for k in (1, 2, 8, 9, 10, 3)))
    print("mma: wave start -> part0 seen %d" % round(float((m[..., 2] - m[..., 0]).mean())))
pl = pose[..., 0] > 0
print("pose role waves: compute %.0f  store+drain %.0f  arrive %.0f ticks (mean over %d waves)" % (
    (pose[..., 1] - pose[..., 0])[pl].mean(), (pose[..., 2] - pose[..., 1])[pl].mean(), (pose[..., 3] - pose[..., 2])[pl].mean(), pl.sum()))
t00 = min(full[..., 12].min(), pose[..., 12][pl].min())  # 100 MHz wall clock, comparable across the chip
us = lambda x: (x - t00) / 100.0
print("wall clock (us after the first wave of the launch): decode waves start %.2f..%.2f, end %.2f..%.2f; pose waves start %.2f..%.2f, end %.2f..%.2f; feeders see the hand-off %.2f..%.2f" % (
    us(full[..., 12].min()), us(full[..., 12].max()), us(full[..., 13].min()), us(full[..., 13].max()),
    us(pose[..., 12][pl].min()), us(pose[..., 12][pl].max()), us(pose[..., 13][pl].min()), us(pose[..., 13][pl].max()),
    us(full[:, 4:, 14].min()), us(full[:, 4:, 14].max())))
if pose[..., 4].max() > 0:
    print("pose fine (ticks from wave start): loads landed + Jdirs in LDS %.0f, joints done %.0f, block computed %.0f" % tuple(
        (pose[..., k] - pose[..., 0])[pl].mean() for k in (4, 5, 1)))
if pose[..., 7].max() > 0:
    print("pose second pass (warm I-cache) of joints+block: %.0f ticks" % (pose[..., 7] - pose[..., 6])[pl].mean())
if pose[..., 9].max() > 0:
    print("  warm pass: dots %.0f, reduce+read %.0f, scalar block %.0f" % tuple((pose[..., k1] - pose[..., k0])[pl].mean() for k0, k1 in ((6, 8), (8, 9), (9, 7))))
if os.environ.get("DAD3D_TRACE_SPREAD"):
    # which workgroups end late? per-XCD and per-phase spread (wall clock, us after the first wave)
    end = us(full[:, :, 13].max(axis=1)); beg = us(full[:, :, 12].min(axis=1))
    gid = np.arange(grid); xcd = gid % 8
    print("end by XCD (min/mean/max):", [(round(float(end[xcd == x].min()), 2), round(float(end[xcd == x].mean()), 2), round(float(end[xcd == x].max()), 2)) for x in range(8)])
    print("start by XCD (mean):", [round(float(beg[xcd == x].mean()), 2) for x in range(8)])
    dm = d[:, :4].mean(axis=1)  # [grid, phase]
    for i, n in enumerate(names):
        q = np.percentile(dm[:, i], [0, 10, 50, 90, 100])
        print(f"  {n:26s} p0/10/50/90/100 = " + " ".join(f"{x:8.0f}" for x in q))
    order = np.argsort(end)
    print("earliest 8 workgroups:", [(int(g_), round(float(end[g_]), 2)) for g_ in order[:8]])
    print("latest 8 workgroups:  ", [(int(g_), round(float(end[g_]), 2)) for g_ in order[-8:]])
    print("corr(end, start) = %.2f; corr(end, GEMM) = %.2f; corr(end, first chunk) = %.2f; corr(end, epilogue) = %.2f" % (
        np.corrcoef(end, beg)[0, 1], np.corrcoef(end, dm[:, 2])[0, 1], np.corrcoef(end, dm[:, 1])[0, 1], np.corrcoef(end, dm[:, 4])[0, 1]))
