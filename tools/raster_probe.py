#!/usr/bin/env python3
"""Diagnostics: per-phase wall-clock profile of the raster kernel (dad3d_mesh_debug_trace) on the config-5 workload
(B=64 decoded heads, 9976 triangles, 256x256x3). Prints, per phase, the mean / max over workgroups of the slowest
wave, in microseconds."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dad_3dheads_amd import _lib, landmarks, synthetic  # noqa: E402
from dad_3dheads_amd.head_mesh import HeadMesh  # noqa: E402
from dad_3dheads_amd.Sim3DR import Mesh  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    st = synthetic.load_static()
    model = synthetic.synthetic_flame_model(0, st)
    hm = HeadMesh(flame_model=model, landmarks=landmarks.canonical("445", st), static=st, device=0)
    p = torch.from_numpy(synthetic.synthetic_params(B, seed=2)).cuda()
    mesh = Mesh(st["faces"], 5023, device=0)
    dec = {}
    hm.flame.decode(p, proj=True, to_2d=False, flip_z=True, out=dec)
    verts = dec["proj"]
    img = torch.zeros((B, 256, 256, 3), dtype=torch.uint8, device="cuda")
    light = mesh.phong_light(verts, mesh.get_normal(verts))
    for _ in range(5):
        mesh.rasterize(verts, light, img)
    tiles, waves = 16, 8
    trace = torch.zeros((B * tiles * 16, waves, 16), dtype=torch.int64, device="cuda")
    lib = _lib.load()
    _lib.check(lib.dad3d_mesh_debug_trace(mesh._handle, trace.data_ptr()))
    mesh.rasterize(verts, light, img)
    torch.cuda.synchronize()
    _lib.check(lib.dad3d_mesh_debug_trace(mesh._handle, None))
    t = trace.cpu().numpy().astype(np.float64)
    n_items = int(t[:, 0, 6].max())
    t = t[:n_items]
    counts = t[:, 0, 7]
    levels = (t[:, 0, 5].astype(np.int64) >> 24) & 3
    print("work items:", n_items, "by split level:", [int((levels == k).sum()) for k in range(3)])
    walkstats = t[:, :, 8:16].copy()
    live = counts > 0
    t, counts, walkstats = t[live], counts[live], walkstats[live]
    traw = t.copy()
    t = t[:, :, :5] / 100.0  # 100 MHz -> us
    t0 = t[:, :, 0].min()
    names = ["start", "list sorted", "fragments done", "barrier", "resolved"]
    print(f"B={B}: {live.sum()} items; listed triangles per tile: mean {counts.mean():.0f} max {counts.max():.0f}")
    print("workgroup start spread: %.2f us; kernel span %.2f us" % (t[:, :, 0].max() - t0, t[:, :, 4].max() - t0))
    for s in range(1, 5):
        d = t[:, :, s] - t[:, :, s - 1]
        print(f"  {names[s - 1]:>15} -> {names[s]:<15} per-wave mean {d.mean():7.2f}  slowest wave/WG mean {d.max(1).mean():7.2f}"
              f"  max {d.max():7.2f} us")
    sub = traw[:, :, [0, 12, 13, 14, 1]] / 100.0
    print("  sort phase split (per-wave mean us): key init %.2f, list load + count %.2f, prefix %.2f, scatter %.2f"
          % tuple(float((sub[:, :, k + 1] - sub[:, :, k]).mean()) for k in range(4)))
    for name, o in (("fragment walk", 0),):
        ws = walkstats[:, :, o:o + 4]
        steps = ws[:, :, 0].sum()
        print(f"  {name}: wave steps/wave mean {ws[:, :, 0].mean():.2f} max {ws[:, :, 0].max():.0f}; per step: wait "
              f"{ws[:, :, 1].sum() / steps / 100:.2f} us, work {ws[:, :, 2].sum() / steps / 100:.2f} us; busiest-lane pixel tests "
              f"per step {ws[:, :, 3].sum() / steps:.1f}")
        heavy = counts.argmax()
        hs = walkstats[heavy, :, o:o + 4]
        print(f"    heaviest tile ({counts[heavy]:.0f} tris): steps/wave {hs[:, 0].mean():.1f}, wait {hs[:, 1].mean() / 100:.2f} us, "
              f"work {hs[:, 2].mean() / 100:.2f} us per wave, busiest-lane tests {hs[:, 3].mean():.0f}")
    st, en = t[:, :, 0].min(1) - t0, t[:, :, 4].max(1) - t0
    print("  item start percentiles (us):", [round(float(np.percentile(st, q)), 1) for q in (0, 10, 25, 50, 75, 90, 100)])
    print("  items running at t =", {tt: int(((st <= tt) & (en > tt)).sum()) for tt in (1, 5, 10, 20, 30, 40, 50, 60, 70, 80)})
    print("  queue position vs start (us), every 100th:", [round(float(st[i]), 1) for i in range(0, len(st), 100)])
    d = t[:, :, 4].max(1) - t[:, :, 0].min(1)
    order = np.argsort(-d)[:5]
    print("  slowest items (queue position, level, list length, us):", [(int(i), int(levels[i]), int(counts[i]), round(float(d[i]), 1)) for i in order])
    for i in order:
        ph = [round(float((t[i, :, k] - t[i, :, k - 1]).max()), 1) for k in range(1, 5)]
        ws = walkstats[i, :, 0:4]
        print(f"    item {int(i)}: phases (slowest wave) {ph}; steps/wave {ws[:, 0].mean():.1f}, work {ws[:, 2].mean() / 100:.1f} us/wave, "
              f"busiest-lane tests {ws[:, 3].mean():.0f}, start {float(st[i]):.1f} us")
    print("  workgroup duration mean %.2f max %.2f us" % (d.mean(), d.max()))


if __name__ == "__main__":
    main()
