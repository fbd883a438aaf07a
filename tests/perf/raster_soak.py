#!/usr/bin/env python3
"""One-off soak of the raster / normals kernels against the compiled reference with FRESH seeds (the hypothesis test in
tests/test_gpu_raster_fuzz.py is derandomised: the same 56 cases every run). Lives under tests/perf/ because it imports the oracle.

    python tests/perf/raster_soak.py [seconds] [first_seed]     -> prints the number of cases and the first mismatch, if any"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_raster_fuzz import PROFILES, build_case  # noqa: E402

from dad_3dheads_amd.Sim3DR import Mesh  # noqa: E402
from oracle.sim3dr_ref import Sim3DROracle  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
orc = Sim3DROracle("best")
n, t0, bad = 0, time.time(), []
while time.time() - t0 < budget and not bad:
    for profile in PROFILES:
        v, t, col, bg, h, w, rev = build_case(seed, profile)
        mesh = Mesh(t, v.shape[0], device=0)
        dv = torch.from_numpy(v).cuda()[None]
        img = torch.from_numpy(bg.copy()).cuda()[None].contiguous()
        depth = torch.full((1, h, w), -1e8, device="cuda")
        mesh.rasterize(dv, torch.from_numpy(col).cuda()[None], img, depth=depth, reverse=rev)
        d, tb, bw = mesh.rasterize_triangles(dv, h, w)
        nrm = mesh.get_normal(dv)
        for alpha in (0.5,) if n % 4 == 0 and col.shape[1] <= 4 else ():
            img_a = torch.from_numpy(bg.copy()).cuda()[None].contiguous()
            mesh.rasterize(dv, torch.from_numpy(col).cuda()[None], img_a, reverse=rev, alpha=alpha)
            with np.errstate(all="ignore"):
                ref_a = orc.rasterize(v, t, col, bg=bg.copy(), reverse=rev, alpha=alpha)
            if not np.array_equal(img_a[0].cpu().numpy(), ref_a):
                bad.append((profile, seed, "alpha"))
        with np.errstate(all="ignore"):
            ref_img, ref_depth = orc.rasterize(v, t, col, bg=bg.copy(), reverse=rev, return_depth=True)
            rd, rtb, rbw = orc.rasterize_triangles(v, t, h, w)
            ref_n = orc.get_normal(v, t)
        ok = (np.array_equal(img[0].cpu().numpy(), ref_img) and np.array_equal(depth[0].cpu().numpy(), ref_depth, equal_nan=True)
              and np.array_equal(d[0].cpu().numpy(), rd, equal_nan=True) and np.array_equal(tb[0].cpu().numpy(), rtb)
              and np.array_equal(bw[0].cpu().numpy(), rbw, equal_nan=True) and np.array_equal(nrm[0].cpu().numpy(), ref_n, equal_nan=True))
        if not ok:
            bad.append((profile, seed, "raster/normals"))
        n += 1
    seed += 1
print(f"SOAK {n} cases in {time.time() - t0:.0f} s over profiles {PROFILES}, seeds from {sys.argv[2] if len(sys.argv) > 2 else 1000000}: "
      + ("all bit-exact" if not bad else f"MISMATCH {bad[:3]}"))
