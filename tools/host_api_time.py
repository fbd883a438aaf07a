#!/usr/bin/env python3
"""Diagnostics: per-call time of the reference-shaped single-image HOST entry points (numpy in, numpy out, staged through the
GPU): Sim3DR.rasterize and Sim3DR.get_normal on one decoded head, 9976 triangles, 256x256x3."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dad_3dheads_amd import Sim3DR, synthetic
st = synthetic.load_static()
g = np.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests/golden/decode_golden.npz"))
verts = np.ascontiguousarray(g["b2_proj3"][0]).copy(); verts[:, 2] *= -1
faces = st["faces"]
col = np.random.default_rng(0).uniform(0, 1, (5023, 3)).astype(np.float32)
for _ in range(5): Sim3DR.rasterize(verts, faces, col, height=256, width=256, channel=3); Sim3DR.get_normal(verts, faces)
t0 = time.perf_counter()
for _ in range(200): Sim3DR.rasterize(verts, faces, col, height=256, width=256, channel=3)
t1 = time.perf_counter()
for _ in range(200): Sim3DR.get_normal(verts, faces)
t2 = time.perf_counter()
print("host API per call: rasterize %.0f us, get_normal %.0f us" % ((t1 - t0) / 200 * 1e6, (t2 - t1) / 200 * 1e6))
