#!/usr/bin/env python3
"""Diagnostics: where the split kernel's outputs leave the fp32 kernel's -- by row, by vertex inside its tile, by tile."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dad_3dheads_amd import landmarks, synthetic
from dad_3dheads_amd.head_mesh import HeadMesh
st = synthetic.load_static()
model = synthetic.synthetic_flame_model(0, st)
lm = landmarks.canonical("445", st)
a = HeadMesh(flame_model=model, landmarks=lm, static=st, device=0); a.flame.select_kernel("split_bf16")
b = HeadMesh(flame_model=model, landmarks=lm, static=st, device=0); b.flame.select_kernel("pipelined")
for B in [int(x) for x in sys.argv[1:]] or [16, 48, 100]:
    p = synthetic.synthetic_params(B, seed=5)
    T2 = os.environ.get("TO2D", "1") == "1"
    x = a.decode(torch.from_numpy(p.copy()).cuda(), to_2d=T2, landmarks=T2)
    y = b.decode(torch.from_numpy(p.copy()).cuda(), to_2d=T2, landmarks=T2)
    torch.cuda.synchronize()
    d = (x["verts3d"] - y["verts3d"]).abs().amax(dim=2).cpu().numpy()  # [B, V]
    print(f"B={B}: max {d.max():.3e}; nan {np.isnan(d).sum()}")
    print(" by row     :", " ".join(f"{v:.0e}" for v in np.nanmax(d, axis=1)[:64]))
    dv = np.full((252 * 20,), 0.0); dv[:5023] = np.nanmax(d, axis=0)
    print(" by vertex in tile:", " ".join(f"{v:.0e}" for v in dv.reshape(252, 20).max(axis=0)))
    bt = np.nonzero(dv.reshape(252, 20).max(axis=1) > 1e-5)[0]
    print(" bad tiles  :", len(bt), bt[:60])
    br = np.nonzero(np.nanmax(d, axis=1) > 1e-5)[0]
    print(" bad rows   :", len(br), br[:80])
    if len(bt):
        t0 = bt[0]
        rows_t = np.nonzero(d[:, t0 * 20:(t0 + 1) * 20].max(axis=1) > 1e-5)[0]
        print(f" tile {t0}: bad rows", rows_t[:80])
        X, Y = x["verts3d"].cpu().numpy(), y["verts3d"].cpu().numpy()
        for t in bt[:3]:
            rr = np.nonzero(d[:, t * 20:(t + 1) * 20].max(axis=1) > 1e-5)[0][:2]
            for r in rr:
                for v in range(t * 20, min(t * 20 + 20, 5023)):
                    if d[r, v] > 1e-5:
                        got = X[r, v]
                        lo, hi = max(0, (r // 16 - 3) * 16), min(X.shape[0], (r // 16 + 4) * 16)
                        cand = Y[lo:hi, t * 20:t * 20 + 20]
                        dist = np.abs(cand - got).max(axis=2)
                        j = np.unravel_index(np.argmin(dist), dist.shape)
                        print(f"    bad (row {r}, vtx {v - t * 20}) got {got} want {Y[r, v]} nearest correct: row {lo + j[0]} vtx {j[1]} dist {dist[j]:.1e}")
                        break
        for t in bt[:4]:  # hypothesis: x_got = h . want with ONE h per (tile, row) -> a stale row of the rotation matrix
            for r in np.nonzero(d[:, t * 20:(t + 1) * 20].max(axis=1) > 1e-5)[0][:2]:
                vs = [v for v in range(t * 20, min(t * 20 + 20, 5023)) if d[r, v] > 1e-5]
                if len(vs) >= 4:
                    A = Y[r, vs].astype(np.float64)
                    for comp in range(3):
                        h, res, *_ = np.linalg.lstsq(A, X[r, vs, comp].astype(np.float64), rcond=None)
                        fit = np.abs(A @ h - X[r, vs, comp]).max()
                        if comp == 0 and fit < 1e-6:
                            a1, a2 = p[:, 403:406].astype(np.float64), p[:, 406:409].astype(np.float64)
                            b1 = a1 / np.linalg.norm(a1, axis=1, keepdims=True)
                            b3 = np.cross(b1, a2); b3 /= np.linalg.norm(b3, axis=1, keepdims=True)
                            b2 = -np.cross(b1, b3)
                            R = np.stack((b1, b2, b3), axis=-1)          # [B, 3, 3], R[b][r][c]
                            g = R[r].T @ h                                # the row of G that was used: h = R g'
                            dist = np.abs(R[:, 0, :] - g).max(axis=1)
                            j = int(np.argmin(dist))
                            print(f"    fit (tile {t} row {r} = phase {r // 16} image {r % 16}, {len(vs)} vertices): used G row 0 of row {j} = phase {j // 16} image {j % 16} (dist {dist[j]:.1e})")
        for t in bt[:0]:
            for r in np.nonzero(d[:, t * 20:(t + 1) * 20].max(axis=1) > 1e-5)[0][:4]:
                print(f"   tile {t} row {r} (phase {r // 16}, image {r % 16}):", " ".join(f"{v:.0e}" for v in d[r, t * 20:(t + 1) * 20]))
    print(" proj       :", (x["proj"] - y["proj"]).abs().max().item())
