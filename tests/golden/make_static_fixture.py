#!/usr/bin/env python3
"""Build dad-3dheads_amd/assets/flame_static.npz (package data) from the reference's static assets.

Runs ONLY in the authoring container (needs /root/reference). The GPU box has no
/root/reference, so everything the tests / bench / smoke need from
`model_training/model/static/` is frozen here into one small fixture:

  faces                int32 [9976,3]  <- static/flame_mesh_faces.pt           (flame.py:128-129 `faces_tensor`)
  faces_wo_ears        int32 [6270,3]  <- static/flame_indices/faces_wo_ears_remapped.npy (pncc_estimator.py:71)
  lmk_445              int64 [445]     <- keypoints_445/*.npy, sorted names, cheeks excluded
                                          (model_training/utils.py:62-105 semantics)
  lmk_565              int64 [565]     <- keypoints_445/*.npy, sorted names, cheeks included
                                          (demo_utils.py:37-47 iterates every file)
  lmk_191              int64 [191]     <- keypoints_191/*.npy sorted == static/indices_2d.npy
  indices_2d           int64 [191]     <- static/indices_2d.npy (flame.py:130-131)
  head_indices         int64 [3669]    <- static/head_indices.npy
  template_geo         f32  [5023,3]   <- NOT reference data: a smooth synthetic head-sized embedding of the
                                          topology (graph-Laplacian eigenvectors per connected component), used
                                          as the geometric template of the seeded synthetic FLAME-shaped model
                                          because the real flame.pkl is absent (.MISSING_LARGE_BLOBS:3).

The index / face arrays are (c) the DAD-3DHeads authors, CC BY-NC-SA 4.0 (see tests/golden/NOTICE.md).
"""
import hashlib
import os
import sys

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
import torch

REF = os.environ.get("DAD3D_REFERENCE_ROOT", "/root/reference")
STATIC = os.path.join(REF, "model_training/model/static")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "dad-3dheads_amd", "assets", "flame_static.npz")


def load_indices_from_npy(path):
    # dict values concatenated in insertion order (model_training/utils.py:99-105)
    data = np.load(path, allow_pickle=True)[()]
    out = []
    for v in data.values():
        out += list(v)
    return out


def keypoint_list(subdir, exclude=()):
    d = os.path.join(STATIC, "face_keypoints", subdir)
    out = []
    for fn in sorted(os.listdir(d)):
        if fn.split(".")[0] in exclude:
            continue
        out += load_indices_from_npy(os.path.join(d, fn))
    return np.asarray(out, dtype=np.int64)


def components(nv, faces):
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]], 0)
    a = sp.coo_matrix((np.ones(len(e)), (e[:, 0], e[:, 1])), shape=(nv, nv)).tocsr()
    a = ((a + a.T) > 0).astype(np.float64)
    n, lab = sp.csgraph.connected_components(a, directed=False)
    return a, n, lab


def spectral_template(nv, faces):
    """Smooth 3-D embedding of each connected component (largest = head, the rest = eyeballs)."""
    a, n, lab = components(nv, faces)
    out = np.zeros((nv, 3), np.float64)
    sizes = sorted(((int((lab == c).sum()), c) for c in range(n)), reverse=True)
    small = 0
    for size, c in sizes:
        idx = np.nonzero(lab == c)[0]
        if size < 8:  # isolated vertices (none in FLAME, kept for safety)
            out[idx] = 0.0
            continue
        sub = a[idx][:, idx]
        deg = np.asarray(sub.sum(1)).ravel()
        lap = sp.diags(deg) - sub
        rng = np.random.default_rng(1234 + size)
        vals, vecs = spla.eigsh(lap.tocsc(), k=4, sigma=-1e-3, which="LM", v0=rng.standard_normal(size))
        order = np.argsort(vals)
        emb = vecs[:, order[1:4]]
        for k in range(3):  # fix the sign ambiguity deterministically
            j = np.argmax(np.abs(emb[:, k]))
            if emb[j, k] < 0:
                emb[:, k] = -emb[:, k]
        emb = emb - emb.mean(0)
        # push onto a slightly flattened ellipsoid so the surface is closed and locally smooth
        r = np.linalg.norm(emb, axis=1, keepdims=True)
        emb = emb / np.maximum(r, 1e-12) * (0.75 + 0.25 * r / r.max())
        if size > 2000:  # head
            out[idx] = emb * np.array([0.085, 0.115, 0.095])
        else:  # eyeballs: small spheres in front of the head centre
            cx = 0.032 if small == 0 else -0.032
            out[idx] = emb * 0.012 + np.array([cx, 0.03, 0.07])
            small += 1
    return out.astype(np.float32)


def main():
    faces = torch.load(os.path.join(STATIC, "flame_mesh_faces.pt")).numpy()
    assert faces.shape == (9976, 3) and faces.dtype == np.int64
    faces_wo = np.load(os.path.join(STATIC, "flame_indices/faces_wo_ears_remapped.npy"))
    lmk445 = keypoint_list("keypoints_445", exclude=("cheeks",))
    lmk565 = keypoint_list("keypoints_445")
    lmk191 = keypoint_list("keypoints_191")
    idx2d = np.load(os.path.join(STATIC, "indices_2d.npy"))
    head = np.load(os.path.join(STATIC, "head_indices.npy"))
    # known-answer digests recorded in SURVEY.md §3.2
    assert len(lmk445) == 445 and lmk445.sum() == 1099433
    assert hashlib.sha256(lmk445.tobytes()).hexdigest().startswith("be0bb07f795b2607")
    assert len(lmk565) == 565 and lmk565.sum() == 1398969
    assert hashlib.sha256(lmk565.tobytes()).hexdigest().startswith("08eb437837dd2d40")
    assert len(lmk191) == 191 and np.array_equal(lmk191, idx2d)
    assert hashlib.sha256(lmk191.tobytes()).hexdigest().startswith("05ef3fc1bd31c95b")
    assert np.array_equal(head, np.load(os.path.join(STATIC, "flame_indices/face_w_ears.npy")))
    tmpl = spectral_template(5023, faces)
    np.savez_compressed(
        OUT,
        faces=faces.astype(np.int32),
        faces_wo_ears=faces_wo.astype(np.int32),
        lmk_445=lmk445,
        lmk_565=lmk565,
        lmk_191=lmk191,
        indices_2d=idx2d.astype(np.int64),
        head_indices=head.astype(np.int64),
        template_geo=tmpl,
    )
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
    print("template range", tmpl.min(0), tmpl.max(0))


if __name__ == "__main__":
    sys.exit(main())
