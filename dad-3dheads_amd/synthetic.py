"""Seeded synthetic FLAME-shaped model and 3DMM parameter batches.

The real FLAME basis (`model_training/model/static/flame.pkl`) and the trained checkpoint are not
redistributed with the reference (`.MISSING_LARGE_BLOBS:3`), so every test / bench here runs on a
deterministic model with exactly the shapes `FLAMELayer.__init__` consumes
(`model_training/model/flame.py:124-180`):

    f [9976,3] int, v_template [5023,3], shapedirs [5023,3,400], posedirs [5023,3,36],
    J_regressor [5,5023], kintree_table [2,5] (row 0 = parents, root = uint32(-1)), weights [5023,5]

A user with a licensed `flame.pkl` passes its path to `FlameModel.from_pickle` instead.
"""
from __future__ import annotations

import hashlib
import os
from types import SimpleNamespace
from typing import Optional

import numpy as np

N_VERTS = 5023
N_BETAS = 400
N_POSE_FEATS = 36
N_JOINTS = 5
N_PARAMS = 413  # shape300 | expr100 | jaw3 | rot6d 6 | trans3 | scale1  (flame.py:48-73)

_HERE = os.path.dirname(os.path.abspath(__file__))


def assets_dir() -> str:
    """Package data: the static index assets of the FLAME topology (dad-3dheads_amd/assets/, see its NOTICE.md)."""
    return os.path.join(_HERE, "assets")


def static_fixture_path() -> str:
    """The frozen static assets (face list, face subset, landmark index lists; built by
    tests/golden/make_static_fixture.py from a reference checkout). `DAD3D_STATIC_NPZ` overrides the packaged copy."""
    return os.environ.get("DAD3D_STATIC_NPZ") or os.path.join(assets_dir(), "flame_static.npz")


def load_static(path: Optional[str] = None) -> dict:
    with np.load(path or static_fixture_path()) as z:
        return {k: z[k] for k in z.files}


def _smooth_features(v: np.ndarray) -> np.ndarray:
    """Low-order polynomial features of the template positions, unit-ish scale. [V,10]"""
    u = v / np.abs(v).max()
    x, y, z = u[:, 0], u[:, 1], u[:, 2]
    return np.stack([np.ones_like(x), x, y, z, x * y, y * z, z * x, x * x, y * y, z * z], 1)


def synthetic_flame_model(seed: int = 0, static: Optional[dict] = None) -> SimpleNamespace:
    """FLAME-shaped constants, float64 like the unpickled original (cast to f32 by the consumer)."""
    st = static if static is not None else load_static()
    rng = np.random.default_rng(seed)
    faces = st["faces"].astype(np.int64)
    v = st["template_geo"].astype(np.float64)
    assert v.shape == (N_VERTS, 3)
    phi = _smooth_features(v)  # [V,10]

    # shape/expression directions: smooth fields with decaying amplitude + a little white noise so that
    # every basis vector is distinct per vertex (a pure low-rank basis would hide column mix-ups).
    amp = 2.5e-3 / (1.0 + np.arange(N_BETAS) / 25.0)
    coef = rng.standard_normal((N_BETAS, 3, phi.shape[1]))
    shapedirs = np.einsum("vf,lkf->vkl", phi, coef) * amp[None, None, :]
    shapedirs += rng.standard_normal(shapedirs.shape) * 2e-5
    coef_p = rng.standard_normal((N_POSE_FEATS, 3, phi.shape[1]))
    posedirs = np.einsum("vf,lkf->vkl", phi, coef_p) * 1.5e-3
    posedirs += rng.standard_normal(posedirs.shape) * 2e-5

    # joints: root, neck, jaw, two eyes -- each regressed from a soft neighbourhood of a seed point
    seeds = np.array(
        [[0.0, -0.02, 0.0], [0.0, -0.08, -0.01], [0.0, -0.03, 0.03], [0.032, 0.03, 0.07], [-0.032, 0.03, 0.07]]
    )
    d2 = ((v[None, :, :] - seeds[:, None, :]) ** 2).sum(-1)  # [5,V]
    jr = np.exp(-d2 / (2 * 0.02**2))
    jr[jr < 1e-4 * jr.max(1, keepdims=True)] = 0.0  # sparse-ish rows, like the real regressor
    jr /= jr.sum(1, keepdims=True)

    # skinning weights: smooth partition of unity, jaw dominant on the lower front of the face
    w = np.exp(-d2.T / (2 * np.array([0.08, 0.05, 0.04, 0.012, 0.012]) ** 2)[None, :])
    w[:, 0] += 0.05
    w /= w.sum(1, keepdims=True)

    kintree = np.array([[np.iinfo(np.uint32).max, 0, 1, 1, 1], [0, 1, 2, 3, 4]], dtype=np.uint32)
    return SimpleNamespace(
        f=faces, v_template=v, shapedirs=shapedirs, posedirs=posedirs, J_regressor=jr, kintree_table=kintree, weights=w
    )


def model_digest(model) -> str:
    """sha256 over the f32 image of the constants: guards golden vectors against RNG drift."""
    h = hashlib.sha256()
    for name in ("v_template", "shapedirs", "posedirs", "J_regressor", "weights"):
        h.update(np.ascontiguousarray(np.asarray(getattr(model, name)), dtype=np.float32).tobytes())
    return h.hexdigest()


def synthetic_params(batch: int, seed: int = 0, profile: str = "crop") -> np.ndarray:
    """Seeded `[B,413]` f32 params shaped like the CNN head's output (flame_regression.py:96-104).

    shape/expr = 3*tanh(N(0,1)); jaw = 0.1 * 3*tanh(N(0,1)) rad; rot6d = N(0,1); tz arbitrary.
    profile "crop":   scale ~ U(5,7), txy ~ U(-0.15,0.15): the head fills the 256x256 crop like a
                      real DAD-3DNet prediction does (head ~0.19 m tall -> ~150-190 px).
    profile "survey": scale ~ U(-0.3,0.3), txy ~ U(-0.2,0.2) (SURVEY.md section 8d wording).
    """
    rng = np.random.default_rng(seed)
    p = np.empty((batch, N_PARAMS), np.float64)
    p[:, :400] = 3.0 * np.tanh(rng.standard_normal((batch, 400)))
    p[:, 400:403] = 0.1 * 3.0 * np.tanh(rng.standard_normal((batch, 3)))
    p[:, 403:409] = rng.standard_normal((batch, 6))
    if profile == "crop":
        p[:, 409:411] = rng.uniform(-0.15, 0.15, (batch, 2))
        p[:, 412] = rng.uniform(5.0, 7.0, batch)
    elif profile == "survey":
        p[:, 409:411] = rng.uniform(-0.2, 0.2, (batch, 2))
        p[:, 412] = rng.uniform(-0.3, 0.3, batch)
    else:
        raise ValueError(f"unknown profile {profile!r}")
    p[:, 411] = rng.standard_normal(batch)  # tz: zeroed by the path (head_mesh.py:41)
    return p.astype(np.float32)
