"""Wire format of the DAD-3DHeads accuracy benchmark, downstream of the decode.

A submission is one JSON object `{item_id: {"68_landmarks_2d": [[x, y]] * 68, "N_landmarks_3d": [[x, y, z]] * N,
"7_landmarks_3d": [[x, y, z]] * 7, "rotation_matrix": 3x3}}` (dad_3dheads_benchmark/README.md:78-95). The 68 3-D landmarks
are points ON the mesh: barycentric combinations of the corners of 68 fixed faces -- 17 contour points (row 0 of the
"dynamic" table: the reference always evaluates it at a zero pose) followed by 51 static points
(`get_68_landmarks`, dad_3dheads_benchmark/utils.py:29-117 == model_training/data/utils.py:120-206); the 7 alignment
landmarks are rows 36, 39, 42, 45, 33, 48, 54 of them (utils.py:143-151). Batched and device-resident here; the evaluation
itself (chamfer distance, Procrustes, kaolin) is out of scope.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Mapping, Optional, Sequence

import numpy as np
import torch
from torch import Tensor

from .synthetic import assets_dir

SEVEN_OF_68 = (36, 39, 42, 45, 33, 48, 54)  # get_7_landmarks_from_68's default (utils.py:145)


def embedding_path() -> str:
    """Packaged copy of the 68-landmark barycentric embedding (`face_idx`, `b_coords`); `DAD3D_LMK68_NPZ` overrides."""
    return os.environ.get("DAD3D_LMK68_NPZ") or os.path.join(assets_dir(), "lmk68_embedding.npz")


class Landmarks68:
    """`get_68_landmarks` for a batch of decoded meshes on any device: `[B,5023,3] -> [B,68,3]`."""

    def __init__(self, faces: np.ndarray, path: Optional[str] = None, device: Optional[torch.device] = None):
        with np.load(path or embedding_path()) as z:
            face_idx, b_coords = z["face_idx"].astype(np.int64), z["b_coords"].astype(np.float32)
        corners = np.asarray(faces).astype(np.int64)[face_idx]          # [68,3] vertex ids of the carrying faces
        self.corners = torch.from_numpy(corners).to(device)
        self.weights = torch.from_numpy(b_coords).to(device)            # [68,3]

    def __call__(self, vertices: Tensor) -> Tensor:
        single = vertices.ndim == 2
        v = vertices[None] if single else vertices
        assert v.shape[1:] == (5023, 3)  # utils.py:109-110
        c, w = self.corners.to(v.device), self.weights.to(v.device)
        tri = v[:, c, :]                                                 # [B,68,3 corners,3 xyz]
        # (verts * b_coords).sum(axis=1) of mesh_points_by_barycentric_coordinates, spelled out in its order
        out = tri[:, :, 0, :] * w[None, :, 0, None] + tri[:, :, 1, :] * w[None, :, 1, None] + tri[:, :, 2, :] * w[None, :, 2, None]
        return out[0] if single else out


def seven_landmarks(lmk68: Tensor, indices: Sequence[int] = SEVEN_OF_68) -> Tensor:
    return lmk68[..., list(indices), :]


def submission_entry(points_68_2d, vertices_3d: Tensor, lmk68_3d: Tensor, rotation_matrix) -> Dict[str, list]:
    """One value of the submission dict from one image's predictions (lists of lists of floats)."""
    to_list = lambda x: (x.detach().cpu() if isinstance(x, Tensor) else torch.as_tensor(np.asarray(x))).to(torch.float64).tolist()  # noqa: E731
    return {"68_landmarks_2d": to_list(points_68_2d), "N_landmarks_3d": to_list(vertices_3d),
            "7_landmarks_3d": to_list(seven_landmarks(lmk68_3d)), "rotation_matrix": to_list(rotation_matrix)}


def write_submission(path: str, entries: Mapping[str, Mapping[str, list]]) -> None:
    with open(path, "w") as f:
        json.dump({str(k): dict(v) for k, v in entries.items()}, f)
