"""Mirrors of the reference's mesh losses (`model_training/losses/vertices_3d_loss.py`, `reprojection_loss.py`) on top
of the differentiable HIP decode. Same constructor arguments and the same forward arithmetic; `weights_and_indices`
is the yaml block `{"weights": {...}, "flame_indices": {"folder": ..., "files": {...}}}` the reference passes, or --
extra to the reference -- ready-made `(weights, indices)` lists (the region `.npy` assets travel with the licensed
FLAME material, so the tests build their own regions).
"""
from __future__ import annotations

import os
from typing import Any, Dict, List, Sequence, Tuple, Union

import numpy as np
import torch
from torch import Tensor, nn

from .head_mesh import HeadMesh

__all__ = ["Vertices3DLoss", "ReprojectionLoss", "indices_reweighing", "normalize_to_cube"]
losses = {"l1": nn.L1Loss, "l2": nn.MSELoss, "smooth_l1": nn.SmoothL1Loss}


def indices_reweighing(weights_and_indices: Union[Dict[str, Any], Tuple[Sequence, Sequence]]) -> Tuple[List, List]:
    """model_training/utils.py:108-117: the regions named in `weights`, in the order of the `files` mapping."""
    if not isinstance(weights_and_indices, dict):
        w, idx = weights_and_indices
        return list(w), [np.asarray(i) for i in idx]
    named, files = weights_and_indices["weights"], weights_and_indices["flame_indices"]
    weights, indices = [], []
    for key, value in files["files"].items():
        if key in named:
            indices.append(np.load(os.path.join(files["folder"], value)))
            weights.append(named[key])
    return weights, indices


def normalize_to_cube(v: Tensor) -> Tensor:
    """model/utils.py:55-68: shift to the positive octant, centre, divide by the largest extent."""
    if v.ndim == 2:
        v = v[None]
    v = v - v.min(1, True)[0]
    v = v - 0.5 * v.max(1, True)[0]
    return v / v.max(-1, True)[0].max(-2, True)[0]


class _MeshLoss(nn.Module):
    def __init__(self, criterion: str, weights_and_indices, head_mesh: HeadMesh) -> None:
        super().__init__()
        if criterion not in losses:
            raise ValueError(f"Unsupported discrepancy loss type {criterion}")
        self.criterion = losses[criterion]()
        self.weights, self.indices = indices_reweighing(weights_and_indices)
        self.head_mesh = head_mesh


class Vertices3DLoss(_MeshLoss):
    """vertices_3d_loss.py:14-49: region-weighted criterion between cube-normalised, un-rotated predicted vertices and
    the target vertices."""

    def __init__(self, criterion, batch_size, consts, weights_and_indices, **head_mesh_kwargs) -> None:
        super().__init__(criterion, weights_and_indices,
                         HeadMesh(flame_config=consts, batch_size=batch_size, **head_mesh_kwargs))

    @torch.autocast("cuda", enabled=False)
    def forward(self, predicted: Tensor, target: Tensor) -> Tensor:
        pred_vertices = self.head_mesh.vertices_3d(params_3dmm=predicted, zero_rotation=True)
        terms = [self.criterion(normalize_to_cube(pred_vertices[:, i]), normalize_to_cube(target[:, i])) * w
                 for w, i in zip(self.weights, self.indices)]
        return torch.stack(terms).sum()


class ReprojectionLoss(_MeshLoss):
    """reprojection_loss.py:13-46: region-weighted criterion between the projected vertices and 2-D targets."""

    def __init__(self, criterion, batch_size, consts, img_size, weights_and_indices, **head_mesh_kwargs) -> None:
        super().__init__(criterion, weights_and_indices,
                         HeadMesh(flame_config=consts, batch_size=batch_size, image_size=img_size, **head_mesh_kwargs))

    @torch.autocast("cuda", enabled=False)
    def forward(self, predicted: Tensor, target: Union[Tensor, List[Tensor]]) -> Tensor:
        projected_vertices = self.head_mesh.reprojected_vertices(params_3dmm=predicted, to_2d=True)
        full_target = target[0] if isinstance(target, list) else target
        terms = [self.criterion(projected_vertices[:, i], full_target[:, i]) * w for w, i in zip(self.weights, self.indices)]
        return torch.stack(terms).sum()
