"""Mirrors of the reference's mesh losses (`model_training/losses/vertices_3d_loss.py`, `reprojection_loss.py`) on top
of the differentiable HIP decode. Same constructor arguments and the same forward arithmetic, evaluated -- value and
gradient -- by the fused kernels of csrc/mesh_losses.hip (three launches per step for both losses instead of ~60 through
torch's gather / reduce / scatter graph; `normalize_to_cube` below stays as the torch statement the tests compare with); `weights_and_indices`
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
from torch.autograd.function import once_differentiable

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


_CRITERION_ID = {"l1": 0, "l2": 1, "smooth_l1": 2}  # DAD3D_LOSS_* of include/dad3d.h


class RegionTables:
    """The region lists of `indices_reweighing` as device tables for the fused loss kernels (csrc/mesh_losses.hip):
    the lists back to back, the same incidence transposed (vertex -> (region, position)) for the gradient gather, and the
    per-vertex weight `sum_r w_r * multiplicity / N_r` the un-normalised point loss collapses into."""

    def __init__(self, weights: Sequence[float], indices: Sequence[np.ndarray], n_verts: int, device: torch.device) -> None:
        idx = [np.asarray(i, dtype=np.int64).reshape(-1) for i in indices]
        for i in idx:
            if i.size and (i.min() < -n_verts or i.max() >= n_verts):
                raise IndexError(f"region index out of range for {n_verts} vertices")
        for k, i in enumerate(idx):
            # the reference's criterion is a mean over the region's vertices: NaN for an empty one (and NaN gradients for the
            # whole step). The kernels would add 0 instead -- refused rather than silently different.
            if i.size == 0:
                raise ValueError(f"region {k} has no vertices (the reference's mean over it is NaN)")
        idx = [np.where(i < 0, i + n_verts, i) for i in idx]  # python-style negative indices, like tensor[:, i]
        ptr = np.zeros(len(idx) + 1, dtype=np.int32)
        ptr[1:] = np.cumsum([i.size for i in idx])
        flat = np.concatenate(idx) if idx else np.zeros(0, dtype=np.int64)
        region_of = np.repeat(np.arange(len(idx)), [i.size for i in idx])
        pos_of = np.concatenate([np.arange(i.size) for i in idx]) if idx else np.zeros(0, dtype=np.int64)
        order = np.lexsort((pos_of, region_of, flat))  # by vertex, then region, then position: a fixed summation order
        vert_ptr = np.zeros(n_verts + 1, dtype=np.int32)
        np.add.at(vert_ptr, flat + 1, 1)
        vert_ptr = np.cumsum(vert_ptr).astype(np.int32)
        point_weight = np.zeros(n_verts, dtype=np.float64)
        for w, i in zip(weights, idx):
            if i.size:
                np.add.at(point_weight, i, float(w) / i.size)
        dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(device)  # noqa: E731
        self.n_regions, self.n_verts, self.device = len(idx), n_verts, device
        self.region_ptr, self.region_idx = dev(ptr, np.int32), dev(flat, np.int32)
        self.region_weight = dev(np.asarray(weights, dtype=np.float32), np.float32)
        self.vert_ptr, self.vert_region, self.vert_pos = dev(vert_ptr, np.int32), dev(region_of[order], np.int32), dev(pos_of[order], np.int32)
        self.point_weight = dev(point_weight, np.float32)


def _stage(t: Tensor, device: torch.device) -> Tensor:
    return t.detach().to(device, torch.float32).contiguous()


def _first_order_pred_only(ctx) -> None:
    """The fused kernels return dL/d(pred) as a constant: no gradient for the target, none of second order (the backward
    passes below are @once_differentiable, so a double backward raises instead of yielding zeros). The reference's torch graph
    gives both; a caller that needs them evaluates `normalize_to_cube` + the torch criterion instead."""
    if ctx.needs_input_grad[1]:
        raise RuntimeError("the HIP mesh losses differentiate with respect to the prediction only: detach the target "
                           "(the reference's targets are dataset tensors)")


class _CubeRegionLoss(torch.autograd.Function):
    """vertices_3d_loss.py:43-49 on decoded vertices: one HIP launch for the value, one more for dL/d(pred)."""

    @staticmethod
    def forward(ctx, pred: Tensor, target: Tensor, tables: RegionTables, criterion: int):
        from . import _lib

        dev = tables.device
        p, t = _stage(pred, dev), _stage(target, dev)
        if p.shape != t.shape or p.ndim != 3 or p.shape[1:] != (tables.n_verts, 3):
            raise ValueError(f"expected two [B,{tables.n_verts},3] tensors, got {tuple(pred.shape)} and {tuple(target.shape)}")
        b = p.shape[0]
        _first_order_pred_only(ctx)
        need_grad = ctx.needs_input_grad[0]
        grad = torch.empty_like(p) if need_grad else None
        stats = torch.empty((tables.n_regions, b, 28), dtype=torch.float32, device=dev)
        terms = torch.zeros((tables.n_regions, b), dtype=torch.float32, device=dev)
        _lib.check(_lib.load().dad3d_cube_region_loss(
            p.data_ptr(), t.data_ptr(), b, tables.n_verts, tables.region_ptr.data_ptr(), tables.region_idx.data_ptr(),
            tables.region_weight.data_ptr(), tables.n_regions, tables.vert_ptr.data_ptr(), tables.vert_region.data_ptr(),
            tables.vert_pos.data_ptr(), criterion, stats.data_ptr(), terms.data_ptr(), grad.data_ptr() if need_grad else None,
            dev.index or 0, torch.cuda.current_stream(dev).cuda_stream))
        ctx.src_device = pred.device
        if need_grad:
            ctx.save_for_backward(grad)
        return terms.sum().to(pred.device)

    @staticmethod
    @once_differentiable
    def backward(ctx, g: Tensor):
        (grad,) = ctx.saved_tensors
        return (grad * g.to(grad.device)).to(ctx.src_device), None, None, None


class _WeightedPointLoss(torch.autograd.Function):
    """reprojection_loss.py:42-46 on projected vertices: one HIP launch, value and dL/d(pred)."""

    @staticmethod
    def forward(ctx, pred: Tensor, target: Tensor, tables: RegionTables, criterion: int):
        from . import _lib

        dev = tables.device
        p, t = _stage(pred, dev), _stage(target, dev)
        if p.shape != t.shape or p.ndim != 3 or p.shape[1] != tables.n_verts:
            raise ValueError(f"expected two [B,{tables.n_verts},C] tensors, got {tuple(pred.shape)} and {tuple(target.shape)}")
        b, comps = p.shape[0], p.shape[2]
        _first_order_pred_only(ctx)
        need_grad = ctx.needs_input_grad[0]
        grad = torch.empty_like(p) if need_grad else None
        lib = _lib.load()
        terms = torch.zeros((b, lib.dad3d_point_loss_terms(tables.n_verts)), dtype=torch.float32, device=dev)
        _lib.check(lib.dad3d_weighted_point_loss(
            p.data_ptr(), t.data_ptr(), b, tables.n_verts, comps, tables.point_weight.data_ptr(), 1.0 / (b * comps), criterion,
            terms.data_ptr(), grad.data_ptr() if need_grad else None, dev.index or 0, torch.cuda.current_stream(dev).cuda_stream))
        ctx.src_device = pred.device
        if need_grad:
            ctx.save_for_backward(grad)
        return terms.sum().to(pred.device)

    @staticmethod
    @once_differentiable
    def backward(ctx, g: Tensor):
        (grad,) = ctx.saved_tensors
        return (grad * g.to(grad.device)).to(ctx.src_device), None, None, None


class _MeshLoss(nn.Module):
    def __init__(self, criterion: str, weights_and_indices, head_mesh: HeadMesh) -> None:
        super().__init__()
        if criterion not in losses:
            raise ValueError(f"Unsupported discrepancy loss type {criterion}")
        self.criterion = losses[criterion]()
        self.criterion_id = _CRITERION_ID[criterion]
        self.weights, self.indices = indices_reweighing(weights_and_indices)
        self.head_mesh = head_mesh
        self.tables = RegionTables(self.weights, self.indices, head_mesh.flame.n_verts, head_mesh.flame.torch_device)


class Vertices3DLoss(_MeshLoss):
    """vertices_3d_loss.py:14-49: region-weighted criterion between cube-normalised, un-rotated predicted vertices and
    the target vertices."""

    def __init__(self, criterion, batch_size, consts, weights_and_indices, **head_mesh_kwargs) -> None:
        super().__init__(criterion, weights_and_indices,
                         HeadMesh(flame_config=consts, batch_size=batch_size, **head_mesh_kwargs))

    @torch.autocast("cuda", enabled=False)
    def forward(self, predicted: Tensor, target: Tensor) -> Tensor:
        pred_vertices = self.head_mesh.vertices_3d(params_3dmm=predicted, zero_rotation=True)
        return _CubeRegionLoss.apply(pred_vertices, target, self.tables, self.criterion_id)


class ReprojectionLoss(_MeshLoss):
    """reprojection_loss.py:13-46: region-weighted criterion between the projected vertices and 2-D targets."""

    def __init__(self, criterion, batch_size, consts, img_size, weights_and_indices, **head_mesh_kwargs) -> None:
        super().__init__(criterion, weights_and_indices,
                         HeadMesh(flame_config=consts, batch_size=batch_size, image_size=img_size, **head_mesh_kwargs))

    @torch.autocast("cuda", enabled=False)
    def forward(self, predicted: Tensor, target: Union[Tensor, List[Tensor]]) -> Tensor:
        projected_vertices = self.head_mesh.reprojected_vertices(params_3dmm=predicted, to_2d=True)
        full_target = target[0] if isinstance(target, list) else target
        return _WeightedPointLoss.apply(projected_vertices, full_target, self.tables, self.criterion_id)
