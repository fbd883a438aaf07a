"""Differentiable `HeadMesh.vertices_3d` / `reprojected_vertices` for the reference's training callers.

The reference's losses differentiate through the decode with torch autograd over `flame.py:182-229` + `smplx.lbs.lbs`
(`model_training/losses/vertices_3d_loss.py:41`, `reprojection_loss.py:33`). Here the forward pass is the fused HIP
decode (`dad3d_flame_decode`) and the backward pass is split where the sizes split:

* everything per VERTEX (5023 x B) runs in the HIP library: `dad3d_flame_decode_backward` (csrc/flame_backward.hip)
  turns dL/d(3d_vertices), dL/d(projected) into dL/d(v_posed) [B,V,3] and the per-image sums dL/d(consts) [B,72];
* v_posed (the skinning's input) is saved by the forward launch itself (`dad3d_flame_decode_posed`, one more store per
  vertex); the one contraction of the backward pass with the blend-shape basis, dL/d[betas | pose feature] =
  dL/d(v_posed) @ basis^T, is `dad3d_flame_grad_inputs` up to batch 96 -- a split-K fp32 MFMA kernel over the 3V axis + a
  fixed-order reduction, 25 % faster than rocBLAS at the reference's training batch of 64 -- and rocBLAS through
  `torch.matmul` above (`GRAD_INPUTS_HIP_MAX_BATCH`);
* the 72 per-image constants (joint transforms, 6-DoF rotation, scale, translation) are a few hundred flops per image
  of Rodrigues / kinematic chain / Gram-Schmidt: `dad3d_flame_pose_chain` evaluates them and
  `dad3d_flame_pose_chain_backward` differentiates them with dual numbers over the same device code (one launch each
  instead of the ~300 small kernels of a torch graph). `pose_chain` below states the same chain with torch ops; the
  tests use it (and torch's autograd over it) to check the two kernels, and on CPU to pin the layout against the oracle.

A backward pass is five launches (+ one that adds partial sums for small batches): chain, per-vertex kernel, GEMM +
its reduction, chain VJP.

Formulas restated from the published smplx algorithm (`lbs.py`: batch_rodrigues, batch_rigid_transform) and
`model_training/model/utils.py:92-101` (rot_mat_from_6dof); checked against the oracle's autograd in the tests.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from . import _lib
from .flame import MAX_EXPRESSION, MAX_SHAPE, FlameParams

N_CONSTS = 72  # csrc/common.hpp kBackwardConsts
# dL/d(v_posed) @ basis^T: the hand-written split-K kernel up to this batch, rocBLAS above. Measured on MI355X
# (profiles/r02_bench_train.json): 24.8 vs 31.6 us at B = 64 (the reference's training batch, train_stage/flame_landmarks.yaml:10),
# 32.8 vs 31.8 at 128, 52 vs 45 at 256, 181 vs 120 at 1024: the library reaches ~0.73 of the fp32-MFMA peak on the large
# shapes, this kernel ~0.5 (its multiplying waves share their SIMDs with the staging waves and re-read the image rows once per
# output quarter); it wins where the launch is short enough for the library's fixed costs to show.
GRAD_INPUTS_HIP_MAX_BATCH = 96


def axis_angle_to_matrix(r: Tensor) -> Tensor:
    """[N,3] -> [N,3,3]; smplx batch_rodrigues: angle = ||r + 1e-8||, R = I + sin K + (1 - cos) K K."""
    angle = (r + 1e-8).norm(dim=1, keepdim=True)
    x, y, z = (r / angle).unbind(dim=1)
    o = torch.zeros_like(x)
    k = torch.stack([o, -z, y, z, o, -x, -y, x, o], dim=1).view(-1, 3, 3)
    sin, cos = torch.sin(angle)[..., None], torch.cos(angle)[..., None]
    return torch.eye(3, dtype=r.dtype, device=r.device) + sin * k + (1.0 - cos) * (k @ k)


def six_dof_to_matrix(v: Tensor) -> Tensor:
    """[B,6] -> [B,3,3] with columns b1, b2, b3 (model/utils.py:92-101; crosses along the last dim)."""
    b1 = torch.nn.functional.normalize(v[:, :3], dim=-1)
    b3 = torch.nn.functional.normalize(torch.linalg.cross(b1, v[:, 3:], dim=-1), dim=-1)
    b2 = -torch.linalg.cross(b1, b3, dim=-1)
    return torch.stack((b1, b2, b3), dim=-1)


def relative_joint_transforms(rot: Tensor, joints: Tensor, parents) -> Tensor:
    """smplx batch_rigid_transform, rows 0..2 of the relative transforms only: rot [B,J,3,3], joints [B,J,3] ->
    [B,J,3,4] with A_j = world_j - [0 | world_j . J_j], world_j = world_parent . [R_j | J_j - J_parent]."""
    world_r, world_t = [rot[:, 0]], [joints[:, 0]]
    for j in range(1, rot.shape[1]):
        p = int(parents[j])
        world_r.append(world_r[p] @ rot[:, j])
        world_t.append((world_r[p] @ (joints[:, j] - joints[:, p])[..., None])[..., 0] + world_t[p])
    wr, wt = torch.stack(world_r, dim=1), torch.stack(world_t, dim=1)
    return torch.cat([wr, (wt - (wr @ joints[..., None])[..., 0])[..., None]], dim=-1)


class DecodeTables:
    """Device tensors the backward pass needs besides the library handle (built once per layer, ~27 MB): the blend-shape
    basis in plain [K, 3V] layout for the library GEMMs and the joint regression folded through it.
    Arguments are the fp32 arrays `FLAMELayer.__init__` keeps (flame.py:124-180): v_template [V,3], shapedirs [V,3,400],
    posedirs [36,3V] (already reshaped + transposed), J_regressor [5,V], parents [5], lbs_weights [V,5]."""

    def __init__(self, v_template, shapedirs, posedirs, j_regressor, parents, lbs_weights, device) -> None:
        v = v_template.shape[0]
        flat = np.ascontiguousarray(shapedirs.reshape(v * 3, -1).T)  # [400, 3V]
        self.n_verts, self.n_betas, self.n_pose = v, flat.shape[0], posedirs.shape[0]
        self.basis = torch.from_numpy(np.concatenate([flat, posedirs], axis=0).astype(np.float32)).to(device)  # [436, 3V]
        self.template = torch.from_numpy(np.ascontiguousarray(v_template, dtype=np.float32).reshape(1, v * 3)).to(device)
        jr = np.asarray(j_regressor, dtype=np.float64)
        self.joints0 = torch.from_numpy((jr @ v_template.astype(np.float64)).astype(np.float32)).to(device)  # [5,3]
        jdirs = np.einsum("jv,vck->jck", jr, shapedirs.astype(np.float64))  # [5,3,400]
        self.joint_dirs = torch.from_numpy(jdirs.reshape(-1, self.n_betas).astype(np.float32)).to(device)  # [15,400]
        self.parents = [int(p) for p in parents]
        self.lbs_weights = torch.from_numpy(np.ascontiguousarray(lbs_weights, dtype=np.float32)).to(device)  # [V,5]

    @classmethod
    def from_layer(cls, layer) -> "DecodeTables":
        return cls(layer._v_template, layer._shapedirs, layer._posedirs, layer._j_regressor, layer._parents,
                   layer._weights, layer.torch_device)


def vertex_stage(tables: DecodeTables, inputs: Tensor, consts: Tensor, zero_rot: bool = False, to_2d: bool = True,
                 image_size: float = 256.0) -> Tuple[Tensor, Tensor]:
    """Plain-torch statement of what the library does per vertex (forward), from the chain's outputs: used to check
    `pose_chain` and the layout of the 72 constants without a GPU; never on the product path."""
    b, v = inputs.shape[0], tables.n_verts
    posed = (tables.template + inputs @ tables.basis).view(b, v, 3)
    a, g = consts[:, :60].view(b, 5, 12), consts[:, 60:69].view(b, 3, 3)
    t = (tables.lbs_weights @ a).view(b, v, 3, 4)
    p = (t[..., :3] @ posed[..., None])[..., 0] + t[..., 3]
    p = p + p.new_tensor([0.0, 0.0, 0.05])
    r = (g[:, None] @ p[..., None])[..., 0]
    trans = torch.cat([consts[:, 70:72], consts.new_zeros((b, 1))], dim=1)
    proj = (r * consts[:, 69, None, None] + trans[:, None] + 1.0) / 2.0 * image_size
    return (p if zero_rot else r), (proj[..., :2] if to_2d else proj)


def pose_chain(tables: DecodeTables, consts: Dict[str, int], params: Tensor) -> Dict[str, Tensor]:
    """Everything of the forward pass that is per IMAGE, as differentiable torch ops on `params` [B,P]:
    `inputs` [B,436] = [betas | pose feature] (the A operand of the blend-shape GEMM) and `consts` [B,72]."""
    fp = FlameParams.from_3dmm(params, consts)
    b = params.shape[0]
    zeros = lambda n: params.new_zeros((b, n))  # noqa: E731
    betas = torch.cat([fp.shape, zeros(MAX_SHAPE - fp.shape.shape[1]), fp.expression,
                       zeros(MAX_EXPRESSION - fp.expression.shape[1])], dim=1)  # flame.py:192-200
    neck = fp.neck if fp.neck.shape[1] else zeros(3)
    eyes = fp.eyeballs if fp.eyeballs.shape[1] else zeros(6)
    jaw = fp.jaw if fp.jaw.shape[1] else zeros(3)
    full_pose = torch.cat([zeros(3), neck, jaw, eyes], dim=1)  # flame.py:201-210: the global rotation stays zero
    rot = axis_angle_to_matrix(full_pose.reshape(-1, 3)).view(b, -1, 3, 3)
    pose_feature = (rot[:, 1:] - torch.eye(3, dtype=params.dtype, device=params.device)).reshape(b, -1)
    joints = tables.joints0 + (betas @ tables.joint_dirs.T).view(b, -1, 3)
    a = relative_joint_transforms(rot, joints, tables.parents)  # [B,5,3,4]
    g = six_dof_to_matrix(fp.rotation)
    s = torch.clamp(fp.scale + 1.0, 1e-8)  # head_mesh.py:39
    c = torch.cat([a.reshape(b, -1), g.reshape(b, 9), s, fp.translation[:, :2]], dim=1)
    return {"inputs": torch.cat([betas, pose_feature], dim=1), "consts": c}


class _Decode(torch.autograd.Function):
    """(params) -> (3d_vertices, projected): forward = one fused HIP launch, backward as described in the module doc."""

    @staticmethod
    def forward(ctx, params: Tensor, layer, want_v3: bool, want_proj: bool, zero_rot: bool, to_2d: bool):
        staged = params.detach().to(layer.torch_device, torch.float32).contiguous()
        if staged.data_ptr() == params.data_ptr():
            # the caller may write into its tensor afterwards (reprojected_vertices zeroes tz in place, head_mesh.py:41):
            # keep our own 1.6 KB per image rather than a view whose version counter that write would bump
            staged = staged.clone()
        dev, b, v = layer.torch_device, staged.shape[0], layer.n_verts
        flags = (_lib.ZERO_ROTATION if zero_rot else 0) | (_lib.TO_2D if to_2d else 0)
        d_v3 = torch.empty((b, v, 3), dtype=torch.float32, device=dev) if want_v3 else None
        d_pj = torch.empty((b, v, 2 if to_2d else 3), dtype=torch.float32, device=dev) if want_proj else None
        posed = torch.empty((b, v * 3), dtype=torch.float32, device=dev)  # v_posed, kept for the backward pass
        _lib.check(layer._lib.dad3d_flame_decode_posed(
            layer._handle, staged.data_ptr(), b, flags, d_v3.data_ptr() if want_v3 else None,
            d_pj.data_ptr() if want_proj else None, posed.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
        ctx.layer, ctx.flags = layer, flags
        ctx.src_device = params.device
        ctx.save_for_backward(staged, posed)
        v3 = d_v3.to(params.device) if want_v3 else params.new_empty(0)
        pj = d_pj.to(params.device) if want_proj else params.new_empty(0)
        if not want_v3:
            ctx.mark_non_differentiable(v3)
        if not want_proj:
            ctx.mark_non_differentiable(pj)
        return v3, pj

    @staticmethod
    def backward(ctx, g_v3: Optional[Tensor], g_pj: Optional[Tensor]):
        layer = ctx.layer
        staged, posed = ctx.saved_tensors
        dev, b, v = layer.torch_device, staged.shape[0], layer.n_verts
        tables = layer.decode_tables()

        def on_dev(g, cols):
            if g is None or g.numel() == 0:
                return None
            return g.detach().to(dev, torch.float32).reshape(b, v, cols).contiguous()

        g_v3 = on_dev(g_v3, 3)
        g_pj = on_dev(g_pj, 2 if ctx.flags & _lib.TO_2D else 3)
        if g_v3 is None and g_pj is None:
            return None, None, None, None, None, None
        lib, handle = layer._lib, layer._handle
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.no_grad():
            inputs = torch.empty((b, tables.basis.shape[0]), dtype=torch.float32, device=dev)
            consts = torch.empty((b, N_CONSTS), dtype=torch.float32, device=dev)
            _lib.check(lib.dad3d_flame_pose_chain(handle, staged.data_ptr(), b, inputs.data_ptr(), consts.data_ptr(), stream))
            g_posed = torch.empty_like(posed)
            g_consts = torch.empty_like(consts)
            _lib.check(lib.dad3d_flame_decode_backward(
                handle, b, ctx.flags, consts.data_ptr(), posed.data_ptr(),
                g_v3.data_ptr() if g_v3 is not None else None, g_pj.data_ptr() if g_pj is not None else None,
                g_posed.data_ptr(), g_consts.data_ptr(), stream))
            if b <= GRAD_INPUTS_HIP_MAX_BATCH:
                g_inputs = torch.empty((b, tables.basis.shape[0]), dtype=torch.float32, device=dev)  # [B,436]
                _lib.check(lib.dad3d_flame_grad_inputs(handle, g_posed.data_ptr(), b, g_inputs.data_ptr(), stream))
            else:
                g_inputs = g_posed @ tables.basis.T  # library GEMM: the split-K partials outgrow their gain (see above)
            g_params = torch.empty_like(staged)
            _lib.check(lib.dad3d_flame_pose_chain_backward(handle, staged.data_ptr(), b, g_inputs.data_ptr(),
                                                           g_consts.data_ptr(), g_params.data_ptr(), stream))
        return g_params.to(ctx.src_device), None, None, None, None, None


def decode_with_grad(layer, params: Tensor, *, verts3d: bool, proj: bool, zero_rot: bool = False,
                     to_2d: bool = True) -> Tuple[Optional[Tensor], Optional[Tensor]]:
    v3, pj = _Decode.apply(params, layer, verts3d, proj, zero_rot, to_2d)
    return (v3 if verts3d else None), (pj if proj else None)
