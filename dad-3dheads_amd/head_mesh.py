"""Drop-in for `model_training.head_mesh.HeadMesh` (head_mesh.py:9-60) backed by the HIP decode.

Same constructor arguments, same method names, same return shapes, same side effects:
`reprojected_vertices` zeroes translation z inside the caller's `params_3dmm` (head_mesh.py:41).
Tensors come back on the device the input lives on: a CPU tensor (what `predictor.py:74,136-137`
passes) is staged through the GPU and returned as a CPU tensor; a CUDA tensor never leaves HBM.

`decode()` is the MI355X-native entry: ONE launch returns `vertices_3d`, the projection and the gathered
landmarks, where the reference runs two full decodes (predictor.py:136-137).

Training callers (`losses/vertices_3d_loss.py:41`, `losses/reprojection_loss.py:33`) pass a tensor that requires grad:
`vertices_3d` / `reprojected_vertices` then go through `autograd.decode_with_grad` (same forward launch; backward = the
library's per-vertex and pose-chain kernels plus ONE GEMM dL/d(v_posed) . basis^T -- the hand-written split-K MFMA kernel up
to batch 96, rocBLAS above) and the result carries a grad_fn like the reference's.

`flame.compat_cross_b3 = True` opts into the reference's batch-of-exactly-three behaviour (`torch.cross` without `dim`,
model_training/model/utils.py:98-99, crosses over the batch axis); by default every image gets its own rotation.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn
from torch import Tensor

from .autograd import decode_with_grad
from .flame import FLAME_CONSTS, FLAMELayer, FlameParams


class HeadMesh(nn.Module):
    def __init__(self, flame_config: Optional[Dict[str, int]] = None, batch_size: int = 1, image_size: int = 256,
                 flame_model=None, flame_path: Optional[str] = None, device: Optional[int] = None,
                 landmarks: Optional[Sequence[int]] = None, static: Optional[dict] = None):
        super().__init__()
        self.flame_constants = FLAME_CONSTS if flame_config is None else flame_config
        self.flame = FLAMELayer(consts=self.flame_constants, batch_size=batch_size, flame_path=flame_path,
                                flame_model=flame_model, device=device, image_size=image_size, static=static)
        self._image_size = image_size
        if landmarks is not None:
            self.flame.set_landmarks(landmarks)

    # -- reference surface -----------------------------------------------------------------------
    def flame_params(self, params_3dmm: Tensor) -> FlameParams:
        return FlameParams.from_3dmm(params_3dmm, self.flame_constants)

    def _stage(self, params_3dmm: Tensor) -> Tensor:
        if params_3dmm.ndim != 2:
            raise AssertionError("tensor_3dmm.ndim == 2 expected")  # flame.py:46
        dev = self.flame.torch_device
        if params_3dmm.device == dev and params_3dmm.dtype == torch.float32 and params_3dmm.is_contiguous():
            return params_3dmm
        return params_3dmm.detach().to(dev, torch.float32).contiguous()

    @staticmethod
    def _needs_grad(params_3dmm: Tensor) -> bool:
        return torch.is_grad_enabled() and params_3dmm.requires_grad

    def vertices_3d(self, params_3dmm: Tensor, zero_rotation: bool = False) -> Tensor:
        if self._needs_grad(params_3dmm):
            if params_3dmm.ndim != 2:
                raise AssertionError("tensor_3dmm.ndim == 2 expected")  # flame.py:46
            return decode_with_grad(self.flame, params_3dmm, verts3d=True, proj=False, zero_rot=zero_rotation)[0]
        staged = self._stage(params_3dmm)
        out = self.flame.decode(staged, verts3d=True, zero_rot=zero_rotation)["verts3d"]
        return out.to(params_3dmm.device)

    def reprojected_vertices(self, params_3dmm: Tensor, to_2d: bool = True) -> Tensor:
        """Returns [B, N, C] (C = 2 or 3) and sets translation z := 0 in `params_3dmm`, like the reference."""
        if self._needs_grad(params_3dmm):
            if params_3dmm.ndim != 2:
                raise AssertionError("tensor_3dmm.ndim == 2 expected")  # flame.py:46
            # head_mesh.py:41, tracked by autograd exactly as in the reference (a leaf that requires grad raises there too)
            self.flame_params(params_3dmm).translation[..., 2] = 0.0
            return decode_with_grad(self.flame, params_3dmm, verts3d=False, proj=True, to_2d=to_2d)[1]
        staged = self._stage(params_3dmm)
        out = self.flame.decode(staged, proj=True, to_2d=to_2d, mutate=True)["proj"]
        if staged is not params_3dmm:  # replay the in-place side effect on the caller's tensor
            with torch.no_grad():
                self.flame_params(params_3dmm).translation[..., 2] = 0.0
        return out.to(params_3dmm.device)

    def adjust_3dmm_to_paddings(self, params_3dmm: Tensor, paddings: List[int]) -> Tensor:
        """head_mesh.py:48-60 (paddings = [top, bottom, left, right]); bug-compatible with
        `to_3dmm_tensor`'s rotation-before-jaw order."""
        fp = self.flame_params(params_3dmm)
        fp.translation = fp.translation + Tensor([[paddings[2], paddings[0], 0]]).to(params_3dmm.device) * 2 / self._image_size
        return fp.to_3dmm_tensor()

    def fork(self) -> "HeadMesh":
        """A HeadMesh for another stream (see `FLAMELayer.fork`): same model in HBM, independent launches."""
        twin = object.__new__(type(self))
        twin.__dict__ = {k: (dict(v) if isinstance(v, dict) else v) for k, v in self.__dict__.items()}
        twin.flame = self.flame.fork()
        return twin

    # -- fused entry ------------------------------------------------------------------------------
    def set_landmarks(self, indices: Sequence[int]) -> None:
        self.flame.set_landmarks(indices)

    def decode(self, params_3dmm: Tensor, *, verts3d: bool = True, proj: bool = True, to_2d: bool = True,
               landmarks: bool = True, landmarks_px: bool = False, zero_rotation: bool = False, flip_z: bool = False,
               mutate: bool = True, out: Optional[Dict[str, Tensor]] = None) -> Dict[str, Tensor]:
        """One launch: {"verts3d" [B,V,3], "proj" [B,V,2|3], "lmk_xy" [B,n,2], "lmk_px" int32 [B,n,2]}.
        `params_3dmm` must already be a contiguous fp32 CUDA tensor on this module's device."""
        want_l = landmarks and self.flame.n_landmarks > 0
        want_lp = landmarks_px and self.flame.n_landmarks > 0
        return self.flame.decode(params_3dmm, verts3d=verts3d, proj=proj, to_2d=to_2d, landmarks=want_l,
                                 landmarks_px=want_lp, zero_rot=zero_rotation, flip_z=flip_z, mutate=mutate, out=out)
