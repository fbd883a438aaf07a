"""PNCC rendering (projected normalised coordinate codes), the step directly after the decode.

Mirror of `inference/pncc_estimator.py`: `compute_ncc_color_codes` (45-60), `pncc` (16-43), `PNCCEstimator` (66-101) with
the same arguments and results, plus the batched device-resident entry `render_batch`: ONE fused decode launch
(`to_2d=False`, z flipped as at pncc_estimator.py:87-88 -- the flip is a flag of the kernel, no extra pass) followed by
the z-buffer raster of the 6270-face subset with the NCC colours; vertices and images never leave HBM.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple, Union

import numpy as np
import torch
from torch import Tensor

from . import Sim3DR
from .head_mesh import HeadMesh
from .Sim3DR.mesh import Mesh
from .synthetic import load_static


def compute_ncc_color_codes(template_face: np.ndarray, subset_indexes: Optional[np.ndarray] = None) -> np.ndarray:
    """pncc_estimator.py:45-60. Note `initial=0` there: zero takes part in the min and the max."""
    if not isinstance(template_face, np.ndarray):
        raise ValueError(f"Argument template_face must be a numpy array, got type {type(template_face)}")
    if len(template_face.shape) != 2 or template_face.shape[1] != 3:
        raise ValueError(f"Argument template_face must have shape [N,3], got shape {template_face.shape}")
    if subset_indexes is not None and not isinstance(subset_indexes, np.ndarray):
        raise ValueError(f"Argument subset_indexes must be a numpy array, got type {type(subset_indexes)}")
    sub = template_face[subset_indexes] if subset_indexes is not None else template_face
    lo = sub.min(axis=0, keepdims=True, initial=0)
    hi = sub.max(axis=0, keepdims=True, initial=0)
    return (template_face - lo) / (hi - lo)


def pncc(img: np.ndarray, vertices: np.ndarray, faces: np.ndarray, colors: np.ndarray, with_bg_flag: bool = True) -> np.ndarray:
    """pncc_estimator.py:16-43: render the coloured mesh over a copy of `img` or over black."""
    overlap = img.copy() if with_bg_flag else np.zeros_like(img)
    c = np.ascontiguousarray
    return Sim3DR.rasterize(c(vertices), c(faces), c(colors), bg=overlap)


class PNCCEstimator:
    def __init__(self, img_size: int = 512, head_mesh: Optional[HeadMesh] = None, static: Optional[dict] = None, **head_mesh_kwargs):
        self.img_size = img_size
        st = static if static is not None else load_static()
        self.head_mesh = head_mesh if head_mesh is not None else HeadMesh(static=st, **head_mesh_kwargs)
        self.faces_wo_back_remapped = np.ascontiguousarray(st["faces_wo_ears"])  # static/flame_indices/faces_wo_ears_remapped.npy
        template = np.asarray(self.head_mesh.flame.flame_model.v_template)
        self.colors = compute_ncc_color_codes(template, np.unique(self.faces_wo_back_remapped))
        self._mesh: Optional[Mesh] = None
        self._colors_dev: Optional[Tensor] = None

    # -- reference surface (single image, host arrays) ---------------------------------------------------
    def _transform_3dmm_to_3d_face_polygons(self, mm_params: Union[Tensor, np.ndarray], flame) -> Tuple[np.ndarray, np.ndarray]:
        with torch.no_grad():
            vertices = self.head_mesh.reprojected_vertices(torch.as_tensor(mm_params), to_2d=False)
            vertices[:, :, 2] *= -1  # pncc_estimator.py:88
        return vertices[0].cpu().numpy(), flame.faces.astype(int)

    def __call__(self, image: np.ndarray, predictions: Dict[str, Tensor], with_background: bool = False) -> np.ndarray:
        v1, _ = self._transform_3dmm_to_3d_face_polygons(predictions["3dmm_params"], self.head_mesh.flame)
        # the Cython binding of the reference takes float32 / int32 buffers; the host wrapper here is as strict
        return pncc(image, v1.astype(np.float32), self.faces_wo_back_remapped.astype(np.int32),
                    self.colors.astype(np.float32), with_background)

    # -- MI355X-native batched entry ----------------------------------------------------------------------
    def render_batch(self, params: Tensor, bg: Optional[Tensor] = None, size: Optional[Tuple[int, int]] = None,
                     mutate: bool = True) -> Tensor:
        """`params [B,413]` fp32 on the GPU -> uint8 `[B,H,W,3]` PNCC images on the GPU. `bg` (rendered into, like
        `Sim3DR.rasterize`'s bg) or `size=(H, W)` for a black background. Two stream-ordered steps: fused decode with
        the z flip, raster of the ear-less face subset with the colour codes."""
        flame = self.head_mesh.flame
        dev = flame.torch_device
        if self._mesh is None:
            self._mesh = Mesh(self.faces_wo_back_remapped.astype(np.int32), int(np.asarray(flame.flame_model.v_template).shape[0]),
                              device=flame.device_index)
            self._colors_dev = torch.from_numpy(self.colors.astype(np.float32)).to(dev)
        b = params.shape[0]
        if bg is None:
            h, w = size if size is not None else (self.img_size, self.img_size)
            bg = torch.zeros((b, h, w, 3), dtype=torch.uint8, device=dev)
        verts = flame.decode(params, proj=True, to_2d=False, flip_z=True, mutate=mutate)["proj"]
        colors = self._colors_dev[None].expand(b, -1, -1).contiguous()
        return self._mesh.rasterize(verts, colors, bg)
