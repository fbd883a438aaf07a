"""Matrix projection of meshes on the MI355X: `P . MV . [v;1]`, perspective divide, image-space flip.

Mirror of the reference's numpy for GT annotations -- `FlameDataset._load_mesh` / `_project_vertices_onto_image`
(model_training/data/flame_dataset.py:115-141) and `get_2d_keypoints` (visualize.py:10-22) -- as ONE HIP launch for a
batch of meshes resident in HBM (`dad3d_project_vertices`).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch
from torch import Tensor

from . import _lib


def project_batch(vertices: Tensor, model_view: Tensor, projection: Tensor, height: Union[float, Tensor] = 0.0,
                  crop_xy: Optional[Tensor] = None, want_world: bool = False, want_int: bool = False) -> Dict[str, Tensor]:
    """`vertices [B,N,3]`, `model_view [B,4,4]`, `projection [B,4,4]` (fp32, CUDA) -> {"xy" [B,N,2],
    "world" [B,N,4] (optional), "xy_int" int32 [B,N,2] (optional)}; `height` scalar or [B], `crop_xy` [B,2] or None."""
    lib = _lib.load()
    dev = vertices.device
    assert vertices.is_cuda and vertices.dtype == torch.float32 and vertices.ndim == 3 and vertices.shape[-1] == 3
    b, n = vertices.shape[:2]
    v = vertices.contiguous()
    mv = model_view.to(dev, torch.float32).reshape(b, 16).contiguous()
    pm = projection.to(dev, torch.float32).reshape(b, 16).contiguous()
    frame = torch.zeros((b, 3), dtype=torch.float32, device=dev)
    frame[:, 0] = torch.as_tensor(height, dtype=torch.float32, device=dev)
    if crop_xy is not None:
        frame[:, 1:] = crop_xy.to(dev, torch.float32)
    out = {"xy": torch.empty((b, n, 2), dtype=torch.float32, device=dev)}
    if want_world:
        out["world"] = torch.empty((b, n, 4), dtype=torch.float32, device=dev)
    if want_int:
        out["xy_int"] = torch.empty((b, n, 2), dtype=torch.int32, device=dev)
    ptr = lambda k: out[k].data_ptr() if k in out else None  # noqa: E731
    _lib.check(lib.dad3d_project_vertices(v.data_ptr(), mv.data_ptr(), pm.data_ptr(), frame.data_ptr(), b, n, ptr("world"),
                                          ptr("xy"), ptr("xy_int"), dev.index or 0, torch.cuda.current_stream(dev).cuda_stream))
    return out


def get_2d_keypoints(data: Dict[str, List], img_height: int, device: int = 0) -> np.ndarray:
    """visualize.py:10-22 for one annotation dict (`vertices`, `model_view_matrix`, `projection_matrix`) -> int [N,2]."""
    dev = torch.device("cuda", device)
    v = torch.tensor(data["vertices"], dtype=torch.float32, device=dev)[None]
    mv = torch.tensor(data["model_view_matrix"], dtype=torch.float32, device=dev)[None]
    pm = torch.tensor(data["projection_matrix"], dtype=torch.float32, device=dev)[None]
    return project_batch(v, mv, pm, float(img_height), want_int=True)["xy_int"][0].cpu().numpy().astype(int)


def project_vertices_onto_image(vertices3d_world_homo: np.ndarray, projection_matrix: np.ndarray, height: int,
                                crop_point_x: int, crop_point_y: int, device: int = 0) -> np.ndarray:
    """flame_dataset.py:130-141: world-space homogeneous vertices [N,4] -> float [N,2] in the cropped image."""
    dev = torch.device("cuda", device)
    w = torch.as_tensor(np.ascontiguousarray(vertices3d_world_homo, dtype=np.float32), device=dev)
    # the world vertices carry w = 1 after a rigid model-view: feed them as points through an identity model-view
    assert torch.all(w[:, 3] == 1.0), "world vertices must be homogeneous points (w == 1)"
    eye = torch.eye(4, device=dev)[None]
    pm = torch.as_tensor(np.asarray(projection_matrix, dtype=np.float32), device=dev)[None]
    crop = torch.tensor([[float(crop_point_x), float(crop_point_y)]], device=dev)
    return project_batch(w[None, :, :3].contiguous(), eye, pm, float(height), crop)["xy"][0].cpu().numpy()
