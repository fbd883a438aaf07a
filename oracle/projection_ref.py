"""CPU restatement of the reference's matrix projection  --  TEST INFRASTRUCTURE (never imported by the product).

`load_mesh` follows model_training/data/flame_dataset.py:115-128 (`_load_mesh`), `project_vertices_onto_image`
flame_dataset.py:130-141, `get_2d_keypoints` visualize.py:10-22. numpy float32 throughout, like the reference.
Pinned by tests/golden/projection_golden.npz = outputs of the reference's own `visualize.get_2d_keypoints`
(tests/golden/make_projection_golden.py).
"""
from typing import Dict, List, Tuple

import numpy as np


def load_mesh(data: Dict[str, List]) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    v = np.array(data["vertices"], dtype=np.float32)
    mv = np.array(data["model_view_matrix"], dtype=np.float32)
    homo = np.concatenate((v, np.ones_like(v[:, [0]])), -1)
    world = np.transpose(np.matmul(mv, np.transpose(homo)))  # rotated and translated (to world coordinates)
    return v, world, np.array(data["projection_matrix"], dtype=np.float32)


def project_vertices_onto_image(world_homo: np.ndarray, projection: np.ndarray, height: int, crop_x: int, crop_y: int) -> np.ndarray:
    clip = np.transpose(np.matmul(projection, np.transpose(world_homo)))
    xy = clip[:, :2] / clip[:, [3]]
    xy = np.stack((xy[:, 0], (height - xy[:, 1])), -1)
    xy -= (crop_x, crop_y)
    return xy


def get_2d_keypoints(data: Dict[str, List], img_height: int) -> np.ndarray:
    _, world, proj = load_mesh(data)
    clip = np.transpose(np.matmul(proj, np.transpose(world)))
    xy = clip[:, :2] / clip[:, [3]]
    return np.stack((xy[:, 0], (img_height - xy[:, 1])), -1).astype(int)
