"""On-disk formats downstream of the decode: the `.obj` mesh text and the FLAME-parameter JSON of the
reference's demo (`demo_utils.py:106-153`). Same bytes as `MeshSaver` / `JsonSaver` write; the batch variants format a
whole batch of decoded meshes with the constant face block rendered once.
"""
from __future__ import annotations

import io
import json
import os
from typing import Any, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .flame import FLAME_CONSTS, FlameParams


def get_mesh(predictions: Dict[str, torch.Tensor], faces: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """demo_utils.py:106-109: vertices [N,3] and faces as FLOATS shifted to 1-based indices (`+ 1.`)."""
    return predictions["3d_vertices"].detach().cpu().numpy(), np.asarray(faces) + 1.0


def get_flame_params(predictions: Dict[str, torch.Tensor], constants: Dict[str, int] = FLAME_CONSTS) -> Dict[str, List[float]]:
    """demo_utils.py:112-116: {"shape": [...], "expression": [...], ...} of the first row."""
    fp = FlameParams.from_3dmm(predictions["3dmm_params"].detach().cpu(), constants)
    return {k: v[0].tolist() for k, v in vars(fp).items()}


def _vertex_block(vertices: np.ndarray) -> str:
    buf = io.StringIO()
    np.savetxt(buf, np.asarray(vertices).reshape(-1, 3), fmt="v %.8f %.8f %.8f")  # '%'-formatting row by row, '\n' ends
    return buf.getvalue()


def _face_block(faces_1based: np.ndarray) -> str:
    buf = io.StringIO()
    np.savetxt(buf, np.asarray(faces_1based).reshape(-1, 3), fmt="f %d %d %d")
    return buf.getvalue()


def obj_text(vertices: np.ndarray, faces_1based: np.ndarray) -> str:
    """The text `MeshSaver.__call__` writes (demo_utils.py:130-144): `v %.8f %.8f %.8f` lines, then `f %d %d %d`."""
    return _vertex_block(vertices) + _face_block(faces_1based)


class MeshSaver:
    def __init__(self) -> None:
        self.extension = ".obj"

    def __call__(self, mesh: Tuple[np.ndarray, np.ndarray], output_path: str) -> None:
        vertices, faces = mesh
        with open(output_path, "w") as f:
            f.write(obj_text(vertices, faces))


class JsonSaver:
    def __init__(self) -> None:
        self.extension = ".json"

    def __call__(self, flame_params: Dict[str, List[float]], output_path: str) -> None:
        with open(output_path, "w") as out:
            json.dump(flame_params, out)


def save_obj_batch(vertices: torch.Tensor, faces: np.ndarray, paths: Sequence[str]) -> None:
    """`vertices [B,N,3]` (any device) -> one `.obj` per row. One device-to-host copy for the batch; the face block is
    the same text for every mesh and is formatted once."""
    v = vertices.detach().cpu().numpy()
    assert v.ndim == 3 and v.shape[0] == len(paths)
    face_text = _face_block(np.asarray(faces) + 1.0)
    for row, path in zip(v, paths):
        with open(path, "w") as f:
            f.write(_vertex_block(row))
            f.write(face_text)


def flame_params_batch(params: torch.Tensor, constants: Dict[str, int] = FLAME_CONSTS) -> List[Dict[str, List[float]]]:
    """`params [B,413]` -> the per-image dictionaries `get_flame_params` produces."""
    fp = FlameParams.from_3dmm(params.detach().cpu(), constants)
    fields = {k: v.tolist() for k, v in vars(fp).items()}
    return [{k: rows[i] for k, rows in fields.items()} for i in range(params.shape[0])]


def get_output_path(input_image_path: str, outputs_folder: str, type_of_output: str, extension: str) -> str:
    """demo_utils.py:156-163: `<outputs_folder>/<stem>_<type_of_output><extension>`."""
    stem = os.path.splitext(os.path.basename(input_image_path))[0]
    return os.path.join(outputs_folder, f"{stem}_{type_of_output}{extension}")
