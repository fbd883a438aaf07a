"""Landmark index lists of the reference (`model_training/utils.py:62-105`, `demo_utils.py:37-47`).

`load_indices_from_npy` / `load_2d_indices` / `get_list_of_npy_files` keep the reference's semantics (dict
values concatenated in insertion order; files sorted; `cheeks` excluded by default) for users that point at
a reference checkout; `canonical(...)` serves the frozen lists from the static fixture:

  "445"  -> keypoints_445, sorted files, cheeks excluded  (the training / benchmark list, 445 unique)
  "565"  -> keypoints_445, sorted files, every file       (what `demo.py 445_landmarks` draws)
  "191"  -> keypoints_191 == static/indices_2d.npy
"""
from __future__ import annotations

import os
from typing import Any, Dict, List, Optional

import numpy as np

from .synthetic import load_static


def load_indices_from_npy(filepath: str) -> List[int]:
    data = np.load(filepath, allow_pickle=True)[()]
    out: List[int] = []
    for value in data.values():
        out += list(value)
    return out


def get_list_of_npy_files(config: Dict[str, Any]) -> List[str]:
    root = str(config.get("2d_subset_path"))
    subset = config.get("2d_keys", "all")
    exclude = config.get("2d_keys_exclude", "cheeks")
    if isinstance(subset, str) and subset == "all":
        names = [f.split(".")[0] for f in os.listdir(root)]
        for feat in ([exclude] if isinstance(exclude, str) else (exclude or [])):
            if feat in names:
                names.remove(feat)
        subset = [os.path.join(root, n + ".npy") for n in names]
    return subset


def load_2d_indices(config: Dict[str, Any]) -> Optional[List[int]]:
    if config["2d_subset_name"] == "multipie_keypoints":
        return None
    out: List[int] = []
    for fn in sorted(get_list_of_npy_files(config)):
        if not os.path.exists(fn):
            raise ValueError(f"[{fn.split('.')[0].split('/')[-1]}] class of keypoints doesn't exist")
        out += load_indices_from_npy(fn)
    return out


def canonical(subset: str = "445", static: Optional[dict] = None) -> np.ndarray:
    st = static if static is not None else load_static()
    key = {"445": "lmk_445", "565": "lmk_565", "191": "lmk_191"}.get(str(subset))
    if key is None:
        raise ValueError("Invalid keypoints subset provided. Available options are: 191, 445, 565")
    return st[key].astype(np.int64)
