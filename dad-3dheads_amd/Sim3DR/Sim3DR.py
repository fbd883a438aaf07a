"""numpy call surface of the reference's `Sim3DR/Sim3DR.py:8-29`, executed on the GPU.

Arrays are host numpy like in the reference (float32 vertices/colours, int32 triangles, uint8 image);
the typed-buffer errors of the Cython layer (Sim3DR/lib/rasterize.pyx:44-102) are reproduced as
ValueError/TypeError. Each call goes through the single-image host entry points of the C ABI
(`dad3d_sim3dr_*`, include/dad3d.h), i.e. the same functions rasterize.pyx would link against.
"""
from __future__ import annotations

import numpy as np

from .. import _lib

_CNAME = {np.dtype(np.float32): "float", np.dtype(np.float64): "double", np.dtype(np.int32): "int",
          np.dtype(np.int64): "long", np.dtype(np.uint8): "unsigned char"}


def _typed(arr, dtype, ndim: int, name: str) -> np.ndarray:
    if arr is None:
        raise TypeError(f"Argument '{name}' must not be None")
    if not isinstance(arr, np.ndarray):
        raise TypeError(f"Argument '{name}' has incorrect type (expected numpy.ndarray, got {type(arr).__name__})")
    if arr.dtype != np.dtype(dtype):
        raise ValueError(f"Buffer dtype mismatch, expected '{_CNAME[np.dtype(dtype)]}' but got "
                         f"'{_CNAME.get(arr.dtype, str(arr.dtype))}'")
    if arr.ndim != ndim:
        raise ValueError(f"Buffer has wrong number of dimensions (expected {ndim}, got {arr.ndim})")
    if not arr.flags.c_contiguous:
        raise ValueError("ndarray is not C-contiguous")
    return arr


def _raise_if_failed(lib) -> None:
    # the reference-shaped entry points return void; a failure leaves a message behind
    msg = lib.dad3d_last_error()
    if msg:
        raise _lib.Dad3dError(_lib.E_HIP, msg.decode("utf-8", "replace"))


def get_normal(vertices, triangles):
    lib = _lib.load()
    _lib.require_gpu()
    v = _typed(vertices, np.float32, 2, "vertices")
    t = _typed(triangles, np.int32, 2, "triangles")
    normal = np.zeros_like(v, dtype=np.float32)  # Sim3DR.py:9
    lib.dad3d_clear_error()
    lib.dad3d_sim3dr_get_normal(normal.ctypes.data, v.ctypes.data, t.ctypes.data, v.shape[0], t.shape[0])
    _raise_if_failed(lib)
    return normal


def rasterize(vertices, triangles, colors, bg=None, height=None, width=None, channel=None, reverse=False):
    lib = _lib.load()
    _lib.require_gpu()
    if bg is not None:
        height, width, channel = bg.shape
    else:
        assert height is not None and width is not None and channel is not None
        bg = np.zeros((height, width, channel), dtype=np.uint8)
    buffer = np.zeros((height, width), dtype=np.float32) - 1e8  # Sim3DR.py:23
    if colors.dtype != np.float32:
        colors = colors.astype(np.float32)
    img = _typed(bg, np.uint8, 3, "image")
    v = _typed(vertices, np.float32, 2, "vertices")
    t = _typed(triangles, np.int32, 2, "triangles")
    c = _typed(colors, np.float32, 2, "colors")
    lib.dad3d_clear_error()
    lib.dad3d_sim3dr_rasterize(img.ctypes.data, v.ctypes.data, t.ctypes.data, c.ctypes.data, buffer.ctypes.data,
                               t.shape[0], height, width, channel, 1.0, int(bool(reverse)))
    _raise_if_failed(lib)
    return bg


def rasterize_triangles(vertices, triangles, height, width, depth_buffer=None):
    """`Sim3DR_Cython.rasterize_triangles` (rasterize.pyx:74-86; exported by the binding, unused in-repo)."""
    lib = _lib.load()
    _lib.require_gpu()
    v = _typed(vertices, np.float32, 2, "vertices")
    t = _typed(triangles, np.int32, 2, "triangles")
    depth = (np.zeros((height, width), np.float32) - 1e8) if depth_buffer is None else _typed(depth_buffer, np.float32, 2, "depth_buffer")
    tri_buf = np.zeros((height, width), np.int32) - 1
    bary = np.zeros((height, width, 3), np.float32)
    lib.dad3d_clear_error()
    lib.dad3d_sim3dr_rasterize_triangles(v.ctypes.data, t.ctypes.data, depth.ctypes.data, tri_buf.ctypes.data,
                                         bary.ctypes.data, t.shape[0], height, width)
    _raise_if_failed(lib)
    return depth, tri_buf, bary
