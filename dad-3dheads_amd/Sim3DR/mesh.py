"""Batched, device-resident Sim3DR: torch CUDA tensors in, torch CUDA tensors out, async on the current stream.

One `Mesh` = one static triangle list (uploaded once together with its vertex->face incidence list).
All methods take a leading batch dimension; nothing is copied to the host.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch
from torch import Tensor

from .. import _lib


def _chk(t: Tensor, dtype, ndim: int, name: str, dev: torch.device) -> Tensor:
    if t.dtype != dtype or t.ndim != ndim or not t.is_contiguous() or t.device != dev:
        raise ValueError(f"{name}: expected a contiguous {dtype} tensor with {ndim} dims on {dev}, "
                         f"got {t.dtype} {tuple(t.shape)} on {t.device}")
    return t


class Mesh:
    def _verts(self, vertices: Tensor, name: str = "vertices") -> Tensor:
        """[B, nver, 3] float32 on this mesh's device -- the kernels index every vertex the triangle list names."""
        v = _chk(vertices, torch.float32, 3, name, self.torch_device)
        if tuple(v.shape[1:]) != (self.nver, 3):
            raise ValueError(f"{name}: expected [B, {self.nver}, 3], got {tuple(v.shape)}")
        return v

    def __init__(self, triangles, nver: int, device: Optional[int] = None):
        tri = np.ascontiguousarray(np.asarray(triangles))
        if tri.dtype != np.int32:
            raise ValueError(f"Buffer dtype mismatch, expected 'int' but got '{tri.dtype}'")  # rasterize.pyx typed buffer
        if tri.ndim != 2 or tri.shape[1] != 3:
            raise ValueError("triangles must have shape [ntri, 3]")
        self._lib = _lib.load()
        _lib.require_gpu()
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.ntri, self.nver = int(tri.shape[0]), int(nver)
        h = C.c_void_p()
        _lib.check(self._lib.dad3d_mesh_create(tri.ctypes.data, self.ntri, self.nver, self.device_index, C.byref(h)))
        self._handle = h

    def __del__(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h:
            try:
                self._lib.dad3d_mesh_destroy(h)
            except Exception:
                pass

    @property
    def torch_device(self) -> torch.device:
        return torch.device("cuda", self.device_index)

    def _stream(self):
        return torch.cuda.current_stream(self.torch_device).cuda_stream

    # -- normals -----------------------------------------------------------------------------------
    def get_normal(self, vertices: Tensor, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
        """`_get_normal` per image: vertices [B,nver,3] -> unit vertex normals [B,nver,3]."""
        v = self._verts(vertices)
        if out is None:
            if accumulate:
                raise ValueError("accumulate=True needs `out`")
            out = torch.empty_like(v)
        self._verts(out, "out")
        if out.shape[0] != v.shape[0]:
            raise ValueError("out: batch mismatch")
        _lib.check(self._lib.dad3d_mesh_get_normal(self._handle, out.data_ptr(), v.data_ptr(), v.shape[0],
                                                   _lib.NORMAL_ACCUMULATE if accumulate else 0, self._stream()))
        return out

    def get_tri_normal(self, vertices: Tensor, norm_flg: bool = False) -> Tensor:
        v = self._verts(vertices)
        out = torch.empty((v.shape[0], self.ntri, 3), dtype=torch.float32, device=v.device)
        _lib.check(self._lib.dad3d_mesh_get_tri_normal(self._handle, out.data_ptr(), v.data_ptr(), v.shape[0], int(norm_flg), self._stream()))
        return out

    def get_ver_normal(self, tri_normal: Tensor, out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
        t = _chk(tri_normal, torch.float32, 3, "tri_normal", self.torch_device)
        if tuple(t.shape[1:]) != (self.ntri, 3):
            raise ValueError(f"tri_normal: expected [B, {self.ntri}, 3], got {tuple(t.shape)}")
        if out is not None and (self._verts(out, "out").shape[0] != t.shape[0]):
            raise ValueError("out: batch mismatch")
        if out is None:
            out = torch.empty((t.shape[0], self.nver, 3), dtype=torch.float32, device=t.device)
        _lib.check(self._lib.dad3d_mesh_get_ver_normal(self._handle, out.data_ptr(), t.data_ptr(), t.shape[0],
                                                       _lib.NORMAL_ACCUMULATE if accumulate else 0, self._stream()))
        return out

    # -- raster ------------------------------------------------------------------------------------
    def rasterize(self, vertices: Tensor, colors: Tensor, bg: Tensor, depth: Optional[Tensor] = None,
                  reverse: bool = False, alpha: float = 1.0) -> Tensor:
        """`_rasterize` per image, in place on `bg` [B,h,w,c] uint8 (returned). colors [B,nver,c] in [0,1]."""
        v = _chk(vertices, torch.float32, 3, "vertices", self.torch_device)
        img = _chk(bg, torch.uint8, 4, "bg", self.torch_device)
        col = _chk(colors, torch.float32, 3, "colors", self.torch_device)
        b, h, w, c = img.shape
        if col.shape != (b, self.nver, c) or v.shape != (b, self.nver, 3):
            raise ValueError("vertices/colors/bg batch or channel mismatch")
        dptr = None
        if depth is not None:
            dptr = _chk(depth, torch.float32, 3, "depth", self.torch_device).data_ptr()
        _lib.check(self._lib.dad3d_mesh_rasterize(self._handle, img.data_ptr(), v.data_ptr(), col.data_ptr(), dptr,
                                                  b, h, w, c, float(alpha), int(reverse), self._stream()))
        return img

    def rasterize_triangles(self, vertices: Tensor, h: int, w: int, depth: Optional[Tensor] = None):
        v = self._verts(vertices)
        b = v.shape[0]
        if depth is None:
            depth = torch.full((b, h, w), -1e8, dtype=torch.float32, device=v.device)
        tri_buf = torch.full((b, h, w), -1, dtype=torch.int32, device=v.device)
        bary = torch.zeros((b, h, w, 3), dtype=torch.float32, device=v.device)
        _lib.check(self._lib.dad3d_mesh_rasterize_triangles(self._handle, v.data_ptr(), depth.data_ptr(), tri_buf.data_ptr(),
                                                            bary.data_ptr(), b, h, w, self._stream()))
        return depth, tri_buf, bary

    # -- lighting ----------------------------------------------------------------------------------
    def phong_light(self, vertices: Tensor, normals: Optional[Tensor] = None, ambient: float = 0.3, directional: float = 0.6,
                    specular: float = 0.1, specular_exp: float = 5, color_ambient: Sequence[float] = (1, 1, 1),
                    color_directional: Sequence[float] = (1, 1, 1), light_pos: Sequence[float] = (0, 0, 5),
                    view_pos: Sequence[float] = (0, 0, 5), normals_out: Optional[Tensor] = None) -> Tensor:
        """Per-vertex Phong light (lighting.py:41-62). `normals=None`: the vertex normals are computed in the same
        launch (and stored into `normals_out` when given) -- RenderPipeline's `_get_normal` + lighting in one pass."""
        v = self._verts(vertices)
        cfg = _lib.LightC(float(ambient), float(directional), float(specular), float(specular_exp),
                          (C.c_float * 3)(*color_ambient), (C.c_float * 3)(*color_directional),
                          (C.c_float * 3)(*light_pos), (C.c_float * 3)(*view_pos))
        out = torch.empty_like(v)
        if normals is None:
            n_out = None if normals_out is None else self._verts(normals_out, "normals_out")
            if n_out is not None and n_out.shape[0] != v.shape[0]:
                raise ValueError("normals_out: batch mismatch")
            _lib.check(self._lib.dad3d_mesh_normal_phong_light(self._handle, out.data_ptr(), None if n_out is None else n_out.data_ptr(),
                                                               v.data_ptr(), v.shape[0], C.byref(cfg), self._stream()))
            return out
        n = self._verts(normals, "normals")
        if n.shape[0] != v.shape[0]:
            raise ValueError("normals: batch mismatch")
        _lib.check(self._lib.dad3d_mesh_phong_light(self._handle, out.data_ptr(), v.data_ptr(), n.data_ptr(), v.shape[0],
                                                    C.byref(cfg), self._stream()))
        return out

    def render(self, vertices: Tensor, bg: Tensor, depth: Optional[Tensor] = None, reverse: bool = False,
               light_out: Optional[Tensor] = None, clear: bool = False, ambient: float = 0.3, directional: float = 0.6, specular: float = 0.1,
               specular_exp: float = 5, color_ambient: Sequence[float] = (1, 1, 1), color_directional: Sequence[float] = (1, 1, 1),
               light_pos: Sequence[float] = (0, 0, 5), view_pos: Sequence[float] = (0, 0, 5)) -> Tensor:
        """RenderPipeline.__call__ (lighting.py:37-71, texture=None) for a batch in TWO launches: the raster's geometry
        kernel also computes normals + Phong light (into `light_out`, allocated when None), the tile kernel rasterises
        with it into the 3-channel `bg`. Same results as `phong_light(v, None)` followed by `rasterize`. `clear=True`:
        `bg` is only a destination, rendered onto black (the geometry launch zeroes it: no fill launch in front)."""
        v = self._verts(vertices)
        img = _chk(bg, torch.uint8, 4, "bg", self.torch_device)
        if img.shape[-1] != 3:
            raise ValueError("render needs a 3-channel image (the light has three components)")
        if img.shape[0] != v.shape[0]:
            raise ValueError("bg: batch mismatch")
        light = torch.empty_like(v) if light_out is None else self._verts(light_out, "light_out")
        if light.shape[0] != v.shape[0]:
            raise ValueError("light_out: batch mismatch")
        if depth is not None and tuple(depth.shape) != tuple(img.shape[:3]):
            raise ValueError("depth: expected [B, h, w] like bg")
        cfg = _lib.LightC(float(ambient), float(directional), float(specular), float(specular_exp),
                          (C.c_float * 3)(*color_ambient), (C.c_float * 3)(*color_directional),
                          (C.c_float * 3)(*light_pos), (C.c_float * 3)(*view_pos))
        dptr = None if depth is None else _chk(depth, torch.float32, 3, "depth", self.torch_device).data_ptr()
        _lib.check(self._lib.dad3d_mesh_render(self._handle, img.data_ptr(), v.data_ptr(), light.data_ptr(), dptr, v.shape[0],
                                               img.shape[1], img.shape[2], C.byref(cfg), int(bool(reverse)) | (2 if clear else 0),
                                               self._stream()))
        return img
