"""ctypes front for the Sim3DR CPU oracles  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Two interchangeable back ends with the same entry points:

  kind="reference": oracle/_ref/libsim3dr_ref.so  = the reference's own Sim3DR/lib/rasterize_kernel.cpp
                    compiled from /root/reference by oracle/Makefile (+ extern "C" shim ref_shim.cpp)
  kind="port":      oracle/_ref/libsim3dr_port.so = oracle/sim3dr_port.c, our C restatement

`get_normal` / `rasterize` mirror the numpy wrappers of Sim3DR/Sim3DR.py:8-29 (zeroed normal
accumulator, depth buffer initialised to -1e8, colours cast to f32, alpha = 1).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF_DIR = os.path.join(_HERE, "_ref")
_F = C.POINTER(C.c_float)
_I = C.POINTER(C.c_int)
_U8 = C.POINTER(C.c_ubyte)


def build(quiet: bool = True) -> None:
    """Compile the oracle libraries (port always; reference when /root/reference is present)."""
    subprocess.run(["make", "-C", _HERE, "all"], check=True, capture_output=quiet)


def available(kind: str) -> bool:
    return os.path.isfile(os.path.join(_REF_DIR, f"libsim3dr_{'ref' if kind == 'reference' else 'port'}.so"))


def best_kind() -> str:
    """"reference" when the reference's own compiled C++ is there (it is built in the authoring container and ships to
    the GPU box inside oracle/_ref/), else the C port that is pinned to it bit for bit (tests/test_oracle_sim3dr.py)."""
    return "reference" if available("reference") else "port"


class Sim3DROracle:
    def __init__(self, kind: str = "port"):
        if kind == "best":
            kind = best_kind()
        assert kind in ("port", "reference")
        self.kind = kind
        name = "libsim3dr_ref.so" if kind == "reference" else "libsim3dr_port.so"
        path = os.path.join(_REF_DIR, name)
        if not os.path.isfile(path):
            build()
        self.lib = C.CDLL(path)
        p = "ref_" if kind == "reference" else "port_"
        self._weight = getattr(self.lib, p + ("get_point_weight" if kind == "reference" else "point_weight"))
        self._in_tri = getattr(self.lib, p + ("is_point_in_tri" if kind == "reference" else "point_in_tri"))
        self._tri_normal = getattr(self.lib, p + "get_tri_normal")
        self._ver_normal = getattr(self.lib, p + "get_ver_normal")
        self._normal = getattr(self.lib, p + "get_normal")
        self._raster = getattr(self.lib, p + "rasterize")
        self._raster_tri = getattr(self.lib, p + "rasterize_triangles")
        f8 = [C.c_float] * 8
        self._weight.argtypes = [_F] + f8
        self._weight.restype = None
        self._in_tri.argtypes = f8
        self._in_tri.restype = C.c_int
        self._tri_normal.argtypes = [_F, _F, _I, C.c_int, C.c_int]
        self._ver_normal.argtypes = [_F, _F, _I, C.c_int, C.c_int]
        self._normal.argtypes = [_F, _F, _I, C.c_int, C.c_int]
        self._raster.argtypes = [_U8, _F, _I, _F, _F, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]
        self._raster_tri.argtypes = [_F, _I, _F, _I, _F, C.c_int, C.c_int, C.c_int]
        for fn in (self._tri_normal, self._ver_normal, self._normal, self._raster, self._raster_tri):
            fn.restype = None

    # -- raw helpers ---------------------------------------------------------------------------
    @staticmethod
    def _f(a):
        assert a.dtype == np.float32 and a.flags.c_contiguous
        return a.ctypes.data_as(_F)

    @staticmethod
    def _i(a):
        assert a.dtype == np.int32 and a.flags.c_contiguous
        return a.ctypes.data_as(_I)

    def point_weight(self, p, p0, p1, p2) -> np.ndarray:
        w = np.zeros(3, np.float32)
        self._weight(self._f(w), p[0], p[1], p0[0], p0[1], p1[0], p1[1], p2[0], p2[1])
        return w

    def point_in_tri(self, p, p0, p1, p2) -> bool:
        return bool(self._in_tri(p[0], p[1], p0[0], p0[1], p1[0], p1[1], p2[0], p2[1]))

    def get_tri_normal(self, vertices, triangles, norm_flg=False) -> np.ndarray:
        out = np.zeros((triangles.shape[0], 3), np.float32)
        self._tri_normal(self._f(out), self._f(vertices), self._i(triangles), triangles.shape[0], int(norm_flg))
        return out

    def get_ver_normal(self, tri_normal, triangles, nver, init=None) -> np.ndarray:
        out = np.zeros((nver, 3), np.float32) if init is None else np.ascontiguousarray(init, np.float32).copy()
        self._ver_normal(self._f(out), self._f(tri_normal), self._i(triangles), nver, triangles.shape[0])
        return out

    # -- Sim3DR/Sim3DR.py mirrors ----------------------------------------------------------------
    def get_normal(self, vertices, triangles, init=None) -> np.ndarray:
        normal = np.zeros_like(vertices, dtype=np.float32) if init is None else np.ascontiguousarray(init, np.float32).copy()
        self._normal(self._f(normal), self._f(vertices), self._i(triangles), vertices.shape[0], triangles.shape[0])
        return normal

    def rasterize(self, vertices, triangles, colors, bg=None, height=None, width=None, channel=None, reverse=False,
                  alpha=1.0, depth=None, return_depth=False):
        if bg is not None:
            height, width, channel = bg.shape
        else:
            bg = np.zeros((height, width, channel), dtype=np.uint8)
        buf = (np.zeros((height, width), dtype=np.float32) - 1e8) if depth is None else depth
        if colors.dtype != np.float32:
            colors = colors.astype(np.float32)
        self._raster(bg.ctypes.data_as(_U8), self._f(vertices), self._i(triangles), self._f(colors), self._f(buf),
                     triangles.shape[0], height, width, channel, float(alpha), int(reverse))
        return (bg, buf) if return_depth else bg

    def rasterize_precast(self, vertices, triangles, colors, height, width, channel=3, reverse=False):
        """Port only (test diagnostic): `rasterize` onto a black image plus the float every written byte was cast from
        (`alpha * 255 * p_color`, rasterize_kernel.cpp:276-281). Returns (image uint8, precast float32, drawn bool)."""
        lib = self.lib if self.kind == "port" else Sim3DROracle("port").lib
        fn = lib.port_rasterize_precast
        fn.argtypes = [_U8, _F, _F, _I, _F, _F, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]
        fn.restype = None
        img = np.zeros((height, width, channel), dtype=np.uint8)
        pre = np.full((height, width, channel), np.nan, dtype=np.float32)
        buf = np.zeros((height, width), dtype=np.float32) - 1e8
        colors = np.ascontiguousarray(colors, dtype=np.float32)
        fn(img.ctypes.data_as(_U8), self._f(pre), self._f(np.ascontiguousarray(vertices, dtype=np.float32)), self._i(triangles),
           self._f(colors), self._f(buf), triangles.shape[0], height, width, channel, 1.0, int(reverse))
        return img, pre, ~np.isnan(pre[..., 0])

    def rasterize_triangles(self, vertices, triangles, h, w, depth=None):
        depth = (np.zeros((h, w), np.float32) - 1e8) if depth is None else depth
        tri_buf = np.zeros((h, w), np.int32) - 1
        bary = np.zeros((h, w, 3), np.float32)
        self._raster_tri(self._f(vertices), self._i(triangles), self._f(depth), self._i(tri_buf), self._f(bary),
                         triangles.shape[0], h, w)
        return depth, tri_buf, bary


def render_pipeline_ref(oracle: Sim3DROracle, vertices, triangles, bg, light_pos=(0, 0, 5), view_pos=(0, 0, 5),
                        ambient=0.3, directional=0.6, specular=0.1, specular_exp=5):
    """Sim3DR/lighting.py:37-71 (`RenderPipeline.__call__`, texture=None) in numpy on top of the oracle.
    Returns (image, per-vertex light) so the shading stage can be checked separately."""
    _norm = lambda a: a / np.sqrt(np.sum(a**2, axis=1))[:, None]  # noqa: E731  lighting.py:6
    normal = oracle.get_normal(vertices, triangles)
    light = np.zeros_like(vertices, dtype=np.float32)
    col = np.array((1, 1, 1), dtype=np.float32)[None, :]
    lp = np.array(light_pos, dtype=np.float32)[None, :]
    vp = np.array(view_pos, dtype=np.float32)[None, :]
    if ambient > 0:
        light += ambient * col
    vn = vertices.copy()  # norm_vertices, lighting.py:9-14
    vn -= vn.min(0)[None, :]
    vn /= vn.max()
    vn *= 2
    vn -= vn.max(0)[None, :] / 2
    if directional > 0:
        direction = _norm(lp - vn)
        cos = np.sum(normal * direction, axis=1)[:, None]
        light += directional * (col * np.clip(cos, 0, 1))
        if specular > 0:
            v2v = _norm(vp - vn)
            reflection = 2 * cos * normal - direction
            spe = np.sum((v2v * reflection) ** specular_exp, axis=1)[:, None]
            spe = np.where(cos != 0, np.clip(spe, 0, 1), np.zeros_like(spe))
            light += specular * col * np.clip(spe, 0, 1)
    light = np.clip(light, 0, 1)
    return oracle.rasterize(vertices, triangles, light, bg=bg), light
