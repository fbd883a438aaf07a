"""ctypes binding of libdad3d_hip.so (the C ABI declared in include/dad3d.h).

There is NO CPU fallback: if the library is missing or a call fails, the caller gets an exception.
Build it with `python __graft_entry__.py build` (or `make -C dad-3dheads_amd/csrc`).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DAD3D_LIB_PATH") or os.path.join(_HERE, "libdad3d_hip.so")  # override: diagnostics builds

OK, E_INVALID, E_HIP, E_UNSUPPORTED, E_NOMEM = range(5)
ZERO_ROTATION, TO_2D, MUTATE_PARAMS, FLIP_Z, COMPAT_CROSS_B3 = 0x1, 0x2, 0x4, 0x8, 0x10
NORMAL_ACCUMULATE = 0x1
KERNEL_AUTO, KERNEL_TWO_ROLE, KERNEL_PIPELINED, KERNEL_SPLIT_BF16, KERNEL_SPLIT_F16 = 0, 1, 2, 3, 4


class Dad3dError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libdad3d_hip status {status}: {message}")
        self.status = status


class UnsupportedError(Dad3dError):
    pass


class FlameModelC(C.Structure):
    _fields_ = [
        ("n_verts", C.c_int32),
        ("n_betas", C.c_int32),
        ("n_joints", C.c_int32),
        ("v_template", C.c_void_p),
        ("shapedirs", C.c_void_p),
        ("posedirs", C.c_void_p),
        ("j_regressor", C.c_void_p),
        ("parents", C.c_void_p),
        ("lbs_weights", C.c_void_p),
    ]


class FlameConstsC(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("shape", "expression", "jaw", "rotation", "eyeballs", "neck", "translation", "scale")]


class LightC(C.Structure):
    _fields_ = [
        ("intensity_ambient", C.c_float),
        ("intensity_directional", C.c_float),
        ("intensity_specular", C.c_float),
        ("specular_exp", C.c_float),
        ("color_ambient", C.c_float * 3),
        ("color_directional", C.c_float * 3),
        ("light_pos", C.c_float * 3),
        ("view_pos", C.c_float * 3),
    ]


# name -> (restype, argtypes); mirrors include/dad3d.h one to one
_P, _I, _U, _F = C.c_void_p, C.c_int, C.c_uint, C.c_float
SIGNATURES = {
    "dad3d_last_error": (C.c_char_p, []),
    "dad3d_clear_error": (None, []),
    "dad3d_version": (_I, []),
    "dad3d_build_info": (C.c_char_p, []),
    "dad3d_device_count": (_I, []),
    "dad3d_flame_create": (_I, [C.POINTER(FlameModelC), C.POINTER(FlameConstsC), _F, _I, C.POINTER(_P)]),
    "dad3d_flame_destroy": (None, [_P]),
    "dad3d_flame_fork": (_I, [_P, C.POINTER(_P)]),
    "dad3d_flame_num_params": (_I, [_P]),
    "dad3d_flame_num_verts": (_I, [_P]),
    "dad3d_flame_set_landmarks": (_I, [_P, _P, _I]),
    "dad3d_flame_num_landmarks": (_I, [_P]),
    "dad3d_flame_num_landmark_vertices": (_I, [_P]),
    "dad3d_flame_decode": (_I, [_P, _P, _I, _U, _P, _P, _P, _P, _P]),
    "dad3d_flame_decode_posed": (_I, [_P, _P, _I, _U, _P, _P, _P, _P]),
    "dad3d_flame_decode_backward": (_I, [_P, _I, _U, _P, _P, _P, _P, _P, _P, _P]),
    "dad3d_flame_grad_inputs": (_I, [_P, _P, _I, _P, _P]),
    "dad3d_flame_num_chain_inputs": (_I, [_P]),
    "dad3d_flame_pose_chain": (_I, [_P, _P, _I, _P, _P, _P]),
    "dad3d_flame_pose_chain_backward": (_I, [_P, _P, _I, _P, _P, _P, _P]),
    "dad3d_flame_decode_host": (_I, [_P, _P, _I, _U, _P, _P, _P, _P]),
    "dad3d_flame_readjust_params": (_I, [_P, _P, _I, _P, _F, _F, _F, _P]),
    "dad3d_flame_profile_begin": (_I, [_P, _P]),
    "dad3d_flame_profile_end": (_I, [_P, _P, C.POINTER(C.c_double), C.POINTER(_I)]),
    "dad3d_flame_select_kernel": (_I, [_P, _I]),
    "dad3d_flame_handoff_timeouts": (_I, [_P, C.POINTER(C.c_uint)]),
    "dad3d_flame_debug_trace": (_I, [_P, _P, C.c_uint64]),
    "dad3d_flame_debug_trace_entries": (C.c_uint64, [_P, _I]),
    "dad3d_mesh_create": (_I, [_P, _I, _I, _I, C.POINTER(_P)]),
    "dad3d_mesh_destroy": (None, [_P]),
    "dad3d_mesh_get_normal": (_I, [_P, _P, _P, _I, _U, _P]),
    "dad3d_mesh_get_tri_normal": (_I, [_P, _P, _P, _I, _I, _P]),
    "dad3d_mesh_get_ver_normal": (_I, [_P, _P, _P, _I, _U, _P]),
    "dad3d_mesh_rasterize": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P]),
    "dad3d_mesh_rasterize_triangles": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "dad3d_mesh_phong_light": (_I, [_P, _P, _P, _P, _I, C.POINTER(LightC), _P]),
    "dad3d_mesh_normal_phong_light": (_I, [_P, _P, _P, _P, _I, C.POINTER(LightC), _P]),
    "dad3d_mesh_render": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, C.POINTER(LightC), _I, _P]),
    "dad3d_mesh_debug_trace": (_I, [_P, _P]),
    "dad3d_project_vertices": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _P, _I, _P]),
    "dad3d_preprocess_images": (_I, [_P, _I, _I, C.POINTER(C.c_float), C.POINTER(C.c_float), _P, _I, _P]),
    "dad3d_nhwc_bias_act": (_I, [_P, _P, _P, C.c_int64, _I, _I, _I, _I, _P]),
    "dad3d_nhwc_resize_sum": (_I, [_P, _I, _I, _I, _I, _I, _I, C.POINTER(_P), C.POINTER(_I), C.POINTER(_I), C.POINTER(C.c_float), _I, _P]),
    "dad3d_cube_region_loss": (_I, [_P, _P, _I, _I, _P, _P, _P, _I, _P, _P, _P, _I, _P, _P, _P, _I, _P]),
    "dad3d_point_loss_terms": (_I, [_I]),
    "dad3d_weighted_point_loss": (_I, [_P, _P, _I, _I, _I, _P, _F, _I, _P, _P, _I, _P]),
    "dad3d_sim3dr_get_tri_normal": (None, [_P, _P, _P, _I, _I]),
    "dad3d_sim3dr_get_ver_normal": (None, [_P, _P, _P, _I, _I]),
    "dad3d_sim3dr_get_normal": (None, [_P, _P, _P, _I, _I]),
    "dad3d_sim3dr_rasterize_triangles": (None, [_P, _P, _P, _P, _P, _I, _I, _I]),
    "dad3d_sim3dr_rasterize": (None, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I]),
}

_lib = None


def load() -> C.CDLL:
    """Load the HIP library or raise. Never falls back to a CPU implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python __graft_entry__.py build`). There is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != OK:
        msg = load().dad3d_last_error().decode("utf-8", "replace")
        raise (UnsupportedError if status == E_UNSUPPORTED else Dad3dError)(status, msg)


def require_gpu() -> None:
    if load().dad3d_device_count() < 1:
        raise RuntimeError("no HIP device visible: the dad-3dheads_amd hot path runs on an MI355X only (no CPU fallback)")
