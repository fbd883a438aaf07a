"""Host-side mirror of `model_training/model/flame.py` for the MI355X decode path.

Keeps the reference's names and argument meaning -- `FLAME_CONSTS`, `FlameParams.from_3dmm` /
`to_3dmm_tensor`, `FLAMELayer(consts, batch_size, flame_path)` with `.faces`, `.faces_tensor`,
`.indices_2d`, `.flame_model`, `forward(flame_params, zero_rot)` -- but the arithmetic of
`FLAMELayer.forward` (flame.py:182-229, i.e. `smplx.lbs.lbs` + offset + 6-DoF rotation) runs in the
HIP library through the C ABI (include/dad3d.h). `decode()` is the raw launch; gradients flow through
`HeadMesh.vertices_3d` / `reprojected_vertices` (autograd.py).
"""
from __future__ import annotations

import ctypes as C
import io
import os
import pickle
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Any, Dict, Optional, Sequence

import numpy as np
import torch
from torch import Tensor

from . import _lib
from .synthetic import load_static

# flame.py:17-26 (note: insertion order there lists rotation before jaw; slicing order is from_3dmm's)
FLAME_CONSTS: Dict[str, int] = {
    "shape": 300,
    "expression": 100,
    "rotation": 6,
    "jaw": 3,
    "eyeballs": 0,
    "neck": 0,
    "translation": 3,
    "scale": 1,
}
_SLICE_ORDER = ("shape", "expression", "jaw", "rotation", "eyeballs", "neck", "translation", "scale")
MAX_SHAPE, MAX_EXPRESSION = 300, 100
MESH_OFFSET_Z = 0.05


@dataclass
class FlameParams:
    """Named views into a `[B, P]` params tensor (flame.py:28-101)."""

    shape: Tensor
    expression: Tensor
    rotation: Tensor
    translation: Tensor
    scale: Tensor
    jaw: Tensor
    eyeballs: Tensor
    neck: Tensor

    @classmethod
    def from_3dmm(cls, tensor_3dmm: Tensor, constants: Dict[str, int], zero_expr: bool = False) -> "FlameParams":
        assert tensor_3dmm.ndim == 2  # flame.py:46
        parts, cur = {}, 0
        for key in _SLICE_ORDER:
            n = int(constants[key])
            parts[key] = tensor_3dmm[:, cur : cur + n]
            cur += n
        if zero_expr:
            parts["expression"] = torch.zeros_like(parts["expression"])
        return cls(**parts)

    def to_3dmm_tensor(self) -> Tensor:
        # flame.py:86-101 concatenates rotation BEFORE jaw (unlike from_3dmm); kept bug-compatible.
        return torch.cat(
            [self.shape, self.expression, self.rotation, self.jaw, self.eyeballs, self.neck, self.translation, self.scale], -1
        )


# ----------------------------------------------------------------------------------------------
# loading a licensed flame.pkl without chumpy (model/utils.py:84-89 `get_flame_model`)
# ----------------------------------------------------------------------------------------------
class _Opaque:
    """Stand-in for classes of modules that are not installed (chumpy): keeps the pickled state."""

    def __init__(self, *a, **k):
        self._args = a

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {"_state": state})


class _TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError):
            return type(name, (_Opaque,), {})


def _as_array(x) -> np.ndarray:
    if hasattr(x, "toarray"):  # scipy sparse (J_regressor)
        return np.asarray(x.toarray())
    if isinstance(x, _Opaque):  # chumpy.Ch keeps its value in `x`
        return np.asarray(x.__dict__.get("x"))
    return np.asarray(x)


def get_flame_model(flame_path: Optional[str] = None) -> SimpleNamespace:
    """Unpickle a FLAME model file into a namespace (`Struct(**pickle.load(...))` in the reference)."""
    flame_path = flame_path or os.environ.get("DAD3D_FLAME_PKL")
    if not flame_path or not os.path.isfile(flame_path):
        raise FileNotFoundError(
            "FLAME model file not found. The reference ships without `model_training/model/static/flame.pkl` "
            "(licence); pass flame_path=..., set DAD3D_FLAME_PKL, or hand a model namespace to HeadMesh(flame_model=...)."
        )
    with open(flame_path, "rb") as f:
        raw = _TolerantUnpickler(io.BytesIO(f.read()), encoding="latin1").load()
    keep = ("f", "v_template", "shapedirs", "posedirs", "J_regressor", "kintree_table", "weights")
    return SimpleNamespace(**{k: _as_array(raw[k]) for k in keep})


class FLAMELayer(torch.nn.Module):
    """`FLAMELayer` whose forward runs on the GPU library.

    Extra to the reference signature: `flame_model` (an already-loaded namespace), `device`, `image_size`
    (the projection lives in the same fused kernel), `static` (index/face assets; defaults to the fixture).
    """

    def __init__(self, consts: Dict[str, Any], batch_size: int = 1, flame_path: Optional[str] = None,
                 flame_model: Any = None, device: Optional[int] = None, image_size: int = 256,
                 static: Optional[dict] = None) -> None:
        super().__init__()
        self.flame_model = flame_model if flame_model is not None else get_flame_model(flame_path)
        self.flame_constants = dict(consts)
        self.batch_size = batch_size
        self.dtype = torch.float32
        m = self.flame_model
        self.faces = np.asarray(m.f)
        self.register_buffer("faces_tensor", torch.tensor(np.asarray(m.f).astype(np.int64), dtype=torch.long), persistent=False)
        st = static if static is not None else load_static()
        self.register_buffer("indices_2d", torch.tensor(st["indices_2d"], dtype=torch.long), persistent=False)  # flame.py:130-131

        f32 = lambda a: np.ascontiguousarray(np.asarray(a), dtype=np.float32)  # noqa: E731
        self._v_template = f32(m.v_template)
        self._shapedirs = f32(m.shapedirs)
        npose = np.asarray(m.posedirs).shape[-1]
        self._posedirs = f32(np.reshape(np.asarray(m.posedirs), [-1, npose]).T)  # flame.py:169-173
        self._j_regressor = f32(_as_array(m.J_regressor))
        parents = np.asarray(m.kintree_table)[0].astype(np.int64)
        parents[0] = -1  # flame.py:176-178
        self._parents = np.ascontiguousarray(parents, dtype=np.int32)
        self._weights = f32(m.weights)
        self.n_verts = int(self._v_template.shape[0])

        lib = _lib.load()
        _lib.require_gpu()
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        model_c = _lib.FlameModelC(
            n_verts=self.n_verts,
            n_betas=int(self._shapedirs.shape[-1]),
            n_joints=int(self._j_regressor.shape[0]),
            v_template=self._v_template.ctypes.data,
            shapedirs=self._shapedirs.ctypes.data,
            posedirs=self._posedirs.ctypes.data,
            j_regressor=self._j_regressor.ctypes.data,
            parents=self._parents.ctypes.data,
            lbs_weights=self._weights.ctypes.data,
        )
        consts_c = _lib.FlameConstsC(**{k: int(consts[k]) for k in _SLICE_ORDER})
        handle = C.c_void_p()
        _lib.check(lib.dad3d_flame_create(C.byref(model_c), C.byref(consts_c), float(image_size), self.device_index, C.byref(handle)))
        self._handle = handle
        self._lib = lib
        self.n_params = lib.dad3d_flame_num_params(handle)
        self.n_landmarks = 0
        self.landmark_indices = np.zeros((0,), dtype=np.int64)

    def __del__(self):
        # plain dict access: nn.Module.__setattr__ is not safe to call while the interpreter shuts down
        h = self.__dict__.get("_handle")
        self.__dict__["_handle"] = None
        if h:
            try:
                self._lib.dad3d_flame_destroy(h)
            except Exception:
                pass

    @property
    def torch_device(self) -> torch.device:
        return torch.device("cuda", self.device_index)

    def fork(self) -> "FLAMELayer":
        """A second layer for ANOTHER stream: shares the model constants on the device (dad3d_flame_fork), owns its
        hand-off buffers and landmark list. One handle serves one stream at a time."""
        handle = C.c_void_p()
        _lib.check(self._lib.dad3d_flame_fork(self._handle, C.byref(handle)))
        twin = object.__new__(type(self))
        twin.__dict__ = {k: (dict(v) if isinstance(v, dict) else v) for k, v in self.__dict__.items()}
        twin.__dict__["_handle"] = handle
        return twin

    def select_kernel(self, which: str = "auto") -> None:
        """Diagnostics / A-B timing: "auto" (default: the pipelined single-role kernel wherever it covers the launch), "two_role"
        (the kernel of rounds 1-3), "pipelined" (raise instead of falling back), "split_bf16" or "split_f16" (the gated exact-product
        splits of the contraction, csrc/flame_decode_split.hip -- three bf16 planes x six products / two fp16 planes x three products:
        not bit-identical to the default, both measured more accurate; raise where the pipelined kernel would).
        dad3d_flame_select_kernel."""
        code = {"auto": _lib.KERNEL_AUTO, "two_role": _lib.KERNEL_TWO_ROLE, "pipelined": _lib.KERNEL_PIPELINED,
                "split_bf16": _lib.KERNEL_SPLIT_BF16, "split_f16": _lib.KERNEL_SPLIT_F16}[which]
        _lib.check(self._lib.dad3d_flame_select_kernel(self._handle, code))

    def decode_tables(self):
        """Device tensors of the backward pass (autograd.DecodeTables), built on first use and shared with forks."""
        t = self.__dict__.get("_decode_tables")
        if t is None:
            from .autograd import DecodeTables

            t = DecodeTables.from_layer(self)
            self.__dict__["_decode_tables"] = t
        return t

    def set_landmarks(self, indices: Sequence[int]) -> None:
        idx = np.ascontiguousarray(np.asarray(indices, dtype=np.int64))
        _lib.check(self._lib.dad3d_flame_set_landmarks(self._handle, idx.ctypes.data, int(idx.size)))
        self.n_landmarks = int(idx.size)
        self.landmark_indices = idx

    # ------------------------------------------------------------------------------------------
    def decode(self, params: Tensor, *, verts3d: bool = False, proj: bool = False, to_2d: bool = True,
               landmarks: bool = False, landmarks_px: bool = False, zero_rot: bool = False, flip_z: bool = False,
               mutate: bool = False, out: Optional[Dict[str, Tensor]] = None) -> Dict[str, Tensor]:
        """One fused launch producing any subset of the outputs; `params` must be a CUDA fp32 [B,P] tensor
        on this layer's device (contiguous). Async on torch's current stream.
        `self.compat_cross_b3` (default False): reproduce the reference's batch-of-exactly-three result, where
        `torch.cross` without `dim` (model/utils.py:98-99) crosses over the batch axis (DAD3D_COMPAT_CROSS_B3)."""
        if params.ndim != 2:
            raise AssertionError("tensor_3dmm.ndim == 2 expected")  # flame.py:46
        if params.shape[1] != self.n_params:
            raise ValueError(f"expected {self.n_params} params per row, got {params.shape[1]}")
        if params.requires_grad and torch.is_grad_enabled():
            raise RuntimeError("FLAMELayer.decode is the raw launch (no grad_fn); differentiate through "
                               "HeadMesh.vertices_3d / reprojected_vertices or autograd.decode_with_grad")
        if params.device != self.torch_device or params.dtype != torch.float32 or not params.is_contiguous():
            raise ValueError("params must be a contiguous float32 tensor on " + str(self.torch_device))
        b, v = params.shape[0], self.n_verts
        res: Dict[str, Tensor] = {} if out is None else out
        dev = params.device

        def buf(key, shape, dtype=torch.float32):
            # a tensor left in `out` by an earlier call is reused only if the kernel can write all of this call's
            # rows into it: same shape, dtype, device, contiguous -- anything else would be written out of bounds
            t = res.get(key)
            if t is not None and (tuple(t.shape) != tuple(shape) or t.dtype != dtype or t.device != dev or not t.is_contiguous()):
                raise ValueError(f"out[{key!r}]: expected a contiguous {dtype} tensor {tuple(shape)} on {dev}, "
                                 f"got {t.dtype} {tuple(t.shape)} on {t.device}")
            if t is None:
                t = torch.empty(shape, dtype=dtype, device=dev)
                res[key] = t
            return t

        p_v = buf("verts3d", (b, v, 3)).data_ptr() if verts3d else None
        p_p = buf("proj", (b, v, 2 if to_2d else 3)).data_ptr() if proj else None
        p_lx = buf("lmk_xy", (b, self.n_landmarks, 2)).data_ptr() if landmarks else None
        p_lp = buf("lmk_px", (b, self.n_landmarks, 2), torch.int32).data_ptr() if landmarks_px else None
        flags = (_lib.ZERO_ROTATION if zero_rot else 0) | (_lib.TO_2D if to_2d else 0) | \
            (_lib.MUTATE_PARAMS if mutate else 0) | (_lib.FLIP_Z if flip_z else 0) | \
            (_lib.COMPAT_CROSS_B3 if getattr(self, "compat_cross_b3", False) else 0)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(self._lib.dad3d_flame_decode(self._handle, params.data_ptr(), b, flags, p_v, p_p, p_lx, p_lp, stream))
        return res

    def forward(self, flame_params: FlameParams, zero_rot: bool = False, zero_jaw: bool = False) -> Tensor:
        """flame.py:182-229 -> vertices [B,V,3]. Accepts the views `from_3dmm` produced (re-assembled into
        one params matrix); output lives on the device of the inputs."""
        jaw = torch.zeros_like(flame_params.jaw) if zero_jaw else flame_params.jaw
        packed = torch.cat(
            [flame_params.shape, flame_params.expression, jaw, flame_params.rotation, flame_params.eyeballs,
             flame_params.neck, flame_params.translation, flame_params.scale], dim=-1)
        src = packed.device
        if torch.is_grad_enabled() and packed.requires_grad:  # training callers: same launch, grad_fn attached
            from .autograd import decode_with_grad

            return decode_with_grad(self, packed, verts3d=True, proj=False, zero_rot=zero_rot)[0]
        with torch.no_grad():
            dev_params = packed.detach().to(self.torch_device, torch.float32).contiguous()
            verts = self.decode(dev_params, verts3d=True, zero_rot=zero_rot)["verts3d"]
        return verts.to(src)


def uint8_to_float32(x: Tensor) -> Tensor:  # flame.py:232-236
    return x.div(255.0).to(dtype=torch.float32) if x.dtype == torch.uint8 else x
