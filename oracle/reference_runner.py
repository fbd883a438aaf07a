"""Run the reference's OWN `model_training/head_mesh.py` unmodified  --  TEST INFRASTRUCTURE.

Authoring-container only (needs /root/reference; the GPU box has none). Used by
tests/golden/make_decode_golden.py to generate the committed fixtures and by
tests/test_oracle_flame.py::test_oracle_bitwise_equals_live_reference (auto-skipped when the reference tree is absent).

The reference cannot be imported as-is here: `hydra`, `smplx`,
`pytorch_toolbelt` are not installed and `static/flame.pkl` is missing. This module registers
minimal `sys.modules` stand-ins for exactly those imports:

    hydra.utils.instantiate        (model_training/model/__init__.py:1)  -> never called on this path
    pytorch_toolbelt.utils         (model_training/model/utils.py:12)    -> never called on this path
    smplx.utils.{Struct,to_tensor,to_np}  (flame.py:6, model/utils.py:2) -> 3 tiny helpers re-stated
    smplx.lbs.lbs                  (flame.py:5)                          -> oracle.flame_ref.lbs

and patches `model_training.model.flame.get_flame_model` to hand back the seeded synthetic model.
Everything else -- `HeadMesh`, `FLAMELayer`, `FlameParams`, `rot_mat_from_6dof` -- is the reference's
code, byte for byte, executed from where it lies. Nothing is copied into this repository.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

REFERENCE_ROOT = os.environ.get("DAD3D_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "model_training", "head_mesh.py"))


def _install_stubs():
    from . import flame_ref

    def _mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    if "hydra" not in sys.modules:
        hydra = _mod("hydra")
        hydra.utils = _mod("hydra.utils")
        hydra.utils.instantiate = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stub"))
    if "pytorch_toolbelt" not in sys.modules:
        ptb = _mod("pytorch_toolbelt")
        ptb.utils = _mod("pytorch_toolbelt.utils")
    if "smplx" not in sys.modules:
        smplx = _mod("smplx")
        su = _mod("smplx.utils")
        sl = _mod("smplx.lbs")
        smplx.utils, smplx.lbs = su, sl

        class Struct:
            def __init__(self, **kw):
                for k, v in kw.items():
                    setattr(self, k, v)

        def to_tensor(array, dtype=torch.float32):
            return (array if torch.is_tensor(array) else torch.tensor(array)).to(dtype)

        def to_np(array, dtype=np.float32):
            if "scipy.sparse" in str(type(array)):
                array = array.todense()
            return np.array(array, dtype=dtype)

        su.Struct, su.to_tensor, su.to_np = Struct, to_tensor, to_np
        sl.lbs = lambda *a, **k: flame_ref.lbs(*a, **k)  # noqa: E731


def load_reference_head_mesh(model, flame_config=None, image_size: int = 256):
    """Return an instance of the reference's HeadMesh driven by `model` (a FLAME-shaped namespace)."""
    if not reference_available():
        raise FileNotFoundError(f"reference tree not found at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True  # the reference tree is read-only
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import model_training.model.flame as ref_flame  # noqa: E402  (reference code)
    from model_training.head_mesh import HeadMesh  # noqa: E402  (reference code)

    Struct = sys.modules["smplx.utils"].Struct
    fields = {k: getattr(model, k) for k in ("f", "v_template", "shapedirs", "posedirs", "J_regressor", "kintree_table", "weights")}
    ref_flame.get_flame_model = lambda flame_path=None: Struct(**fields)
    with torch.no_grad():
        return HeadMesh(flame_config=flame_config, image_size=image_size)


def load_reference_losses(model):
    """The reference's own `Vertices3DLoss` / `ReprojectionLoss` classes (model_training/losses/*.py, unmodified),
    wired to `model` like `load_reference_head_mesh`. The package `__init__` (which pulls in unrelated losses with more
    uninstalled imports) is bypassed by registering an empty package module; `model_training/utils.py` gets stand-ins
    for `omegaconf`, `coloredlogs` and `hydra.utils.get_original_cwd`, none of which the loss path calls."""
    load_reference_head_mesh(model)  # stubs, sys.path and the patched get_flame_model

    def _mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    if "omegaconf" not in sys.modules:
        oc = _mod("omegaconf")
        oc.OmegaConf, oc.DictConfig = type("OmegaConf", (), {}), dict
    if "coloredlogs" not in sys.modules:
        cl = _mod("coloredlogs")
        cl.DEFAULT_FIELD_STYLES, cl.install = {}, (lambda *a, **k: None)
    if not hasattr(sys.modules["hydra.utils"], "get_original_cwd"):
        sys.modules["hydra.utils"].get_original_cwd = os.getcwd
    if "model_training.losses" not in sys.modules:
        pkg = _mod("model_training.losses")
        pkg.__path__ = [os.path.join(REFERENCE_ROOT, "model_training", "losses")]
    import importlib

    v3d = importlib.import_module("model_training.losses.vertices_3d_loss")
    rep = importlib.import_module("model_training.losses.reprojection_loss")
    return v3d.Vertices3DLoss, rep.ReprojectionLoss
