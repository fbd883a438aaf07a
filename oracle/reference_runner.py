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


def _pytorchcv_resnet50_features():
    """Stand-in for `pytorchcv.model_provider.get_model("resnet50").features` (encoders.py:5,22; the package is not
    installed here, pinned by the reference's requirements as a third-party dependency): a module tree with pytorchcv's
    published resnet50 layout and NAMES -- `init_block.conv.{conv,bn}`, `stage{1..4}.unit{k}.body.conv{1,2,3}.{conv,bn}`,
    `unit1.identity_conv.{conv,bn}`, stride on conv1 of a unit's bottleneck (`conv1_stride=True`) -- written from the
    architecture, random-initialised. It exists so that the reference's own FlameRegression / BiFPN / heads can be
    executed and so that a state dict with the checkpoint's key names exists; it pins nothing about pytorchcv itself."""
    from torch import nn

    class ConvBlock(nn.Module):
        def __init__(self, cin, cout, k, stride=1, activ=True):
            super().__init__()
            self.conv = nn.Conv2d(cin, cout, k, stride, k // 2, bias=False)
            self.bn = nn.BatchNorm2d(cout)
            self.activ = nn.ReLU(inplace=True) if activ else None

        def forward(self, x):
            x = self.bn(self.conv(x))
            return x if self.activ is None else self.activ(x)

    class ResBottleneck(nn.Module):
        def __init__(self, cin, cout, stride):
            super().__init__()
            mid = cout // 4
            self.conv1 = ConvBlock(cin, mid, 1, stride)
            self.conv2 = ConvBlock(mid, mid, 3)
            self.conv3 = ConvBlock(mid, cout, 1, activ=False)

        def forward(self, x):
            return self.conv3(self.conv2(self.conv1(x)))

    class ResUnit(nn.Module):
        def __init__(self, cin, cout, stride):
            super().__init__()
            self.resize_identity = cin != cout or stride != 1
            self.body = ResBottleneck(cin, cout, stride)
            if self.resize_identity:
                self.identity_conv = ConvBlock(cin, cout, 1, stride, activ=False)
            self.activ = nn.ReLU(inplace=True)

        def forward(self, x):
            identity = self.identity_conv(x) if self.resize_identity else x
            return self.activ(self.body(x) + identity)

    class ResInitBlock(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = ConvBlock(3, 64, 7, 2)
            self.pool = nn.MaxPool2d(3, 2, 1)

        def forward(self, x):
            return self.pool(self.conv(x))

    features = nn.Sequential()
    features.add_module("init_block", ResInitBlock())
    cin = 64
    for i, (cout, units) in enumerate(((256, 3), (512, 4), (1024, 6), (2048, 3))):
        stage = nn.Sequential()
        for j in range(units):
            stage.add_module(f"unit{j + 1}", ResUnit(cin, cout, 2 if (j == 0 and i != 0) else 1))
            cin = cout
        features.add_module(f"stage{i + 1}", stage)
    return features


def load_reference_regressor(seed: int = 0, num_classes: int = 68):
    """The reference's own `FlameRegression` (model_training/model/flame_regression.py:62-105, with its bifpn.py,
    layers.py and encoders.py, unmodified) on the resnet50 configuration of config/model/resnet_regression.yaml,
    random-initialised from `seed`. Stand-ins: `pytorchcv.model_provider.get_model` (see above),
    `pytorch_toolbelt.modules` (imported by layers.py:8, used only by heads this model does not build), `hydra`."""
    if not reference_available():
        raise FileNotFoundError(f"reference tree not found at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    if "pytorch_toolbelt.modules" not in sys.modules:
        mods = types.ModuleType("pytorch_toolbelt.modules")
        sys.modules["pytorch_toolbelt.modules"] = mods
        sys.modules["pytorch_toolbelt"].modules = mods
    if "pytorchcv" not in sys.modules:
        cv = types.ModuleType("pytorchcv")
        provider = types.ModuleType("pytorchcv.model_provider")
        cv.model_provider = provider
        sys.modules["pytorchcv"], sys.modules["pytorchcv.model_provider"] = cv, provider

        def get_model(name, pretrained=False, **kwargs):
            if name != "resnet50":
                raise ValueError(f"stand-in only declares resnet50, not {name}")
            return types.SimpleNamespace(features=_pytorchcv_resnet50_features())

        provider.get_model = get_model
    if "model_training.data" not in sys.modules:  # its __init__ pulls in the datasets (albumentations, cv2): bypassed,
        pkg = types.ModuleType("model_training.data")  # only data/config.py (string constants) is needed
        pkg.__path__ = [os.path.join(REFERENCE_ROOT, "model_training", "data")]
        sys.modules["model_training.data"] = pkg
    from model_training.model.flame_regression import FlameRegression  # noqa: E402  (reference code)

    state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    try:
        cfg = {"backbone": "resnet50", "pretrained": False, "num_filters": 256, "num_channels": 3,
               "num_classes": num_classes, "img_size": 256, "conv_block": "regular", "limit_value": 3}
        return FlameRegression(cfg, {}, num_classes=num_classes)
    finally:
        torch.random.set_rng_state(state)
