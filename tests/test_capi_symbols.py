"""CPU: libdad3d_hip.so loads without a GPU and exports every function include/dad3d.h declares
(no compute calls here). Also the C++-linkage doubles of the reference's Sim3DR prototypes."""
import ctypes
import os
import re
import subprocess

import pytest

from dad_3dheads_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "dad3d.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dad3d_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built():
    assert os.path.isfile(_lib.LIB_PATH), "run `python __graft_entry__.py build`"


def test_every_declared_symbol_is_exported_and_bound():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/dad3d.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in _lib.py"
    assert set(_lib.SIGNATURES) == set(names)


def test_reference_cpp_prototypes_are_exported():
    out = subprocess.run(["nm", "-D", "--defined-only", "-C", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    for proto in ("_get_tri_normal(float*, float*, int*, int, bool)", "_get_ver_normal(float*, float*, int*, int, int)",
                  "_get_normal(float*, float*, int*, int, int)",
                  "_rasterize_triangles(float*, int*, float*, int*, float*, int, int, int)",
                  "_rasterize(unsigned char*, float*, int*, float*, float*, int, int, int, int, float, bool)"):
        assert proto in out, proto


def test_nothing_else_is_exported():
    """The dynamic symbol table is the C ABI and the five Sim3DR doubles, nothing else (-fvisibility=hidden + csrc/exports.map):
    no launcher, kernel handle, libstdc++ instantiation or __hip_cuid_* leaks out of a drop-in library."""
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = {line.split()[-1] for line in out.splitlines() if line.strip()}
    five = {"_Z15_get_tri_normalPfS_Piib", "_Z15_get_ver_normalPfS_Piii", "_Z11_get_normalPfS_Piii",
            "_Z20_rasterize_trianglesPfPiS_S0_S_iii", "_Z10_rasterizePhPfPiS0_S0_iiiifb"}
    assert five <= names
    assert names - five == set(declared_functions()), sorted(names - five - set(declared_functions()))


def test_host_only_calls_work_without_gpu():
    lib = _lib.load()
    assert lib.dad3d_version() == 100
    assert lib.dad3d_device_count() >= 0
    lib.dad3d_clear_error()
    assert lib.dad3d_last_error() == b""
    # argument validation happens before any device work
    assert lib.dad3d_flame_create(None, None, 256.0, 0, None) == _lib.E_INVALID
    assert b"null" in lib.dad3d_last_error()


def test_product_fails_loudly_without_library(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "missing.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "dad-3dheads_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                src = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dirpath, fn)
                assert "oracle/" not in src or fn.endswith(".md"), os.path.join(dirpath, fn)


def test_rccl_binding_resolves_its_entry_points():
    """dad_3dheads_amd/rccl.py binds RCCL's C API by name from the librccl.so PyTorch-ROCm ships (no GPU needed to look)."""
    from dad_3dheads_amd import rccl

    lib = rccl._load()
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclAllGather", "ncclCommDestroy", "ncclGetErrorString"):
        assert hasattr(lib, name), name
    assert ctypes.sizeof(rccl._UniqueId) == 128


def test_python_constants_mirror_the_header():
    """The flag / kernel-choice / dtype numbers `_lib.py` passes through ctypes are the header's `#define`s (a new constant on one side only
    would select another kernel without an error)."""
    text = open(os.path.join(ROOT, "include", "dad3d.h")).read()
    defines = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"#define\s+DAD3D_([A-Z0-9_]+)\s+(0x[0-9a-fA-F]+|\d+)u?\b", text)}
    checked = 0
    for name, value in vars(_lib).items():
        if name.isupper() and isinstance(value, int) and name in defines:
            assert defines[name] == value, (name, defines[name], value)
            checked += 1
    assert checked >= 10, checked
    for must in ("KERNEL_AUTO", "KERNEL_TWO_ROLE", "KERNEL_PIPELINED", "KERNEL_SPLIT_BF16", "KERNEL_SPLIT_F16", "TO_2D", "MUTATE_PARAMS", "FLIP_Z"):
        assert must in defines and getattr(_lib, must) == defines[must], must
