"""The reference's OWN Cython binding (`Sim3DR/lib/rasterize.pyx`, module "Sim3DR_Cython", Sim3DR/setup.py:12-18) built
by oracle/Makefile against libdad3d_hip.so instead of rasterize_kernel.cpp -- the drop-in of SURVEY 8(b), linked for real:
its five C++ prototypes resolve to the mangled symbols sim3dr_compat.cpp exports. Every call below goes numpy -> the
reference's typed-buffer prologue (rasterize.pyx:44-102) -> the C ABI -> HIP kernels -> back, and is held to the goldens
produced by the reference's own C++ (tests/golden/sim3dr_golden.npz), bit for bit.

The module is built in the authoring container (`__graft_entry__.build()` -> `make -C oracle`, needs /root/reference for the
.pyx) and travels to the GPU box as a built file under oracle/_ref/hip/ like the other oracle libraries."""
import glob
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load():
    hits = glob.glob(os.path.join(ROOT, "oracle", "_ref", "hip", "Sim3DR_Cython*.so"))
    if not hits:
        pytest.skip("oracle/_ref/hip/Sim3DR_Cython*.so not built (needs the reference checkout: make -C oracle cython_hip)")
    spec = importlib.util.spec_from_file_location("Sim3DR_Cython", hits[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_binding_links_against_the_hip_library_and_exposes_the_reference_surface():
    """CPU: the module imports (its NEEDED libdad3d_hip.so resolves through the RUNPATH) and has the five entry points."""
    mod = _load()
    for name in ("get_tri_normal", "get_ver_normal", "get_normal", "rasterize_triangles", "rasterize"):
        assert callable(getattr(mod, name))
    import subprocess

    so = glob.glob(os.path.join(ROOT, "oracle", "_ref", "hip", "Sim3DR_Cython*.so"))[0]
    dyn = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
    assert "libdad3d_hip.so" in dyn
    und = subprocess.run(["nm", "-D", "--undefined-only", so], capture_output=True, text=True).stdout
    for sym in ("_Z10_rasterizePhPfPiS0_S0_iiiifb", "_Z11_get_normalPfS_Piii", "_Z15_get_tri_normalPfS_Piib",
                "_Z15_get_ver_normalPfS_Piii", "_Z20_rasterize_trianglesPfPiS_S0_S_iii"):
        assert sym in und  # resolved by libdad3d_hip.so at load time, not compiled in


@pytest.mark.gpu
def test_reference_cython_prologue_drives_the_gpu_bit_exact(static, decode_golden, sim3dr_golden):
    mod, g = _load(), sim3dr_golden
    verts = np.ascontiguousarray(decode_golden["b2_proj3"][0]).copy()
    verts[:, 2] *= -1
    faces = static["faces"]
    nver, ntri = verts.shape[0], faces.shape[0]
    # Sim3DR.py:8-12 get_normal
    normal = np.zeros_like(verts, dtype=np.float32)
    mod.get_normal(normal, verts, faces, nver, ntri)
    assert np.array_equal(normal, g["head_normals"])
    # the two halves (rasterize.pyx:44-63)
    tn = np.zeros((ntri, 3), np.float32)
    mod.get_tri_normal(tn, verts, faces, ntri, True)
    assert np.array_equal(tn, g["head_tri_normals_unit"])
    tn_raw = np.zeros((ntri, 3), np.float32)
    mod.get_tri_normal(tn_raw, verts, faces, ntri, False)
    vn = np.zeros_like(verts)
    mod.get_ver_normal(vn, tn_raw, faces, nver, ntri)
    assert np.array_equal(vn, g["head_normals"])
    # Sim3DR.py:15-29 rasterize (bg zeros, depth -1e8), both orientations
    col = np.clip(normal * 0.5 + 0.5, 0, 1).astype(np.float32)
    for reverse, key in ((False, "head_image"), (True, "head_image_reverse")):
        img = np.zeros((256, 256, 3), np.uint8)
        depth = np.zeros((256, 256), np.float32) - 1e8
        mod.rasterize(img, verts, faces, col, depth, ntri, 256, 256, 3, 1.0, reverse)
        assert np.array_equal(img, g[key])
        if not reverse:
            assert np.array_equal(depth, g["head_depth"])
    d = np.zeros((256, 256), np.float32) - 1e8
    tb = np.zeros((256, 256), np.int32) - 1
    bw = np.zeros((256, 256, 3), np.float32).reshape(256, -1)  # the prologue types it ndim=2
    mod.rasterize_triangles(verts, faces, d, tb, bw, ntri, 256, 256)
    assert np.array_equal(tb, g["head_tri_buf"]) and np.array_equal(bw.reshape(256, 256, 3), g["head_bary"])
    assert np.array_equal(d, g["head_depth_tri"])
    # the "soup" golden: 4 channels, pre-filled background and depth, off-screen and degenerate triangles, accumulate path
    sv, stri = g["soup_vertices"], g["soup_triangles"]
    img, depth = g["soup_bg"].copy(), g["soup_depth_in"].copy()
    mod.rasterize(img, sv, stri, g["soup_colors"], depth, stri.shape[0], 48, 64, 4, 1.0, False)
    assert np.array_equal(img, g["soup_image"]) and np.array_equal(depth, g["soup_depth"])
    acc = g["soup_normal_init"].copy()
    mod.get_normal(acc, sv, stri, sv.shape[0], stri.shape[0])
    assert np.array_equal(acc, g["soup_normals_accum"])
    # the prologue's own type checks are the reference's (typed memoryview buffers)
    with pytest.raises(ValueError, match="Buffer dtype mismatch"):
        mod.get_normal(np.zeros_like(verts), verts.astype(np.float64), faces, nver, ntri)
    with pytest.raises((ValueError, TypeError)):
        mod.get_normal(np.zeros_like(verts), verts, faces.astype(np.int64), nver, ntri)
