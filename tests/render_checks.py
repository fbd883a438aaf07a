"""Shared checker for `RenderPipeline` byte output.

`Sim3DR/lighting.py:59` raises to `specular_exp` with numpy's float32 `**`, which goes through the host's libm or SVML
(np.power has exact fast paths for exponents 1 and 2 only): the reference's own bytes differ between hosts by that one
function, so they cannot be reproduced bit for bit on a GPU. What CAN be shown, and is shown here:

* with `specular_exp` 1 or 2 (np.power == multiplication, the kernel uses the same products) the light and the rendered
  bytes are bit-identical to the numpy restatement -- every operation of the pipeline except pow;
* with the default exponent the per-vertex light is within 2e-5 of numpy on this host, coverage is identical, and EVERY
  differing byte is explained by truncation: the float the oracle cast that byte from (`alpha * 255 * p_color`,
  rasterize_kernel.cpp:276-281, recorded by the port's `port_rasterize_precast`) lies within 255 * 2e-5 (+ slack) of an
  integer -- `(unsigned char)` flips there and nowhere else.
"""
import numpy as np


def assert_render_bytes_explained(img_gpu, img_ref, oracle, vertices, triangles, light_ref, light_tol=2e-5):
    """img_* [h,w,3] uint8 rendered onto black. Coverage identical, |byte difference| <= 1, and every differing byte sits
    on a truncation boundary of the ORACLE's own pre-cast float. Returns the number of differing bytes."""
    assert img_gpu.shape == img_ref.shape and img_gpu.dtype == np.uint8
    h, w, c = img_ref.shape
    img_chk, pre, drawn = oracle.rasterize_precast(vertices, triangles, light_ref, h, w, c)
    assert np.array_equal(img_chk, img_ref)  # the diagnostic raster IS the oracle's raster, plus the floats
    assert not img_gpu[~drawn].any()         # nothing drawn outside the oracle's coverage
    diff = img_gpu.astype(int) - img_ref.astype(int)
    assert np.abs(diff).max() <= 1
    bad = diff != 0
    assert not (bad & ~drawn[..., None]).any()
    if bad.any():
        x = pre[bad].astype(np.float64)
        dist = np.abs(x - np.round(x))
        assert dist.max() < 255.0 * light_tol + 1e-3, f"{int(bad.sum())} bytes differ, farthest from a truncation boundary: {dist.max():.4f}"
    return int(bad.sum())
