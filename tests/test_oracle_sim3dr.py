"""CPU: the Sim3DR C restatement (oracle/sim3dr_port.c) against the reference's known answers, against the
goldens produced by the reference's own C++, and -- when it could be built here -- against that C++ live."""
import numpy as np
import pytest

from oracle import sim3dr_ref
from oracle.sim3dr_ref import Sim3DROracle


def test_known_answers_from_reference_tests(port_oracle, sim3dr_golden):
    # Sim3DR/tests/test.cpp:10-48 inputs; expected values in SURVEY.md section 4
    assert port_oracle.point_in_tri((0.2, 0.2), (0, 0), (1, 0), (1, 1)) is True
    assert np.allclose(port_oracle.point_weight((0.2, 0.2), (0, 0), (1, 0), (1, 1)), [0.8, 0.0, 0.2], atol=1e-7)
    v = np.array([[1, 1.1, 0], [0, 0, 0], [0, 0.6, 0.7]], np.float32)
    t = np.array([[0, 1, 2]], np.int32)
    assert np.allclose(port_oracle.get_tri_normal(v, t), [[-0.77, 0.7, -0.6]], atol=1e-6)
    assert np.allclose(port_oracle.get_tri_normal(v, t, True), [[-0.6410215, 0.5827468, -0.4994973]], atol=1e-7)
    assert np.array_equal(port_oracle.point_weight((0.2, 0.2), (0, 0), (1, 0), (1, 1)), sim3dr_golden["ka_weight"])
    assert np.array_equal(port_oracle.get_tri_normal(v, t, True), sim3dr_golden["ka_tri_normal_unit"])


def test_single_triangle_staircase(port_oracle, sim3dr_golden):
    tv = np.array([[1, 1, 0.5], [6, 1, 0.5], [1, 6, 0.5]], np.float32)
    img = port_oracle.rasterize(tv, np.array([[0, 1, 2]], np.int32), np.ones((3, 3), np.float32), height=8, width=8, channel=3)
    assert np.array_equal(img, sim3dr_golden["tri8_image"])
    cov = img[..., 0] > 0
    # strictly interior: x>1, y>1, x+y<7; pixels exactly on the hypotenuse (x+y==7) depend on fp32 rounding of
    # 1-u-v (the reference's own result, frozen in the golden, covers them), so only the rest is asserted here
    for y in range(8):
        for x in range(8):
            if x + y != 7:
                assert cov[y, x] == (x > 1 and y > 1 and x + y < 7)


def _head_inputs(static, decode_golden):
    verts = np.ascontiguousarray(decode_golden["b2_proj3"][0]).copy()
    verts[:, 2] *= -1
    return verts, static["faces"]


def test_port_bitwise_equals_reference_golden_head(port_oracle, sim3dr_golden, decode_golden, static):
    verts, faces = _head_inputs(static, decode_golden)
    n = port_oracle.get_normal(verts, faces)
    assert np.array_equal(n, sim3dr_golden["head_normals"])
    assert np.array_equal(port_oracle.get_tri_normal(verts, faces, True), sim3dr_golden["head_tri_normals_unit"])
    col = np.clip(n * 0.5 + 0.5, 0, 1).astype(np.float32)
    img, depth = port_oracle.rasterize(verts, faces, col, height=256, width=256, channel=3, return_depth=True)
    assert np.array_equal(img, sim3dr_golden["head_image"]) and np.array_equal(depth, sim3dr_golden["head_depth"])
    rev = port_oracle.rasterize(verts, faces, col, height=256, width=256, channel=3, reverse=True)
    assert np.array_equal(rev, sim3dr_golden["head_image_reverse"]) and np.array_equal(rev, img[::-1])
    d, tb, bw = port_oracle.rasterize_triangles(verts, faces, 256, 256)
    assert np.array_equal(tb, sim3dr_golden["head_tri_buf"]) and np.array_equal(bw, sim3dr_golden["head_bary"])
    assert np.array_equal(d, sim3dr_golden["head_depth_tri"])
    pncc = port_oracle.rasterize(verts, static["faces_wo_ears"], sim3dr_golden["pncc_colors"], bg=np.zeros((256, 256, 3), np.uint8))
    assert np.array_equal(pncc, sim3dr_golden["pncc_image"])


def test_port_bitwise_equals_reference_golden_soup(port_oracle, sim3dr_golden):
    g = sim3dr_golden
    v, t, col = g["soup_vertices"], g["soup_triangles"], g["soup_colors"]
    img, dep = port_oracle.rasterize(v, t, col, bg=g["soup_bg"].copy(), depth=g["soup_depth_in"].copy(), return_depth=True)
    assert np.array_equal(img, g["soup_image"]) and np.array_equal(dep, g["soup_depth"])
    d, tb, bw = port_oracle.rasterize_triangles(v, t, 48, 64, depth=g["soup_depth_in"].copy())
    assert np.array_equal(tb, g["soup_tri_buf"]) and np.array_equal(bw, g["soup_bary"]) and np.array_equal(d, g["soup_depth_tri"])
    assert np.array_equal(port_oracle.get_normal(v, t), g["soup_normals"])
    assert np.array_equal(port_oracle.get_normal(v, t, init=g["soup_normal_init"]), g["soup_normals_accum"])


@pytest.mark.skipif(not sim3dr_ref.available("reference"), reason="oracle/_ref/libsim3dr_ref.so not built (no reference tree)")
def test_port_bitwise_equals_live_reference_random():
    R, P = Sim3DROracle("reference"), Sim3DROracle("port")
    rng = np.random.default_rng(0)
    for trial in range(20):
        nver, ntri = int(rng.integers(3, 60)), int(rng.integers(1, 120))
        h, w, c = int(rng.integers(1, 40)), int(rng.integers(1, 40)), int(rng.integers(1, 5))
        v = rng.uniform(-5, 45, (nver, 3)).astype(np.float32)
        if trial % 3 == 0:
            v[:, 2] = np.round(v[:, 2] / 10)
        if trial % 4 == 0:
            v[:, :2] = np.round(v[:, :2])
        t = rng.integers(0, nver, (ntri, 3)).astype(np.int32)
        col = rng.uniform(0, 1, (nver, c)).astype(np.float32)
        bg = rng.integers(0, 255, (h, w, c)).astype(np.uint8)
        rev = bool(trial % 2)
        a = R.rasterize(v, t, col, bg=bg.copy(), reverse=rev, return_depth=True)
        b = P.rasterize(v, t, col, bg=bg.copy(), reverse=rev, return_depth=True)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        for x, y in zip(R.rasterize_triangles(v, t, h, w), P.rasterize_triangles(v, t, h, w)):
            assert np.array_equal(x, y)
        assert np.array_equal(R.get_normal(v, t), P.get_normal(v, t))
        tn = R.get_tri_normal(v, t, bool(trial % 2))
        assert np.array_equal(tn, P.get_tri_normal(v, t, bool(trial % 2)))
        assert np.array_equal(R.get_ver_normal(tn, t, nver), P.get_ver_normal(tn, t, nver))
    # alpha != 1 is order dependent in the reference; the port reproduces the serial semantics
    a = R.rasterize(v, t, col, bg=bg.copy(), alpha=0.4)
    b = P.rasterize(v, t, col, bg=bg.copy(), alpha=0.4)
    assert np.array_equal(a, b)


def test_empty_and_offscreen(port_oracle):
    v = np.array([[-5, -5, 1], [-1, -5, 1], [-5, -1, 1]], np.float32)
    t = np.array([[0, 1, 2]], np.int32)
    bg = np.full((4, 4, 3), 7, np.uint8)
    assert np.array_equal(port_oracle.rasterize(v, t, np.ones((3, 3), np.float32), bg=bg.copy()), bg)
    # unreferenced vertices get the (0,0,0)/1e-6 normal = 0
    n = port_oracle.get_normal(np.zeros((5, 3), np.float32), np.array([[0, 1, 2]], np.int32))
    assert np.array_equal(n, np.zeros((5, 3), np.float32))


def test_precast_diagnostic_is_the_port_raster_plus_floats(static, decode_golden, port_oracle):
    """`port_rasterize_precast` (tests/render_checks.py builds on it) writes the same bytes as `port_rasterize` -- which is
    pinned to the reference's C++ above -- and every byte is the x86 cast of the float it reports; the byte checker accepts
    a light that is off by 1.5e-5 and rejects one that is off by 4e-4."""
    from oracle.sim3dr_ref import render_pipeline_ref
    from render_checks import assert_render_bytes_explained

    faces = static["faces"]
    v = decode_golden["b2_proj3"][0].copy()
    v[:, 2] *= -1.0
    ref, light = render_pipeline_ref(port_oracle, v.copy(), faces, np.zeros((256, 256, 3), np.uint8))
    img, pre, drawn = port_oracle.rasterize_precast(v, faces, light, 256, 256, 3)
    assert np.array_equal(img, ref) and drawn.sum() > 10000
    assert np.array_equal(pre[drawn].astype(np.int32).astype(np.uint8), ref[drawn])  # cvttss2si, low 8 bits
    assert assert_render_bytes_explained(ref.copy(), ref, port_oracle, v, faces, light) == 0
    near = port_oracle.rasterize(v.copy(), faces, np.clip(light + np.float32(1.5e-5), 0, 1), bg=np.zeros((256, 256, 3), np.uint8))
    assert 0 < assert_render_bytes_explained(near, ref, port_oracle, v, faces, light) < 500
    far = port_oracle.rasterize(v.copy(), faces, np.clip(light + np.float32(4e-4), 0, 1), bg=np.zeros((256, 256, 3), np.uint8))
    with pytest.raises(AssertionError):
        assert_render_bytes_explained(far, ref, port_oracle, v, faces, light)
