"""GPU: `_rasterize` with alpha != 1 (rasterize_kernel.cpp:268-284) bit for bit against the reference's compiled C++.

The reference blends every fragment that passes the RUNNING depth test, in triangle-index order:
`(unsigned char)((1 - alpha) * old + alpha * 255 * colour)`. `raster_blend_kernel` replays that chain per pixel, one link per pass.
Unreachable from the reference's Python (Sim3DR.py:27-28 passes alpha = 1), exposed by the Cython `def` (rasterize.pyx:95) and the
C++ prototype (rasterize.h:98-100) -- hence tested through both the C ABI and the reference's own binding."""
import numpy as np
import pytest
import torch

from dad_3dheads_amd import _lib
from dad_3dheads_amd.Sim3DR import Mesh

pytestmark = pytest.mark.gpu
ALPHAS = (0.3, 0.5, 0.75)


def _check(oracle, v, t, col, bg, alpha, reverse, depth0=None, tag=""):
    h, w, _ = bg.shape
    d0 = np.full((h, w), -1e8, np.float32) if depth0 is None else depth0
    mesh = Mesh(t, v.shape[0], device=0)
    img = torch.from_numpy(bg.copy()).cuda()[None].contiguous()
    depth = torch.from_numpy(d0.copy()).cuda()[None].contiguous()
    mesh.rasterize(torch.from_numpy(v).cuda()[None], torch.from_numpy(col).cuda()[None], img, depth=depth, reverse=reverse, alpha=alpha)
    with np.errstate(all="ignore"):
        ref_img, ref_depth = oracle.rasterize(v, t, col, bg=bg.copy(), reverse=reverse, alpha=alpha, depth=d0.copy(), return_depth=True)
    assert np.array_equal(img[0].cpu().numpy(), ref_img), (tag, alpha)
    assert np.array_equal(depth[0].cpu().numpy(), ref_depth, equal_nan=True), (tag, alpha)
    return ref_img


@pytest.mark.parametrize("alpha", ALPHAS)
def test_random_meshes_blend_bit_exact(sim3dr_oracle, alpha):
    rng = np.random.default_rng(4242)
    changed = 0
    for trial in range(12):
        nver, ntri = int(rng.integers(3, 80)), int(rng.integers(1, 300))
        h, w, c = int(rng.integers(1, 300)), int(rng.integers(1, 300)), int(rng.integers(1, 5))
        v = rng.uniform(-20, max(h, w) + 20, (nver, 3)).astype(np.float32)
        if trial % 3 == 0:
            v[:, 2] = np.round(v[:, 2] / 40)  # depth ties: only a strictly deeper fragment blends
        if trial % 4 == 0:
            v[:, :2] = np.round(v[:, :2])
        t = rng.integers(0, nver, (ntri, 3)).astype(np.int32)
        col = rng.uniform(-0.1, 1.1, (nver, c)).astype(np.float32)
        bg = rng.integers(0, 255, (h, w, c)).astype(np.uint8)
        d0 = None
        if trial % 5 == 0:
            d0 = rng.uniform(-50, 300, (h, w)).astype(np.float32)  # something already drawn
        ref = _check(sim3dr_oracle, v, t, col, bg, alpha, bool(trial % 2), d0, f"trial {trial}")
        changed += int((ref != bg).any())
    assert changed >= 10


def test_head_mesh_and_a_batch_blend_bit_exact(static, decode_golden, sim3dr_oracle):
    faces = static["faces"]
    verts = np.ascontiguousarray(decode_golden["b2_proj3"]).copy()  # two heads
    verts[..., 2] *= -1
    rng = np.random.default_rng(7)
    col = rng.uniform(0, 1, (2, verts.shape[1], 3)).astype(np.float32)
    bg = rng.integers(0, 255, (2, 256, 256, 3)).astype(np.uint8)
    mesh = Mesh(faces, verts.shape[1], device=0)
    for alpha in (0.5, 0.9):
        img = torch.from_numpy(bg.copy()).cuda()
        mesh.rasterize(torch.from_numpy(verts).cuda(), torch.from_numpy(col).cuda(), img, alpha=alpha)
        for b in range(2):
            ref = sim3dr_oracle.rasterize(np.ascontiguousarray(verts[b]), faces, np.ascontiguousarray(col[b]), bg=bg[b].copy(), alpha=alpha)
            assert np.array_equal(img[b].cpu().numpy(), ref), (alpha, b)
    # alpha == 1 through the same entry is the z-buffer path: unchanged
    img = torch.from_numpy(bg.copy()).cuda()
    mesh.rasterize(torch.from_numpy(verts).cuda(), torch.from_numpy(col).cuda(), img, alpha=1.0)
    assert np.array_equal(img[0].cpu().numpy(), sim3dr_oracle.rasterize(np.ascontiguousarray(verts[0]), faces, np.ascontiguousarray(col[0]), bg=bg[0].copy()))


@pytest.mark.parametrize("profile", ["pixel_centres", "slivers", "tie_planes", "tile_straddle", "mixed_sizes", "split_tile", "wild", "layers"])
def test_adversarial_profiles_blend_bit_exact(sim3dr_oracle, profile):
    from test_gpu_raster_fuzz import build_case

    for k, alpha in enumerate(ALPHAS):
        for seed in (11 + 100 * k, 12 + 100 * k):
            v, t, col, bg, h, w, rev = build_case(seed, profile)
            d0 = None
            if profile == "layers":
                d0 = np.full((h, w), -1e8, np.float32)
                d0[h // 4 : h // 2, :] = 0.0
                d0[:, w // 3 : w // 2] = 2.5
            _check(sim3dr_oracle, v, t, col, bg, alpha, rev, d0, f"{profile} {seed}")


def test_dense_tile_with_split_parts_blends_bit_exact(sim3dr_oracle):
    """20 000 small triangles in one screen region: tile lists longer than 4096 entries, tiles split 2 x 2 and 4 x 4."""
    rng = np.random.default_rng(1)
    h, w, c, nver, ntri = 256, 256, 3, 9000, 20000
    v = np.empty((nver, 3), np.float32)
    v[:, :2] = rng.uniform(70, 170, (nver, 2))
    v[:, 2] = rng.uniform(-5, 5, nver)
    base = rng.integers(0, nver, ntri)
    order = np.lexsort((v[:, 1] // 4, v[:, 0] // 4))
    pos = np.empty(nver, np.int64)
    pos[order] = np.arange(nver)
    t = np.stack([base, order[np.minimum(pos[base] + 1, nver - 1)], order[np.minimum(pos[base] + 2, nver - 1)]], 1).astype(np.int32)
    col = rng.uniform(0, 1, (nver, c)).astype(np.float32)
    bg = rng.integers(0, 255, (h, w, c)).astype(np.uint8)
    _check(sim3dr_oracle, v, t, col, bg, 0.3, False, None, "dense")


def test_blend_through_the_reference_cython_binding(static, decode_golden, sim3dr_oracle):
    from test_gpu_cython_binding import _load

    mod = _load()
    verts = np.ascontiguousarray(decode_golden["b2_proj3"][0]).copy()
    verts[:, 2] *= -1
    faces = static["faces"]
    rng = np.random.default_rng(3)
    col = rng.uniform(0, 1, (verts.shape[0], 3)).astype(np.float32)
    for alpha, reverse in ((0.3, False), (0.75, True)):
        bg = rng.integers(0, 255, (256, 256, 3)).astype(np.uint8)
        img, depth = bg.copy(), np.zeros((256, 256), np.float32) - 1e8
        mod.rasterize(img, verts, faces, col, depth, faces.shape[0], 256, 256, 3, alpha, reverse)  # rasterize.pyx:95
        ref, ref_depth = sim3dr_oracle.rasterize(verts, faces, col, bg=bg.copy(), alpha=alpha, reverse=reverse, return_depth=True)
        assert np.array_equal(img, ref) and np.array_equal(depth, ref_depth), alpha


def test_blend_argument_checks():
    mesh = Mesh(np.array([[0, 1, 2]], np.int32), 3, device=0)
    v = torch.zeros((1, 3, 3), device="cuda")
    with pytest.raises(_lib.Dad3dError, match="channels"):
        mesh.rasterize(v, torch.zeros((1, 3, 5), device="cuda"), torch.zeros((1, 4, 4, 5), dtype=torch.uint8, device="cuda"), alpha=0.5)
    with pytest.raises(_lib.Dad3dError, match="NaN"):
        mesh.rasterize(v, torch.zeros((1, 3, 3), device="cuda"), torch.zeros((1, 4, 4, 3), dtype=torch.uint8, device="cuda"), alpha=float("nan"))
