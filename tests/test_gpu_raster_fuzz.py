"""GPU: hypothesis-driven adversarial meshes for the z-buffer raster and the normals, bit-exact against the compiled reference
(oracle/_ref/libsim3dr_ref.so when present, the C port otherwise).

The cases aim at where a parallel restatement of Sim3DR/lib/rasterize_kernel.cpp:219-353 can go wrong while a head mesh never
shows it: vertices exactly on pixel centres (the strict `> 0` interior test of _rasterize against the `>= 0` of
_rasterize_triangles), zero-area and sliver triangles (inv = 0 or huge: NaN / inf barycentrics), exact depth ties between
triangles that land in DIFFERENT 64 x 64 tiles and in split tiles (tie -> lowest triangle index), boxes that straddle tile
borders and the image edge, triangles as large as the image next to hundreds of tiny ones (every area class of the lane
allocation in one list), repeated indices, huge coordinates, stacked sheets of either winding (the two-pass culling of
long tile lists). Shrunk counter-examples, if any ever appear, belong in
tests/golden/ (none so far)."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from dad_3dheads_amd.Sim3DR import Mesh

pytestmark = pytest.mark.gpu

PROFILES = ("pixel_centres", "slivers", "tie_planes", "tile_straddle", "mixed_sizes", "split_tile", "wild", "layers")


def build_case(seed: int, profile: str):
    rng = np.random.default_rng(seed)
    h, w = int(rng.integers(40, 200)), int(rng.integers(40, 280))
    if profile == "split_tile":
        h, w = 128, 192
    nver = int(rng.integers(6, 120))
    ntri = int(rng.integers(4, 260))
    v = np.empty((nver, 3), np.float32)
    v[:, 0] = rng.uniform(-12, w + 12, nver)
    v[:, 1] = rng.uniform(-12, h + 12, nver)
    v[:, 2] = rng.uniform(-4, 4, nver)
    t = rng.integers(0, nver, (ntri, 3)).astype(np.int32)
    if profile == "pixel_centres":  # every vertex on a pixel centre, half-integer depths: edges run through pixel centres
        v[:, :2] = np.round(v[:, :2])
        v[:, 2] = np.round(v[:, 2] * 2) / 2
    elif profile == "slivers":  # collinear and nearly collinear corners, repeated corners, tiny areas
        base = rng.integers(0, nver, ntri)
        d = rng.uniform(-30, 30, (ntri, 2)).astype(np.float32)
        extra = nver + 2 * ntri
        v2 = np.empty((extra, 3), np.float32)
        v2[:nver] = v
        for k in range(ntri):
            a = v[base[k]]
            e = float(rng.choice([0.0, 1e-6, 1e-4, 1e-2, 0.3]))
            v2[nver + 2 * k] = [a[0] + d[k, 0], a[1] + d[k, 1], a[2] + 1]
            v2[nver + 2 * k + 1] = [a[0] + 2 * d[k, 0] - e * d[k, 1], a[1] + 2 * d[k, 1] + e * d[k, 0], a[2] - 1]
        t = np.stack([base, nver + 2 * np.arange(ntri), nver + 2 * np.arange(ntri) + 1], 1).astype(np.int32)
        t[::7, 2] = t[::7, 1]  # a repeated corner: exactly zero area
        v, nver = v2, extra
    elif profile == "tie_planes":  # constant-depth planes: every overlap is an exact tie, decided by the triangle index
        v[:, 2] = np.round(v[:, 2])
        v[rng.integers(0, nver, nver // 2), 2] = 1.0
    elif profile == "tile_straddle":  # corners hugging the 64-pixel tile borders and the image edge
        snap = rng.choice([0.0, 63.0, 64.0, 127.0, 128.0, float(w - 1), float(w)], nver)
        v[:, 0] = snap + rng.choice([-1.0, -0.5, -1e-3, 0.0, 1e-3, 0.5, 1.0], nver) + rng.uniform(-3, 3, nver) * (rng.random(nver) < 0.5)
        snap = rng.choice([0.0, 63.0, 64.0, float(h - 1), float(h)], nver)
        v[:, 1] = snap + rng.choice([-1.0, -0.5, 0.0, 0.5, 1.0], nver) + rng.uniform(-40, 40, nver) * (rng.random(nver) < 0.5)
        v = v.astype(np.float32)
    elif profile == "mixed_sizes":  # a few image-sized triangles among many small ones, equal depths between the two kinds
        near = rng.integers(0, nver, ntri)
        t = np.stack([near, (near + 1) % nver, (near + 2) % nver], 1).astype(np.int32)
        order = np.argsort(v[:, 0] // 6 * 1000 + v[:, 1])
        v = v[order]
        big = np.array([[-50, -50, 0.5], [w + 60, -40, 0.5], [w / 2, h + 90, 0.5], [-30, h + 20, -0.5]], np.float32)
        v = np.concatenate([v, big]).astype(np.float32)
        t = np.concatenate([t, np.array([[nver, nver + 1, nver + 2], [nver + 3, nver + 1, nver + 2], [nver, nver + 2, nver + 3]], np.int32)])
        v[: nver // 3, 2] = 0.5
        nver += 4
    elif profile == "split_tile":  # one tile far beyond the split threshold: 2x2 and 4x4 parts, lists re-classed per part
        nver, ntri = 400, 900
        v = np.empty((nver, 3), np.float32)
        v[:, 0] = rng.uniform(60, 132, nver)
        v[:, 1] = rng.uniform(-4, 70, nver)
        v[:, 2] = np.round(rng.uniform(-3, 3, nver))
        t = rng.integers(0, nver, (ntri, 3)).astype(np.int32)
    elif profile == "layers":  # sheets of opposite winding over the same pixels, tile lists long enough for the two-pass
        # path of the tile kernel (far-facing triangles culled against what the near-facing ones left): exact depth ties
        # between the sheets, sheets that cross, either winding in front, a sheet of mixed winding, a pre-filled depth buffer
        h, w = int(rng.integers(70, 140)), int(rng.integers(70, 200))
        g = int(rng.integers(12, 18))
        sheets_v, sheets_t = [], []
        level = rng.permutation([0.0, 0.0, 1.0, -1.0])[:3]  # two sheets may share a depth plane
        for k in range(3):
            gx, gy = np.meshgrid(np.linspace(-4, w * rng.uniform(0.6, 1.05), g), np.linspace(-4, h * rng.uniform(0.6, 1.05), g))
            jit = rng.uniform(-1.2, 1.2, (2,) + gx.shape) * (rng.random(gx.shape) < 0.7)
            px, py = gx + jit[0], gy + jit[1]
            if k == 1:
                px, py = np.round(px), np.round(py)
            z = level[k] + np.round(rng.uniform(-1, 1, gx.shape) * 2) / 2 * (rng.random(gx.shape) < 0.4) + float(rng.choice([0.0, 0.02])) * gx
            sv = np.stack([px.ravel(), py.ravel(), z.ravel()], 1)
            i = (np.arange(g - 1)[:, None] * g + np.arange(g - 1)[None]).ravel()
            st_ = np.concatenate([np.stack([i, i + 1, i + g], 1), np.stack([i + 1, i + g + 1, i + g], 1)])
            if k == 1:
                st_ = st_[:, ::-1]
            if k == 2:
                flip = rng.random(len(st_)) < 0.5
                st_[flip] = st_[flip][:, ::-1]
            sheets_t.append(st_ + sum(len(q) for q in sheets_v))
            sheets_v.append(sv)
        v = np.concatenate(sheets_v).astype(np.float32)
        t = np.concatenate(sheets_t).astype(np.int32)
        t = t[rng.permutation(len(t))]
        nver = len(v)
    elif profile == "wild":  # huge and non-finite-producing coordinates next to ordinary ones
        v[rng.integers(0, nver, 3), 0] = rng.choice([1e7, -1e7, 3e38, -3e38, 65535.5])
        v[rng.integers(0, nver, 2), 1] = rng.choice([1e7, -3e38])
        v[rng.integers(0, nver, 2), 2] = rng.choice([3e38, -3e38, 1e-40])
    c = int(rng.choice([1, 3, 3, 4]))
    col = rng.uniform(-0.2, 1.2, (nver, c)).astype(np.float32)  # colours a little outside [0, 1]: the (uchar) cast wraps
    bg = rng.integers(0, 255, (h, w, c)).astype(np.uint8)
    return np.ascontiguousarray(v, np.float32), np.ascontiguousarray(t), col, bg, h, w, bool(rng.integers(0, 2))


@settings(max_examples=56, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(seed=st.integers(0, 2**31 - 1), profile=st.sampled_from(PROFILES))
def test_raster_and_normals_fuzz_bit_exact(sim3dr_oracle, seed, profile):
    v, t, col, bg, h, w, rev = build_case(seed, profile)
    mesh = Mesh(t, v.shape[0], device=0)
    dv = torch.from_numpy(v).cuda()[None]
    img = torch.from_numpy(bg.copy()).cuda()[None].contiguous()
    depth0 = np.full((h, w), -1e8, np.float32)
    if profile == "layers" and seed % 2:  # something already drawn in front of / level with / behind the sheets
        depth0[h // 4 : h // 2, :] = 0.0
        depth0[:, w // 3 : w // 2] = 2.5
        depth0[h // 2 :, : w // 4] = np.float32(seed % 5) / 4 - 1
    depth = torch.from_numpy(depth0.copy()).cuda()[None]
    mesh.rasterize(dv, torch.from_numpy(col).cuda()[None], img, depth=depth, reverse=rev)
    with np.errstate(all="ignore"):
        ref_img, ref_depth = sim3dr_oracle.rasterize(v, t, col, bg=bg.copy(), reverse=rev, depth=depth0.copy(), return_depth=True)
        rd, rtb, rbw = sim3dr_oracle.rasterize_triangles(v, t, h, w)
        ref_n = sim3dr_oracle.get_normal(v, t)
    assert np.array_equal(img[0].cpu().numpy(), ref_img), (profile, seed)
    assert np.array_equal(depth[0].cpu().numpy(), ref_depth, equal_nan=True), (profile, seed)
    d, tb, bw = mesh.rasterize_triangles(dv, h, w)
    # EVERY pixel: where the reference has no winner its buffers keep their initial values (-1, zeros) -- a stray write would show
    assert np.array_equal(d[0].cpu().numpy(), rd, equal_nan=True), (profile, seed)
    assert np.array_equal(tb[0].cpu().numpy(), rtb), (profile, seed)
    assert np.array_equal(bw[0].cpu().numpy(), rbw, equal_nan=True), (profile, seed)
    assert np.array_equal(mesh.get_normal(dv)[0].cpu().numpy(), ref_n, equal_nan=True), (profile, seed)


@pytest.mark.parametrize("profile", ["pixel_centres", "tie_planes", "tile_straddle", "mixed_sizes", "split_tile", "layers"])
def test_batch_of_three_different_adversarial_meshes(sim3dr_oracle, profile):
    """One topology, three different vertex sets in one launch (the batched entries share the triangle list): image b must equal the
    reference run on image b alone -- per-image tile lists, work items and scratch do not leak between images."""
    v, t, col, bg, h, w, rev = build_case(20240 + len(profile), profile)
    rng = np.random.default_rng(5)
    v2 = v[::-1].copy()  # the same points under other indices: every triangle changes
    v3 = v.copy()
    v3[:, :2] = np.round(v3[:, :2] + rng.uniform(-3, 3, v3[:, :2].shape))  # on pixel centres, shifted
    v3[:, 2] = np.round(v3[:, 2])
    vs = np.stack([v, v2, v3]).astype(np.float32)
    cols = np.stack([col, col[::-1], 1.0 - col]).astype(np.float32)
    bgs = np.stack([bg, bg[::-1], 255 - bg]).astype(np.uint8)
    mesh = Mesh(t, v.shape[0], device=0)
    img = torch.from_numpy(bgs.copy()).cuda().contiguous()
    depth = torch.full((3, h, w), -1e8, device="cuda")
    dv = torch.from_numpy(vs).cuda().contiguous()
    mesh.rasterize(dv, torch.from_numpy(cols).cuda().contiguous(), img, depth=depth, reverse=rev)
    d, tb, bw = mesh.rasterize_triangles(dv, h, w)
    normals = mesh.get_normal(dv)
    for b in range(3):
        with np.errstate(all="ignore"):
            ref_img, ref_depth = sim3dr_oracle.rasterize(np.ascontiguousarray(vs[b]), t, np.ascontiguousarray(cols[b]), bg=bgs[b].copy(),
                                                       reverse=rev, return_depth=True)
            rd, rtb, rbw = sim3dr_oracle.rasterize_triangles(np.ascontiguousarray(vs[b]), t, h, w)
            ref_n = sim3dr_oracle.get_normal(np.ascontiguousarray(vs[b]), t)
        assert np.array_equal(img[b].cpu().numpy(), ref_img), (profile, b)
        assert np.array_equal(depth[b].cpu().numpy(), ref_depth, equal_nan=True), (profile, b)
        assert np.array_equal(tb[b].cpu().numpy(), rtb) and np.array_equal(bw[b].cpu().numpy(), rbw, equal_nan=True), (profile, b)
        assert np.array_equal(d[b].cpu().numpy(), rd, equal_nan=True), (profile, b)
        assert np.array_equal(normals[b].cpu().numpy(), ref_n, equal_nan=True), (profile, b)
