"""GPU: Sim3DR normals / rasterisation (through the C ABI) -- bit-exact against the CPU oracle and the
goldens produced by the reference's own C++."""
import numpy as np
import pytest
import torch

from dad_3dheads_amd import Sim3DR, _lib
from dad_3dheads_amd.Sim3DR import Mesh
from oracle.sim3dr_ref import render_pipeline_ref
from render_checks import assert_render_bytes_explained

pytestmark = pytest.mark.gpu


def head_inputs(static, decode_golden):
    verts = np.ascontiguousarray(decode_golden["b2_proj3"][0]).copy()
    verts[:, 2] *= -1
    return verts, static["faces"]


def test_numpy_surface_matches_reference_goldens(static, decode_golden, sim3dr_golden):
    g = sim3dr_golden
    verts, faces = head_inputs(static, decode_golden)
    n = Sim3DR.get_normal(verts, faces)
    assert np.array_equal(n, g["head_normals"])
    col = np.clip(n * 0.5 + 0.5, 0, 1).astype(np.float32)
    img = Sim3DR.rasterize(verts, faces, col, height=256, width=256, channel=3)
    assert np.array_equal(img, g["head_image"])
    assert np.array_equal(Sim3DR.rasterize(verts, faces, col, height=256, width=256, channel=3, reverse=True), g["head_image_reverse"])
    d, tb, bw = Sim3DR.rasterize_triangles(verts, faces, 256, 256)
    assert np.array_equal(tb, g["head_tri_buf"]) and np.array_equal(bw, g["head_bary"]) and np.array_equal(d, g["head_depth_tri"])
    pncc = Sim3DR.rasterize(verts, static["faces_wo_ears"], g["pncc_colors"], bg=np.zeros((256, 256, 3), np.uint8))
    assert np.array_equal(pncc, g["pncc_image"])
    tv = np.array([[1, 1, 0.5], [6, 1, 0.5], [1, 6, 0.5]], np.float32)
    tri8 = Sim3DR.rasterize(tv, np.array([[0, 1, 2]], np.int32), np.ones((3, 3), np.float32), height=8, width=8, channel=3)
    assert np.array_equal(tri8, g["tri8_image"])


def test_typed_buffer_errors_like_cython():
    v = np.zeros((3, 3), np.float32)
    t = np.array([[0, 1, 2]], np.int32)
    with pytest.raises(ValueError, match="expected 'float' but got 'double'"):
        Sim3DR.get_normal(v.astype(np.float64), t)
    with pytest.raises(ValueError, match="expected 'int' but got 'long'"):
        Sim3DR.get_normal(v, t.astype(np.int64))
    with pytest.raises(TypeError):
        Sim3DR.get_normal(None, t)


def test_soup_case_bit_exact(sim3dr_golden):
    g = sim3dr_golden
    v, t, col = g["soup_vertices"], g["soup_triangles"], g["soup_colors"]
    mesh = Mesh(t, v.shape[0], device=0)
    dv = torch.from_numpy(v).cuda()[None]
    img = torch.from_numpy(g["soup_bg"].copy()).cuda()[None].contiguous()
    depth = torch.from_numpy(g["soup_depth_in"].copy()).cuda()[None].contiguous()
    mesh.rasterize(dv, torch.from_numpy(col).cuda()[None], img, depth=depth)
    assert np.array_equal(img[0].cpu().numpy(), g["soup_image"])
    assert np.array_equal(depth[0].cpu().numpy(), g["soup_depth"])
    d, tb, bw = mesh.rasterize_triangles(dv, 48, 64, depth=torch.from_numpy(g["soup_depth_in"].copy()).cuda()[None].contiguous())
    won = g["soup_tri_buf"] >= 0
    assert np.array_equal(tb[0].cpu().numpy()[won], g["soup_tri_buf"][won])
    assert np.array_equal(bw[0].cpu().numpy()[won], g["soup_bary"][won]) and np.array_equal(d[0].cpu().numpy(), g["soup_depth_tri"])
    assert np.array_equal(mesh.get_normal(dv)[0].cpu().numpy(), g["soup_normals"])
    acc = torch.from_numpy(g["soup_normal_init"].copy()).cuda()[None].contiguous()
    mesh.get_normal(dv, out=acc, accumulate=True)
    assert np.array_equal(acc[0].cpu().numpy(), g["soup_normals_accum"])
    tn = mesh.get_tri_normal(dv, norm_flg=True)
    assert np.array_equal(mesh.get_ver_normal(mesh.get_tri_normal(dv))[0].cpu().numpy(), g["soup_normals"])
    assert tn.shape == (1, 400, 3)


def test_random_meshes_bit_exact_vs_oracle(sim3dr_oracle):
    rng = np.random.default_rng(42)
    for trial in range(12):
        nver, ntri = int(rng.integers(3, 80)), int(rng.integers(1, 300))
        h, w, c = int(rng.integers(1, 300)), int(rng.integers(1, 300)), int(rng.integers(1, 5))
        batch = int(rng.integers(1, 4))
        v = rng.uniform(-20, max(h, w) + 20, (batch, nver, 3)).astype(np.float32)
        if trial % 3 == 0:
            v[..., 2] = np.round(v[..., 2] / 40)  # depth ties
        if trial % 4 == 0:
            v[..., :2] = np.round(v[..., :2])  # vertices on pixel centres
        t = rng.integers(0, nver, (ntri, 3)).astype(np.int32)
        col = rng.uniform(0, 1, (batch, nver, c)).astype(np.float32)
        bg = rng.integers(0, 255, (batch, h, w, c)).astype(np.uint8)
        rev = bool(trial % 2)
        mesh = Mesh(t, nver, device=0)
        img = torch.from_numpy(bg.copy()).cuda()
        depth = torch.full((batch, h, w), -1e8, device="cuda")
        mesh.rasterize(torch.from_numpy(v).cuda(), torch.from_numpy(col).cuda(), img, depth=depth, reverse=rev)
        normals = mesh.get_normal(torch.from_numpy(v).cuda()).cpu().numpy()
        for b in range(batch):
            ref_img, ref_depth = sim3dr_oracle.rasterize(np.ascontiguousarray(v[b]), t, np.ascontiguousarray(col[b]), bg=bg[b].copy(),
                                                       reverse=rev, return_depth=True)
            assert np.array_equal(img[b].cpu().numpy(), ref_img), (trial, b)
            assert np.array_equal(depth[b].cpu().numpy(), ref_depth)
            assert np.array_equal(normals[b], sim3dr_oracle.get_normal(np.ascontiguousarray(v[b]), t))


def test_batched_head_render_and_full_size_properties(static, decode_golden, sim3dr_oracle):
    """BASELINE config 5 shape per GPU (B=64, 9976 triangles, 256x256x3): every image of the batch equals the
    single-image oracle for the first few, plus size-independent properties for all: idempotence of a
    second draw, untouched background outside coverage, reverse == vertical flip."""
    verts, faces = head_inputs(static, decode_golden)
    B = 64
    rng = np.random.default_rng(1)
    shift = rng.uniform(-30, 30, (B, 1, 3)).astype(np.float32)
    shift[..., 2] = 0
    v = torch.from_numpy(verts[None] + shift).cuda().contiguous()
    mesh = Mesh(faces, 5023, device=0)
    normals = mesh.get_normal(v)
    col = (normals * 0.5 + 0.5).clamp(0, 1).contiguous()
    bg = torch.full((B, 256, 256, 3), 9, dtype=torch.uint8, device="cuda")
    img = mesh.rasterize(v, col, bg.clone())
    again = mesh.rasterize(v, col, img.clone())  # same fragments win again: image unchanged
    rev = mesh.rasterize(v, col, bg.clone(), reverse=True)
    torch.cuda.synchronize()
    assert torch.equal(img, again) and torch.equal(rev, img.flip(1))
    _, tri_buf, _ = mesh.rasterize_triangles(v, 256, 256)
    for b in range(3):
        vb = np.ascontiguousarray(v[b].cpu().numpy())
        assert np.array_equal(normals[b].cpu().numpy(), sim3dr_oracle.get_normal(vb, faces))
        ref = sim3dr_oracle.rasterize(vb, faces, col[b].cpu().numpy(), bg=np.full((256, 256, 3), 9, np.uint8))
        assert np.array_equal(img[b].cpu().numpy(), ref)
    # unit normals wherever a vertex has faces
    nn = normals.norm(dim=-1)
    assert torch.all((nn - 1).abs() < 1e-5)


def test_render_pipeline_matches_numpy_lighting(static, decode_golden, sim3dr_oracle):
    verts, faces = head_inputs(static, decode_golden)
    ref_img, ref_light = render_pipeline_ref(sim3dr_oracle, verts.copy(), faces, np.zeros((256, 256, 3), np.uint8))
    img = Sim3DR.RenderPipeline()(verts.copy(), faces, np.zeros((256, 256, 3), np.uint8))
    mesh = Mesh(faces, 5023, device=0)
    dv = torch.from_numpy(verts).cuda()[None]
    normals = mesh.get_normal(dv)
    light = mesh.phong_light(dv, normals)[0].cpu().numpy()
    assert np.abs(light - ref_light).max() < 2e-5  # float pow / normalisation differ by rounding only
    # the one-launch variant (normals computed inside) is the same arithmetic: identical bits, normals included
    n_out = torch.empty_like(dv)
    fused = mesh.phong_light(dv, None, normals_out=n_out)
    assert torch.equal(n_out, normals) and np.array_equal(fused[0].cpu().numpy(), light)
    b8 = dv.expand(8, -1, -1).contiguous() * torch.linspace(0.9, 1.1, 8, device="cuda")[:, None, None]
    assert torch.equal(mesh.phong_light(b8, None), mesh.phong_light(b8, mesh.get_normal(b8)))
    # two-launch render (lighting inside the raster's geometry kernel) == light then rasterize, bit for bit
    want_light = mesh.phong_light(b8, None)
    want_img = mesh.rasterize(b8, want_light, torch.zeros((8, 256, 256, 3), dtype=torch.uint8, device="cuda"))
    got_light = torch.empty_like(b8)
    got_img = mesh.render(b8, torch.zeros((8, 256, 256, 3), dtype=torch.uint8, device="cuda"), light_out=got_light)
    assert torch.equal(got_light, want_light) and torch.equal(got_img, want_img) and got_img.any()
    # clear=True: the destination's old contents are irrelevant (the geometry launch zeroes it), also for a buffer that
    # is not 16-byte aligned / sized (memset path) and with reverse
    dirty = torch.full((8, 256, 256, 3), 77, dtype=torch.uint8, device="cuda")
    assert torch.equal(mesh.render(b8, dirty, clear=True), want_img)
    odd = torch.full((8 * 256 * 256 * 3 + 5,), 99, dtype=torch.uint8, device="cuda")
    view = odd[5:].view(8, 256, 256, 3)
    assert torch.equal(mesh.render(b8, view, clear=True), want_img) and int(odd[:5].sum()) == 5 * 99
    rev = mesh.render(b8, torch.full_like(dirty, 3), reverse=True, clear=True)
    assert torch.equal(rev, mesh.render(b8, torch.zeros_like(dirty), reverse=True))
    # bytes: identical coverage, and every byte that differs is explained by `(unsigned char)(255 * c)` flipping where the
    # oracle's own float colour is within 255 * 2e-5 of an integer (tests/render_checks.py) -- no tolerance on bytes
    n_diff = assert_render_bytes_explained(img, ref_img, sim3dr_oracle, verts, faces, ref_light)
    assert n_diff < 0.02 * img.size
    # every operation but pow is bit-identical: np.power is exact multiplication for exponents 1 and 2, and so is the kernel
    for e in (1, 2):
        _, l_ref = render_pipeline_ref(sim3dr_oracle, verts.copy(), faces, np.zeros((256, 256, 3), np.uint8), specular_exp=e)
        l_gpu = mesh.phong_light(dv, None, specular_exp=e)[0].cpu().numpy()
        assert np.array_equal(l_gpu, l_ref), e
        pipe = Sim3DR.RenderPipeline(specular_exp=e)(verts.copy(), faces, np.zeros((256, 256, 3), np.uint8))
        ref_e, _ = render_pipeline_ref(sim3dr_oracle, verts.copy(), faces, np.zeros((256, 256, 3), np.uint8), specular_exp=e)
        assert np.array_equal(pipe, ref_e), e  # RenderPipeline bytes bit-exact when pow is out of the picture


def test_empty_inputs():  # (alpha != 1: tests/test_gpu_raster_alpha.py)
    empty = Mesh(np.zeros((0, 3), np.int32), 4, device=0)
    out = empty.rasterize(torch.zeros((2, 4, 3), device="cuda"), torch.zeros((2, 4, 3), device="cuda"),
                          torch.full((2, 5, 5, 3), 3, dtype=torch.uint8, device="cuda"))
    assert torch.all(out == 3)
    assert torch.all(empty.get_normal(torch.ones((2, 4, 3), device="cuda")) == 0)
    with pytest.raises(_lib.Dad3dError):
        Mesh(np.array([[0, 1, 7]], np.int32), 3, device=0)


@pytest.mark.parametrize("case", ["dense_tile", "huge_triangles", "large_odd_image", "large_packed_image"])
def test_raster_work_queue_paths_bit_exact(sim3dr_oracle, case):
    """The paths a head mesh at 256x256 does not reach: tile lists longer than one sorting round (> 4096 entries),
    tiles split 2x2 and 4x4 by the cost model, boxes as large as a tile (the 512-lane class), hundreds of tiles,
    parts beyond the image edge, and the bytewise colour path (width not a multiple of 4)."""
    rng = np.random.default_rng({"dense_tile": 1, "huge_triangles": 2, "large_odd_image": 3, "large_packed_image": 4}[case])
    if case == "dense_tile":
        h, w, c, nver, ntri = 256, 256, 3, 9000, 20000
        v = np.empty((nver, 3), np.float32)
        v[:, :2] = rng.uniform(70, 170, (nver, 2))
        v[:, 2] = rng.uniform(-5, 5, nver)
        base = rng.integers(0, nver, ntri)
        # small triangles: the three corners are close neighbours in a sorted order of the vertices
        order = np.lexsort((v[:, 1] // 4, v[:, 0] // 4))
        pos = np.empty(nver, np.int64)
        pos[order] = np.arange(nver)
        t = np.stack([base, order[np.minimum(pos[base] + 1, nver - 1)], order[np.minimum(pos[base] + 2, nver - 1)]], 1).astype(np.int32)
    elif case == "huge_triangles":
        h, w, c, nver, ntri = 384, 512, 4, 40, 60
        v = rng.uniform(-300, 800, (nver, 3)).astype(np.float32)
        v[:, 2] = np.round(v[:, 2] / 200)  # depth ties between whole-image triangles
        t = rng.integers(0, nver, (ntri, 3)).astype(np.int32)
    else:
        h, w, c = (777, 1030, 3) if case == "large_odd_image" else (700, 1032, 3)
        nver, ntri = 3000, 6000
        v = rng.uniform(-50, 1100, (nver, 3)).astype(np.float32)
        near = rng.integers(0, nver, ntri)
        t = np.stack([near, (near + rng.integers(1, 40, ntri)) % nver, (near + rng.integers(1, 40, ntri)) % nver], 1).astype(np.int32)
        v[:, :2] = np.sort(v[:, :2], axis=0)  # neighbours in index are neighbours on screen: mid-sized triangles
    col = rng.uniform(0, 1, (nver, c)).astype(np.float32)
    bg = rng.integers(0, 255, (h, w, c)).astype(np.uint8)
    mesh = Mesh(t, nver, device=0)
    dv = torch.from_numpy(v).cuda()[None]
    img = torch.from_numpy(bg.copy()).cuda()[None].contiguous()
    depth = torch.full((1, h, w), -1e8, device="cuda")
    mesh.rasterize(dv, torch.from_numpy(col).cuda()[None], img, depth=depth)
    ref_img, ref_depth = sim3dr_oracle.rasterize(v, t, col, bg=bg.copy(), return_depth=True)
    assert np.array_equal(img[0].cpu().numpy(), ref_img)
    assert np.array_equal(depth[0].cpu().numpy(), ref_depth)
    assert (ref_img != bg).any()
    d, tb, bw = mesh.rasterize_triangles(dv, h, w)
    rd, rtb, rbw = sim3dr_oracle.rasterize_triangles(v, t, h, w)
    won = rtb >= 0
    assert np.array_equal(d[0].cpu().numpy(), rd) and np.array_equal(tb[0].cpu().numpy()[won], rtb[won])
    assert np.array_equal(bw[0].cpu().numpy()[won], rbw[won])
    # a second batch shape on the same handle re-plans the scratch; the first result must be reproducible afterwards
    two = torch.cat([dv, dv]).contiguous()
    img2 = torch.from_numpy(np.stack([bg, bg])).cuda().contiguous()
    mesh.rasterize(two, torch.from_numpy(np.stack([col, col])).cuda(), img2)
    assert np.array_equal(img2[0].cpu().numpy(), ref_img) and np.array_equal(img2[1].cpu().numpy(), ref_img)


def test_back_to_back_rasterisations_are_stable(static, decode_golden):
    """The work queue of a launch is built by the last block of the geometry kernel from counters other blocks
    updated (agent-scope atomics, no fence) and consumed by the next kernel; 60 launches queued without any host
    synchronisation, alternating two batch shapes on one handle, must all give the same images."""
    verts, faces = head_inputs(static, decode_golden)
    mesh = Mesh(faces, 5023, device=0)
    base = torch.from_numpy(verts).cuda()
    scale = torch.linspace(0.6, 1.3, 16, device="cuda")[:, None, None]
    v16 = (base[None] * scale).contiguous()
    v5 = v16[3:8].contiguous()
    col = torch.rand((16, 5023, 3), device="cuda")
    first16 = mesh.rasterize(v16, col, torch.zeros((16, 256, 256, 3), dtype=torch.uint8, device="cuda")).clone()
    first5 = mesh.rasterize(v5, col[3:8].contiguous(), torch.zeros((5, 200, 312, 3), dtype=torch.uint8, device="cuda")).clone()
    assert first16.any() and first5.any()
    outs = []
    for i in range(30):
        outs.append(mesh.rasterize(v16, col, torch.zeros((16, 256, 256, 3), dtype=torch.uint8, device="cuda")))
        outs.append(mesh.rasterize(v5, col[3:8].contiguous(), torch.zeros((5, 200, 312, 3), dtype=torch.uint8, device="cuda")))
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        assert torch.equal(o, first16 if i % 2 == 0 else first5), i


def test_config4_and_config5_batch_sizes(static, flame_model, flame_consts, sim3dr_oracle):
    """BASELINE configs 4 and 5 at their single-node totals on ONE GPU: decode of 2048 rows (sampled against the oracle)
    and the render chain on 512 of them (sampled bit-exact against the reference raster on the same vertices)."""
    from dad_3dheads_amd import landmarks, synthetic
    from dad_3dheads_amd.head_mesh import HeadMesh
    from oracle import flame_ref

    hm = HeadMesh(flame_model=flame_model, landmarks=landmarks.canonical("445", static), static=static, device=0)
    p = torch.from_numpy(synthetic.synthetic_params(2048, seed=77)).cuda()
    out = hm.decode(p, to_2d=False, flip_z=True, landmarks=False, landmarks_px=True)
    idx = [0, 1, 63, 64, 1000, 2047]
    ref = flame_ref.vertices_3d(flame_consts, p[idx].cpu().clone())
    assert (out["verts3d"][idx].cpu() - ref).abs().max() < 5e-6
    assert out["lmk_px"].shape == (2048, 445, 2)
    mesh = Mesh(static["faces"], 5023, device=0)
    verts = out["proj"][:512].contiguous()
    light = mesh.phong_light(verts, None)
    img = mesh.rasterize(verts, light, torch.zeros((512, 256, 256, 3), dtype=torch.uint8, device="cuda"))
    for i in (0, 255, 511):
        want = sim3dr_oracle.rasterize(np.ascontiguousarray(verts[i].cpu().numpy()), static["faces"],
                                     np.ascontiguousarray(light[i].cpu().numpy()), bg=np.zeros((256, 256, 3), np.uint8))
        assert np.array_equal(img[i].cpu().numpy(), want), i
