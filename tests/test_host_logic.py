"""CPU: host-side mirrors of the reference interface that need no GPU."""
import os
import pickle
import sys
import types

import numpy as np
import pytest
import torch

from dad_3dheads_amd import flame, landmarks, synthetic
from oracle import flame_ref


def test_from_3dmm_views_and_order():
    p = torch.arange(2 * 413, dtype=torch.float32).reshape(2, 413)
    consts = {"shape": 300, "expression": 100, "jaw": 3, "rotation": 6, "eyeballs": 0, "neck": 0, "translation": 3, "scale": 1}
    fp = flame.FlameParams.from_3dmm(p, consts)
    assert fp.shape.shape == (2, 300) and fp.expression.shape == (2, 100) and fp.jaw.shape == (2, 3)
    assert fp.rotation.shape == (2, 6) and fp.eyeballs.shape == (2, 0) and fp.neck.shape == (2, 0)
    assert fp.jaw[0, 0] == 400 and fp.rotation[0, 0] == 403 and fp.translation[0, 0] == 409 and fp.scale[0, 0] == 412
    fp.translation[..., 2] = -1.0  # views alias the parent like in the reference
    assert (p[:, 411] == -1).all()
    ref = flame_ref.split_3dmm(p, consts)
    for k in ref:
        assert torch.equal(getattr(fp, k), ref[k])
    with pytest.raises(AssertionError):
        flame.FlameParams.from_3dmm(p[0], consts)
    # to_3dmm_tensor is bug-compatible: rotation before jaw (flame.py:86-101)
    t = fp.to_3dmm_tensor()
    assert torch.equal(t[:, 400:406], p[:, 403:409]) and torch.equal(t[:, 406:409], p[:, 400:403])
    assert torch.equal(flame.FlameParams.from_3dmm(p, consts, zero_expr=True).expression, torch.zeros(2, 100))


def test_canonical_landmark_lists(static):
    assert np.array_equal(landmarks.canonical("445", static), static["lmk_445"])
    assert len(landmarks.canonical("565", static)) == 565 and len(landmarks.canonical(191, static)) == 191
    with pytest.raises(ValueError):
        landmarks.canonical("68", static)


def test_load_indices_from_npy_semantics(tmp_path):
    d = tmp_path / "kp"
    d.mkdir()
    np.save(d / "b.npy", {"x": np.array([5, 6]), "y": np.array([7])}, allow_pickle=True)
    np.save(d / "a.npy", {"q": np.array([1, 2])}, allow_pickle=True)
    np.save(d / "cheeks.npy", {"c": np.array([9])}, allow_pickle=True)
    assert landmarks.load_indices_from_npy(str(d / "b.npy")) == [5, 6, 7]
    cfg = {"2d_subset_name": "keypoints", "2d_subset_path": str(d)}
    assert landmarks.load_2d_indices(cfg) == [1, 2, 5, 6, 7]  # sorted files, cheeks excluded
    cfg["2d_keys_exclude"] = None
    assert landmarks.load_2d_indices(cfg) == [1, 2, 5, 6, 7, 9]
    assert landmarks.load_2d_indices({"2d_subset_name": "multipie_keypoints"}) is None


def test_synthetic_model_shapes(flame_model):
    m = flame_model
    assert m.v_template.shape == (5023, 3) and m.shapedirs.shape == (5023, 3, 400) and m.posedirs.shape == (5023, 3, 36)
    assert m.J_regressor.shape == (5, 5023) and m.weights.shape == (5023, 5) and m.f.shape == (9976, 3)
    assert np.allclose(m.J_regressor.sum(1), 1) and np.allclose(m.weights.sum(1), 1)
    assert m.kintree_table[0].tolist() == [4294967295, 0, 1, 1, 1]
    p = synthetic.synthetic_params(8, seed=3)
    assert p.shape == (8, 413) and p.dtype == np.float32 and np.abs(p[:, :400]).max() <= 3.0
    assert np.array_equal(p, synthetic.synthetic_params(8, seed=3))


def test_flame_pickle_loader_without_chumpy(tmp_path, flame_model):
    # build a pickle whose shapedirs is an instance of a class from a module that will not exist at load time
    mod = types.ModuleType("chumpy_fake")

    class Ch:
        def __init__(self, x):
            self.x = x

    Ch.__module__ = "chumpy_fake"
    Ch.__qualname__ = "Ch"
    mod.Ch = Ch
    sys.modules["chumpy_fake"] = mod
    import scipy.sparse as sp

    small = dict(f=flame_model.f[:4], v_template=flame_model.v_template[:7], shapedirs=Ch(flame_model.shapedirs[:7]),
                 posedirs=flame_model.posedirs[:7], J_regressor=sp.csc_matrix(flame_model.J_regressor[:, :7]),
                 kintree_table=flame_model.kintree_table, weights=flame_model.weights[:7], extra="ignored")
    path = tmp_path / "flame.pkl"
    with open(path, "wb") as f:
        pickle.dump(small, f, protocol=2)
    del sys.modules["chumpy_fake"]
    m = flame.get_flame_model(str(path))
    assert np.array_equal(m.shapedirs, flame_model.shapedirs[:7])
    assert np.array_equal(m.J_regressor, flame_model.J_regressor[:, :7])
    with pytest.raises(FileNotFoundError, match="flame.pkl"):
        flame.get_flame_model(str(tmp_path / "nope.pkl"))


def test_dad3dnet_declaration_matches_the_reference_output_contract():
    """network.py: output dict of flame_regression.py:100-104 with the resnet50 config shapes, on CPU."""
    import torch

    from dad_3dheads_amd.network import DAD3DNet, InferenceNet

    net = DAD3DNet(seed=0).eval()  # eval: the stage walk below must not move the BatchNorm statistics
    assert 32e6 < sum(p.numel() for p in net.parameters()) < 34e6  # ResNet-50 23.5 M + fusion + BiFPN + three heads
    x = torch.zeros(1, 3, 256, 256)
    feats, f = [], x
    for st in net.encoder.stages:
        f = st(f)
        feats.append(tuple(f.shape[1:]))
    assert feats == [(64, 64, 64), (256, 64, 64), (512, 32, 32), (1024, 16, 16), (2048, 8, 8)]
    out = InferenceNet(net, torch.float32)(torch.randn(1, 3, 256, 256, generator=torch.Generator().manual_seed(0)))
    assert out["OUTPUT_3DMM_PARAMS"].shape == (1, 413) and out["OUTPUT_2D_LANDMARKS"].shape == (1, 68, 2)
    assert out["OUTPUT_LANDMARKS_HEATMAP"].shape == (1, 68, 64, 64)
    assert out["OUTPUT_3DMM_PARAMS"][:, :403].abs().max() <= 3.0 and (out["OUTPUT_2D_LANDMARKS"] >= 0).all()
    again = InferenceNet(DAD3DNet(seed=0), torch.float32)(torch.randn(1, 3, 256, 256, generator=torch.Generator().manual_seed(0)))
    assert torch.equal(out["OUTPUT_3DMM_PARAMS"], again["OUTPUT_3DMM_PARAMS"])  # seeded initialisation
    # BatchNorm folding is an exact rewrite up to rounding (non-trivial statistics to make it a real check)
    raw = DAD3DNet(seed=0).eval()
    g = torch.Generator().manual_seed(3)
    for mod in raw.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(mod.num_features, generator=g) + 0.5)
            mod.weight.data.copy_(torch.rand(mod.num_features, generator=g) + 0.5)
            mod.bias.data.copy_(torch.randn(mod.num_features, generator=g) * 0.1)
    import copy

    xin = torch.randn(1, 3, 256, 256, generator=torch.Generator().manual_seed(1))
    ref = InferenceNet(copy.deepcopy(raw), torch.float32, fold_bn=False)(xin)
    folded = InferenceNet(raw, torch.float32, fold_bn=True)(xin)
    assert not any(isinstance(mod, torch.nn.BatchNorm2d) for mod in raw.modules())
    for k in ref:
        assert torch.allclose(ref[k], folded[k], atol=2e-3, rtol=1e-3), k


def test_obj_and_json_writers_produce_the_reference_bytes(tmp_path):
    """writers.py vs the formatting statements of demo_utils.py:106-153 restated line by line."""
    import json

    import torch

    from dad_3dheads_amd import writers
    from dad_3dheads_amd.flame import FLAME_CONSTS

    rng = np.random.default_rng(0)
    verts = (rng.standard_normal((50, 3)) * 0.1).astype(np.float32)
    verts[0] = [0.0, -0.0, 1e-9]
    faces = rng.integers(0, 50, (20, 3))
    v, f = writers.get_mesh({"3d_vertices": torch.from_numpy(verts)}, faces)
    assert f.dtype == np.float64 and np.array_equal(f, faces + 1.0)
    expected = "".join("v %.8f %.8f %.8f\n" % tuple(row) for row in v) + "".join("f %d %d %d\n" % tuple(row) for row in f)
    path = tmp_path / "m.obj"
    writers.MeshSaver()(( v, f), str(path))
    assert path.read_text() == expected and writers.obj_text(v, f) == expected
    batch = torch.from_numpy(np.stack([verts, verts * 2]))
    paths = [str(tmp_path / "a.obj"), str(tmp_path / "b.obj")]
    writers.save_obj_batch(batch, faces, paths)
    assert open(paths[0]).read() == expected and open(paths[1]).read().startswith("v %.8f %.8f %.8f\n" % tuple(verts[0] * 2))
    params = torch.arange(2 * 413, dtype=torch.float32).reshape(2, 413)
    d = writers.get_flame_params({"3dmm_params": params})
    assert list(d) == ["shape", "expression", "rotation", "translation", "scale", "jaw", "eyeballs", "neck"]
    assert d["shape"] == list(map(float, range(300))) and d["jaw"] == [400.0, 401.0, 402.0] and d["rotation"] == [403.0 + i for i in range(6)]
    assert d["eyeballs"] == [] and d["scale"] == [412.0] and sum(FLAME_CONSTS.values()) == 413
    assert writers.flame_params_batch(params)[0] == d and writers.flame_params_batch(params)[1]["scale"] == [825.0]
    jp = tmp_path / "p.json"
    writers.JsonSaver()(d, str(jp))
    assert json.loads(jp.read_text()) == d
    assert writers.get_output_path("/x/y/1.jpeg", "out", "head_mesh", ".obj") == "out/1_head_mesh.obj"


def test_writers_reproduce_the_bytes_of_the_reference_demo_utils(tmp_path, static, flame_consts):
    """writers.py vs the reference's OWN `demo_utils.py` (MeshSaver / JsonSaver / get_mesh / get_flame_params /
    get_output_path run unmodified by tests/golden/make_writers_golden.py): byte-identical .obj and .json files for seeded
    decodes, single and batch entry points."""
    import torch

    from dad_3dheads_amd import synthetic, writers
    from oracle import flame_ref

    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "writers_golden.npz")) as g:
        g = {k: g[k] for k in g.files}
    batch, seed = int(g["batch"]), int(g["seed"])
    params = torch.from_numpy(synthetic.synthetic_params(batch, seed=seed))
    verts = flame_ref.vertices_3d(flame_consts, params.clone())
    assert bool(g["faces_equal_static"]) and str(g["faces_plus_one_dtype"]) == "float64"
    for i in range(batch):
        pred = {"3d_vertices": verts[i], "3dmm_params": params[i : i + 1]}
        mesh = writers.get_mesh(pred, static["faces"])
        assert mesh[1].dtype == np.float64
        po, pj = tmp_path / f"m{i}.obj", tmp_path / f"m{i}.json"
        writers.MeshSaver()(mesh, str(po))
        writers.JsonSaver()(writers.get_flame_params(pred), str(pj))
        assert po.read_bytes() == g[f"obj_{i}"].tobytes(), i
        assert pj.read_bytes() == g[f"json_{i}"].tobytes(), i
    paths = [str(tmp_path / f"b{i}.obj") for i in range(batch)]
    writers.save_obj_batch(verts, static["faces"], paths)
    fl = writers.flame_params_batch(params)
    for i in range(batch):
        assert open(paths[i], "rb").read() == g[f"obj_{i}"].tobytes()
        pj = tmp_path / f"b{i}.json"
        writers.JsonSaver()(fl[i], str(pj))
        assert pj.read_bytes() == g[f"json_{i}"].tobytes()
    assert writers.get_output_path("/data/in/some.image.jpeg", "outputs", "head_mesh", writers.MeshSaver().extension) == str(g["output_path"])
    assert [".png", writers.MeshSaver().extension, writers.JsonSaver().extension] == list(g["extensions"])


def test_ncc_color_codes():
    from dad_3dheads_amd.pncc import compute_ncc_color_codes

    t = np.array([[1.0, -2.0, 3.0], [2.0, -1.0, 5.0], [4.0, -4.0, 4.0]])
    c = compute_ncc_color_codes(t, np.array([0, 1]))
    # `initial=0` (pncc_estimator.py:54-55): zero joins the subset's min and max
    lo, hi = np.array([[0.0, -2.0, 0.0]]), np.array([[2.0, 0.0, 5.0]])
    assert np.array_equal(c, (t - lo) / (hi - lo))
    with pytest.raises(ValueError):
        compute_ncc_color_codes(t.tolist())
    with pytest.raises(ValueError):
        compute_ncc_color_codes(t[:, :2])


def test_68_landmark_embedding_matches_reference_goldens(tmp_path):
    """benchmark_export.Landmarks68 vs outputs of the reference's own get_68_landmarks / get_7_landmarks_from_68
    (tests/golden/make_lmk68_fixture.py), and the submission wire format of the benchmark README."""
    import json

    import torch

    from dad_3dheads_amd import benchmark_export as bx
    from dad_3dheads_amd import synthetic

    st = synthetic.load_static()
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lmk68_embedding.npz")
    with np.load(golden) as z, np.load(bx.embedding_path()) as packaged:
        verts, want68, want7 = torch.from_numpy(z["verts"]), z["lmk68"], z["lmk7"]
        # the packaged embedding (dad-3dheads_amd/assets/) is the one the goldens were produced with
        assert np.array_equal(packaged["face_idx"], z["face_idx"]) and np.array_equal(packaged["b_coords"], z["b_coords"])
    lm = bx.Landmarks68(st["faces"])
    got = lm(verts)
    assert got.shape == (3, 68, 3) and np.array_equal(got.numpy(), want68)  # same three products, same order: same bits
    assert np.array_equal(lm(verts[1]).numpy(), want68[1])
    assert np.array_equal(bx.seven_landmarks(got).numpy(), want7)
    with pytest.raises(AssertionError):
        lm(verts[:, :100])
    rot = torch.eye(3)
    entry = bx.submission_entry(torch.zeros(68, 2), verts[0][:10], got[0], rot)
    assert list(entry) == ["68_landmarks_2d", "N_landmarks_3d", "7_landmarks_3d", "rotation_matrix"]
    assert len(entry["68_landmarks_2d"]) == 68 and len(entry["7_landmarks_3d"]) == 7 and len(entry["rotation_matrix"]) == 3
    p = tmp_path / "sub.json"
    bx.write_submission(str(p), {"item_0": entry})
    back = json.loads(p.read_text())
    assert back["item_0"]["7_landmarks_3d"] == [[float(x) for x in row] for row in want7[0]]


def test_projection_oracle_matches_reference_goldens():
    """oracle/projection_ref.py == the reference's own visualize.get_2d_keypoints (tests/golden/make_projection_golden.py)."""
    import os

    from oracle import projection_ref

    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "projection_golden.npz")) as z:
        for i in range(3):
            data = {"vertices": z[f"vertices_{i}"].tolist(), "model_view_matrix": z[f"model_view_{i}"].tolist(),
                    "projection_matrix": z[f"projection_{i}"].tolist()}
            assert np.array_equal(projection_ref.get_2d_keypoints(data, int(z[f"height_{i}"])), z[f"keypoints_{i}"])


@pytest.mark.skipif(not os.path.isdir("/root/reference/dad_3dheads_benchmark"), reason="reference tree not present on this machine")
@pytest.mark.parametrize("script,fixture", [("make_lmk68_fixture.py", "lmk68_embedding.npz"), ("make_projection_golden.py", "projection_golden.npz"),
                                            ("make_loss_golden.py", "loss_golden.npz"), ("make_writers_golden.py", "writers_golden.npz"),
                                            ("make_decode_b3_golden.py", "decode_b3_golden.npz")])
def test_committed_goldens_are_what_the_reference_produces_here(tmp_path, script, fixture):
    """Authoring container only: re-run the generator (the reference's own functions, imported from where they lie)
    and compare every array with the committed fixture."""
    import subprocess
    import sys

    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    out = tmp_path / fixture
    subprocess.run([sys.executable, os.path.join(here, script), str(out)], check=True, capture_output=True, timeout=300)
    with np.load(out) as fresh, np.load(os.path.join(here, fixture)) as committed:
        assert sorted(fresh.files) == sorted(committed.files)
        for k in fresh.files:
            assert np.array_equal(fresh[k], committed[k]), k


def test_runtime_assets_are_package_data_not_test_fixtures():
    """the package must not reach into tests/ -- static index assets ship under dad-3dheads_amd/assets/."""
    from dad_3dheads_amd import benchmark_export as bx
    from dad_3dheads_amd import synthetic

    pkg = os.path.dirname(os.path.abspath(synthetic.__file__))
    for path in (synthetic.static_fixture_path(), bx.embedding_path()):
        assert os.path.commonpath([pkg, os.path.abspath(path)]) == pkg and os.path.isfile(path), path
    assert os.path.isfile(os.path.join(pkg, "assets", "NOTICE.md"))  # licence of the bundled data


def test_region_tables_for_the_fused_losses():
    """losses.RegionTables (host logic of csrc/mesh_losses.hip): the region lists back to back, the same incidence
    transposed per vertex in (vertex, region, position) order, negative indices like `tensor[:, i]`, and the per-vertex
    weight the reprojection loss collapses into."""
    import torch

    from dad_3dheads_amd.losses import RegionTables

    n = 50
    idx = [np.array([3, 7, 7, 49]), np.array([-1, 0, 3]), np.arange(10, 20)]
    w = [1.0, 0.5, 2.0]
    t = RegionTables(w, idx, n, torch.device("cpu"))
    assert t.n_regions == 3 and t.region_ptr.tolist() == [0, 4, 7, 17]
    assert t.region_idx.tolist()[:7] == [3, 7, 7, 49, 49, 0, 3]
    ptr, reg, pos = t.vert_ptr.numpy(), t.vert_region.numpy(), t.vert_pos.numpy()
    assert ptr[0] == 0 and ptr[-1] == 17 and (np.diff(ptr) >= 0).all()
    flat, rptr = t.region_idx.numpy(), t.region_ptr.numpy()
    seen = set()
    for v in range(n):
        entries = [(int(reg[e]), int(pos[e])) for e in range(ptr[v], ptr[v + 1])]
        assert entries == sorted(entries)  # fixed summation order: by region, then position
        for r, p in entries:
            assert flat[rptr[r] + p] == v
            seen.add((r, p))
    assert len(seen) == 17  # every (region, position) exactly once
    expect = np.zeros(n)
    for wi, i in zip(w, idx):
        for v in np.where(i < 0, i + n, i):
            expect[v] += wi / max(len(i), 1)
    assert np.allclose(t.point_weight.numpy(), expect, rtol=1e-6)
    with pytest.raises(IndexError):
        RegionTables([1.0], [np.array([50])], n, torch.device("cpu"))
    with pytest.raises(ValueError, match="no vertices"):  # the reference's mean over an empty region is NaN: refused
        RegionTables([1.0, 1.0], [np.array([3]), np.array([], dtype=np.int64)], n, torch.device("cpu"))


def test_grad_inputs_batch_limit_is_the_same_on_both_sides_of_the_c_abi():
    """include/dad3d.h's DAD3D_GRAD_INPUTS_MAX_BATCH (up to which a training forward prepares the hand-written backward GEMM's
    pack and scratch) is the threshold the host mirror switches to the library GEMM at."""
    import re

    from dad_3dheads_amd import autograd

    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "dad3d.h")).read()
    assert int(re.search(r"#define\s+DAD3D_GRAD_INPUTS_MAX_BATCH\s+(\d+)", text).group(1)) == autograd.GRAD_INPUTS_HIP_MAX_BATCH
