"""GPU: the gated exact-product splits of the blend-shape contraction (csrc/flame_decode_split.hip, round 6;
`select_kernel("split_bf16")`: three bf16 planes, six products; `select_kernel("split_f16")`: two fp16 planes, three products) through
the C ABI, every test on both forms. Held to the SAME bars as the default fp32 kernel -- reference-generated goldens,
the CPU oracle at every phase boundary, exact gather, integer pixels -- and, first of all, to its measured error against the
float64 arbiter (oracle/lbs_independent.py) next to the fp32 kernel's: the table goes to gpurun_out/r06_split_error.md (committed as
profiles/r06_split_error.md). Contraction being replaced: model_training/model/flame.py:212-221 (smplx.lbs.blend_shapes + correctives)."""
import os

import numpy as np
import pytest
import torch

from dad_3dheads_amd import _lib, landmarks, synthetic
from dad_3dheads_amd.head_mesh import HeadMesh
from oracle import flame_ref
from oracle.lbs_independent import projected_pixels_subset

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL_V, TOL_PX = 5e-6, 1e-3  # the bars of tests/test_gpu_decode.py (north star: 1e-4 abs)


FORMS = ("split_bf16", "split_f16")


def _mesh(flame_model, static, kernel):
    hm = HeadMesh(flame_model=flame_model, landmarks=landmarks.canonical("445", static), static=static, device=0)
    hm.flame.select_kernel(kernel)  # the split forms raise where they do not cover the launch, never fall back
    hm.split_form = kernel
    return hm


@pytest.fixture(scope="module")
def pipe_mesh(flame_model, static):
    return _mesh(flame_model, static, "pipelined")


@pytest.fixture(scope="module", params=FORMS)
def meshes(request, flame_model, static, pipe_mesh):
    return _mesh(flame_model, static, request.param), pipe_mesh


@pytest.fixture(scope="module")
def both_forms(flame_model, static):
    return {form: _mesh(flame_model, static, form) for form in FORMS}


def _args64(fc):
    return tuple(np.asarray(a, np.float64) if a.dtype.kind == "f" else a for a in
                 (fc.v_template.numpy(), fc.shapedirs.numpy(), fc.posedirs.numpy(), fc.j_regressor.numpy(), fc.parents.numpy(),
                  fc.lbs_weights.numpy()))


def _float64_truth(params, verts, args64):
    """float64 3-D vertices (rotated, model units) and pixels of the listed vertices of every row."""
    px = np.stack([projected_pixels_subset(params[b], verts, *args64) for b in range(params.shape[0])])  # [B, n, 3]
    p = params.astype(np.float64)
    s = np.maximum(p[:, 412] + 1.0, 1e-8)[:, None, None]
    t = np.stack([p[:, 409], p[:, 410], np.zeros(len(p))], axis=-1)[:, None, :]
    v3d = ((px / 128.0 - 1.0) - t) / s
    return v3d, px


def test_error_against_float64_next_to_the_fp32_kernel(both_forms, pipe_mesh, flame_consts):
    """Error before speed: max and rms distance to the float64 evaluation of the same formula, fp32 kernel and both split forms, at
    B = 64 / 256 / 2048 and both camera profiles, on 48 vertices of every row. A split must not be worse than 1.5x the fp32
    kernel on any line (both are measured better: every product enters the accumulator exactly)."""
    pipe = pipe_mesh
    args64 = _args64(flame_consts)
    rng = np.random.default_rng(6)
    verts = np.sort(rng.choice(5023, 48, replace=False))
    vsel = torch.from_numpy(verts).cuda()
    lines = ["| camera profile | batch | kernel | 3-D max | 3-D rms | px max | px rms |", "|---|---|---|---|---|---|---|"]
    for profile in ("crop", "survey"):
        for batch in (64, 256, 2048):
            params = synthetic.synthetic_params(batch, seed=600 + batch, profile=profile)
            v64, px64 = _float64_truth(params, verts, args64)
            err = {}
            for name, hm in (("fp32 (pipelined)", pipe), ("bf16x3 split", both_forms["split_bf16"]), ("fp16x2 split", both_forms["split_f16"])):
                out = hm.decode(torch.from_numpy(params.copy()).cuda(), to_2d=False, landmarks=False)
                torch.cuda.synchronize()
                dv = out["verts3d"][:, vsel].cpu().numpy().astype(np.float64) - v64
                dp = out["proj"][:, vsel].cpu().numpy().astype(np.float64) - px64
                err[name] = (np.abs(dv).max(), np.sqrt((dv ** 2).mean()), np.abs(dp).max(), np.sqrt((dp ** 2).mean()))
                lines.append(f"| {profile} | {batch} | {name} | {err[name][0]:.3e} | {err[name][1]:.3e} | {err[name][2]:.3e} | {err[name][3]:.3e} |")
                assert err[name][0] < TOL_V and err[name][2] < TOL_PX
            for form in ("bf16x3 split", "fp16x2 split"):
                a, b = err[form], err["fp32 (pipelined)"]
                for k in range(4):
                    assert a[k] <= 1.5 * b[k], (profile, batch, form, k, a, b)
    text = ("# Error of the decode kernels against float64 (oracle/lbs_independent.py), fp32 MFMA chain vs the exact-product splits\n\n"
            "Written by tests/test_gpu_decode_split.py on the GPU box: 48 vertices of every row, `3-D` in FLAME model units (north star 1e-4,\n"
            "test bar 5e-6), `px` in pixels of the 256 x 256 frame (test bar 1e-3). Both kernels share the epilogue and the per-image constants;\n"
            "they differ in the blend-shape contraction only (flame.py:212-221): fp32 MFMA chain | three bf16 planes, six products | two fp16 planes,\n"
            "three products (and the split forms fold the projection into one fma per component).\n\n" + "\n".join(lines) + "\n")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06_split_error.md"), "w") as fh:
        fh.write(text)
    print(text)


def test_split_kernel_matches_reference_goldens(meshes, decode_golden, static):
    split, _ = meshes
    g = decode_golden
    dev = torch.from_numpy(g["b2_params"].copy()).cuda()
    out = split.decode(dev, to_2d=False)
    torch.cuda.synchronize()
    assert np.abs(out["verts3d"].cpu().numpy() - g["b2_v3d"]).max() < TOL_V
    assert np.abs(out["proj"].cpu().numpy() - g["b2_proj3"]).max() < TOL_PX
    assert np.array_equal(dev.cpu().numpy(), g["b2_params_after"])
    dev = torch.from_numpy(g["b64_params"].copy()).cuda()
    out = split.decode(dev, to_2d=True, landmarks_px=True)
    torch.cuda.synchronize()
    sub = g["b64_subset"]
    assert np.abs(out["verts3d"].cpu().numpy()[:, sub] - g["b64_v3d_sub"]).max() < TOL_V
    assert np.abs(out["proj"].cpu().numpy()[:, sub] - g["b64_proj_sub"]).max() < TOL_PX
    assert np.abs(out["lmk_xy"].cpu().numpy() - g["b64_lmk_xy"]).max() < TOL_PX
    diff = out["lmk_px"].cpu().numpy() - g["b64_lmk_px"]
    near_int = np.abs(g["b64_lmk_xy"] - np.round(g["b64_lmk_xy"])) < 2 * TOL_PX
    assert np.all((diff == 0) | ((np.abs(diff) == 1) & near_int))
    # edge rows: zero jaw, zero expression, scale clamp, degenerate 6-DoF inputs, 4x coefficients
    dev = torch.from_numpy(g["edge_params"].copy()).cuda()
    out = split.decode(dev, to_2d=False, landmarks=False)
    torch.cuda.synchronize()
    sub = g["edge_subset"]
    assert np.abs(out["verts3d"].cpu().numpy()[:, sub] - g["edge_v3d_sub"]).max() < 4 * TOL_V
    err = np.abs(out["proj"].cpu().numpy()[:, sub] - g["edge_proj3_sub"])
    assert err[:5].max() < TOL_PX and err[5].max() < 4 * TOL_PX
    assert np.array_equal(dev.cpu().numpy(), g["edge_params_after"])
    assert torch.all(out["verts3d"][3] == 0) and torch.all(out["verts3d"][4] == 0)  # degenerate 6-DoF -> R = 0
    # configs[2]: every row of the batch-256 golden
    g256 = np.load(os.path.join(ROOT, "tests", "golden", "decode_b256_golden.npz"))
    dev = torch.from_numpy(synthetic.synthetic_params(256, seed=104)).cuda()
    out = split.decode(dev, to_2d=False, landmarks=False)
    sel = torch.from_numpy(g256["subset"]).cuda()
    assert np.abs(out["verts3d"][:, sel].cpu().numpy() - g256["v3d_sub"]).max() < TOL_V
    assert np.abs(out["proj"][:, sel].cpu().numpy() - g256["proj3_sub"]).max() < TOL_PX
    assert bool((dev[:, 411] == 0).all())


# ragged and aligned batch sizes around every boundary of the 16-image phases
@pytest.mark.parametrize("batch", [1, 3, 15, 16, 17, 31, 32, 33, 47, 48, 49, 64, 65, 100, 128, 129, 259, 385, 641, 1030])
@pytest.mark.parametrize("to_2d", [True, False])
def test_split_kernel_matches_oracle_and_the_fp32_kernel(meshes, flame_consts, static, batch, to_2d):
    split, pipe = meshes
    params = synthetic.synthetic_params(batch, seed=6100 + batch)
    a_in, b_in = torch.from_numpy(params.copy()).cuda(), torch.from_numpy(params.copy()).cuda()
    a = split.decode(a_in, to_2d=to_2d, landmarks=True, landmarks_px=True)
    b = pipe.decode(b_in, to_2d=to_2d, landmarks=True, landmarks_px=True)
    torch.cuda.synchronize()
    assert torch.equal(a_in, b_in) and bool((a_in[:, 411] == 0).all())  # tz := 0 written back by both, nothing else touched
    p = torch.from_numpy(params.copy())
    v_ref = flame_ref.vertices_3d(flame_consts, p).numpy()
    pr_ref = flame_ref.reprojected_vertices(flame_consts, p, to_2d=to_2d).numpy()
    assert np.abs(a["verts3d"].cpu().numpy() - v_ref).max() < TOL_V
    assert np.abs(a["proj"].cpu().numpy() - pr_ref).max() < TOL_PX
    assert float((a["verts3d"] - b["verts3d"]).abs().max()) < 1e-6  # two contractions of the same numbers
    assert float((a["proj"] - b["proj"]).abs().max()) < 5e-4  # ~1e-7 of a vertex x s x 128 (both within TOL_PX of the oracle)
    lm = torch.from_numpy(landmarks.canonical("445", static)).cuda()
    assert torch.equal(a["lmk_xy"], a["proj"][:, lm, :2])  # the gather is exact
    assert torch.equal(a["lmk_px"], a["proj"][:, lm, :2].to(torch.int32))
    d = (a["lmk_px"] - b["lmk_px"]).abs()
    frac = (b["lmk_xy"] - torch.round(b["lmk_xy"])).abs()
    assert bool(((d == 0) | ((d == 1) & (frac < 1e-3))).all())


def test_split_kernel_flags_null_outputs_and_landmark_only(meshes, flame_consts):
    split, _ = meshes
    params = synthetic.synthetic_params(70, seed=6200)
    p = torch.from_numpy(params.copy())
    v_ref = flame_ref.vertices_3d(flame_consts, p).numpy()
    p3_ref = flame_ref.reprojected_vertices(flame_consts, p, to_2d=False).numpy()
    out = split.decode(torch.from_numpy(params.copy()).cuda(), to_2d=False, flip_z=True)
    only_v = split.decode(torch.from_numpy(params.copy()).cuda(), proj=False, landmarks=False)
    only_p = split.decode(torch.from_numpy(params.copy()).cuda(), verts3d=False, to_2d=True, landmarks=False)
    keep = torch.from_numpy(params.copy()).cuda()
    only_l = split.decode(keep, verts3d=False, proj=False, landmarks=True, landmarks_px=True, mutate=False)  # pinned: the whole mesh
    full = split.decode(torch.from_numpy(params.copy()).cuda(), to_2d=True, landmarks=True, landmarks_px=True)
    torch.cuda.synchronize()
    flipped = p3_ref.copy()
    flipped[..., 2] *= -1.0  # inference/pncc_estimator.py:88
    assert np.abs(out["verts3d"].cpu().numpy() - v_ref).max() < TOL_V
    assert np.abs(out["proj"].cpu().numpy() - flipped).max() < TOL_PX
    assert torch.equal(only_v["verts3d"], out["verts3d"]) and torch.equal(only_p["proj"], out["proj"][..., :2])
    assert torch.equal(only_l["lmk_xy"], full["lmk_xy"]) and torch.equal(only_l["lmk_px"], full["lmk_px"])
    assert np.array_equal(keep.cpu().numpy(), params)  # mutate=False: the caller's rows are untouched


def test_split_kernel_duplicate_rows_are_bit_identical_wherever_they_land(meshes):
    split, _ = meshes
    base = synthetic.synthetic_params(5, seed=6300)
    idx = np.array([0, 1, 2, 3, 4] * 60 + [2, 0])  # 302 rows: 19 phases, ragged end
    out = split.decode(torch.from_numpy(base[idx]).cuda(), to_2d=True, landmarks=True, landmarks_px=True)
    torch.cuda.synchronize()
    for k in ("verts3d", "proj", "lmk_xy", "lmk_px"):
        t = out[k]
        for r in range(5):
            rows = t[torch.from_numpy(np.nonzero(idx == r)[0]).cuda()]
            assert torch.equal(rows, rows[:1].expand_as(rows)), (k, r)


def test_split_kernel_poisoned_row_stays_in_its_row(meshes, flame_consts):
    """A NaN or an infinity in one params row reaches that row's outputs only (torch semantics: the reference would return NaN there).
    The fp16 form's documented limit is a row's business too: an entry beyond +-4094 makes THAT row inf/NaN (the bf16 form carries
    fp32's range)."""
    split, _ = meshes
    params = synthetic.synthetic_params(40, seed=6400)
    bad = params.copy()
    bad[7, 13] = np.nan
    bad[21, 350] = np.inf
    if split.split_form == "split_f16":
        bad[33, 5] = 1.0e5
        big = split.decode(torch.from_numpy(bad).cuda(), to_2d=True, landmarks=False)
        assert not bool(torch.isfinite(big["verts3d"][33]).all())
        bad[33, 5] = 4000.0  # inside the range: finite
    good = split.decode(torch.from_numpy(params.copy()).cuda(), to_2d=True, landmarks=False)
    out = split.decode(torch.from_numpy(bad).cuda(), to_2d=True, landmarks=False)
    torch.cuda.synchronize()
    rest = [i for i in range(40) if i not in (7, 21, 33)]
    assert torch.equal(out["verts3d"][rest], good["verts3d"][rest]) and torch.equal(out["proj"][rest], good["proj"][rest])
    assert bool(torch.isfinite(out["verts3d"][33]).all())
    assert bool(torch.isnan(out["verts3d"][7]).all()) and bool(torch.isnan(out["verts3d"][21]).all())


def test_split_kernel_refuses_what_it_does_not_cover_and_replays_from_a_graph(meshes, flame_model, static):
    split, _ = meshes
    p = torch.from_numpy(synthetic.synthetic_params(48, seed=6500)).cuda()
    with pytest.raises(_lib.UnsupportedError):
        split.decode(p.clone(), proj=False, landmarks=False, zero_rotation=True)  # DAD3D_ZERO_ROTATION: the two-role kernel's
    want = split.decode(p.clone(), to_2d=True, landmarks=True)  # warm-up: the scratch of this batch size exists now
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    static_in = p.clone()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            out = split.decode(static_in, to_2d=True, landmarks=True)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    for k in ("verts3d", "proj", "lmk_xy"):
        assert torch.equal(out[k], want[k])
    # the scratch grows with a larger batch; the graph captured at 48 rows keeps the one it was captured with
    big = split.decode(torch.from_numpy(synthetic.synthetic_params(400, seed=6501)).cuda(), to_2d=True, landmarks=False)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(big["verts3d"]).all())
    out["verts3d"].zero_()
    g.replay()
    torch.cuda.synchronize()
    for k in ("verts3d", "proj", "lmk_xy"):
        assert torch.equal(out[k], want[k])
    # a fork shares the basis and owns its scratch
    twin = split.fork()
    twin.flame.select_kernel(split.split_form)
    other = twin.decode(p.clone(), to_2d=True, landmarks=True)
    assert torch.equal(other["proj"], want["proj"])


def test_split_kernel_two_forks_on_two_streams(meshes):
    """Two batches in flight on one GPU (bench.py --streams 2 in split mode): each fork owns its scratch (planes + constants), the basis is
    shared; launches interleaved on two streams return what one stream returns."""
    split, _ = meshes
    twin = split.fork()
    twin.flame.select_kernel(split.split_form)
    pa = torch.from_numpy(synthetic.synthetic_params(200, seed=6600)).cuda()
    pb = torch.from_numpy(synthetic.synthetic_params(136, seed=6601)).cuda()
    want_a = split.decode(pa.clone(), to_2d=True, landmarks=True)
    want_b = split.decode(pb.clone(), to_2d=True, landmarks=True)
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for _ in range(6):
        with torch.cuda.stream(sa):
            oa = split.decode(pa.clone(), to_2d=True, landmarks=True)
        with torch.cuda.stream(sb):
            ob = twin.decode(pb.clone(), to_2d=True, landmarks=True)
        outs.append((oa, ob))
    sa.synchronize(), sb.synchronize()
    for oa, ob in outs:
        for k in ("verts3d", "proj", "lmk_xy"):
            assert torch.equal(oa[k], want_a[k]) and torch.equal(ob[k], want_b[k]), k


def test_write_back_and_write_through_launches_return_the_same_bits(meshes):
    """From 640 (fp16x2) / 1024 (bf16x3) images up the tile kernel's vertex stores are write-back instead of write-through (a cache policy:
    csrc/flame_decode_split.hip, WB): the first 100 rows of a launch of 1100, decoded again on their own, come back bit for bit."""
    split, _ = meshes
    params = synthetic.synthetic_params(1100, seed=6700)
    big = split.decode(torch.from_numpy(params.copy()).cuda(), to_2d=True, landmarks=True, landmarks_px=True)
    small = split.decode(torch.from_numpy(params[:100].copy()).cuda(), to_2d=True, landmarks=True, landmarks_px=True)
    torch.cuda.synchronize()
    for k in ("verts3d", "proj", "lmk_xy", "lmk_px"):
        assert torch.equal(big[k][:100], small[k]), k


@pytest.mark.parametrize("batch", [1, 16, 17, 100, 256, 700, 2048])
def test_landmark_only_launches_of_a_split_handle_run_the_sub_model_and_return_the_full_launch_bits(meshes, batch):
    """A handle on a split form keeps its arithmetic: landmark outputs only -> the sub-model of the listed vertices on the SAME form
    (23 tiles, the launch's phases dealt over up to 11 workgroups per tile: split_chunking in csrc/common.hpp), bit for bit the landmark
    rows of the whole-mesh launch -- duplicates in the list, a custom list, the write-back included or not."""
    split, _ = meshes
    rng = np.random.default_rng(batch)
    for lm in (None, np.concatenate([rng.integers(0, 5023, 90), [7, 7, 5022, 0]]).astype(np.int64)):
        if lm is not None:
            split.set_landmarks(lm)
        params = synthetic.synthetic_params(batch, seed=6800 + batch)
        keep = torch.from_numpy(params.copy()).cuda()
        only = split.decode(keep, verts3d=False, proj=False, landmarks=True, landmarks_px=True, mutate=(batch % 2 == 0))
        full = split.decode(torch.from_numpy(params.copy()).cuda(), to_2d=True, landmarks=True, landmarks_px=True)
        torch.cuda.synchronize()
        assert torch.equal(only["lmk_xy"], full["lmk_xy"]) and torch.equal(only["lmk_px"], full["lmk_px"]), (batch, lm is None)
        after = keep.cpu().numpy()
        want = params.copy()
        if batch % 2 == 0:
            want[:, 411] = 0.0
        assert np.array_equal(after, want)
    assert _lib.load().dad3d_flame_num_landmark_vertices(split.flame._handle) > 0
    split.set_landmarks(landmarks.canonical("445", synthetic.load_static()))


def test_landmark_only_launch_of_a_split_handle_replays_from_a_graph_behind_a_full_warm_up(meshes):
    """The sub-model borrows the parent's scratch: a landmark-only launch captured behind a full-output warm-up of the same batch size allocates
    nothing (the first split launch at a batch size does, and refuses inside a capture)."""
    split, _ = meshes
    p = torch.from_numpy(synthetic.synthetic_params(300, seed=6900)).cuda()
    want = split.decode(p.clone(), to_2d=True, landmarks=True, landmarks_px=True)  # warm-up: scratch of 19 phases
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    static_in = p.clone()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            out = split.decode(static_in, verts3d=False, proj=False, landmarks=True, landmarks_px=True)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out["lmk_xy"], want["lmk_xy"]) and torch.equal(out["lmk_px"], want["lmk_px"])
