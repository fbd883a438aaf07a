"""GPU: the pipelined single-role decode kernel (csrc/flame_decode_pipe.hip, round 4) against the two-role kernel of rounds 1-3
and the CPU oracle, through the C ABI. Both kernels restate model_training/model/flame.py:191-228 + smplx.lbs; the pipelined one
drops the translations of the joints that cannot rotate (they cancel to fp32 rounding), so the two agree to ~1e-7, not bit for bit."""
import numpy as np
import pytest
import torch

from dad_3dheads_amd import _lib, landmarks, synthetic
from dad_3dheads_amd.head_mesh import HeadMesh
from oracle import flame_ref

pytestmark = pytest.mark.gpu
TOL_V, TOL_PX = 5e-6, 1e-3  # the bars of tests/test_gpu_decode.py (north star: 1e-4 abs)


@pytest.fixture(scope="module")
def meshes(flame_model, static):
    lm = landmarks.canonical("445", static)
    pipe = HeadMesh(flame_model=flame_model, landmarks=lm, static=static, device=0)
    pipe.flame.select_kernel("pipelined")  # raises instead of falling back
    two = HeadMesh(flame_model=flame_model, landmarks=lm, static=static, device=0)
    two.flame.select_kernel("two_role")
    return pipe, two


# ragged and aligned batch sizes around every boundary of the 32-image half-blocks and the 128-image constants rounds
@pytest.mark.parametrize("batch", [1, 3, 16, 17, 31, 32, 33, 48, 63, 64, 65, 96, 127, 128, 129, 200, 259, 385])
@pytest.mark.parametrize("to_2d", [True, False])
def test_pipelined_kernel_agrees_with_the_two_role_kernel(meshes, static, batch, to_2d):
    pipe, two = meshes
    params = synthetic.synthetic_params(batch, seed=7000 + batch)
    a_in, b_in = torch.from_numpy(params).cuda(), torch.from_numpy(params).cuda()
    a = pipe.decode(a_in, to_2d=to_2d, landmarks=True, landmarks_px=True)
    b = two.decode(b_in, to_2d=to_2d, landmarks=True, landmarks_px=True)
    torch.cuda.synchronize()
    assert torch.equal(a_in, b_in)  # tz := 0 written back by both, nothing else touched
    assert float((a["verts3d"] - b["verts3d"]).abs().max()) < 1e-6
    assert float((a["proj"] - b["proj"]).abs().max()) < 3e-4
    lm = torch.from_numpy(landmarks.canonical("445", static)).cuda()
    assert torch.equal(a["lmk_xy"], a["proj"][:, lm, :2])  # the gather is exact in both
    assert torch.equal(a["lmk_px"], a["proj"][:, lm, :2].to(torch.int32))
    d = (a["lmk_px"] - b["lmk_px"]).abs()
    frac = (b["lmk_xy"] - torch.round(b["lmk_xy"])).abs()
    assert bool(((d == 0) | ((d == 1) & (frac < 1e-3))).all())


@pytest.mark.parametrize("batch", [32, 64, 70, 256])
def test_pipelined_kernel_matches_oracle_with_every_flag(meshes, flame_consts, batch):
    pipe, _ = meshes
    params = synthetic.synthetic_params(batch, seed=7100 + batch)
    p = torch.from_numpy(params.copy())
    v_ref = flame_ref.vertices_3d(flame_consts, p).numpy()
    p3_ref = flame_ref.reprojected_vertices(flame_consts, p, to_2d=False).numpy()
    out = pipe.decode(torch.from_numpy(params).cuda(), to_2d=False, flip_z=True)
    only_v = pipe.decode(torch.from_numpy(params).cuda(), proj=False, landmarks=False)  # a null output pointer
    only_p = pipe.decode(torch.from_numpy(params).cuda(), verts3d=False, to_2d=True, landmarks=False)
    torch.cuda.synchronize()
    assert np.abs(out["verts3d"].cpu().numpy() - v_ref).max() < TOL_V
    flipped = p3_ref.copy()
    flipped[..., 2] *= -1.0  # inference/pncc_estimator.py:88
    assert np.abs(out["proj"].cpu().numpy() - flipped).max() < TOL_PX
    assert np.abs(only_v["verts3d"].cpu().numpy() - v_ref).max() < TOL_V
    assert np.abs(only_p["proj"].cpu().numpy() - p3_ref[..., :2]).max() < TOL_PX


def test_duplicate_rows_are_bit_identical_wherever_they_land(meshes):
    """The same params row in different half-blocks, constants rounds and lanes decodes to the same bits (the epilogue and the
    constants code are inlined at several places; they are compiled without fp contraction and with explicit FMAs)."""
    pipe, _ = meshes
    base = synthetic.synthetic_params(5, seed=7200)
    idx = np.array([0, 1, 2, 3, 4] * 60 + [2, 0])  # 302 rows: 10 half-blocks, 3 rounds, ragged end
    out = pipe.decode(torch.from_numpy(base[idx]).cuda(), to_2d=True, landmarks=True, landmarks_px=True)
    torch.cuda.synchronize()
    for k in ("verts3d", "proj", "lmk_xy", "lmk_px"):
        t = out[k]
        for r in range(5):
            rows = t[torch.from_numpy(np.nonzero(idx == r)[0]).cuda()]
            assert torch.equal(rows, rows[:1].expand_as(rows)), (k, r)


def test_unsupported_launches_fall_back_or_refuse(flame_model, static, flame_consts):
    lm = landmarks.canonical("445", static)
    auto = HeadMesh(flame_model=flame_model, landmarks=lm, static=static, device=0)
    params = synthetic.synthetic_params(40, seed=7300)
    z = auto.decode(torch.from_numpy(params).cuda(), proj=False, landmarks=False, zero_rotation=True)["verts3d"]  # two-role kernel
    torch.cuda.synchronize()
    ref = flame_ref.vertices_3d(flame_consts, torch.from_numpy(params.copy()), zero_rotation=True).numpy()
    assert np.abs(z.cpu().numpy() - ref).max() < TOL_V
    auto.flame.select_kernel("pipelined")
    with pytest.raises(_lib.UnsupportedError):
        auto.decode(torch.from_numpy(params).cuda(), proj=False, landmarks=False, zero_rotation=True)


def test_back_to_back_launches_of_changing_sizes_and_a_graph_replay(meshes, flame_consts):
    """The kernel keeps no state between launches (no hand-off buffer, no counters): sizes can change from launch to launch,
    and a captured launch replays (nothing per-launch comes from the host)."""
    pipe, _ = meshes
    sizes = [64, 1, 33, 256, 17, 130, 64]
    ins = [torch.from_numpy(synthetic.synthetic_params(b, seed=7400 + i)).cuda() for i, b in enumerate(sizes)]
    outs = [pipe.decode(x, to_2d=True, landmarks_px=True) for x in ins]
    torch.cuda.synchronize()
    for x, o in zip(ins, outs):
        v_ref = flame_ref.vertices_3d(flame_consts, x.cpu().clone()).numpy()
        assert np.abs(o["verts3d"].cpu().numpy() - v_ref).max() < TOL_V
    p = torch.from_numpy(synthetic.synthetic_params(96, seed=7500)).cuda()
    bufs = {k: torch.empty(s, dtype=dt, device="cuda") for k, s, dt in (("verts3d", (96, 5023, 3), torch.float32),
                                                                          ("proj", (96, 5023, 2), torch.float32))}
    eager = pipe.decode(p.clone(), to_2d=True, landmarks=False)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        pipe.decode(p, to_2d=True, landmarks=False, out=bufs)  # warm-up on the capture stream
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            pipe.decode(p, to_2d=True, landmarks=False, out=bufs)
        for k in bufs:
            bufs[k].zero_()
        g.replay()
        s.synchronize()
    for k in bufs:
        assert torch.equal(bufs[k], eager[k]), k


def test_duplicate_landmark_indices_the_565_list_and_a_forked_handle(flame_model, static):
    """A vertex listed three times (the slot chain beyond the two slots a lane keeps in registers), the last vertex of the partial
    last tile, a list swapped on a live handle (set_landmarks rewrites the per-vertex table), a fork with its own list."""
    idx = np.array([5, 5, 0, 5022, 5, 17, 5021, 5020], dtype=np.int64)
    hm = HeadMesh(flame_model=flame_model, landmarks=idx, static=static, device=0)
    hm.flame.select_kernel("pipelined")
    p = torch.from_numpy(synthetic.synthetic_params(70, seed=7600)).cuda()
    out = hm.decode(p.clone(), landmarks_px=True)
    assert torch.equal(out["lmk_xy"], out["proj"][:, torch.from_numpy(idx).cuda()])
    assert torch.equal(out["lmk_px"], out["lmk_xy"].to(torch.int32))
    twin = hm.fork()  # shares the basis, owns its landmark table
    hm.set_landmarks(landmarks.canonical("565", static))
    out565 = hm.decode(p.clone())
    assert out565["lmk_xy"].shape == (70, 565, 2)
    assert torch.equal(out565["lmk_xy"], out565["proj"][:, torch.from_numpy(landmarks.canonical("565", static)).cuda()])
    out_twin = twin.decode(p.clone())
    assert out_twin["lmk_xy"].shape == (70, len(idx), 2) and torch.equal(out_twin["lmk_xy"], out["lmk_xy"])
    assert torch.equal(out_twin["verts3d"], out["verts3d"])


def test_reference_edge_cases_in_every_half_block_position(meshes, decode_golden):
    """The reference-generated edge rows (tests/golden/make_decode_golden.py: zero jaw, zero expression, scale clamp, degenerate
    6-DoF vectors, 4x coefficients) through the FORCED pipelined kernel, tiled over a batch that puts every one of them into the
    first, a middle and the ragged last half-block and into two constants rounds."""
    pipe, _ = meshes
    g = decode_golden
    edge = g["edge_params"]
    reps = 27  # 162 rows: five full half-blocks and one of two rows
    params = np.tile(edge, (reps, 1))
    dev = torch.from_numpy(params.copy()).cuda()
    out = pipe.decode(dev, to_2d=False, landmarks=False)
    torch.cuda.synchronize()
    sub = g["edge_subset"]
    v = out["verts3d"].cpu().numpy().reshape(reps, edge.shape[0], -1, 3)
    p = out["proj"].cpu().numpy().reshape(reps, edge.shape[0], -1, 3)
    assert np.abs(v[:, :, sub] - g["edge_v3d_sub"][None]).max() < 4 * TOL_V
    err = np.abs(p[:, :, sub] - g["edge_proj3_sub"][None])
    assert err[:, :5].max() < TOL_PX and err[:, 5].max() < 4 * TOL_PX
    assert np.array_equal(v, np.broadcast_to(v[:1], v.shape)) and np.array_equal(p, np.broadcast_to(p[:1], p.shape))  # same bits everywhere
    assert np.array_equal(dev.cpu().numpy(), np.tile(g["edge_params_after"], (reps, 1)))
    assert np.all(v[:, 3] == 0) and np.all(v[:, 4] == 0)  # degenerate 6-DoF -> R = 0 (F.normalize's eps)


@pytest.mark.parametrize("profile", ["crop", "survey"])
def test_full_range_jaw_rotations_and_both_camera_profiles(meshes, flame_consts, profile):
    """The CNN head emits 3 tanh(.) for the jaw's axis-angle too (flame_regression.py:96-104): rotations of up to 5.2 rad, far
    outside the 0.3 rad the synthetic workload uses -- the kernel's own sine / cosine (flame_math.hpp) must hold the bar there --
    plus axis-angle vectors that are exactly zero or denormal-small (smplx's `+ 1e-8` inside the norm) and the SURVEY 8d camera
    (scale + 1 in 0.7..1.3) beside the crop-filling one."""
    pipe, two = meshes
    rng = np.random.default_rng(991)
    params = synthetic.synthetic_params(96, seed=7300, profile=profile)
    params[:, 400:403] = rng.uniform(-3.0, 3.0, (96, 3)).astype(np.float32)
    params[0, 400:403] = 0.0
    params[1, 400:403] = [1e-9, -2e-9, 5e-10]
    params[2, 400:403] = [3.0, 3.0, 3.0]
    params[3, 400:403] = [np.pi, 0.0, 0.0]
    params[4, 400:403] = [0.0, -2.0 * np.pi, 0.0]
    p = torch.from_numpy(params.copy())
    v_ref = flame_ref.vertices_3d(flame_consts, p).numpy()
    p2_ref = flame_ref.reprojected_vertices(flame_consts, p, to_2d=True).numpy()
    for hm in (pipe, two):
        out = hm.decode(torch.from_numpy(params.copy()).cuda(), to_2d=True)
        torch.cuda.synchronize()
        assert np.abs(out["verts3d"].cpu().numpy() - v_ref).max() < TOL_V
        assert np.abs(out["proj"].cpu().numpy() - p2_ref).max() < TOL_PX


def test_huge_jaw_angles_take_the_large_argument_path(meshes, flame_consts):
    """A garbage / early-training jaw vector (the head's output is unbounded before its tanh saturates in fp32 only by range):
    beyond ~8e3 rad the kernel's three-piece argument reduction is no longer exact and flame_math.hpp branches to OCML's sincosf,
    the function the two-role kernel and the oracle use. Angles are powers of two so that the kernel's v_rsq_f32 norm and the
    oracle's sqrt give the SAME fp32 angle (one ulp of 1e6 rad is 0.06 rad: any other choice tests the norm, not the sine)."""
    pipe, two = meshes
    params = synthetic.synthetic_params(40, seed=7700)
    params[0, 400:403] = [2.0 ** 20, 0.0, 0.0]
    params[1, 400:403] = [0.0, 2.0 ** 16, 0.0]
    params[2, 400:403] = [0.0, 0.0, -(2.0 ** 14)]
    params[35, 400:403] = [-(2.0 ** 22), 0.0, 0.0]  # second half-block
    p = torch.from_numpy(params.copy())
    v_ref = flame_ref.vertices_3d(flame_consts, p).numpy()
    for hm in (pipe, two):
        out = hm.decode(torch.from_numpy(params.copy()).cuda(), to_2d=True)
        torch.cuda.synchronize()
        got = out["verts3d"].cpu().numpy()
        assert np.isfinite(got).all()
        assert np.abs(got - v_ref).max() < TOL_V


@pytest.mark.parametrize("kernel", ["pipelined", "two_role"])
def test_poisoned_rows_do_not_leak_into_their_neighbours(meshes, kernel):
    """A serving batch with garbage rows (NaN / +-Inf anywhere in the 413 floats: an upstream network that diverged on one frame)
    must cost exactly those rows: the GEMM's image rows are independent, the jaw-joint columns and the per-image constants are per
    row, so every OTHER row decodes to the same bits as in a clean batch -- whichever half-block, constants round or lane the poisoned
    rows sit in. (The reference gives NaN for those rows too: torch propagates them through einsum and lbs.)"""
    pipe, two = meshes
    hm = pipe if kernel == "pipelined" else two
    clean = synthetic.synthetic_params(200, seed=7800)
    dirty = clean.copy()
    bad = {0: (5, np.nan), 31: (401, np.inf), 32: (405, -np.inf), 63: (412, np.nan), 64: (299, np.inf), 130: (409, np.nan), 199: (350, np.nan)}
    for row, (col, val) in bad.items():
        dirty[row, col] = val
    a = hm.decode(torch.from_numpy(clean).cuda(), to_2d=True, landmarks_px=True)
    b = hm.decode(torch.from_numpy(dirty).cuda(), to_2d=True, landmarks_px=True)
    torch.cuda.synchronize()
    good = torch.tensor([r for r in range(200) if r not in bad], device="cuda")
    for k in ("verts3d", "proj", "lmk_xy", "lmk_px"):
        assert torch.equal(a[k][good], b[k][good]), k
    for row in bad:
        assert not bool(torch.isfinite(b["proj"][row]).all())  # the poisoned row itself is not silently "repaired"


@pytest.mark.parametrize("batch", [35600, 36000])
def test_maximum_sizes_either_side_of_the_2_gb_output_boundary(flame_model, static, flame_consts, batch):
    """The pipelined kernel addresses its outputs with 32-bit buffer offsets and covers launches whose largest output stays below
    2 GB (35 628 rows of [5023, 3] floats); one row more and the C ABI hands the launch to the two-role kernel (64-bit addressing).
    Both sides of that boundary, 4.3 GB per output: every row equals the row it repeats (the batch tiles 64 distinct rows) and the
    first / last / boundary rows match the oracle -- an offset that wrapped would show as a wrong or untouched row near the end."""
    lm = landmarks.canonical("445", static)
    hm = HeadMesh(flame_model=flame_model, landmarks=lm, static=static, device=0)  # automatic kernel choice, as a caller gets it
    base = synthetic.synthetic_params(64, seed=7900)
    reps = (batch + 63) // 64
    params = torch.from_numpy(np.tile(base, (reps, 1))[:batch].copy()).cuda()
    out = hm.decode(params, to_2d=False, landmarks_px=True)
    torch.cuda.synchronize()
    ref_v = flame_ref.vertices_3d(flame_consts, torch.from_numpy(base.copy())).numpy()
    ref_p = flame_ref.reprojected_vertices(flame_consts, torch.from_numpy(base.copy()), to_2d=False).numpy()
    for row in (0, 31, 32, 63, 64, batch // 2, batch - 65, batch - 33, batch - 2, batch - 1):
        assert np.abs(out["verts3d"][row].cpu().numpy() - ref_v[row % 64]).max() < TOL_V, row
        assert np.abs(out["proj"][row].cpu().numpy() - ref_p[row % 64]).max() < TOL_PX, row
    idx = torch.arange(batch, device="cuda") % 64
    for k in ("verts3d", "proj", "lmk_px"):
        first = out[k][:64]
        for lo in range(0, batch, 4096):  # chunked: no second 4 GB tensor
            hi = min(batch, lo + 4096)
            assert torch.equal(out[k][lo:hi], first[idx[lo:hi]]), (k, lo)
    assert bool((params[:, 411] == 0).all())  # tz := 0 written back for every row
    del out
    torch.cuda.empty_cache()
