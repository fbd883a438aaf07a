"""GPU: landmark-only launches run on the sub-model of the listed vertices (csrc/capi.cpp: build_landmark_subset; SURVEY 7.1's
landmark-only fast path, BASELINE configs[3]'s per-GPU work) -- on the SAME kernel a full-output launch of the handle takes, with the
batch cut into chunks across workgroups. Held to: the landmark rows of a full-output launch of the same handle -- bit for bit, at every
batch size --, a whole-mesh launch of a pinned handle (same kernel, no sub-model) -- bit for bit --, the two-role kernel to fp32
rounding, and the CPU oracle within the decode's bars."""
import numpy as np
import pytest
import torch

from dad_3dheads_amd import _lib, landmarks, synthetic
from dad_3dheads_amd.head_mesh import HeadMesh
from oracle import flame_ref

pytestmark = pytest.mark.gpu
TOL_PX = 1e-3


def _pair(flame_model, static, idx, pin="pipelined"):
    auto = HeadMesh(flame_model=flame_model, landmarks=idx, static=static, device=0)
    full = HeadMesh(flame_model=flame_model, landmarks=idx, static=static, device=0)
    full.flame.select_kernel(pin)  # a pinned handle never takes the sub-model
    return auto, full


def _n_sub(hm):
    return _lib.load().dad3d_flame_num_landmark_vertices(hm.flame._handle)


@pytest.mark.parametrize("which", ["445", "565", "dups", "one"])
@pytest.mark.parametrize("batch", [1, 17, 32, 33, 64, 65, 256, 300, 1000])
def test_landmark_only_launch_equals_the_whole_mesh_launch(flame_model, flame_consts, static, which, batch):
    idx = {"445": landmarks.canonical("445", static), "565": landmarks.canonical("565", static),
           "dups": np.array([5, 5, 0, 5022, 5, 17, 5021, 5020, 0], dtype=np.int64), "one": np.array([4711], dtype=np.int64)}[which]
    auto, full = _pair(flame_model, static, idx)
    assert _n_sub(auto) == len(np.unique(idx)) and _n_sub(full) == len(np.unique(idx))  # built for both, used by the unpinned one
    params = synthetic.synthetic_params(batch, seed=8100 + batch)
    a_in, f_in = torch.from_numpy(params.copy()).cuda(), torch.from_numpy(params.copy()).cuda()
    a = auto.decode(a_in, verts3d=False, proj=False, landmarks=True, landmarks_px=True, mutate=True)
    f = full.decode(f_in, verts3d=False, proj=False, landmarks=True, landmarks_px=True, mutate=True)
    torch.cuda.synchronize()
    assert set(a) == {"lmk_xy", "lmk_px"}
    assert torch.equal(a["lmk_xy"], f["lmk_xy"]) and torch.equal(a["lmk_px"], f["lmk_px"])  # same arithmetic, same bits
    assert torch.equal(a["lmk_px"], a["lmk_xy"].to(torch.int32))
    assert torch.equal(a_in, f_in) and bool((a_in[:, 411] == 0).all())  # tz := 0 written back by the sub-model launch as well
    # ... and the landmark outputs of a full-output launch of the SAME handle (default kernel, whole mesh), both projection widths
    for to_2d in (True, False):
        whole = auto.decode(torch.from_numpy(params.copy()).cuda(), to_2d=to_2d, landmarks=True, landmarks_px=True)
        assert torch.equal(whole["lmk_xy"], a["lmk_xy"]) and torch.equal(whole["lmk_px"], a["lmk_px"])
        assert torch.equal(whole["lmk_xy"], whole["proj"][:, torch.from_numpy(idx).cuda(), :2])
    ref = flame_ref.reprojected_vertices(flame_consts, torch.from_numpy(params.copy()), to_2d=True).numpy()[:, idx]
    assert np.abs(a["lmk_xy"].cpu().numpy() - ref).max() < TOL_PX
    only_px = auto.decode(torch.from_numpy(params.copy()).cuda(), verts3d=False, proj=False, landmarks=False, landmarks_px=True)
    assert torch.equal(only_px["lmk_px"], a["lmk_px"])


@pytest.mark.parametrize("batch", [5, 64, 200])
def test_landmark_only_launch_against_the_two_role_kernel(flame_model, static, batch):
    """Another kernel (OCML sine/cosine of the jaw, another summation order of the joints): agreement to fp32 rounding, integer pixels
    equal except where the coordinate sits on an integer to within that rounding."""
    idx = landmarks.canonical("445", static)
    auto, two = _pair(flame_model, static, idx, pin="two_role")
    params = synthetic.synthetic_params(batch, seed=8150 + batch)
    a = auto.decode(torch.from_numpy(params.copy()).cuda(), verts3d=False, proj=False, landmarks=True, landmarks_px=True)
    t = two.decode(torch.from_numpy(params.copy()).cuda(), verts3d=False, proj=False, landmarks=True, landmarks_px=True)
    assert float((a["lmk_xy"] - t["lmk_xy"]).abs().max()) < 3e-4
    d = (a["lmk_px"] - t["lmk_px"]).abs()
    frac = (t["lmk_xy"] - torch.round(t["lmk_xy"])).abs()
    assert bool(((d == 0) | ((d == 1) & (frac < TOL_PX))).all())


def test_subset_follows_the_list_forks_and_pins(flame_model, static):
    idx = landmarks.canonical("445", static)
    hm = HeadMesh(flame_model=flame_model, landmarks=idx, static=static, device=0)
    p = torch.from_numpy(synthetic.synthetic_params(70, seed=8200)).cuda()
    first = hm.decode(p.clone(), verts3d=False, proj=False, landmarks=True)["lmk_xy"].clone()
    twin = hm.fork()  # shares the sub-model's constants, owns its lists and hand-off buffers
    assert _n_sub(twin) == _n_sub(hm) == len(np.unique(idx))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        from_twin = twin.decode(p.clone(), verts3d=False, proj=False, landmarks=True)["lmk_xy"]
    s.synchronize()
    assert torch.equal(from_twin, first)
    # a list naming more than a third of the mesh: no sub-model, the launch decodes the whole mesh -- same results either way
    wide = np.arange(0, 5023, 2, dtype=np.int64)
    hm.set_landmarks(wide)
    assert _n_sub(hm) == 0 and _n_sub(twin) == len(np.unique(idx))  # the fork keeps its own list
    out = hm.decode(p.clone(), landmarks=True)
    assert torch.equal(out["lmk_xy"], out["proj"][:, torch.from_numpy(wide).cuda()])
    lm_only = hm.decode(p.clone(), verts3d=False, proj=False, landmarks=True)["lmk_xy"]
    assert torch.equal(lm_only, out["lmk_xy"])  # no sub-model: the whole mesh on the same kernel
    hm.set_landmarks(idx)  # back: rebuilt
    assert _n_sub(hm) == len(np.unique(idx))
    again = hm.decode(p.clone(), verts3d=False, proj=False, landmarks=True)["lmk_xy"]
    assert torch.equal(again, first)
    hm.set_landmarks(np.zeros((0,), dtype=np.int64))
    assert _n_sub(hm) == 0


def test_full_output_launches_are_untouched_and_agree_with_the_landmark_only_one(flame_model, static):
    idx = landmarks.canonical("445", static)
    hm = HeadMesh(flame_model=flame_model, landmarks=idx, static=static, device=0)
    p = torch.from_numpy(synthetic.synthetic_params(128, seed=8300)).cuda()
    whole = hm.decode(p.clone(), landmarks=True, landmarks_px=True)  # pipelined kernel, every output
    only = hm.decode(p.clone(), verts3d=False, proj=False, landmarks=True, landmarks_px=True)
    torch.cuda.synchronize()
    assert torch.equal(whole["lmk_xy"], whole["proj"][:, torch.from_numpy(idx).cuda()])
    assert torch.equal(only["lmk_xy"], whole["lmk_xy"]) and torch.equal(only["lmk_px"], whole["lmk_px"])  # one arithmetic


def test_landmark_only_launch_is_capturable_after_a_full_output_warm_up(flame_model, static):
    """the sub-model needs no scratch on the pipelined kernel, so a graph may capture a landmark-only launch whose warm-up
    was a full-output launch of the same batch."""
    idx = landmarks.canonical("445", static)
    hm = HeadMesh(flame_model=flame_model, landmarks=idx, static=static, device=0)
    p = torch.from_numpy(synthetic.synthetic_params(192, seed=8400)).cuda()
    want = hm.decode(p.clone(), landmarks=True)["lmk_xy"].clone()  # warm-up: every output
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    static_in = p.clone()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            out = hm.decode(static_in, verts3d=False, proj=False, landmarks=True)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out["lmk_xy"], want)
