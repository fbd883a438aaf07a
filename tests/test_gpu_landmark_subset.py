"""GPU: landmark-only launches run on the sub-model of the listed vertices (csrc/capi.cpp: build_landmark_subset; SURVEY 7.1's
landmark-only fast path, BASELINE configs[3]'s per-GPU work). Held to: the same call on a handle pinned to the two-role kernel (whole
mesh) -- bit for bit -- and the CPU oracle within the decode's bars."""
import numpy as np
import pytest
import torch

from dad_3dheads_amd import _lib, landmarks, synthetic
from dad_3dheads_amd.head_mesh import HeadMesh
from oracle import flame_ref

pytestmark = pytest.mark.gpu
TOL_PX = 1e-3


def _pair(flame_model, static, idx):
    auto = HeadMesh(flame_model=flame_model, landmarks=idx, static=static, device=0)
    full = HeadMesh(flame_model=flame_model, landmarks=idx, static=static, device=0)
    full.flame.select_kernel("two_role")  # a pinned handle never takes the sub-model
    return auto, full


def _n_sub(hm):
    return _lib.load().dad3d_flame_num_landmark_vertices(hm.flame._handle)


@pytest.mark.parametrize("which", ["445", "565", "dups", "one"])
@pytest.mark.parametrize("batch", [1, 17, 64, 65, 256, 300])
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
    ref = flame_ref.reprojected_vertices(flame_consts, torch.from_numpy(params.copy()), to_2d=True).numpy()[:, idx]
    assert np.abs(a["lmk_xy"].cpu().numpy() - ref).max() < TOL_PX
    only_px = auto.decode(torch.from_numpy(params.copy()).cuda(), verts3d=False, proj=False, landmarks=False, landmarks_px=True)
    assert torch.equal(only_px["lmk_px"], a["lmk_px"])


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
    assert float((lm_only - out["lmk_xy"]).abs().max()) < 3e-4  # pipelined kernel, whole mesh
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
    assert float((only["lmk_xy"] - whole["lmk_xy"]).abs().max()) < 3e-4  # two kernels: agreement to fp32 rounding
    d = (only["lmk_px"] - whole["lmk_px"]).abs()
    frac = (whole["lmk_xy"] - torch.round(whole["lmk_xy"])).abs()
    assert bool(((d == 0) | ((d == 1) & (frac < TOL_PX))).all())
