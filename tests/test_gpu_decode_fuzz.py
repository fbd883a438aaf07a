"""GPU: hypothesis-driven decode calls -- any batch size, any subset of outputs, either projection width, z flip, the side effect on
or off, landmark lists with duplicates and of any length, both camera profiles, any of the four kernels -- against the CPU oracle through the
same Python surface a caller uses. The fixed-case tests pin the known boundaries; this one looks for interplay between options
(a null output pointer with landmarks on, a ragged last half-block with a one-entry landmark list, ...)."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from dad_3dheads_amd import synthetic
from dad_3dheads_amd.head_mesh import HeadMesh
from oracle import flame_ref

pytestmark = pytest.mark.gpu
TOL_V, TOL_PX = 5e-6, 1e-3


@pytest.fixture(scope="module")
def heads(flame_model, static):
    out = {}
    for kernel in ("auto", "two_role", "split_bf16", "split_f16"):
        hm = HeadMesh(flame_model=flame_model, landmarks=np.arange(3, dtype=np.int64), static=static, device=0)
        hm.flame.select_kernel(kernel)
        out[kernel] = hm
    return out


@settings(max_examples=240, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(seed=st.integers(0, 2**31 - 1), batch=st.one_of(st.integers(1, 70), st.integers(71, 330)), kernel=st.sampled_from(["auto", "two_role", "split_bf16", "split_f16"]),
       want_v=st.booleans(), want_p=st.booleans(), to_2d=st.booleans(), flip_z=st.booleans(), want_lx=st.booleans(), want_lp=st.booleans(),
       mutate=st.booleans(), n_lmk=st.integers(1, 600), profile=st.sampled_from(["crop", "survey"]))
def test_any_decode_call_matches_the_oracle(heads, flame_consts, seed, batch, kernel, want_v, want_p, to_2d, flip_z, want_lx, want_lp, mutate,
                                            n_lmk, profile):
    if not (want_v or want_p or want_lx or want_lp):
        want_v = True
    flip_z = flip_z and not to_2d  # DAD3D_FLIP_Z needs a 3-component projection (the C ABI refuses the combination)
    hm = heads[kernel]
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, 5023, n_lmk).astype(np.int64)
    idx[rng.integers(0, n_lmk)] = idx[0]  # at least one duplicate (when n_lmk > 1)
    hm.set_landmarks(idx)
    params = synthetic.synthetic_params(batch, seed=seed % 100000, profile=profile)
    dev = torch.from_numpy(params.copy()).cuda()
    out = hm.decode(dev, verts3d=want_v, proj=want_p, to_2d=to_2d, landmarks=want_lx, landmarks_px=want_lp, flip_z=flip_z, mutate=mutate)
    torch.cuda.synchronize()
    p = torch.from_numpy(params.copy())
    v_ref = flame_ref.vertices_3d(flame_consts, p).numpy()
    pr_ref = flame_ref.reprojected_vertices(flame_consts, p, to_2d=False).numpy()  # zeroes tz in `p`, like the reference
    if flip_z:
        pr_ref[..., 2] *= -1.0
    assert set(out) == {k for k, w in (("verts3d", want_v), ("proj", want_p), ("lmk_xy", want_lx), ("lmk_px", want_lp)) if w}
    if want_v:
        assert np.abs(out["verts3d"].cpu().numpy() - v_ref).max() < TOL_V
    if want_p:
        got = out["proj"].cpu().numpy()
        assert got.shape[-1] == (2 if to_2d else 3) and np.abs(got - pr_ref[..., : got.shape[-1]]).max() < TOL_PX
    if want_lx:
        lx = out["lmk_xy"].cpu().numpy()
        assert lx.shape == (batch, n_lmk, 2) and np.abs(lx - pr_ref[:, idx, :2]).max() < TOL_PX
        if want_p:
            assert np.array_equal(lx, out["proj"].cpu().numpy()[:, idx, :2])  # the gather itself is exact
    if want_lp:
        lp = out["lmk_px"].cpu().numpy()
        want = pr_ref[:, idx, :2].astype(int)
        diff = lp != want
        near = np.abs(pr_ref[:, idx, :2] - np.round(pr_ref[:, idx, :2])) < TOL_PX
        assert lp.shape == (batch, n_lmk, 2) and bool(near[diff].all()) and np.abs(lp - want).max() <= 1
    after = dev.cpu().numpy()
    assert np.array_equal(after, p.numpy() if mutate else params)  # tz := 0 exactly when asked, nothing else ever touched
