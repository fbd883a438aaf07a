"""CPU: the FLAME oracle (oracle/flame_ref.py) against the committed goldens produced by the reference's own
head_mesh.py, and -- when /root/reference is present -- against the reference modules executed live."""
import hashlib

import numpy as np
import pytest
import torch

from dad_3dheads_amd import synthetic
from oracle import flame_ref, reference_runner


def test_model_digest_matches_golden(flame_model, decode_golden):
    # the synthetic model is re-generated from a seed on every machine; the goldens are only valid for this one
    assert synthetic.model_digest(flame_model) == bytes(decode_golden["model_digest"]).hex()


def test_params_are_deterministic(decode_golden):
    assert np.array_equal(synthetic.synthetic_params(2, seed=101), decode_golden["b2_params"])
    assert np.array_equal(synthetic.synthetic_params(64, seed=102), decode_golden["b64_params"])


def test_oracle_bitwise_equals_reference_golden_b2(flame_consts, decode_golden):
    p = torch.from_numpy(decode_golden["b2_params"].copy())
    v3d = flame_ref.vertices_3d(flame_consts, p)
    v3d0 = flame_ref.vertices_3d(flame_consts, p, zero_rotation=True)
    proj = flame_ref.reprojected_vertices(flame_consts, p, to_2d=False)
    assert np.array_equal(v3d.numpy(), decode_golden["b2_v3d"])
    assert np.array_equal(v3d0.numpy(), decode_golden["b2_v3d_zero_rot"])
    assert np.array_equal(proj.numpy(), decode_golden["b2_proj3"])
    assert np.array_equal(p.numpy(), decode_golden["b2_params_after"])  # tz := 0 side effect
    assert (p.numpy()[:, 411] == 0).all() and (decode_golden["b2_params"][:, 411] != 0).all()


def test_oracle_matches_reference_golden_b64(flame_consts, decode_golden, static):
    # torch CPU matmul blocking may differ between B=64 runs on different hosts -> tolerance, not bitwise
    p = torch.from_numpy(decode_golden["b64_params"].copy())
    sub = decode_golden["b64_subset"]
    v3d = flame_ref.vertices_3d(flame_consts, p).numpy()
    proj_t = flame_ref.reprojected_vertices(flame_consts, p, to_2d=True)
    proj = proj_t.numpy()
    assert np.abs(v3d[:, sub] - decode_golden["b64_v3d_sub"]).max() < 1e-6
    assert np.abs(proj[:, sub] - decode_golden["b64_proj_sub"]).max() < 1e-4
    lmk = static["lmk_445"]
    assert np.abs(proj[:, lmk] - decode_golden["b64_lmk_xy"]).max() < 1e-4
    px = flame_ref.gather_landmarks_int(proj_t, lmk)
    diff = px - decode_golden["b64_lmk_px"]
    # int truncation may flip by one only where the float coordinate sits within tolerance of an integer
    frac = np.abs(decode_golden["b64_lmk_xy"] - np.round(decode_golden["b64_lmk_xy"]))
    assert np.all((diff == 0) | ((np.abs(diff) == 1) & (frac < 1e-3)))


def test_oracle_edge_cases_match_golden(flame_consts, decode_golden):
    p = torch.from_numpy(decode_golden["edge_params"].copy())
    sub = decode_golden["edge_subset"]
    v3d = flame_ref.vertices_3d(flame_consts, p).numpy()
    v3d0 = flame_ref.vertices_3d(flame_consts, p, zero_rotation=True).numpy()
    proj = flame_ref.reprojected_vertices(flame_consts, p, to_2d=False).numpy()
    assert np.abs(v3d[:, sub] - decode_golden["edge_v3d_sub"]).max() < 1e-6
    assert np.abs(v3d0[:, sub] - decode_golden["edge_v3d_zero_rot_sub"]).max() < 1e-6
    assert np.abs(proj[:, sub] - decode_golden["edge_proj3_sub"]).max() < 1e-4
    # scale clamp row: s = max(-1.5 + 1, 1e-8) -> the mesh collapses onto (t + 1) * 128
    t = decode_golden["edge_params"][2, 409:411]
    assert np.abs(proj[2, :, :2] - (t + 1.0) * 128.0).max() < 1e-3
    # degenerate 6-DoF rows: R = 0 -> rotated vertices are exactly 0
    assert np.all(v3d[3] == 0) and np.all(v3d[4] == 0)


@pytest.mark.skipif(not reference_runner.reference_available(), reason="reference tree not present on this machine")
def test_oracle_bitwise_equals_live_reference(flame_model, flame_consts):
    hm = reference_runner.load_reference_head_mesh(flame_model)
    for b, seed in ((1, 5), (4, 6)):
        p_ref = torch.from_numpy(synthetic.synthetic_params(b, seed=seed))
        p_or = p_ref.clone()
        with torch.no_grad():
            v_ref = hm.vertices_3d(p_ref)
            pr_ref = hm.reprojected_vertices(p_ref, to_2d=True)
        v = flame_ref.vertices_3d(flame_consts, p_or)
        pr = flame_ref.reprojected_vertices(flame_consts, p_or, to_2d=True)
        assert torch.equal(v, v_ref) and torch.equal(pr, pr_ref) and torch.equal(p_or, p_ref)


def test_rodrigues_zero_pose_is_exact_identity():
    r = flame_ref.batch_rodrigues(torch.zeros(3, 3))
    assert torch.equal(r, torch.eye(3).expand(3, 3, 3))


def test_landmark_lists_known_answers(static):
    # digests recorded in SURVEY.md section 3.2 (int64 little-endian bytes)
    for key, n, total, sha in (("lmk_445", 445, 1099433, "be0bb07f795b2607"), ("lmk_565", 565, 1398969, "08eb437837dd2d40"),
                               ("lmk_191", 191, 450370, "05ef3fc1bd31c95b")):
        a = static[key].astype(np.int64)
        assert len(a) == n and int(a.sum()) == total and len(set(a.tolist())) == n
        assert hashlib.sha256(a.tobytes()).hexdigest().startswith(sha)
    assert static["lmk_445"][:5].tolist() == [570, 694, 3865, 17, 16]
    assert static["lmk_445"][-3:].tolist() == [1437, 1164, 1154]
    assert np.array_equal(static["lmk_191"], static["indices_2d"])
    assert static["faces"].shape == (9976, 3) and static["faces"].max() == 5022
    assert static["faces_wo_ears"].shape == (6270, 3)


def test_predictor_geometry_helpers():
    # config 1 image (766 x 954, W x H): scale 256/954, resized 206 x 256, pads [0, 0, 25, 25]
    pads, scale = flame_ref.get_paddings((954, 766))
    assert pads == [0, 0, 25, 25] and abs(scale - 256 / 954) < 1e-12
    assert flame_ref.py3round(0.5) == 0 and flame_ref.py3round(1.5) == 2 and flame_ref.py3round(2.5) == 2
    assert flame_ref.calculate_paddings(256, 206) == [0, 0, 25, 25]
    p = torch.zeros(1, 413)
    flame_ref.readjust_3dmm(p, pads, scale)
    assert abs(p[0, 412].item() - (1 / scale - 1)) < 1e-6
    assert abs(p[0, 409].item() - ((1 - 25 * 2 / 256) / scale - 1)) < 1e-6


def test_lbs_restatement_agrees_with_an_independent_float64_formulation(flame_model, flame_consts, decode_golden):
    """VERDICT r1 missing #7: `flame_ref.lbs` (the restatement of third-party smplx.lbs that every decode golden passes
    through) against oracle/lbs_independent.py -- written from the SMPL paper, float64, per vertex, scipy's exponential
    map -- on the edge-case rows of the goldens (zero jaw, zero expression, large coefficients with a big jaw rotation) and
    on full-pose inputs (neck, jaw, both eyeballs rotating) that the 413-parameter layout never produces."""
    import torch

    from oracle import flame_ref
    from oracle.lbs_independent import lbs_per_vertex

    fc = flame_consts
    m = flame_model
    posedirs = fc.posedirs.numpy().astype(np.float64)
    args = (fc.v_template.numpy(), fc.shapedirs.numpy(), posedirs, fc.j_regressor.numpy(), fc.parents.numpy(), fc.lbs_weights.numpy())
    rng = np.random.default_rng(12)
    cases = []
    for row in (0, 1, 5):  # zero jaw / zero expression / 4x coefficients + jaw (1.2, -0.7, 0.4)
        p = decode_golden["edge_params"][row]
        pose = np.zeros((5, 3), np.float32)
        pose[2] = p[400:403]
        cases.append((p[:400].copy(), pose))
    for _ in range(2):  # every joint rotates (the generic 36-feature path), root included
        cases.append(((rng.standard_normal(400) * 0.6).astype(np.float32), (rng.standard_normal((5, 3)) * 0.5).astype(np.float32)))
    for betas, pose in cases:
        v32, j32 = flame_ref.lbs(torch.from_numpy(betas)[None], torch.from_numpy(pose).reshape(1, 15), fc.v_template, fc.shapedirs,
                                 fc.posedirs, fc.j_regressor, fc.parents, fc.lbs_weights)
        v64, j64 = lbs_per_vertex(betas, pose, *args)
        scale = max(1.0, float(np.abs(v64).max()))
        assert np.abs(v32[0].numpy() - v64).max() < 1e-6 * scale, np.abs(v32[0].numpy() - v64).max()
        assert np.abs(j32[0].numpy() - j64).max() < 1e-6 * scale
    assert m.v_template.shape == (5023, 3)


def test_float64_pixel_arbiter_is_the_independent_formulation_and_agrees_with_the_oracle(flame_consts):
    """oracle/lbs_independent.lbs_vertex_subset (the float64 arbiter of tests/test_gpu_parity_pixels.py) is the per-vertex
    formulation restricted to a vertex list -- equal to `lbs_per_vertex` there to float64 rounding -- and
    `projected_pixels_subset` (lbs + flame.py:224-228 + head_mesh.py:39-45 in float64) agrees with the float32 oracle to a few ulp(256)."""
    import torch

    from dad_3dheads_amd import synthetic
    from oracle import flame_ref
    from oracle.lbs_independent import lbs_per_vertex, lbs_vertex_subset, projected_pixels_subset

    fc = flame_consts
    args = (fc.v_template.numpy(), fc.shapedirs.numpy(), fc.posedirs.numpy().astype(np.float64), fc.j_regressor.numpy(), fc.parents.numpy(),
            fc.lbs_weights.numpy())
    ids = np.array([0, 3, 1777, 3931, 4477, 5022])
    rng = np.random.default_rng(5)
    betas = (rng.standard_normal(400) * 0.6).astype(np.float32)
    pose = (rng.standard_normal((5, 3)) * 0.5).astype(np.float32)
    full, _ = lbs_per_vertex(betas, pose, *args)
    assert np.abs(lbs_vertex_subset(betas, pose, *args, ids) - full[ids]).max() < 1e-12
    for profile in ("crop", "survey"):
        p = synthetic.synthetic_params(3, seed=31, profile=profile)
        ref = flame_ref.reprojected_vertices(fc, torch.from_numpy(p.copy()), to_2d=False).numpy()
        got = projected_pixels_subset(p[1], ids, *args)
        assert np.abs(got - ref[1, ids]).max() < 2e-4
