"""CPU: the per-image half of the differentiable decode (`autograd.pose_chain`) and the layout of its 72 constants,
against the oracle's forward AND the oracle's torch autograd -- no GPU, no library call. The per-vertex half runs in the
HIP library and is checked in tests/test_gpu_autograd.py."""
import numpy as np
import pytest
import torch

from dad_3dheads_amd import autograd as ag
from dad_3dheads_amd import synthetic
from dad_3dheads_amd.flame import FLAME_CONSTS
from oracle import flame_ref


@pytest.fixture(scope="module")
def tables(flame_model):
    m = flame_model
    npose = np.asarray(m.posedirs).shape[-1]
    parents = np.asarray(m.kintree_table)[0].astype(np.int64)
    parents[0] = -1
    f32 = lambda a: np.ascontiguousarray(np.asarray(a), dtype=np.float32)  # noqa: E731
    jr = m.J_regressor.toarray() if hasattr(m.J_regressor, "toarray") else m.J_regressor
    return ag.DecodeTables(f32(m.v_template), f32(m.shapedirs), f32(np.reshape(np.asarray(m.posedirs), [-1, npose]).T),
                           f32(jr), parents, f32(m.weights), "cpu")


@pytest.mark.parametrize("zero_rot,to_2d", [(False, True), (True, False)])
def test_chain_plus_vertex_stage_equals_oracle_forward_and_gradient(tables, flame_consts, zero_rot, to_2d):
    c = flame_consts
    params = torch.from_numpy(synthetic.synthetic_params(3, seed=77))
    gen = torch.Generator().manual_seed(5)
    wv, wp = torch.randn((3, 5023, 3), generator=gen), torch.randn((3, 5023, 2 if to_2d else 3), generator=gen)

    p_ref = params.clone().requires_grad_(True)
    v_ref = flame_ref.vertices_3d(c, p_ref, zero_rotation=zero_rot)
    pr_ref = flame_ref.reprojected_vertices(c, p_ref * 1.0, to_2d=to_2d)  # * 1.0: the in-place tz write needs a non-leaf
    ((v_ref * wv).sum() + (pr_ref * wp).sum() * 1e-2).backward()

    p = params.clone().requires_grad_(True)
    chain = ag.pose_chain(tables, FLAME_CONSTS, p)
    assert chain["inputs"].shape == (3, 436) and chain["consts"].shape == (3, ag.N_CONSTS)
    v, pr = ag.vertex_stage(tables, chain["inputs"], chain["consts"], zero_rot=zero_rot, to_2d=to_2d)
    ((v * wv).sum() + (pr * wp).sum() * 1e-2).backward()

    assert (v - v_ref).abs().max() < 5e-6
    assert (pr - pr_ref).abs().max() < 1e-3
    scale = p_ref.grad.abs().max()
    assert (p.grad - p_ref.grad).abs().max() < 2e-4 * scale
    assert float(p.grad[:, 411].abs().max()) == 0.0  # translation z never reaches an output (head_mesh.py:41)


def test_loss_helpers_match_the_reference_definitions(tmp_path):
    from dad_3dheads_amd.losses import indices_reweighing, normalize_to_cube

    np.save(tmp_path / "a.npy", np.array([1, 2, 3]))
    np.save(tmp_path / "b.npy", np.array([7, 8]))
    cfg = {"weights": {"b": 0.5, "a": 2.0}, "flame_indices": {"folder": str(tmp_path), "files": {"a": "a.npy", "b": "b.npy", "c": "c.npy"}}}
    w, idx = indices_reweighing(cfg)  # order of `files`, only the regions that have a weight (utils.py:108-117)
    assert w == [2.0, 0.5] and [i.tolist() for i in idx] == [[1, 2, 3], [7, 8]]
    v = torch.tensor([[[0.0, 0.0, 0.0], [2.0, 4.0, 1.0], [1.0, 1.0, 1.0]]])
    n = normalize_to_cube(v)  # model/utils.py:55-68: min -> 0, centre, divide by the largest half extent
    assert torch.allclose(n, torch.tensor([[[-0.5, -1.0, -0.25], [0.5, 1.0, 0.25], [0.0, -0.5, 0.25]]]))
    assert normalize_to_cube(v[0]).shape == (1, 3, 3)


def test_loss_mirrors_equal_the_references_own_loss_modules(tmp_path, flame_model, flame_consts):
    """Live reference (authoring container only): `model_training/losses/vertices_3d_loss.py` and `reprojection_loss.py`
    executed unmodified on the synthetic model, against the same losses assembled from this package's helpers over the
    oracle decode -- values and d/d(params). The GPU tests compare the HIP path with exactly that assembly."""
    from oracle import reference_runner as rr

    if not rr.reference_available():
        pytest.skip("reference tree not present")
    from dad_3dheads_amd.losses import indices_reweighing, normalize_to_cube

    RefV3D, RefRep = rr.load_reference_losses(flame_model)
    np.save(tmp_path / "a.npy", np.arange(0, 5023, 7))
    np.save(tmp_path / "b.npy", np.arange(3000, 3600))
    cfg = {"weights": {"a": 1.0, "b": 0.5}, "flame_indices": {"folder": str(tmp_path), "files": {"a": "a.npy", "b": "b.npy"}}}
    weights, indices = indices_reweighing(cfg)
    batch = 4  # not 3: the reference's `torch.cross` without `dim` (model/utils.py:98-99) then crosses over the batch axis
    params = torch.from_numpy(synthetic.synthetic_params(batch, seed=31))
    tgt3d = flame_ref.vertices_3d(flame_consts, torch.from_numpy(synthetic.synthetic_params(batch, seed=32)), zero_rotation=True)
    tgt2d = flame_ref.reprojected_vertices(flame_consts, torch.from_numpy(synthetic.synthetic_params(batch, seed=33)))
    for crit, fn in (("l1", torch.nn.L1Loss()), ("l2", torch.nn.MSELoss()), ("smooth_l1", torch.nn.SmoothL1Loss())):
        p_ref = params.clone().requires_grad_(True)
        val_ref = RefV3D(crit, batch, FLAME_CONSTS, cfg)(p_ref * 1.0, tgt3d) + RefRep(crit, batch, FLAME_CONSTS, 256, cfg)(p_ref * 1.0, tgt2d) * 1e-2
        val_ref.backward()
        p = params.clone().requires_grad_(True)
        v = flame_ref.vertices_3d(flame_consts, p * 1.0, zero_rotation=True)
        pr = flame_ref.reprojected_vertices(flame_consts, p * 1.0)
        val = torch.stack([fn(normalize_to_cube(v[:, i]), normalize_to_cube(tgt3d[:, i])) * w for w, i in zip(weights, indices)]).sum() \
            + torch.stack([fn(pr[:, i], tgt2d[:, i]) * w for w, i in zip(weights, indices)]).sum() * 1e-2
        val.backward()
        assert float(val.detach()) == float(val_ref.detach())
        assert torch.equal(p.grad, p_ref.grad)


def test_losses_over_the_oracle_reproduce_the_reference_goldens(flame_consts):
    """tests/golden/loss_golden.npz holds values and gradients of the reference's OWN loss modules
    (tests/golden/make_loss_golden.py); the losses assembled from this package's helpers over the oracle decode -- the
    assembly the GPU tests hold the HIP path to -- reproduce them. Runs anywhere (no reference tree needed)."""
    import os

    from dad_3dheads_amd.losses import indices_reweighing, normalize_to_cube

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loss_golden.npz"))
    batch, seeds = int(g["batch"]), [int(x) for x in g["seeds"]]
    weights, indices = indices_reweighing(([float(w) for w in g["region_weights"]], [g["region_" + str(n)] for n in g["region_names"]]))
    params = torch.from_numpy(synthetic.synthetic_params(batch, seed=seeds[0]))
    tgt3d = flame_ref.vertices_3d(flame_consts, torch.from_numpy(synthetic.synthetic_params(batch, seed=seeds[1])), zero_rotation=True)
    tgt2d = flame_ref.reprojected_vertices(flame_consts, torch.from_numpy(synthetic.synthetic_params(batch, seed=seeds[2])))
    for crit, fn in (("l1", torch.nn.L1Loss()), ("l2", torch.nn.MSELoss()), ("smooth_l1", torch.nn.SmoothL1Loss())):
        p = params.clone().requires_grad_(True)
        v = flame_ref.vertices_3d(flame_consts, p * 1.0, zero_rotation=True)
        v3 = torch.stack([fn(normalize_to_cube(v[:, i]), normalize_to_cube(tgt3d[:, i])) * w for w, i in zip(weights, indices)]).sum()
        pr = flame_ref.reprojected_vertices(flame_consts, p * 1.0)
        rp = torch.stack([fn(pr[:, i], tgt2d[:, i]) * w for w, i in zip(weights, indices)]).sum()
        (g3,) = torch.autograd.grad(v3, p, retain_graph=True)
        (g2,) = torch.autograd.grad(rp, p)
        # bit-identical on the authoring host; a different CPU may pick other BLAS kernels, hence a few ulp of slack
        assert abs(float(v3.detach()) - float(g[crit + "_vertices3d"])) <= 1e-6 * abs(float(g[crit + "_vertices3d"]))
        assert abs(float(rp.detach()) - float(g[crit + "_reprojection"])) <= 1e-6 * abs(float(g[crit + "_reprojection"]))
        for mine, ref in ((g3.numpy(), g[crit + "_vertices3d_grad"]), (g2.numpy(), g[crit + "_reprojection_grad"])):
            assert np.abs(mine - ref).max() <= 1e-5 * np.abs(ref).max()
