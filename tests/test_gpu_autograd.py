"""GPU: gradients through `HeadMesh.vertices_3d` / `reprojected_vertices` (HIP forward, HIP + rocBLAS backward) against
torch autograd over the CPU oracle -- the way the reference's own losses obtain them (vertices_3d_loss.py:41,
reprojection_loss.py:33). Tolerance: 2e-4 of the largest gradient entry (fp32 sums over 15069 terms in two different
orders); the forward values keep the decode's tolerances."""
import ctypes as C

import numpy as np
import pytest
import torch

from dad_3dheads_amd import _lib, landmarks, synthetic
from dad_3dheads_amd.flame import FLAME_CONSTS
from dad_3dheads_amd.head_mesh import HeadMesh
from dad_3dheads_amd.losses import ReprojectionLoss, Vertices3DLoss, normalize_to_cube
from oracle import flame_ref

pytestmark = pytest.mark.gpu
RTOL = 2e-4


@pytest.fixture(scope="module")
def hm(flame_model, static):
    return HeadMesh(flame_model=flame_model, landmarks=landmarks.canonical("445", static), static=static, device=0)


def close(g, g_ref):
    g, g_ref = g.detach().cpu(), g_ref.detach().cpu()
    return float((g - g_ref).abs().max()) <= RTOL * float(g_ref.abs().max())


@pytest.mark.parametrize("batch,zero_rot", [(3, True), (3, False), (70, True)])
def test_vertices_3d_gradient_matches_oracle_autograd(hm, flame_consts, batch, zero_rot):
    params = torch.from_numpy(synthetic.synthetic_params(batch, seed=300 + batch))
    w = torch.randn((batch, 5023, 3), generator=torch.Generator().manual_seed(1))
    p_ref = params.clone().requires_grad_(True)
    (flame_ref.vertices_3d(flame_consts, p_ref, zero_rotation=zero_rot) * w).sum().backward()

    p = params.clone().cuda().requires_grad_(True)
    v = hm.vertices_3d(p, zero_rotation=zero_rot)
    assert v.requires_grad and v.device == p.device and v.shape == (batch, 5023, 3)
    with torch.no_grad():
        # zero_rotation: both take the two-role kernel (the same launch, bit for bit); otherwise the inference call takes the
        # pipelined kernel at every batch size (round 5: no crossover table) and the two agree to fp32 rounding
        infer = hm.vertices_3d(p.detach(), zero_rotation=zero_rot)
        assert torch.equal(v, infer) if zero_rot else float((v - infer).abs().max()) < 1e-6
    (v * w.cuda()).sum().backward()
    assert close(p.grad, p_ref.grad)
    assert float(p.grad[:, 409:412].abs().max()) == 0.0 and float(p.grad[:, 412].abs().max()) == 0.0  # t, s unused here


@pytest.mark.parametrize("to_2d", [True, False])
def test_reprojection_gradient_and_in_place_side_effect(hm, flame_consts, to_2d):
    params = torch.from_numpy(synthetic.synthetic_params(5, seed=41))
    w = torch.randn((5, 5023, 2 if to_2d else 3), generator=torch.Generator().manual_seed(2))
    p_ref = params.clone().requires_grad_(True)
    q_ref = p_ref * 1.0  # the network output of the reference's training loop: a non-leaf the method writes into
    (flame_ref.reprojected_vertices(flame_consts, q_ref, to_2d=to_2d) * w).sum().backward()

    p = params.clone().cuda().requires_grad_(True)
    q = p * 1.0
    proj = hm.reprojected_vertices(q, to_2d=to_2d)
    assert float(q.detach()[:, 411].abs().max()) == 0.0  # head_mesh.py:41
    (proj * w.cuda()).sum().backward()
    assert close(p.grad, p_ref.grad)
    assert float(p.grad[:, 411].abs().max()) == 0.0
    with pytest.raises(RuntimeError):  # a leaf that requires grad cannot be written in place -- in the reference either
        hm.reprojected_vertices(params.clone().cuda().requires_grad_(True))


def test_both_losses_on_one_prediction_tensor(hm, flame_consts):
    """The training loop hands the SAME network output to both losses: vertices_3d saves it, reprojected_vertices then
    writes tz := 0 into it in place -- that must not invalidate the first graph."""
    params = torch.from_numpy(synthetic.synthetic_params(4, seed=21))
    gen = torch.Generator().manual_seed(6)
    wv, wp = torch.randn((4, 5023, 3), generator=gen), torch.randn((4, 5023, 2), generator=gen)
    p_ref = params.clone().requires_grad_(True)
    q_ref = p_ref * 1.0
    v_ref = flame_ref.vertices_3d(flame_consts, q_ref, zero_rotation=True)
    pr_ref = flame_ref.reprojected_vertices(flame_consts, q_ref)
    ((v_ref * wv).sum() + (pr_ref * wp).sum() * 1e-2).backward()
    p = params.clone().cuda().requires_grad_(True)
    q = p * 1.0
    v = hm.vertices_3d(q, zero_rotation=True)
    pr = hm.reprojected_vertices(q)
    ((v * wv.cuda()).sum() + (pr * wp.cuda()).sum() * 1e-2).backward()
    assert close(p.grad, p_ref.grad)


def test_cpu_tensors_round_trip_and_backward_is_reproducible(hm, flame_consts):
    params = torch.from_numpy(synthetic.synthetic_params(4, seed=9))
    w = torch.randn((4, 5023, 3), generator=torch.Generator().manual_seed(3))
    grads = []
    for _ in range(2):
        p = params.clone().requires_grad_(True)  # CPU leaf, like predictor-side callers: staged through the GPU
        v = hm.vertices_3d(p)
        assert v.device.type == "cpu"
        (v * w).sum().backward()
        grads.append(p.grad.clone())
    assert torch.equal(grads[0], grads[1])  # one workgroup per image, no atomics
    p_ref = params.clone().requires_grad_(True)
    (flame_ref.vertices_3d(flame_consts, p_ref) * w).sum().backward()
    assert close(grads[0], p_ref.grad)
    # FLAMELayer.forward (losses / pncc call style) carries the same graph
    p = params.clone().cuda().requires_grad_(True)
    (hm.flame.forward(hm.flame_params(p), zero_rot=True) * w.cuda()).sum().backward()
    p_ref = params.clone().requires_grad_(True)
    (flame_ref.vertices_3d(flame_consts, p_ref, zero_rotation=True) * w).sum().backward()
    assert close(p.grad, p_ref.grad)


def test_reference_losses_values_and_gradients(flame_model, flame_consts, static):
    regions = ([1.0, 0.5, 2.0], [np.arange(0, 5023, 7), landmarks.canonical("445", static)[:200], np.arange(3000, 3600)])
    kw = dict(flame_model=flame_model, static=static, device=0)
    batch = 6
    params = torch.from_numpy(synthetic.synthetic_params(batch, seed=11))
    tgt3d = flame_ref.vertices_3d(flame_consts, torch.from_numpy(synthetic.synthetic_params(batch, seed=12)), zero_rotation=True)
    tgt2d = flame_ref.reprojected_vertices(flame_consts, torch.from_numpy(synthetic.synthetic_params(batch, seed=13)))

    for crit in ("l1", "l2", "smooth_l1"):
        fn = {"l1": torch.nn.L1Loss, "l2": torch.nn.MSELoss, "smooth_l1": torch.nn.SmoothL1Loss}[crit]()
        # Vertices3DLoss (vertices_3d_loss.py:31-49)
        loss = Vertices3DLoss(crit, batch, FLAME_CONSTS, regions, **kw)
        p = params.clone().cuda().requires_grad_(True)
        val = loss(p * 1.0, tgt3d.cuda())
        val.backward()
        p_ref = params.clone().requires_grad_(True)
        v_ref = flame_ref.vertices_3d(flame_consts, p_ref * 1.0, zero_rotation=True)
        ref = torch.stack([fn(normalize_to_cube(v_ref[:, i]), normalize_to_cube(tgt3d[:, i])) * w for w, i in zip(*regions)]).sum()
        ref.backward()
        assert abs(float(val.detach()) - float(ref.detach())) <= 1e-5 * max(1.0, abs(float(ref.detach())))
        assert close(p.grad, p_ref.grad)
        # ReprojectionLoss (reprojection_loss.py:22-46), list target form
        loss = ReprojectionLoss(crit, batch, FLAME_CONSTS, 256, regions, **kw)
        p = params.clone().cuda().requires_grad_(True)
        val = loss(p * 1.0, [tgt2d.cuda()])
        val.backward()
        p_ref = params.clone().requires_grad_(True)
        pr_ref = flame_ref.reprojected_vertices(flame_consts, p_ref * 1.0)
        ref = torch.stack([fn(pr_ref[:, i], tgt2d[:, i]) * w for w, i in zip(*regions)]).sum()
        ref.backward()
        assert abs(float(val.detach()) - float(ref.detach())) <= 1e-4 * max(1.0, abs(float(ref.detach())))
        assert close(p.grad, p_ref.grad)
    with pytest.raises(ValueError, match="Unsupported discrepancy loss type"):
        Vertices3DLoss("huber", batch, FLAME_CONSTS, regions, **kw)


def test_c_abi_backward_argument_errors(hm):
    lib = _lib.load()
    buf = torch.zeros((1, 5023, 3), device="cuda")
    c72 = torch.zeros((1, 72), device="cuda")
    h = hm.flame._handle
    assert lib.dad3d_flame_decode_backward(h, 0, 0, None, None, None, None, None, None, None) == 0  # empty batch
    st = lib.dad3d_flame_decode_backward(h, 1, 0, c72.data_ptr(), buf.data_ptr(), None, None, buf.data_ptr(), c72.data_ptr(), None)
    assert st != 0 and b"no upstream gradient" in lib.dad3d_last_error()
    st = lib.dad3d_flame_decode_backward(None, 1, 0, c72.data_ptr(), buf.data_ptr(), buf.data_ptr(), None, buf.data_ptr(), c72.data_ptr(), None)
    assert st != 0


def _configs(flame_model, static):
    """(HeadMesh, params) for the shipped constants, a full pose (neck + eyeballs) and narrow shape / expression widths."""
    rng = np.random.default_rng(3)
    base = synthetic.synthetic_params(5, seed=9)
    full = {"shape": 300, "expression": 100, "jaw": 3, "rotation": 6, "eyeballs": 6, "neck": 3, "translation": 3, "scale": 1}
    narrow = {"shape": 100, "expression": 50, "jaw": 3, "rotation": 6, "eyeballs": 0, "neck": 0, "translation": 3, "scale": 1}
    kw = dict(flame_model=flame_model, static=static, device=0)
    yield HeadMesh(**kw), base, FLAME_CONSTS
    yield (HeadMesh(flame_config=full, **kw),
           np.concatenate([base[:, :409], 0.2 * rng.standard_normal((5, 9)).astype(np.float32), base[:, 409:]], axis=1), full)
    yield HeadMesh(flame_config=narrow, **kw), np.concatenate([base[:, :100], base[:, 300:350], base[:, 400:]], axis=1), narrow


def test_chain_kernels_match_the_torch_statement_of_the_chain(flame_model, static):
    """`dad3d_flame_pose_chain` against `autograd.pose_chain` (itself pinned to the oracle on CPU) and its dual-number
    backward against torch autograd over that statement, for three constants layouts."""
    from dad_3dheads_amd import autograd as ag

    lib = _lib.load()
    for mesh, params_np, consts in _configs(flame_model, static):
        layer = mesh.flame
        tables = layer.decode_tables()
        p = torch.from_numpy(np.ascontiguousarray(params_np)).cuda().requires_grad_(True)
        b = p.shape[0]
        chain = ag.pose_chain(tables, consts, p)
        k = lib.dad3d_flame_num_chain_inputs(layer._handle)
        assert k == chain["inputs"].shape[1] == 436
        inputs = torch.empty((b, k), device="cuda")
        c72 = torch.empty((b, 72), device="cuda")
        _lib.check(lib.dad3d_flame_pose_chain(layer._handle, p.data_ptr(), b, inputs.data_ptr(), c72.data_ptr(), None))
        torch.cuda.synchronize()
        assert (inputs - chain["inputs"]).abs().max() < 2e-6
        assert (c72 - chain["consts"]).abs().max() < 2e-6
        gen = torch.Generator().manual_seed(8)
        g_in, g_c = torch.randn((b, k), generator=gen).cuda(), torch.randn((b, 72), generator=gen).cuda()
        (g_ref,) = torch.autograd.grad([chain["inputs"], chain["consts"]], [p], [g_in, g_c])
        g = torch.full_like(p.detach(), float("nan"))  # every entry must be written
        _lib.check(lib.dad3d_flame_pose_chain_backward(layer._handle, p.data_ptr(), b, g_in.data_ptr(), g_c.data_ptr(), g.data_ptr(), None))
        torch.cuda.synchronize()
        assert bool(torch.isfinite(g).all())
        assert float((g - g_ref).abs().max()) <= 2e-5 * float(g_ref.abs().max())


def test_full_pose_and_narrow_layout_gradients_match_oracle(flame_model, flame_consts, static):
    for mesh, params_np, consts in list(_configs(flame_model, static))[1:]:
        params = torch.from_numpy(np.ascontiguousarray(params_np))
        w = torch.randn((params.shape[0], 5023, 3), generator=torch.Generator().manual_seed(4))
        p_ref = params.clone().requires_grad_(True)
        (flame_ref.reprojected_vertices(flame_consts, p_ref * 1.0, to_2d=False, consts=consts) * w).sum().backward()
        p = params.clone().cuda().requires_grad_(True)
        (mesh.reprojected_vertices(p * 1.0, to_2d=False) * w.cuda()).sum().backward()
        assert close(p.grad, p_ref.grad)


def test_training_step_replays_from_a_hip_graph(hm, flame_consts):
    """Forward and backward of both mesh terms on one prediction tensor, captured ONCE into a hipGraph and replayed on new
    parameters: the captured forward runs the device-epoch instantiation of the training kernel (no per-launch argument),
    and the replays must give the gradients the eager step gives -- and the oracle's."""
    batch = 6
    gen = torch.Generator().manual_seed(8)
    wv, wp = torch.randn((batch, 5023, 3), generator=gen).cuda(), torch.randn((batch, 5023, 2), generator=gen).cuda()

    def backward_of(p):
        q = p * 1.0
        v = hm.vertices_3d(q, zero_rotation=True)
        pr = hm.reprojected_vertices(q)
        ((v * wv).sum() + (pr * wp).sum() * 1e-2).backward()

    static_p = torch.from_numpy(synthetic.synthetic_params(batch, seed=50)).cuda().requires_grad_(True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):  # rocBLAS workspace, the library's partial-sum buffer, autograd's buffers: before the capture
            static_p.grad = None
            backward_of(static_p)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    static_p.grad = None
    with torch.cuda.graph(graph):
        backward_of(static_p)
    for seed in (51, 52, 53):
        params = torch.from_numpy(synthetic.synthetic_params(batch, seed=seed))
        with torch.no_grad():
            static_p.copy_(params.cuda())
        graph.replay()
        torch.cuda.synchronize()
        replayed = static_p.grad.clone()
        p = params.clone().cuda().requires_grad_(True)
        backward_of(p)
        assert close(replayed, p.grad)
        p_ref = params.clone().requires_grad_(True)
        q_ref = p_ref * 1.0
        ((flame_ref.vertices_3d(flame_consts, q_ref, zero_rotation=True) * wv.cpu()).sum()
         + (flame_ref.reprojected_vertices(flame_consts, q_ref) * wp.cpu()).sum() * 1e-2).backward()
        assert close(replayed, p_ref.grad)
    n = C.c_uint()
    _lib.check(hm.flame._lib.dad3d_flame_handoff_timeouts(hm.flame._handle, C.byref(n)))
    assert n.value == 0


@pytest.mark.parametrize("crit", ["l1", "l2", "smooth_l1"])
def test_fused_loss_kernels_match_the_torch_statement(crit):
    """csrc/mesh_losses.hip against the reference's own formulation through torch (vertices_3d_loss.py:43-49,
    reprojection_loss.py:42-46, model/utils.py:55-68) on random tensors: overlapping regions, a region that repeats a
    vertex, a single-vertex-wide region and negative indices; value and dL/d(pred) incl. the arg-position terms of the
    normalisation."""
    from dad_3dheads_amd.losses import RegionTables, _CubeRegionLoss, _WeightedPointLoss, _CRITERION_ID

    n_verts, batch = 700, 5
    rng = np.random.default_rng(17)
    indices = [np.arange(0, n_verts, 3), rng.permutation(n_verts)[:257], np.array([5, 9, 9, 300, 12, 5, 640]),
               np.arange(100, 400), np.array([-1, -2, 3, 4])]
    weights = [1.0, 0.5, 2.0, 0.25, 1.5]
    fn = {"l1": torch.nn.L1Loss, "l2": torch.nn.MSELoss, "smooth_l1": torch.nn.SmoothL1Loss}[crit]()
    tables = RegionTables(weights, indices, n_verts, torch.device("cuda", 0))
    gen = torch.Generator().manual_seed(4)
    scale = 3.0 if crit == "smooth_l1" else 1.0  # differences on both sides of SmoothL1's knee
    for comps, fused, stated in (
            (3, _CubeRegionLoss, lambda p, t: torch.stack([fn(normalize_to_cube(p[:, i]), normalize_to_cube(t[:, i])) * w
                                                           for w, i in zip(weights, indices)]).sum()),
            (3, _WeightedPointLoss, lambda p, t: torch.stack([fn(p[:, i], t[:, i]) * w for w, i in zip(weights, indices)]).sum()),
            (2, _WeightedPointLoss, lambda p, t: torch.stack([fn(p[:, i], t[:, i]) * w for w, i in zip(weights, indices)]).sum())):
        pred = (torch.randn((batch, n_verts, comps), generator=gen) * scale).cuda()
        target = (torch.randn((batch, n_verts, comps), generator=gen) * scale).cuda()
        p_ref = pred.double().requires_grad_(True)  # float64 torch statement
        want = stated(p_ref, target.double())
        want.backward()
        p = pred.clone().requires_grad_(True)
        got = fused.apply(p, target, tables, _CRITERION_ID[crit])
        (got * 1.75).backward()
        assert abs(float(got) - float(want)) <= 2e-6 * max(1.0, abs(float(want)))
        ref_grad = p_ref.grad.float() * 1.75
        assert float((p.grad - ref_grad).abs().max()) <= 2e-5 * float(ref_grad.abs().max())
        with torch.no_grad():  # value-only path (no gradient buffer)
            assert float(fused.apply(pred, target, tables, _CRITERION_ID[crit])) == float(got)


def test_fused_loss_argument_errors():
    lib = _lib.load()
    buf = torch.zeros((1, 8, 3), device="cuda")
    assert lib.dad3d_cube_region_loss(None, None, 0, 8, None, None, None, 0, None, None, None, 0, None, None, None, 0, None) == 0
    st = lib.dad3d_cube_region_loss(buf.data_ptr(), buf.data_ptr(), 1, 8, None, None, None, 1, None, None, None, 0, None, None, None, 0, None)
    assert st != 0 and b"null argument" in lib.dad3d_last_error()
    st = lib.dad3d_weighted_point_loss(buf.data_ptr(), buf.data_ptr(), 1, 8, 3, buf.data_ptr(), 1.0, 7, buf.data_ptr(), None, 0, None)
    assert st != 0 and b"unknown criterion" in lib.dad3d_last_error()
    with pytest.raises(IndexError):
        from dad_3dheads_amd.losses import RegionTables

        RegionTables([1.0], [np.array([0, 8])], 8, torch.device("cuda", 0))
    # first-order gradients of the prediction only: a target that wants one, or a double backward, raises (the reference's
    # torch graph would give both; silently returning None / zeros would not be the reference's result)
    from dad_3dheads_amd.losses import _CRITERION_ID, RegionTables, _WeightedPointLoss

    tables = RegionTables([1.0], [np.array([0, 3, 5])], 8, torch.device("cuda", 0))
    pred = torch.randn((2, 8, 3), device="cuda", requires_grad=True)
    with pytest.raises(RuntimeError, match="detach the target"):
        _WeightedPointLoss.apply(pred, torch.randn((2, 8, 3), device="cuda", requires_grad=True), tables, _CRITERION_ID["l2"])
    loss = _WeightedPointLoss.apply(pred, torch.randn((2, 8, 3), device="cuda"), tables, _CRITERION_ID["l2"])
    (g,) = torch.autograd.grad(loss, pred, create_graph=True)
    with pytest.raises(RuntimeError):
        g.sum().backward()


def test_grad_inputs_kernel_matches_the_plain_contraction(flame_model, static):
    """`dad3d_flame_grad_inputs` (split-K fp32 MFMA + fixed-order reduction, basis^T packed on the device from the forward
    pack) against dL/d(v_posed) @ basis^T in float64 with the plain [436, 3V] basis the host keeps, for the three constants
    layouts (jaw-only 416-row pack, full 448-row pack, narrow shape / expression widths) and ragged batches."""
    lib = _lib.load()
    gen = torch.Generator().manual_seed(12)
    for mesh, params_np, consts in _configs(flame_model, static):
        layer = mesh.flame
        tables = layer.decode_tables()
        n_in = lib.dad3d_flame_num_chain_inputs(layer._handle)
        assert n_in == tables.basis.shape[0]
        live = torch.ones(n_in, dtype=torch.bool)  # pose features of joints that are not inputs are exactly zero: no gradient
        if consts["neck"] == 0 and consts["eyeballs"] == 0:
            live[400:409] = False
            live[418:436] = False
        for batch in (1, 5, 64, 70, 130):
            p = torch.from_numpy(np.resize(params_np, (batch, params_np.shape[1])).astype(np.float32)).cuda()
            v3 = torch.empty((batch, 5023, 3), device="cuda")
            posed = torch.empty((batch, 15069), device="cuda")
            _lib.check(lib.dad3d_flame_decode_posed(layer._handle, p.data_ptr(), batch, 0, v3.data_ptr(), None, posed.data_ptr(), None))
            g_posed = torch.randn((batch, 15069), generator=gen).cuda()
            got = torch.full((batch, n_in), float("nan"), device="cuda")
            _lib.check(lib.dad3d_flame_grad_inputs(layer._handle, g_posed.data_ptr(), batch, got.data_ptr(), None))
            again = torch.empty_like(got)
            _lib.check(lib.dad3d_flame_grad_inputs(layer._handle, g_posed.data_ptr(), batch, again.data_ptr(), None))
            want = (g_posed.double() @ tables.basis.double().T).float()
            assert torch.equal(got, again)  # no atomics: bit-reproducible
            err = (got - want)[:, live].abs().max().item()
            assert err <= 2e-5 * want[:, live].abs().max().item(), (batch, err)
            assert float(got[:, ~live].abs().max()) == 0.0 if (~live).any() else True
