"""GPU: the FaceMeshPredictor drop-in (call surface of predictor.py:68-211) with a stand-in CNN -- the trained
TorchScript checkpoint is not available offline, so the network is a deterministic module with the real
output contract (flame_regression.py:96-104); everything after it is compared with the reference's CPU path."""
import numpy as np
import pytest
import torch

from dad_3dheads_amd import synthetic
from dad_3dheads_amd.config import load_default_config
from dad_3dheads_amd.predictor import FaceMeshPredictor, calculate_paddings, py3round
from oracle import flame_ref

pytestmark = pytest.mark.gpu


class StandInRegressor(torch.nn.Module):
    """Output contract of DAD-3DNet: {"OUTPUT_3DMM_PARAMS": [B,413], "OUTPUT_2D_LANDMARKS": [B,68,2] in [0,1]} -- the key
    strings of model_training/data/config.py:16-23, which the reference's TorchScript checkpoint returns."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        self.register_buffer("w", torch.randn(3, 413, generator=g) * 0.5)
        self.register_buffer("base", torch.from_numpy(synthetic.synthetic_params(1, seed=8))[0])
        self.register_buffer("ramp", torch.linspace(0.2, 0.9, 68)[None, :, None])  # a buffer: follows .to() when traced

    def forward(self, x):
        feat = x.mean(dim=(2, 3))  # [B,3]
        p = self.base[None] + 0.05 * torch.tanh(feat @ self.w)
        lm = torch.sigmoid(feat[:, :2])[:, None, :].expand(-1, 68, -1) * self.ramp
        return {"OUTPUT_3DMM_PARAMS": p, "OUTPUT_2D_LANDMARKS": lm}


@pytest.fixture(scope="module")
def predictor(flame_model):
    return FaceMeshPredictor(load_default_config(), cuda_id=0, model=StandInRegressor(), flame_model=flame_model)


def reference_postprocess(pred, image, flame_consts):
    """predictor.py:102-176 on the CPU oracle, from the SAME network output."""
    cache = {}
    x = pred.preprocess(image, cache)
    out = pred.process(x)
    params = out["OUTPUT_3DMM_PARAMS"].detach().cpu().clone()
    lm = out["OUTPUT_2D_LANDMARKS"].detach().cpu().numpy() * 256.0
    pads, scale = flame_ref.get_paddings(image.shape[:2])
    pts = lm.clip(min=0, max=256) - np.array([[pads[2], pads[0]]])
    pts = (pts / scale).astype(int).reshape(-1, 2)
    params = flame_ref.readjust_3dmm(params, pads, scale)
    v3d = flame_ref.vertices_3d(flame_consts, params)[0].squeeze()
    proj = flame_ref.reprojected_vertices(flame_consts, params, to_2d=True)
    return pts, proj, v3d, params


@pytest.mark.parametrize("hw", [(256, 256), (954, 766), (300, 500), (100, 80)])
def test_single_image_call_surface(predictor, flame_consts, hw):
    rng = np.random.default_rng(hw[0])
    image = rng.integers(0, 255, (hw[0], hw[1], 3), dtype=np.uint8)
    res = predictor(image)
    assert set(res) == {"points", "projected_vertices", "3d_vertices", "3dmm_params"}
    assert res["points"].shape == (68, 2) and res["points"].dtype.kind == "i"
    assert res["projected_vertices"].shape == (1, 5023, 2) and res["3d_vertices"].shape == (5023, 3)
    assert res["3dmm_params"].shape == (1, 413) and res["3dmm_params"].device.type == "cpu"
    pts, proj, v3d, params = reference_postprocess(predictor, image, flame_consts)
    assert np.array_equal(res["points"], pts)
    assert (res["3dmm_params"] - params).abs().max() < 1e-5 and res["3dmm_params"][0, 411] == 0
    assert (res["3d_vertices"] - v3d).abs().max() < 5e-6
    # pixels scale with 1/scale of the input frame (up to ~3.7x for the 954-pixel image)
    assert (res["projected_vertices"] - proj).abs().max() < 1e-3 * max(1.0, max(hw) / 256)


def test_geometry_of_demo_image(predictor):
    pads, scale, (nh, nw) = predictor._geometry((954, 766))  # images/demo_heads/1.jpeg is 766x954 (WxH)
    assert pads == [0, 0, 25, 25] and (nh, nw) == (256, 206) and abs(scale - 256 / 954) < 1e-12
    assert py3round(2.5) == 2 and py3round(3.5) == 4 and calculate_paddings(206, 256) == [25, 25, 0, 0]
    x = predictor.preprocess(np.zeros((954, 766, 3), np.uint8), {})
    assert x.shape == (1, 3, 256, 256)
    mean, std = np.array([0.485, 0.456, 0.406]), np.array([0.229, 0.224, 0.225])
    assert np.allclose(x[0, :, 0, 0].cpu().numpy(), (0 - 255 * mean) / (255 * std), atol=1e-5)


def test_batched_predictor_equals_single_calls(predictor):
    rng = np.random.default_rng(3)
    images = [rng.integers(0, 255, (h, w, 3), dtype=np.uint8) for h, w in ((256, 256), (300, 200), (128, 512), (256, 256), (90, 90))]
    batched = predictor.predict_batch(images)
    for im, b in zip(images, batched):
        s = predictor(im)
        assert np.array_equal(s["points"], b["points"])
        for k in ("projected_vertices", "3d_vertices", "3dmm_params"):
            assert torch.allclose(s[k], b[k], atol=2e-4, rtol=0)  # batch-size dependent conv/reduction order in the CNN
    dev = predictor.predict_batch(images[:2], device_outputs=True)
    assert dev[0]["3d_vertices"].is_cuda and dev[0]["projected_vertices"].is_cuda


def test_missing_checkpoint_is_a_loud_error(flame_model, tmp_path, monkeypatch):
    monkeypatch.setenv("HOME", str(tmp_path))
    with pytest.raises(FileNotFoundError, match="no network"):
        FaceMeshPredictor(load_default_config(), cuda_id=0, flame_model=flame_model)


def test_torchscript_checkpoint_path(flame_model, flame_consts, tmp_path, monkeypatch):
    """The reference's construction path (predictor.py:72): `torch.jit.load($HOME/<model_path>)` of a scripted module that
    returns the reference's output keys (model_training/data/config.py:16-23) -- no `model=` argument."""
    monkeypatch.setenv("HOME", str(tmp_path))
    config = load_default_config()
    path = tmp_path / config["model_path"]
    path.parent.mkdir(parents=True, exist_ok=True)
    example = torch.zeros(1, 3, 256, 256)
    torch.jit.trace(StandInRegressor(), example, strict=False).save(str(path))
    pred = FaceMeshPredictor(config, cuda_id=0, flame_model=flame_model)
    assert isinstance(pred.model, torch.jit.ScriptModule)
    image = np.random.default_rng(3).integers(0, 255, (300, 260, 3), dtype=np.uint8)
    res = pred(image)
    pts, proj, v3d, params = reference_postprocess(pred, image, flame_consts)
    assert np.array_equal(res["points"], pts)
    assert (res["3dmm_params"] - params).abs().max() < 1e-5
    assert (res["3d_vertices"] - v3d).abs().max() < 5e-6


# ---- DAD-3DNet front half declared for PyTorch-ROCm ------------------------------------------------
@pytest.fixture(scope="module")
def net_predictor(flame_model):
    from dad_3dheads_amd import landmarks

    return FaceMeshPredictor.random_init(cuda_id=0, flame_model=flame_model, landmarks=landmarks.canonical("445"))


def test_network_output_contract(net_predictor):
    x = torch.randn(3, 3, 256, 256, device="cuda")
    out = net_predictor.process(x)
    assert out["OUTPUT_3DMM_PARAMS"].shape == (3, 413) and out["OUTPUT_3DMM_PARAMS"].dtype == torch.float32
    assert out["OUTPUT_2D_LANDMARKS"].shape == (3, 68, 2) and (out["OUTPUT_2D_LANDMARKS"] >= 0).all()
    assert out["OUTPUT_LANDMARKS_HEATMAP"].shape == (3, 68, 64, 64)
    assert out["OUTPUT_3DMM_PARAMS"][:, :403].abs().max() <= 3.0  # tanh * limit_value (flame_regression.py:94)


@pytest.mark.parametrize("hw", [(256, 256), (320, 240)])
def test_device_resident_batch_matches_reference_postprocess(net_predictor, flame_consts, hw, monkeypatch):
    """predict_tensor: nothing leaves the GPU; from the SAME CNN output the reference's CPU post-processing
    (re-adjust, two decodes, astype(int) gather) must agree."""
    g = torch.Generator().manual_seed(5)
    images = torch.randint(0, 255, (4, hw[0], hw[1], 3), dtype=torch.uint8, generator=g).cuda()
    seen = {}
    real_process = net_predictor.process

    def recording_process(x):  # keep the network output this very call produced (reduced-precision convolutions need
        seen["x"], seen["out"] = x, real_process(x)  # not be bit-reproducible between two calls)
        return seen["out"]

    monkeypatch.setattr(net_predictor, "process", recording_process)
    res = net_predictor.predict_tensor(images)
    monkeypatch.undo()
    assert all(v.is_cuda for v in res.values())
    assert res["points"].shape == (4, 68, 2) and res["points"].dtype == torch.int32
    assert res["landmarks"].shape == (4, 445, 2) and res["landmarks"].dtype == torch.int32
    # the batched device-side normalisation equals the per-image preprocess of the single-image path
    x = torch.cat([net_predictor.preprocess(im, {}) for im in images.cpu().numpy()])
    assert torch.allclose(seen["x"], x, atol=1e-5)
    params = seen["out"]["OUTPUT_3DMM_PARAMS"].detach().cpu().clone()
    lm68 = seen["out"]["OUTPUT_2D_LANDMARKS"].detach().cpu().numpy() * 256.0
    pads, scale = flame_ref.get_paddings(hw)
    params = flame_ref.readjust_3dmm(params, pads, scale)
    v3d = flame_ref.vertices_3d(flame_consts, params)
    proj = flame_ref.reprojected_vertices(flame_consts, params, to_2d=True)
    assert x.shape == (4, 3, 256, 256)
    pts = ((lm68.clip(min=0, max=256) - np.array([[pads[2], pads[0]]])) / scale).astype(int)
    assert np.array_equal(res["points"].cpu().numpy(), pts)
    assert (res["3dmm_params"].cpu() - params).abs().max() < 1e-5
    assert (res["3d_vertices"].cpu() - v3d).abs().max() < 5e-6
    assert (res["projected_vertices"].cpu() - proj).abs().max() < 2e-3
    idx = torch.from_numpy(np.asarray(net_predictor.head_mesh.flame.landmark_indices)).long()
    assert torch.equal(res["landmarks"].cpu(), res["projected_vertices"].cpu()[:, idx, :].to(torch.int32))


def test_68_landmarks_on_device_equal_host(net_predictor, static):
    from dad_3dheads_amd.benchmark_export import Landmarks68, seven_landmarks

    params = torch.from_numpy(synthetic.synthetic_params(5, seed=44)).cuda()
    verts = net_predictor.head_mesh.decode(params, proj=False, landmarks=False)["verts3d"]
    lm = Landmarks68(static["faces"], device=verts.device)
    on_dev = lm(verts)
    assert on_dev.is_cuda and on_dev.shape == (5, 68, 3)
    assert torch.equal(on_dev.cpu(), Landmarks68(static["faces"])(verts.cpu()))
    assert seven_landmarks(on_dev).shape == (5, 7, 3)


def test_cnn_glue_kernels_equal_the_framework_ops():
    """csrc/cnn_glue.hip through its C ABI: bias (+ residual) (+ ReLU) in place and the BiFPN's weighted nearest-resize-and-sum,
    against the torch statements they replace (layers of model_training/model/layers.py; bifpn.py:98-125), for the three
    element types, up- and down-sampling, odd extents."""
    import torch.nn.functional as F

    from dad_3dheads_amd import _glue

    g = torch.Generator().manual_seed(11)
    for dtype, tol in ((torch.float32, 1e-6), (torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)):
        for shape in ((3, 64, 17, 9), (2, 256, 8, 8), (1, 2048, 1, 1)):
            y = torch.randn(shape, generator=g).cuda().to(dtype).contiguous(memory_format=torch.channels_last)
            z = torch.randn(shape, generator=g).cuda().to(dtype).contiguous(memory_format=torch.channels_last)
            b = torch.randn(shape[1], generator=g).cuda().to(dtype)
            assert _glue.supported(y, z)
            for zz in (None, z):
                for relu in (False, True):
                    want = y.float() + b.float().view(1, -1, 1, 1) + (zz.float() if zz is not None else 0)
                    want = F.relu(want) if relu else want
                    got = _glue.bias_act_(y.clone(memory_format=torch.channels_last), b, zz, relu)
                    assert got.is_contiguous(memory_format=torch.channels_last)
                    assert float((got.float() - want).abs().max()) <= tol * max(float(want.abs().max()), 1.0)
        xs = [torch.randn(2, 64, h, w, generator=g).cuda().to(dtype).contiguous(memory_format=torch.channels_last)
              for h, w in ((16, 16), (8, 8), (32, 31))]
        ws = (0.43, 0.31, 0.27)
        for size in ((16, 16), (8, 8), (5, 7), (32, 31)):
            for k in (1, 2, 3):
                want = sum(ws[j] * F.interpolate(xs[j].float(), size=size) for j in range(k))
                got = _glue.resize_sum(ws[:k], xs[:k], size)
                assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
                assert float((got.float() - want).abs().max()) <= tol * max(float(want.abs().max()), 1.0), (dtype, size, k)
    assert not _glue.supported(torch.zeros(1, 68, 4, 4, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last))
    assert not _glue.supported(torch.zeros(1, 64, 4, 4, device="cuda"))  # NCHW


def test_inference_net_equals_the_plain_module_on_the_gpu():
    """InferenceNet's serving rewrite -- BatchNorm folded, BiFPN weights frozen, channels-last, weights converted ONCE to the
    serving dtype instead of autocast -- against the plain fp32 module on the same device (fp32: rounding only; bf16: the
    precision of the format)."""
    import copy

    from dad_3dheads_amd.network import DAD3DNet, InferenceNet

    base = DAD3DNet(seed=3).eval()
    with torch.no_grad():  # BatchNorm statistics away from their initial values so that the folding does something
        for m in base.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.uniform_(-0.2, 0.2), m.running_var.uniform_(0.6, 1.4), m.weight.uniform_(0.7, 1.3), m.bias.uniform_(-0.1, 0.1)
    x = torch.randn(2, 3, 256, 256, generator=torch.Generator().manual_seed(5)).cuda()
    with torch.no_grad():
        want = copy.deepcopy(base).cuda()(x)
    got = InferenceNet(copy.deepcopy(base), torch.float32).cuda()(x)
    for k in want:
        assert float((got[k] - want[k]).abs().max()) < 2e-3 * max(float(want[k].abs().max()), 1.0), k
    from dad_3dheads_amd.network import ConvBiasAct

    plain = InferenceNet(copy.deepcopy(base), torch.float32, glue=False).cuda()(x)  # the same rewrite on the framework's own ops
    for k in want:
        assert float((got[k] - plain[k]).abs().max()) < 1e-4 * max(float(want[k].abs().max()), 1.0), k
    low_net = InferenceNet(copy.deepcopy(base), torch.bfloat16).cuda()
    assert sum(isinstance(m, ConvBiasAct) for m in low_net.modules()) > 60
    assert all(p.dtype == torch.bfloat16 for p in low_net.parameters())
    low = low_net(x)
    for k in want:
        assert low[k].dtype == torch.float32 and torch.isfinite(low[k]).all()
        assert float((low[k] - want[k]).abs().max()) < 0.15 * max(float(want[k].abs().max()), 1.0), k


def test_graphed_network_equals_eager(flame_model):
    """GraphedNet: the frozen CNN replayed from a hipGraph (one per input shape) gives the eager outputs."""
    from dad_3dheads_amd.network import DAD3DNet, GraphedNet, InferenceNet

    net = InferenceNet(DAD3DNet(seed=1), torch.bfloat16).cuda()
    graphed = GraphedNet(net)
    for shape in ((1, 3, 256, 256), (3, 3, 256, 256), (1, 3, 256, 256)):
        for seed in (0, 1):
            x = torch.randn(*shape, generator=torch.Generator().manual_seed(seed)).cuda()
            want, got = net(x), graphed(x)
            for k in want:
                assert torch.allclose(want[k], got[k], atol=2e-2, rtol=2e-2), k  # bf16 kernels may differ eager/captured
    assert len(graphed._graphs) == 2
    pred = FaceMeshPredictor.random_init(graph=True, cuda_id=0, flame_model=flame_model)
    rng = np.random.default_rng(9)
    for hw in ((256, 256), (300, 200), (256, 256)):
        res = pred(rng.integers(0, 255, (hw[0], hw[1], 3), dtype=np.uint8))
        assert res["3d_vertices"].shape == (5023, 3) and torch.isfinite(res["3d_vertices"]).all()
