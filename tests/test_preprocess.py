"""Predictor preprocessing (SURVEY 8 rows f1 / g): the restated cv2 + albumentations arithmetic (oracle/preprocess_ref.py)
against its frozen outputs for the reference's demo image and against float bilinear sampling; the HIP kernel against the
oracle bit for bit; BASELINE configs[0] end to end -- demo image -> FaceMeshPredictor with a stand-in regressor ->
flame_params JSON bytes."""
import io
import json
import os

import numpy as np
import pytest
import torch

from oracle import preprocess_ref as pr

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "demo_image.npz")


def demo_image():
    from PIL import Image

    with np.load(GOLDEN) as z:
        g = {k: z[k] for k in z.files}
    img = np.asarray(Image.open(io.BytesIO(g["jpeg"].tobytes())).convert("RGB"))
    return img, g


def test_oracle_reproduces_the_frozen_demo_image_outputs():
    img, g = demo_image()
    assert tuple(img.shape) == tuple(g["shape"]) == (954, 766, 3)
    nh, nw, top, left, scale = pr.geometry(*img.shape[:2])
    assert [nh, nw, top, left] == list(g["geometry"]) == [256, 206, 0, 25] and scale == float(g["scale"])
    assert np.array_equal(pr.resize_linear_u8(img, nh, nw), g["resized"])
    x = pr.transform(img)
    assert x.shape == (3, 256, 256) and x.dtype == np.float32
    assert np.array_equal(x[:, :4, 23:29], g["transformed_corner"])
    assert abs(float(x.astype(np.float64).sum()) - float(g["transformed_sum"])) < 1e-6
    # the padding columns are (0 - 255 mean) / (255 std) in float32, the image columns hold the resized bytes
    m, s = np.float32(0.485) * np.float32(255), np.float32(0.229) * np.float32(255)
    assert x[0, 0, 0] == (np.float32(0) - m) * np.reciprocal(s, dtype=np.float32)
    assert x[0, 10, 25] == (np.float32(g["resized"][10, 0, 0]) - m) * np.reciprocal(s, dtype=np.float32)


@pytest.mark.parametrize("hw", [(954, 766), (100, 180), (300, 500), (77, 31), (256, 256), (512, 512), (255, 257)])
def test_fixed_point_resize_is_float_bilinear_to_within_rounding(hw):
    """The restated OpenCV 8-bit path against half-pixel-centre float bilinear sampling (torch, no antialias): one LSB."""
    import torch.nn.functional as F

    rng = np.random.default_rng(hw[0] * 1000 + hw[1])
    img = rng.integers(0, 256, (hw[0], hw[1], 3), dtype=np.uint8)
    nh, nw, top, left, _ = pr.geometry(*hw)
    assert max(nh, nw) == 256 and top == int((256 - nh) / 2) and left == int((256 - nw) / 2)
    small = pr.resize_linear_u8(img, nh, nw)
    fl = F.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None].float(), size=(nh, nw), mode="bilinear", align_corners=False,
                       antialias=False)[0].permute(1, 2, 0).numpy()
    assert small.shape == (nh, nw, 3) and np.abs(small.astype(np.float64) - fl).max() < 1.0
    if (nh, nw) == hw:
        assert np.array_equal(small, img)


# ---- GPU ---------------------------------------------------------------------------------------------------------------
class StandInRegressor(torch.nn.Module):
    """Output contract of DAD-3DNet: {"OUTPUT_3DMM_PARAMS": [B,413], "OUTPUT_2D_LANDMARKS": [B,68,2] in [0,1]} (flame_regression.py:96-104)."""

    def __init__(self):
        super().__init__()
        from dad_3dheads_amd import synthetic

        g = torch.Generator().manual_seed(0)
        self.register_buffer("w", torch.randn(3, 413, generator=g) * 0.5)
        self.register_buffer("base", torch.from_numpy(synthetic.synthetic_params(1, seed=8))[0])

    def forward(self, x):
        feat = x.mean(dim=(2, 3))
        p = self.base[None] + 0.05 * torch.tanh(feat @ self.w)
        lm = torch.sigmoid(feat[:, :2])[:, None, :].expand(-1, 68, -1) * torch.linspace(0.2, 0.9, 68, device=x.device)[None, :, None]
        return {"OUTPUT_3DMM_PARAMS": p, "OUTPUT_2D_LANDMARKS": lm}


@pytest.fixture(scope="module")
def predictor(flame_model):
    from dad_3dheads_amd.config import load_default_config
    from dad_3dheads_amd.predictor import FaceMeshPredictor

    return FaceMeshPredictor(load_default_config(), cuda_id=0, model=StandInRegressor(), flame_model=flame_model)


@pytest.mark.gpu
def test_preprocess_kernel_equals_the_oracle_bit_for_bit(predictor):
    img, g = demo_image()
    x = predictor.preprocess(img, {})
    torch.cuda.synchronize()
    assert x.shape == (1, 3, 256, 256) and x.dtype == torch.float32
    assert np.array_equal(x[0].cpu().numpy(), pr.transform(img))
    rng = np.random.default_rng(5)
    sizes = [(954, 766), (100, 180), (300, 500), (77, 31), (256, 256), (512, 512), (255, 257), (1, 1), (2, 600), (1080, 1920)]
    images = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    for im in images:
        assert np.array_equal(predictor.preprocess(im, {})[0].cpu().numpy(), pr.transform(im)), im.shape
    # one launch for a batch of different sizes == the single-image launches
    staged = [torch.from_numpy(im).cuda() for im in images]
    batch = predictor._preprocess_launch([(t.data_ptr(), t.shape[0], t.shape[1], t.shape[1] * 3) for t in staged])
    torch.cuda.synchronize()
    for i, im in enumerate(images):
        assert np.array_equal(batch[i].cpu().numpy(), pr.transform(im)), im.shape
    # device-resident uniform batch (predict_tensor's entry)
    same = torch.from_numpy(np.stack([images[1], images[1][::-1].copy()])).cuda()
    out = predictor._preprocess_launch([(same.data_ptr() + i * same[0].numel(), 100, 180, 540) for i in range(2)])
    assert np.array_equal(out[1].cpu().numpy(), pr.transform(images[1][::-1].copy()))
    with pytest.raises(ValueError):
        predictor.preprocess(images[0].astype(np.float32), {})


@pytest.mark.gpu
def test_config0_demo_image_to_flame_params_json(predictor, flame_consts, tmp_path):
    """BASELINE configs[0] (`demo.py <image> <out> flame_params`, demo.py:24-50 + demo_utils.py:112-153) with a stand-in
    regressor: decode the demo JPEG -> FaceMeshPredictor.__call__ -> get_flame_params -> JsonSaver. The JSON bytes must be
    the ones the reference's CPU post-processing (predictor.py:117-176, through the oracle) produces from the SAME network
    output for the SAME preprocessed tensor."""
    from dad_3dheads_amd import writers
    from oracle import flame_ref

    img, g = demo_image()
    res = predictor(img)
    assert set(res) == {"points", "projected_vertices", "3d_vertices", "3dmm_params"}
    out_path = tmp_path / writers.get_output_path("images/demo_heads/1.jpeg", "", "flame_params", writers.JsonSaver().extension).lstrip("/")
    writers.JsonSaver()(writers.get_flame_params(res), str(out_path))
    assert out_path.name == "1_flame_params.json"
    # the reference's path on the CPU from the same network output
    x = torch.from_numpy(pr.transform(img))[None].cuda()
    net_out = predictor.process(x)
    params = net_out["OUTPUT_3DMM_PARAMS"].detach().cpu().clone()
    pads, scale = flame_ref.get_paddings(img.shape[:2])
    assert pads == [0, 0, 25, 25] and abs(scale - 256 / 954) < 1e-15
    params = flame_ref.readjust_3dmm(params, pads, scale)
    flame_ref.reprojected_vertices(flame_consts, params, to_2d=True)  # zeroes translation z like head_mesh.py:41
    want = {k: v[0].tolist() for k, v in flame_ref.split_3dmm(params).items()}
    got = json.loads(out_path.read_text())
    assert list(got) == ["shape", "expression", "rotation", "translation", "scale", "jaw", "eyeballs", "neck"]
    for k in got:
        assert np.allclose(got[k], want[k], rtol=0, atol=2e-6), k
    assert got["translation"][2] == 0.0 and got["eyeballs"] == [] and len(got["shape"]) == 300
    lm = net_out["OUTPUT_2D_LANDMARKS"].detach().cpu().numpy() * 256.0
    pts = ((lm.clip(min=0, max=256) - np.array([[pads[2], pads[0]]])) / scale).astype(int).reshape(-1, 2)
    assert np.array_equal(res["points"], pts)
