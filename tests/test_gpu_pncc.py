"""GPU: PNCC rendering (inference/pncc_estimator.py) -- decode with z flip + raster of the 6270-face subset with NCC
colours, against the CPU oracle (FLAME restatement for the vertices, the reference's Sim3DR C++ for the raster)."""
import numpy as np
import pytest
import torch

from dad_3dheads_amd import synthetic
from dad_3dheads_amd.pncc import PNCCEstimator
from oracle import flame_ref, sim3dr_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def estimator(flame_model, static):
    return PNCCEstimator(img_size=256, flame_model=flame_model, device=0, static=static)


def test_render_batch_matches_oracle(estimator, flame_consts):
    b = 6
    params = torch.from_numpy(synthetic.synthetic_params(b, seed=21)).cuda()
    ref_v = flame_ref.reprojected_vertices(flame_consts, params.cpu().clone(), to_2d=False)
    ref_v[:, :, 2] *= -1
    images = estimator.render_batch(params, size=(256, 256))
    assert images.shape == (b, 256, 256, 3) and images.dtype == torch.uint8 and images.is_cuda
    assert params[:, 411].abs().max() == 0  # reprojected_vertices' side effect (head_mesh.py:41)
    verts = estimator.head_mesh.flame.decode(params, proj=True, to_2d=False, flip_z=True)["proj"]
    assert (verts.cpu() - ref_v).abs().max() < 1e-3  # pixel units
    oracle = sim3dr_ref.Sim3DROracle("best")
    faces = estimator.faces_wo_back_remapped.astype(np.int32)
    colors = estimator.colors.astype(np.float32)
    for i in range(b):  # the raster itself is bit-exact on identical vertices
        want = oracle.rasterize(np.ascontiguousarray(verts[i].cpu().numpy()), faces, colors, bg=np.zeros((256, 256, 3), np.uint8))
        assert np.array_equal(images[i].cpu().numpy(), want)
    assert (images.cpu().numpy().reshape(b, -1).max(1) > 0).all()


def test_single_image_call_equals_batch_row(estimator):
    params = torch.from_numpy(synthetic.synthetic_params(3, seed=22)).cuda()
    batch = estimator.render_batch(params.clone(), size=(200, 240)).cpu().numpy()
    rng = np.random.default_rng(0)
    image = rng.integers(0, 255, (200, 240, 3), dtype=np.uint8)
    for i in range(3):
        black = estimator(image, {"3dmm_params": params[i : i + 1].cpu().clone()}, with_background=False)
        assert np.array_equal(black, batch[i])
    over = estimator(image, {"3dmm_params": params[:1].cpu().clone()}, with_background=True)
    covered = batch[0].any(-1)
    assert np.array_equal(over[~covered], image[~covered]) and covered.any()
