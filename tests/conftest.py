import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests must never pass on a silent fallback: without a device they are skipped, never emulated.
    try:
        import torch

        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def static():
    from dad_3dheads_amd import synthetic

    return synthetic.load_static()


@pytest.fixture(scope="session")
def flame_model(static):
    from dad_3dheads_amd import synthetic

    return synthetic.synthetic_flame_model(0, static)


@pytest.fixture(scope="session")
def flame_consts(flame_model):
    from oracle import flame_ref

    return flame_ref.FlameConstants.from_model(flame_model)


@pytest.fixture(scope="session")
def decode_golden():
    with np.load(os.path.join(GOLDEN, "decode_golden.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def sim3dr_golden():
    with np.load(os.path.join(GOLDEN, "sim3dr_golden.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def port_oracle():
    from oracle.sim3dr_ref import Sim3DROracle

    return Sim3DROracle("port")


@pytest.fixture(scope="session")
def sim3dr_oracle():
    """The checker of the GPU tests: the reference's own rasterize_kernel.cpp (oracle/_ref/libsim3dr_ref.so, compiled in the
    authoring container, shipped to the GPU box) whenever it is present, the C port otherwise."""
    from oracle.sim3dr_ref import Sim3DROracle

    return Sim3DROracle("best")
