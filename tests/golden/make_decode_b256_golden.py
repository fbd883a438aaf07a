#!/usr/bin/env python3
"""Generate tests/golden/decode_b256_golden.npz by running the REFERENCE's own HeadMesh on BASELINE configs[2]
(batch = 256, head_mesh / .obj-vertices path: `vertices_3d` + `reprojected_vertices(to_2d=False)`).

Authoring-container only (needs /root/reference; see make_decode_golden.py for how the reference is imported). Every one of
the 256 rows is kept, on a fixed subset of 40 vertices (4 of them in the partial last tile of the pipelined kernel, 4 in the
first), so bench.py's `secondary.decode_b256` leg and tests/test_gpu_decode.py can hold every row the timed launches wrote to
the reference without /root/reference on the GPU box.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dad_3dheads_amd import synthetic  # noqa: E402
from oracle import reference_runner  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "decode_b256_golden.npz")
SEED = 104


def main():
    static = synthetic.load_static()
    model = synthetic.synthetic_flame_model(0, static)
    hm = reference_runner.load_reference_head_mesh(model)
    params = synthetic.synthetic_params(256, seed=SEED)
    with torch.no_grad():
        p = torch.from_numpy(params.copy())
        v3d = hm.vertices_3d(p).numpy()
        proj3 = hm.reprojected_vertices(p, to_2d=False).numpy()
    rng = np.random.default_rng(9)
    sub = np.sort(np.concatenate([np.arange(4), rng.choice(np.arange(4, 5019), 32, replace=False), np.arange(5019, 5023)])).astype(np.int64)
    np.savez_compressed(OUT, model_digest=np.frombuffer(bytes.fromhex(synthetic.model_digest(model)), dtype=np.uint8),
                        seed=np.int64(SEED), subset=sub, v3d_sub=v3d[:, sub], proj3_sub=proj3[:, sub],
                        params_after_changed_cols=np.nonzero((p.numpy() != params).any(axis=0))[0], tz_after=p.numpy()[:, 411].copy())
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
