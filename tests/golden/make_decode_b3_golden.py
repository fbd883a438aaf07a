#!/usr/bin/env python3
"""Generate tests/golden/decode_b3_golden.npz: the REFERENCE's own HeadMesh on a batch of exactly three rows.

model_training/model/utils.py:98-99 calls `torch.cross(b1, vy)` without `dim`; for [3,3] operands torch's legacy rule
takes the first axis of size 3 -- the batch axis -- so the three images' 6-DoF rotations mix. The
drop-in gives every image its own rotation by default and reproduces the reference's batch-of-three output only with
DAD3D_COMPAT_CROSS_B3. This file freezes what the reference itself returns (authoring container only; same machinery as
make_decode_golden.py), plus the same three rows decoded one at a time (batch 1: no mixing) for contrast."""
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dad_3dheads_amd import synthetic  # noqa: E402
from oracle import reference_runner  # noqa: E402

OUT = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "decode_b3_golden.npz")


def main():
    static = synthetic.load_static()
    model = synthetic.synthetic_flame_model(0, static)
    hm = reference_runner.load_reference_head_mesh(model)
    p3 = synthetic.synthetic_params(3, seed=333)
    sub = np.arange(0, 5023, 11, dtype=np.int64)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        p = torch.from_numpy(p3.copy())
        v3d = hm.vertices_3d(p).numpy()
        proj = hm.reprojected_vertices(p, to_2d=False).numpy()
        rows = [hm.vertices_3d(torch.from_numpy(p3[i:i + 1].copy())).numpy()[0] for i in range(3)]
    np.savez_compressed(OUT, params=p3, subset=sub, v3d_batch3=v3d[:, sub], proj3_batch3=proj[:, sub],
                        v3d_one_at_a_time=np.stack(rows)[:, sub])
    print("wrote", OUT, os.path.getsize(OUT), "bytes; max |batch3 - one at a time| =", float(np.abs(v3d - np.stack(rows)).max()))


if __name__ == "__main__":
    main()
