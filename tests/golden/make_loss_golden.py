#!/usr/bin/env python3
"""tests/golden/loss_golden.npz: values and d/d(params) of the reference's OWN `Vertices3DLoss` and `ReprojectionLoss`
(model_training/losses/vertices_3d_loss.py, reprojection_loss.py, imported unmodified through oracle/reference_runner.py)
on the seeded synthetic FLAME model, for the three criteria. Targets are decodes of other seeded params rows, so only
the seeds travel. Authoring container only."""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "loss_golden.npz")
BATCH, SEEDS = 4, (31, 32, 33)  # batch != 3: the reference's torch.cross without `dim` (model/utils.py:98-99)
REGIONS = {"a": (1.0, np.arange(0, 5023, 7)), "b": (0.5, np.arange(3000, 3600))}


def main():
    from dad_3dheads_amd import synthetic
    from dad_3dheads_amd.flame import FLAME_CONSTS
    from oracle import flame_ref, reference_runner as rr

    st = synthetic.load_static()
    model = synthetic.synthetic_flame_model(0, st)
    fc = flame_ref.FlameConstants.from_model(model)
    RefV3D, RefRep = rr.load_reference_losses(model)
    params = torch.from_numpy(synthetic.synthetic_params(BATCH, seed=SEEDS[0]))
    tgt3d = flame_ref.vertices_3d(fc, torch.from_numpy(synthetic.synthetic_params(BATCH, seed=SEEDS[1])), zero_rotation=True)
    tgt2d = flame_ref.reprojected_vertices(fc, torch.from_numpy(synthetic.synthetic_params(BATCH, seed=SEEDS[2])))
    out = {"batch": BATCH, "seeds": np.array(SEEDS), "region_names": np.array(list(REGIONS)),
           "region_weights": np.array([w for w, _ in REGIONS.values()])}
    with tempfile.TemporaryDirectory() as d:
        for k, (_, idx) in REGIONS.items():
            np.save(os.path.join(d, k + ".npy"), idx)
            out["region_" + k] = idx
        cfg = {"weights": {k: w for k, (w, _) in REGIONS.items()},
               "flame_indices": {"folder": d, "files": {k: k + ".npy" for k in REGIONS}}}
        for crit in ("l1", "l2", "smooth_l1"):
            p = params.clone().requires_grad_(True)
            v3 = RefV3D(crit, BATCH, FLAME_CONSTS, cfg)(p * 1.0, tgt3d)
            rp = RefRep(crit, BATCH, FLAME_CONSTS, 256, cfg)(p * 1.0, tgt2d)
            (g3,) = torch.autograd.grad(v3, p, retain_graph=True)
            (g2,) = torch.autograd.grad(rp, p)
            out[crit + "_vertices3d"], out[crit + "_reprojection"] = float(v3.detach()), float(rp.detach())
            out[crit + "_vertices3d_grad"], out[crit + "_reprojection_grad"] = g3.numpy(), g2.numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: out[k] for k in out if k.endswith(("vertices3d", "reprojection"))})


if __name__ == "__main__":
    main()
