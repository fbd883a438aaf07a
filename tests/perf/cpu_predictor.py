"""The reference CPU predictor's call sequence on the host (the north star's comparator), used by `tools/bench_e2e.py --cpu`.
Lives under tests/perf/ because it imports `oracle/` (test infrastructure: CPU restatements of the reference); never imported by the product."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def cpu_reference_predictor(model, lmk_idx, n_images: int, threads=None, budget_s: float = 45.0):
    """The reference predictor's call sequence on the host CPU (predictor.py:97-145), one 256 x 256 image per call."""
    import numpy as np

    from dad_3dheads_amd.network import DAD3DNet
    from oracle import flame_ref, preprocess_ref

    net = DAD3DNet(seed=0).eval()  # the declaration the GPU leg runs, fp32 on the CPU (the reference: a TorchScript of the same graph)
    consts = flame_ref.FlameConstants.from_model(model)
    rng = np.random.default_rng(0)
    frames = [rng.integers(0, 255, (256, 256, 3), dtype=np.uint8) for _ in range(8)]
    ncpu = os.cpu_count() or 1

    def one(img):
        x = torch.from_numpy(preprocess_ref.transform(img))[None]                       # predictor.py:86-95
        with torch.no_grad():
            out = net(x)                                                                 # predictor.py:97-100
        params = out["OUTPUT_3DMM_PARAMS"].detach().cpu().float()                       # predictor.py:104
        post = flame_ref.predictor_postprocess(consts, params.clone(), lmk_idx, input_hw=img.shape[:2])  # :125-145 + demo_utils.py:42-46
        pts = (out["OUTPUT_2D_LANDMARKS"].detach().cpu().numpy() * 256.0).clip(0, 256).astype(int)     # :147-152 (identity frame)
        return post, pts

    tried, counted = {}, {}
    for threads in sorted({t for t in (threads or (1, 8, 32, ncpu)) if t <= ncpu} or {ncpu}):
        torch.set_num_threads(threads)
        for i in range(2):
            one(frames[i])
        # n_images per setting, but never more than ~45 s of it: torch's default of ALL cores is pathologically slow for one 256 x 256
        # image on a many-core host (bench.py's cpu_baseline found the same for the decode alone), and the run must stay bounded
        n, t0 = 0, time.perf_counter()
        while n < n_images and (n < 3 or time.perf_counter() - t0 < budget_s):
            one(frames[n % len(frames)])
            n += 1
        tried[threads], counted[threads] = n / (time.perf_counter() - t0), n
        print(f"cpu predictor, {threads} threads: {tried[threads]:.2f} img/s over {n} images", file=sys.stderr, flush=True)
    best = max(tried, key=tried.get)
    return {"images_per_s": tried[best], "threads": best, "images_per_s_by_threads": {str(k): v for k, v in tried.items()},
            "images_timed_by_threads": {str(k): v for k, v in counted.items()}, "host_logical_cores": ncpu, "dtype": "f32",
            "what": "DAD3DNet (network.py declaration, fp32, eval) + oracle preprocess + oracle predictor_postprocess (readjust, 2 FLAME decodes, "
                    "445 int landmarks) + 68 landmarks, ONE image per call (predictor.py:97-145), torch CPU"}
