#!/usr/bin/env python3
"""End-to-end predictor throughput, reported SEPARATELY from the headline metric ("never mixed into the
metric"): uint8 frames resident in HBM -> normalise -> DAD-3DNet (hand-declared ResNet-50 + BiFPN + heads, random
weights, PyTorch-ROCm bf16 channels-last) -> re-adjust kernel -> fused decode + 445 landmarks, all on one stream with
no host copy in between. Prints one JSON object with the split between the CNN and the decode hot path."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dad_3dheads_amd import landmarks, synthetic  # noqa: E402
from dad_3dheads_amd.predictor import FaceMeshPredictor  # noqa: E402


def timed(fn, iters, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    st = synthetic.load_static()
    model = synthetic.synthetic_flame_model(0, st)
    out = {"batch": batch, "data": "synthetic uint8 256x256x3, random-init weights"}
    for name, dtype in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
        pred = FaceMeshPredictor.random_init(dtype=dtype, tune=True, cuda_id=0, flame_model=model, landmarks=landmarks.canonical("445", st))
        g = torch.Generator().manual_seed(0)
        images = torch.randint(0, 255, (batch, 256, 256, 3), dtype=torch.uint8, generator=g).cuda()
        x = torch.randn(batch, 3, 256, 256, device="cuda")
        t_all = timed(lambda: pred.predict_tensor(images), 20, 5)
        t_cnn = timed(lambda: pred.process(x), 20, 5)
        params = pred.process(x)["OUTPUT_3DMM_PARAMS"].contiguous()
        t_dec = timed(lambda: pred.head_mesh.decode(params, landmarks=False, landmarks_px=True), 200, 20)
        out[name] = {"images_per_s_end_to_end": batch / t_all, "ms_per_batch_end_to_end": t_all * 1e3,
                     "ms_cnn_only": t_cnn * 1e3, "ms_decode_only": t_dec * 1e3,
                     "decode_share_of_batch_time": t_dec / t_all}
    # the whole batch replayed from ONE hipGraph (the CNN's ~200 kernels + the glue + the decode's neighbours; bf16)
    pred = FaceMeshPredictor.random_init(dtype=torch.bfloat16, tune=True, graph=True, cuda_id=0, flame_model=model,
                                         landmarks=landmarks.canonical("445", st))
    t_graph = timed(lambda: pred.predict_tensor(images), 20, 5)
    out["bf16"]["images_per_s_end_to_end_hipgraph"] = batch / t_graph
    out["bf16"]["ms_per_batch_end_to_end_hipgraph"] = t_graph * 1e3
    # single image, the reference's call pattern: ~200 launch-bound kernels at batch 1 -> replay them from a hipGraph
    import numpy as np

    img1 = np.random.default_rng(0).integers(0, 255, (256, 256, 3), dtype=np.uint8)
    for name, graph in (("eager", False), ("hipgraph", True)):
        pred = FaceMeshPredictor.random_init(dtype=torch.bfloat16, tune=True, graph=graph, cuda_id=0, flame_model=model,
                                             landmarks=landmarks.canonical("445", st))
        t = timed(lambda: pred(img1), 50, 10)
        out.setdefault("single_image_ms", {})[name] = t * 1e3
    print(json.dumps(out))


if __name__ == "__main__":
    main()
