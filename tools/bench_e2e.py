#!/usr/bin/env python3
"""End-to-end predictor throughput, reported SEPARATELY from the headline metric ("never mixed into the
metric"): uint8 frames resident in HBM -> normalise -> DAD-3DNet (hand-declared ResNet-50 + BiFPN + heads, random
weights, PyTorch-ROCm bf16 channels-last) -> re-adjust kernel -> fused decode + 445 landmarks, all on one stream with
no host copy in between. Prints one JSON object with the split between the CNN and the decode hot path.

    python tools/bench_e2e.py [batch] [--cpu] [--cpu-images N] [--quick]

`--cpu`: the north star's literal comparator in the same run -- "the reference CPU predictor's images/sec": the SAME DAD-3DNet
declaration (fp32, eval, random weights) on the host CPU, one image per call like predictor.py:97-145 (preprocess -> CNN -> `.cpu()`
-> readjust -> vertices_3d + reprojected_vertices -> 68 landmarks; plus the 445-landmark int gather of demo_utils.py:42-46), through
the CPU oracle (oracle/preprocess_ref.py, oracle/flame_ref.py), with torch.set_num_threads in {1, 8, all}; >= 30 images each. Reported
beside the GPU figures with the ratios; never mixed into the headline metric."""
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
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    want_cpu = "--cpu" in sys.argv
    quick = "--quick" in sys.argv  # bf16 only, no batch-64 hipGraph, single image through the hipGraph only (each variant re-tunes MIOpen: minutes)
    cpu_images = int(sys.argv[sys.argv.index("--cpu-images") + 1]) if "--cpu-images" in sys.argv else 30
    if "--cpu-images" in sys.argv:
        argv = [a for a in argv if a != str(cpu_images)] or argv
    batch = int(argv[0]) if argv else 64
    st = synthetic.load_static()
    model = synthetic.synthetic_flame_model(0, st)
    out = {"batch": batch, "data": "synthetic uint8 256x256x3, random-init weights"}
    for name, dtype in (("bf16", torch.bfloat16), ("fp16", torch.float16))[: 1 if quick else 2]:
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
    if not quick:
        pred = FaceMeshPredictor.random_init(dtype=torch.bfloat16, tune=True, graph=True, cuda_id=0, flame_model=model,
                                             landmarks=landmarks.canonical("445", st))
        t_graph = timed(lambda: pred.predict_tensor(images), 20, 5)
        out["bf16"]["images_per_s_end_to_end_hipgraph"] = batch / t_graph
        out["bf16"]["ms_per_batch_end_to_end_hipgraph"] = t_graph * 1e3
    # single image, the reference's call pattern: ~200 launch-bound kernels at batch 1 -> replay them from a hipGraph
    import numpy as np

    img1 = np.random.default_rng(0).integers(0, 255, (256, 256, 3), dtype=np.uint8)
    for name, graph in (("eager", False), ("hipgraph", True))[1 if quick else 0:]:
        pred = FaceMeshPredictor.random_init(dtype=torch.bfloat16, tune=True, graph=graph, cuda_id=0, flame_model=model,
                                             landmarks=landmarks.canonical("445", st))
        t = timed(lambda: pred(img1), 50, 10)
        out.setdefault("single_image_ms", {})[name] = t * 1e3
    if want_cpu:
        # the CPU leg imports the oracle: it lives under tests/perf/ like every other script that may (tests/perf/README.md)
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "perf"))
        from cpu_predictor import cpu_reference_predictor

        cpu = cpu_reference_predictor(model, landmarks.canonical("445", st), cpu_images)
        out["cpu_reference_predictor"] = cpu
        out["ratio_gpu_batch_end_to_end_vs_cpu_predictor"] = out["bf16"]["images_per_s_end_to_end"] / cpu["images_per_s"]
        out["ratio_gpu_single_image_call_vs_cpu_predictor"] = (1e3 / out["single_image_ms"]["hipgraph"]) / cpu["images_per_s"]
        out["north_star_target_ratio"] = 200
    print(json.dumps(out))


if __name__ == "__main__":
    main()
