#!/usr/bin/env python3
"""Diagnostics: the DAD-3DNet forward (PyTorch-ROCm plumbing, bf16 channels-last, random weights) at batch 64, 30 times, for
`rocprofv3 --kernel-trace --stats` -- which kernels the 7 ms are made of."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dad_3dheads_amd import landmarks, synthetic
from dad_3dheads_amd.predictor import FaceMeshPredictor

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64
st = synthetic.load_static()
pred = FaceMeshPredictor.random_init(dtype=torch.bfloat16, tune=True, cuda_id=0, flame_model=synthetic.synthetic_flame_model(0, st),
                                     landmarks=landmarks.canonical("445", st))
x = torch.randn(batch, 3, 256, 256, device="cuda")
for _ in range(5):
    pred.process(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(60):
    pred.process(x)
torch.cuda.synchronize()
print("CNN ms per batch", (time.perf_counter() - t0) / 60 * 1e3)
