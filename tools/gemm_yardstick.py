#!/usr/bin/env python3
"""Yardstick, never product: the bare fp32 GEMM [B,416] x [416,15069] of the decode (smplx blend shapes, called at
model_training/model/flame.py:212-221) through the vendor library (torch.matmul -> rocBLAS / hipBLASLt) at B = 64 / 256 /
1024, beside the fused kernel's time. Says how much of the fused kernel's distance to the fp32-MFMA peak is this kernel and
how much is what a library GEMM of this shape reaches at all."""
import json, sys
import torch

PEAK = 157.3
out = {}
torch.backends.cuda.matmul.allow_tf32 = False
for b in (64, 256, 1024):
    a = torch.randn(b, 416, device="cuda")
    w = torch.randn(416, 15069, device="cuda")
    wt = torch.randn(15069, 416, device="cuda")
    c = torch.empty(b, 15069, device="cuda")
    res = {}
    for name, fn in (("A[B,416] @ W[416,15069]", lambda: torch.matmul(a, w, out=c)),
                     ("A[B,416] @ Wt[15069,416].T", lambda: torch.matmul(a, wt.t(), out=c))):
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(300):
                fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 300 * 1e3)
        flops = 2.0 * b * 416 * 15069
        res[name] = {"us": best, "tflops": flops / best / 1e6, "frac_fp32_mfma_peak": flops / best / 1e6 / PEAK}
    out[f"B={b}"] = res
print("YARDSTICK " + json.dumps(out))
