#!/usr/bin/env python3
"""Diagnostics: A/B timing of the fused decode through the C ABI for the library in $DAD3D_LIB_PATH (default: product).

    DAD3D_LIB_PATH=tools/_variants/lib_x.so python tools/ab_decode.py [tag]

Prints one line: hipEvent us/launch at B = 64 / 128 / 256 / 1024 (445-landmark path, all outputs), golden check of the B = 64
outputs (tests/golden/decode_golden.npz, the reference's own HeadMesh), hand-off time-outs."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dad_3dheads_amd import _lib, landmarks, synthetic  # noqa: E402
from dad_3dheads_amd.head_mesh import HeadMesh  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("DAD3D_LIB_PATH", "product"))
st = synthetic.load_static()
hm = HeadMesh(flame_model=synthetic.synthetic_flame_model(0, st), landmarks=landmarks.canonical("445", st), static=st, device=0)
lib = _lib.load()
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "decode_golden.npz"))
res = {}
for b, iters in ((64, 3000), (128, 1500), (256, 1000), (1024, 300)):
    p = torch.from_numpy(g["b64_params"] if b == 64 else synthetic.synthetic_params(b, seed=b)).cuda()
    v3 = torch.empty((b, 5023, 3), device="cuda"); pr = torch.empty((b, 5023, 2), device="cuda")
    lp = torch.empty((b, 445, 2), dtype=torch.int32, device="cuda")
    call = (hm.flame._handle, p.data_ptr(), b, _lib.TO_2D | _lib.MUTATE_PARAMS, v3.data_ptr(), pr.data_ptr(), None, lp.data_ptr(), None)
    for _ in range(300):
        _lib.check(lib.dad3d_flame_decode(*call))
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            lib.dad3d_flame_decode(*call)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    res[b] = best
    if b == 256:  # lets two variants be compared beyond the B = 64 goldens
        res["sum256"] = f"{float(v3.double().abs().sum()):.6e}/{float(pr.double().abs().sum()):.6e}/{int(lp.long().sum())}"
    if b == 64:
        sub = g["b64_subset"]
        ev = float(np.abs(v3.cpu().numpy()[:, sub] - g["b64_v3d_sub"]).max()); ep = float(np.abs(pr.cpu().numpy()[:, sub] - g["b64_proj_sub"]).max())
        d = lp.cpu().numpy() != g["b64_lmk_px"]
        frac = np.abs(g["b64_lmk_xy"] - np.round(g["b64_lmk_xy"]))
        ok = ev < 5e-6 and ep < 1e-3 and bool((frac[d] < 1e-3).all())
        res["golden"] = f"{'OK' if ok else 'MISMATCH'} dv={ev:.1e} dp={ep:.1e} lmk_diff={int(d.sum())}"
n = C.c_uint()
_lib.check(lib.dad3d_flame_handoff_timeouts(hm.flame._handle, C.byref(n)))
print(f"AB {tag:28s} B64 {res[64]:6.2f} us  B128 {res[128]:6.2f} us  B256 {res[256]:6.2f} us  B1024 {res[1024]:7.2f} us  golden {res['golden']}  sum256 {res['sum256']}  timeouts {n.value}", flush=True)
