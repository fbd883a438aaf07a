#!/usr/bin/env python3
"""Same-call A/B of dL/d[betas | pose feature] = dL/d(v_posed) . basis^T (the transpose of the blend-shape GEMM, flame.py:212-221; caller
vertices_3d_loss.py:14-49): the repo's split-K fp32-MFMA kernel (dad3d_flame_grad_inputs) against the library GEMM torch.matmul dispatches
(rocBLAS / hipBLASLt) at the batch sizes either side of the host mirror's crossover (autograd.GRAD_INPUTS_HIP_MAX_BATCH = 96).
    python tools/grad_inputs_ab.py > profiles/r06_grad_inputs_ab.txt"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dad_3dheads_amd import _lib, landmarks, synthetic  # noqa: E402
from dad_3dheads_amd.head_mesh import HeadMesh  # noqa: E402

st = synthetic.load_static()
hm = HeadMesh(flame_model=synthetic.synthetic_flame_model(0, st), landmarks=landmarks.canonical("445", st), static=st, device=0)
lib, h = _lib.load(), hm.flame._handle
tables = hm.flame.decode_tables()
stream = torch.cuda.current_stream().cuda_stream
print("batch | own split-K kernel us | library GEMM us | max |diff| / max |value| over the 400 betas | faster")
for b in (16, 64, 96, 128, 192, 256, 512, 1024):
    g = torch.randn((b, 5023 * 3), device="cuda") * 1e-3
    out = torch.empty((b, tables.basis.shape[0]), device="cuda")
    # the kernel's pack and scratch are created by a training forward of this batch size
    p = torch.from_numpy(synthetic.synthetic_params(b, seed=b)).cuda()
    v3 = torch.empty((b, 5023, 3), device="cuda"); posed = torch.empty((b, 5023, 3), device="cuda")
    _lib.check(lib.dad3d_flame_decode_posed(h, p.data_ptr(), b, 0, v3.data_ptr(), None, posed.data_ptr(), stream))
    def own():
        _lib.check(lib.dad3d_flame_grad_inputs(h, g.data_ptr(), b, out.data_ptr(), stream))
    def blas():
        return g @ tables.basis.T
    res = []
    for fn in (own, blas):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 10)
        res.append(best)
    own(); ref = blas(); torch.cuda.synchronize()
    rel = float((out[:, :400] - ref[:, :400]).abs().max() / ref[:, :400].abs().max())  # the betas (pose-feature rows the model cannot move are zero in the kernel)
    print(f"{b:5d} | {res[0]:8.1f} | {res[1]:8.1f} | {rel:.1e} | {'own' if res[0] < res[1] else 'library'}")
