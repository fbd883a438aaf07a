#!/usr/bin/env python3
"""Training-side measurement: forward + backward through the reference's two mesh losses
(`Vertices3DLoss` with zero_rotation, `ReprojectionLoss` to_2d) on one MI355X, and the same through torch autograd
over the CPU oracle (= how the reference obtains these gradients) on this host. Prints one JSON object.

    python tests/perf/bench_train.py [batch ...]      default 64 256
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dad_3dheads_amd import landmarks, synthetic  # noqa: E402
from dad_3dheads_amd.flame import FLAME_CONSTS  # noqa: E402
from dad_3dheads_amd.head_mesh import HeadMesh  # noqa: E402
from dad_3dheads_amd.losses import normalize_to_cube  # noqa: E402
from oracle import flame_ref  # noqa: E402


def gpu_time(fn, iters=100, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    batches = [int(x) for x in sys.argv[1:]] or [64, 256]
    st = synthetic.load_static()
    model = synthetic.synthetic_flame_model(0, st)
    hm = HeadMesh(flame_model=model, landmarks=landmarks.canonical("445", st), static=st, device=0)
    regions = ([1.0, 0.5], [torch.arange(0, 5023, 3).cuda(), torch.from_numpy(landmarks.canonical("445", st)).cuda()])
    l1 = torch.nn.L1Loss()
    from dad_3dheads_amd.losses import RegionTables, _CubeRegionLoss, _WeightedPointLoss  # the modules' fused kernels

    region_tables = RegionTables(regions[0], [i.cpu().numpy() for i in regions[1]], 5023, torch.device("cuda", 0))
    out = {}
    for b in batches:
        params = torch.from_numpy(synthetic.synthetic_params(b, seed=1)).cuda()
        tgt3 = torch.randn((b, 5023, 3), device="cuda")
        tgt2 = torch.randn((b, 5023, 2), device="cuda") * 128 + 128

        def loss_of(q, fused):  # Vertices3DLoss + 1e-2 ReprojectionLoss on one prediction tensor
            v = hm.vertices_3d(q, zero_rotation=True)
            pr = hm.reprojected_vertices(q, to_2d=True)
            if fused:  # what dad_3dheads_amd.losses.{Vertices3DLoss,ReprojectionLoss} run (csrc/mesh_losses.hip)
                return _CubeRegionLoss.apply(v, tgt3, region_tables, 0) + _WeightedPointLoss.apply(pr, tgt2, region_tables, 0) * 1e-2
            loss = sum(l1(normalize_to_cube(v[:, i]), normalize_to_cube(tgt3[:, i])) * w for w, i in zip(*regions))
            return loss + sum(l1(pr[:, i], tgt2[:, i]) * w for w, i in zip(*regions)) * 1e-2

        def step(fused=True):
            p = params.clone().requires_grad_(True)
            loss_of(p * 1.0, fused).backward()
            return p.grad

        gv, gp = torch.randn((b, 5023, 3), device="cuda"), torch.randn((b, 5023, 2), device="cuda")

        def decode_only():  # the two forward launches + the two backward passes, no loss arithmetic
            p = params.clone().requires_grad_(True)
            q = p * 1.0
            v = hm.vertices_3d(q, zero_rotation=True)
            pr = hm.reprojected_vertices(q, to_2d=True)
            torch.autograd.backward([v, pr], [gv, gp])

        t_step, t_dec = gpu_time(step), gpu_time(decode_only)
        t_step_torch = gpu_time(lambda: step(False))
        fused_vs_torch = float((step(True) - step(False)).abs().max() / step(False).abs().max())

        # the same step captured once into a hipGraph (its launches replayed without host work)
        static_p = params.clone().requires_grad_(True)

        def graph_body():
            loss_of(static_p * 1.0, True).backward()

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                static_p.grad = None
                graph_body()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        static_p.grad = None
        with torch.cuda.graph(graph):
            graph_body()
        t_graph = gpu_time(graph.replay)
        step()  # eager gradient of the same parameters
        g_eager = step()
        graph.replay()
        torch.cuda.synchronize()
        graph_err = float((static_p.grad - g_eager).abs().max() / g_eager.abs().max())
        # pieces of one backward pass
        layer = hm.flame
        tables = layer.decode_tables()
        from dad_3dheads_amd import _lib  # noqa: E402

        lib = _lib.load()
        inputs = torch.empty((b, 436), device="cuda")
        c72 = torch.empty((b, 72), device="cuda")
        posed = torch.empty((b, 15069), device="cuda")
        g_posed, g_c, g_params = torch.empty_like(posed), torch.empty_like(c72), torch.empty_like(params)
        g_in = torch.empty((b, 436), device="cuda")
        h = layer._handle
        t_chain = gpu_time(lambda: lib.dad3d_flame_pose_chain(h, params.data_ptr(), b, inputs.data_ptr(), c72.data_ptr(), None))
        v3 = torch.empty((b, 5023, 3), device="cuda")
        t_fwd = gpu_time(lambda: lib.dad3d_flame_decode_posed(h, params.data_ptr(), b, _lib.ZERO_ROTATION, v3.data_ptr(), None, posed.data_ptr(), None))
        t_fwd0 = gpu_time(lambda: lib.dad3d_flame_decode(h, params.data_ptr(), b, _lib.ZERO_ROTATION, v3.data_ptr(), None, None, None, None))
        t_vert = gpu_time(lambda: lib.dad3d_flame_decode_backward(h, b, _lib.ZERO_ROTATION, c72.data_ptr(), posed.data_ptr(), gv.data_ptr(),
                                                                  None, g_posed.data_ptr(), g_c.data_ptr(), None))
        t_gemm2 = gpu_time(lambda: torch.matmul(g_posed, tables.basis.T, out=g_in))
        t_gemm_hip = gpu_time(lambda: lib.dad3d_flame_grad_inputs(h, g_posed.data_ptr(), b, g_in.data_ptr(), None))
        t_vjp = gpu_time(lambda: lib.dad3d_flame_pose_chain_backward(h, params.data_ptr(), b, g_in.data_ptr(), g_c.data_ptr(),
                                                                     g_params.data_ptr(), None))
        out[f"b{b}"] = {
            "losses_fwd_bwd_us": t_step * 1e6, "images_per_s": b / t_step,
            "two_decodes_fwd_bwd_us": t_dec * 1e6,
            "losses_fwd_bwd_torch_loss_graph_us": t_step_torch * 1e6, "fused_vs_torch_grad_rel_err": fused_vs_torch,
            "losses_fwd_bwd_hipgraph_us": t_graph * 1e6, "images_per_s_hipgraph": b / t_graph,
            "hipgraph_vs_eager_grad_rel_err": graph_err,
            "forward_launch_us": {"3d_vertices + v_posed saved": t_fwd * 1e6, "3d_vertices only (inference)": t_fwd0 * 1e6},
            "backward_pieces_us": {"pose_chain": t_chain * 1e6, "vertex_backward": t_vert * 1e6,
                                   "grad_inputs split-K MFMA kernel + reduction (product)": t_gemm_hip * 1e6,
                                   "grad_inputs through rocBLAS (round 1)": t_gemm2 * 1e6, "pose_chain_vjp": t_vjp * 1e6},
            "vertex_backward_GBps": b * 4 * 60276 / t_vert / 1e9,
        }
    # CPU: torch autograd over the oracle, the reference's way, bounded sample
    fc = flame_ref.FlameConstants.from_model(model)
    b = 16
    params = torch.from_numpy(synthetic.synthetic_params(b, seed=1))
    tgt3, tgt2 = torch.randn((b, 5023, 3)), torch.randn((b, 5023, 2)) * 128 + 128
    reg = ([1.0, 0.5], [torch.arange(0, 5023, 3), torch.from_numpy(landmarks.canonical("445", st))])
    l1 = torch.nn.L1Loss()

    def cpu_step():
        p = params.clone().requires_grad_(True)
        q = p * 1.0
        v = flame_ref.vertices_3d(fc, q, zero_rotation=True)
        pr = flame_ref.reprojected_vertices(fc, q, to_2d=True)
        loss = sum(l1(normalize_to_cube(v[:, i]), normalize_to_cube(tgt3[:, i])) * w for w, i in zip(*reg))
        loss = loss + sum(l1(pr[:, i], tgt2[:, i]) * w for w, i in zip(*reg)) * 1e-2
        loss.backward()

    best = None
    for threads in (1, 8, 32):
        torch.set_num_threads(threads)
        cpu_step()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 4.0:
            cpu_step()
            n += 1
        ips = b * n / (time.perf_counter() - t0)
        if best is None or ips > best[1]:
            best = (threads, ips)
    out["cpu_oracle_autograd"] = {"batch": b, "threads": best[0], "images_per_s": best[1], "host_cores": os.cpu_count()}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
