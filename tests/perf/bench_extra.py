#!/usr/bin/env python3
"""Secondary measurements (not the headline bench.py line): BASELINE configs 3 and 5 on one MI355X.

  config 3: batch 256 head_mesh decode (3d_vertices only + projected 3-component), fused decode kernel
  config 5 (per-GPU share, B=64): decode(to_2d=False, z flipped) -> vertex normals -> Phong light -> z-buffer raster
            of 9976 triangles onto 256x256x3, plus the PNCC variant (6270 triangles, NCC colours)
CPU references timed on this host: the reference's own Sim3DR C++ when oracle/_ref/libsim3dr_ref.so was shipped,
else the C port. Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dad_3dheads_amd import landmarks, synthetic  # noqa: E402
from dad_3dheads_amd.head_mesh import HeadMesh  # noqa: E402
from dad_3dheads_amd.Sim3DR import Mesh  # noqa: E402
from oracle import sim3dr_ref  # noqa: E402


def gpu_time(fn, iters=200, warm=20):
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
    st = synthetic.load_static()
    model = synthetic.synthetic_flame_model(0, st)
    hm = HeadMesh(flame_model=model, landmarks=landmarks.canonical("445", st), static=st, device=0)
    out = {}
    # ---- config 3 --------------------------------------------------------------------------------------
    p256 = torch.from_numpy(synthetic.synthetic_params(256, seed=0)).cuda()
    bufs = {}
    t = gpu_time(lambda: hm.flame.decode(p256, verts3d=True, proj=False, out=bufs))
    out["config3_decode_b256"] = {"images_per_s": 256 / t, "us_per_batch": t * 1e6,
                                   "tflops_algorithmic": 14.5e6 * 256 / t / 1e12}
    for b in (64, 1024, 2048):
        pb = torch.from_numpy(synthetic.synthetic_params(b, seed=1)).cuda()
        bb = {}
        t = gpu_time(lambda: hm.decode(pb, to_2d=True, landmarks=False, landmarks_px=True, out=bb), iters=100)
        out[f"decode_b{b}_full_outputs"] = {"images_per_s": b / t, "us_per_batch": t * 1e6,
                                            "tflops_algorithmic": 14.5e6 * b / t / 1e12}
    # ---- config 5 share --------------------------------------------------------------------------------
    B = 64
    p = torch.from_numpy(synthetic.synthetic_params(B, seed=2)).cuda()
    faces, fw = st["faces"], st["faces_wo_ears"]
    mesh, mesh_pncc = Mesh(faces, 5023, device=0), Mesh(fw, 5023, device=0)
    dec = {}
    hm.flame.decode(p, proj=True, to_2d=False, flip_z=True, out=dec)
    verts = dec["proj"]
    img = torch.zeros((B, 256, 256, 3), dtype=torch.uint8, device="cuda")
    normals = mesh.get_normal(verts)
    light = mesh.phong_light(verts, normals)
    tmpl = torch.from_numpy(st["template_geo"]).cuda()
    ncc = ((tmpl - tmpl.min(0).values) / (tmpl.max(0).values - tmpl.min(0).values)).float()[None].expand(B, -1, -1).contiguous()
    t_norm = gpu_time(lambda: mesh.get_normal(verts, out=normals))
    t_light = gpu_time(lambda: mesh.phong_light(verts, normals))
    t_nlight = gpu_time(lambda: mesh.phong_light(verts, None))
    t_rast = gpu_time(lambda: mesh.rasterize(verts, light, img))
    t_pncc = gpu_time(lambda: mesh_pncc.rasterize(verts, ncc, img))

    def pipeline():
        hm.flame.decode(p, proj=True, to_2d=False, flip_z=True, out=dec)
        lt = mesh.phong_light(dec["proj"], None)  # normals + light in one launch
        mesh.rasterize(dec["proj"], lt, img)

    t_pipe = gpu_time(pipeline, iters=100)
    lbuf = torch.empty_like(verts)

    def pipeline3():  # three launches: decode, geometry (+ normals + light), tiles
        hm.flame.decode(p, proj=True, to_2d=False, flip_z=True, out=dec)
        mesh.render(dec["proj"], img, light_out=lbuf)

    t_pipe3 = gpu_time(pipeline3, iters=100)
    t_render = gpu_time(lambda: mesh.render(verts, img, light_out=lbuf))

    # the same chain with two batches in flight: one forked decode handle, one mesh handle and one buffer set per stream
    lanes = []
    for i in range(2):
        lanes.append({"hm": hm if i == 0 else hm.fork(), "mesh": Mesh(faces, 5023, device=0), "stream": torch.cuda.Stream(),
                      "p": torch.from_numpy(synthetic.synthetic_params(B, seed=2 + i)).cuda(), "dec": {}, "img": torch.zeros_like(img),
                      "light": torch.empty_like(verts)})
    torch.cuda.synchronize()
    turn = [0]

    def pipeline2():
        ln = lanes[turn[0] % 2]
        turn[0] += 1
        with torch.cuda.stream(ln["stream"]):
            ln["hm"].flame.decode(ln["p"], proj=True, to_2d=False, flip_z=True, out=ln["dec"])
            ln["mesh"].render(ln["dec"]["proj"], ln["img"], light_out=ln["light"])

    def wall(fn, iters=200, warm=20):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters

    t_pipe2 = wall(pipeline2)
    v0 = np.ascontiguousarray(verts[0].cpu().numpy())
    x, y = v0[:, 0][faces], v0[:, 1][faces]
    tests = float(np.sum(np.clip(np.floor(x.max(1)) - np.ceil(x.min(1)) + 1, 0, None) * np.clip(np.floor(y.max(1)) - np.ceil(y.min(1)) + 1, 0, None)))
    out["config5_share_b64"] = {
        "get_normal": {"images_per_s": B / t_norm, "us_per_batch": t_norm * 1e6, "GBps_algorithmic": B * 120552 / t_norm / 1e9},
        "phong_light": {"images_per_s": B / t_light, "us_per_batch": t_light * 1e6},
        "normals+phong_one_launch": {"images_per_s": B / t_nlight, "us_per_batch": t_nlight * 1e6},
        "rasterize_9976": {"images_per_s": B / t_rast, "us_per_batch": t_rast * 1e6, "GBps_algorithmic": B * 513768 / t_rast / 1e9,
                           "bbox_pixel_tests_image0": tests, "Gtests_per_s": tests * B / t_rast / 1e9},
        "pncc_6270": {"images_per_s": B / t_pncc, "us_per_batch": t_pncc * 1e6},
        "decode+normals+light+raster": {"images_per_s": B / t_pipe, "us_per_batch": t_pipe * 1e6},
        "render_two_launches": {"images_per_s": B / t_render, "us_per_batch": t_render * 1e6},
        "decode+render_three_launches": {"images_per_s": B / t_pipe3, "us_per_batch": t_pipe3 * 1e6},
        "decode+normals+light+raster_two_streams": {"images_per_s": B / t_pipe2, "us_per_batch": t_pipe2 * 1e6},
    }
    # ---- CPU references ----------------------------------------------------------------------------------
    kind = "reference" if sim3dr_ref.available("reference") else "port"
    orc = sim3dr_ref.Sim3DROracle(kind)
    col = np.ascontiguousarray(light[0].cpu().numpy())
    def cpu_time(fn, budget=3.0):
        fn(); n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget:
            fn(); n += 1
        return (time.perf_counter() - t0) / n
    tn = cpu_time(lambda: orc.get_normal(v0, faces))
    tr = cpu_time(lambda: orc.rasterize(v0, faces, col, height=256, width=256, channel=3))
    tp = cpu_time(lambda: sim3dr_ref.render_pipeline_ref(orc, v0.copy(), faces, np.zeros((256, 256, 3), np.uint8)))
    out["cpu_sim3dr"] = {"kind": kind, "cores": 1, "get_normal_images_per_s": 1 / tn, "rasterize_images_per_s": 1 / tr,
                         "render_pipeline_images_per_s": 1 / tp}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
