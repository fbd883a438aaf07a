#!/usr/bin/env python3
"""tests/golden/projection_golden.npz: outputs of the reference's OWN `visualize.get_2d_keypoints` (visualize.py:10-22),
imported unmodified from /root/reference with stand-ins for its uninstalled imports (`fire`, `cv2`, `demo_utils`: none is
touched by that function), on seeded annotation-shaped inputs (5023 vertices, a rigid model-view matrix, an OpenGL-style
perspective matrix scaled to pixels like the dataset's). Authoring container only."""
import os
import sys
import types

import numpy as np

REF = os.environ.get("DAD3D_REFERENCE_ROOT", "/root/reference")
OUT = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "projection_golden.npz")


def annotation(rng, n=5023):
    v = (rng.standard_normal((n, 3)) * 0.08).astype(np.float32)
    a = rng.uniform(-0.6, 0.6, 3)
    rx = np.array([[1, 0, 0], [0, np.cos(a[0]), -np.sin(a[0])], [0, np.sin(a[0]), np.cos(a[0])]])
    ry = np.array([[np.cos(a[1]), 0, np.sin(a[1])], [0, 1, 0], [-np.sin(a[1]), 0, np.cos(a[1])]])
    rz = np.array([[np.cos(a[2]), -np.sin(a[2]), 0], [np.sin(a[2]), np.cos(a[2]), 0], [0, 0, 1]])
    mv = np.eye(4)
    mv[:3, :3] = rz @ ry @ rx
    mv[:3, 3] = [rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1), -rng.uniform(0.6, 1.2)]
    w, h, f = 800.0, 640.0, rng.uniform(900, 1400)
    proj = np.array([[f, 0, -w / 2, 0], [0, f, -h / 2, 0], [0, 0, 1.01, 0.1], [0, 0, -1, 0]])  # clip.w = -z_world > 0
    return {"vertices": v.tolist(), "model_view_matrix": mv.astype(np.float32).tolist(), "projection_matrix": proj.astype(np.float32).tolist()}, int(h)


def main():
    for name in ("fire", "cv2", "demo_utils"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["fire"].Fire = lambda *a, **k: None
    sys.modules["demo_utils"].draw_points = sys.modules["demo_utils"].get_output_path = None
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import visualize  # the reference's module

    rng = np.random.default_rng(2024)
    out = {}
    for i in range(3):
        data, h = annotation(rng)
        out[f"vertices_{i}"] = np.array(data["vertices"], np.float32)
        out[f"model_view_{i}"] = np.array(data["model_view_matrix"], np.float32)
        out[f"projection_{i}"] = np.array(data["projection_matrix"], np.float32)
        out[f"height_{i}"] = np.int64(h)
        out[f"keypoints_{i}"] = visualize.get_2d_keypoints(data, h)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes", out["keypoints_0"][:3], out["keypoints_0"].min(0), out["keypoints_0"].max(0))


if __name__ == "__main__":
    main()
