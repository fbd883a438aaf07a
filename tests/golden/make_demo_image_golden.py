#!/usr/bin/env python3
"""tests/golden/demo_image.npz: BASELINE configs[0]'s input, the reference's demo image `images/demo_heads/1.jpeg` (766 x 954,
(c) the DAD-3DHeads authors, CC BY-NC-SA 4.0 -- a data asset, carried as the JPEG's own bytes so the GPU box decodes it with
the same PIL), and what the preprocessing oracle (oracle/preprocess_ref.py) makes of it: the 206 x 256 uint8 image after
LongestMaxSize, the geometry, and a float checksum of the normalised tensor. Authoring container only. cv2 is absent, so the
frozen resize is the RESTATED OpenCV fixed-point path, cross-checked here against float bilinear sampling (+-1 LSB)."""
import io
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = os.environ.get("DAD3D_REFERENCE_ROOT", "/root/reference")
OUT = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "demo_image.npz")


def main():
    from oracle import preprocess_ref as pr

    raw = open(os.path.join(REF, "images", "demo_heads", "1.jpeg"), "rb").read()
    img = np.asarray(Image.open(io.BytesIO(raw)).convert("RGB"))
    nh, nw, top, left, scale = pr.geometry(*img.shape[:2])
    small = pr.resize_linear_u8(img, nh, nw)
    fl = F.interpolate(torch.from_numpy(img.copy()).permute(2, 0, 1)[None].float(), size=(nh, nw), mode="bilinear",
                       align_corners=False, antialias=False)[0].permute(1, 2, 0).numpy()
    assert np.abs(small.astype(np.float64) - fl).max() < 1.0  # the fixed-point path is float bilinear to within rounding
    x = pr.transform(img)
    np.savez_compressed(OUT, jpeg=np.frombuffer(raw, dtype=np.uint8), shape=np.array(img.shape), geometry=np.array([nh, nw, top, left]),
                        scale=np.float64(scale), resized=small, transformed_sum=np.float64(x.astype(np.float64).sum()),
                        transformed_corner=x[:, :4, 23:29].copy())
    print("wrote", OUT, os.path.getsize(OUT), "bytes", img.shape, (nh, nw, top, left), scale)


if __name__ == "__main__":
    main()
