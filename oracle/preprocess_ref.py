"""CPU restatement of `FaceMeshPredictor._transform` + `_array_to_batch`  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The reference's preprocessing (predictor.py:80-95,195-203) is three calls into third-party packages that are absent from
this image: albumentations `LongestMaxSize` (-> `cv2.resize(..., INTER_LINEAR)` on the uint8 image, new size via
`py3round`), `PadIfNeeded` (centred, BORDER_CONSTANT, value 0) and `Normalize`. PARITY UNPINNED against the cv2 / albumentations
binaries: they cannot be run here, and cv2's wheels may route 8-bit linear resizing through IPP, which differs from OpenCV's
own C++ by one LSB on some pixels. What is restated is the published C++ path and albumentations' functional code:

  OpenCV modules/imgproc/src/resize.cpp, resize() with INTER_LINEAR on CV_8UC3:
      inv_scale = dst / src (double); scale = 1 / inv_scale
      per destination coordinate d: f = (float)((d + 0.5) * scale - 0.5); s = cvFloor(f); f -= s
          columns only: s < 0 -> (f, s) = (0, 0);  s >= src - 1 -> (f, s) = (0, src - 1); rows keep f, the row loop clamps
          the two row indices (round 3, ADVICE: rounds 1-2 zeroed the fraction for the rows too)
          coefficients = saturate_cast<short>((1 - f) * 2048), saturate_cast<short>(f * 2048)        (round half to even)
      HResizeLinear<uchar,int,short>:  row[d] = S[s] * c0 + S[s + 1] * c1
      VResizeLinear<uchar,int,short>:  dst = (((b0 * (row0 >> 4)) >> 16) + ((b1 * (row1 >> 4)) >> 16) + 2) >> 2
  albumentations functional.normalize: mean = float32(mean) * 255; denominator = reciprocal(float32(std) * 255);
      img = float32(img); img -= mean; img *= denominator
  albumentations py3round, model_training/model/utils.py:71-77 calculate_paddings: see dad_3dheads_amd/predictor.py

Cross-checks kept in tests/: against float bilinear sampling with the same half-pixel rule (torch, +-1 LSB) and frozen outputs
for the reference's demo image. The HIP kernel (csrc/preprocess.hip) is held to this module bit for bit."""
from __future__ import annotations

import numpy as np

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def py3round(x: float) -> int:
    if abs(round(x) - x) == 0.5:
        return int(2.0 * round(x / 2.0))
    return int(round(x))


def geometry(h: int, w: int, size: int = 256):
    """-> (new_h, new_w, pad_top, pad_left, scale) of predictor.py:117-123 / LongestMaxSize / PadIfNeeded."""
    scale = size / float(max(h, w))
    nh, nw = py3round(h * scale), py3round(w * scale)
    m = max(nh, nw)  # calculate_paddings pads to the longer side (== size)
    return nh, nw, int((m - nh) / 2), int((m - nw) / 2), scale


def _taps(dst_n: int, src_n: int, horizontal: bool):
    """resize()'s coefficient set-up. Only the HORIZONTAL taps get (f, s) = (0, border) at the image borders; for the rows the
    fraction is kept and the two row indices are clamped in the row loop (`clip(sy + k, 0, ssize.height)`)."""
    scale = 1.0 / (float(dst_n) / float(src_n))
    d = np.arange(dst_n, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    if horizontal:
        lo, hi = s < 0, s >= src_n - 1
        f = np.where(lo | hi, np.float32(0), f).astype(np.float32)
        s = np.where(lo, 0, np.where(hi, src_n - 1, s))
    c0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)  # cvRound: half to even
    c1 = np.rint(f * np.float32(2048)).astype(np.int64)
    return np.clip(s, 0, src_n - 1), np.clip(s + 1, 0, src_n - 1), c0, c1


def resize_linear_u8(img: np.ndarray, nh: int, nw: int) -> np.ndarray:
    """cv2.resize(img, (nw, nh), interpolation=cv2.INTER_LINEAR) for uint8 HxWxC, OpenCV's C++ fixed-point path."""
    assert img.dtype == np.uint8 and img.ndim == 3
    h, w = img.shape[:2]
    if (nh, nw) == (h, w):
        return img.copy()
    sx, sx1, a0, a1 = _taps(nw, w, True)
    sy, sy1, b0, b1 = _taps(nh, h, False)
    src = img.astype(np.int64)
    rows = src[:, sx, :] * a0[None, :, None] + src[:, sx1, :] * a1[None, :, None]  # [h, nw, c]
    r0, r1 = rows[sy], rows[sy1]
    out = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def transform(img: np.ndarray, size: int = 256, mean=MEAN, std=STD) -> np.ndarray:
    """uint8 RGB HxWx3 -> float32 [3, size, size]: `_transform` then the HWC->CHW of `_array_to_batch`."""
    h, w = img.shape[:2]
    nh, nw, top, left, _ = geometry(h, w, size)
    small = resize_linear_u8(np.ascontiguousarray(img), nh, nw)
    canvas = np.zeros((size, size, 3), dtype=np.uint8)
    canvas[top : top + nh, left : left + nw] = small
    m = np.array(mean, dtype=np.float32)
    m *= 255.0
    s = np.array(std, dtype=np.float32)
    s *= 255.0
    den = np.reciprocal(s, dtype=np.float32)
    out = canvas.astype(np.float32)
    out -= m
    out *= den
    return np.ascontiguousarray(np.transpose(out, (2, 0, 1)))
