"""ctypes front of csrc/cnn_glue.hip: the two fused streaming kernels InferenceNet puts between the framework's convolutions.
Device tensors only (there is no CPU fallback in this package: callers keep the framework's own ops for CPU tensors and for
shapes the kernels do not take -- `supported()` says which)."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch
from torch import Tensor

from . import _lib

_DTYPE = {torch.float32: (0, 4), torch.float16: (1, 8), torch.bfloat16: (2, 8)}


def _nhwc(t: Tensor) -> bool:
    return t.ndim == 4 and t.is_contiguous(memory_format=torch.channels_last)


def supported(y: Tensor, *others: Optional[Tensor]) -> bool:
    """A channels-last CUDA tensor of a dtype the kernels take, channel count a multiple of 16 bytes' worth; `others` (the
    residual, further inputs) the same dtype, device and layout."""
    if not (y.is_cuda and y.dtype in _DTYPE and _nhwc(y) and y.shape[1] % _DTYPE[y.dtype][1] == 0 and y.data_ptr() % 16 == 0):
        return False
    for o in others:
        if o is not None and not (o.is_cuda and o.dtype == y.dtype and o.device == y.device and _nhwc(o) and o.shape[1] == y.shape[1]
                                  and o.data_ptr() % 16 == 0):
            return False
    return True


def bias_act_(y: Tensor, bias: Tensor, z: Optional[Tensor] = None, relu: bool = True) -> Tensor:
    """In place: y = act(y + bias[c] (+ z)). `supported(y, z)` must hold, bias is [C] of y's dtype, z has y's shape."""
    if z is not None and z.shape != y.shape:
        raise ValueError("bias_act_: the residual must have the output's shape")
    if bias.dtype != y.dtype or bias.numel() != y.shape[1] or not bias.is_contiguous():
        raise ValueError("bias_act_: bias must be a contiguous [C] tensor of the output's dtype")
    n, c, h, w = y.shape
    _lib.check(_lib.load().dad3d_nhwc_bias_act(y.data_ptr(), bias.data_ptr(), z.data_ptr() if z is not None else None, n * h * w, c,
                                               _DTYPE[y.dtype][0], int(relu), y.device.index or 0,
                                               torch.cuda.current_stream(y.device).cuda_stream))
    return y


def resize_sum(weights: Sequence[float], xs: Sequence[Tensor], size) -> Tensor:
    """sum_k weights[k] * F.interpolate(xs[k], size=size) (nearest) in one pass; up to three inputs, `supported(x)` for each."""
    k = len(xs)
    if not (1 <= k <= 3 and len(weights) == k):
        raise ValueError("resize_sum: one to three weighted inputs")
    x0 = xs[0]
    if not supported(x0, *xs[1:]) or any(x.dim() != 4 or x.shape[0] != x0.shape[0] for x in xs):
        raise ValueError("resize_sum: every input must be a channels-last CUDA tensor of one dtype, device, batch and channel count, "
                         "16-byte aligned (the kernel reads raw NHWC rows)")
    n, c = x0.shape[0], x0.shape[1]
    oh, ow = int(size[0]), int(size[1])
    out = torch.empty((n, c, oh, ow), dtype=x0.dtype, device=x0.device, memory_format=torch.channels_last)
    ptrs = (C.c_void_p * k)(*[x.data_ptr() for x in xs])
    hs, ws = (C.c_int * k)(*[x.shape[2] for x in xs]), (C.c_int * k)(*[x.shape[3] for x in xs])
    wt = (C.c_float * k)(*[float(v) for v in weights])
    _lib.check(_lib.load().dad3d_nhwc_resize_sum(out.data_ptr(), n, oh, ow, c, _DTYPE[x0.dtype][0], k, ptrs, hs, ws, wt,
                                                 x0.device.index or 0, torch.cuda.current_stream(x0.device).cuda_stream))
    return out
