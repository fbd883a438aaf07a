"""Drop-in for the reference's `predictor.FaceMeshPredictor` (predictor.py:68-211) on the MI355X.

Same constructor (`config` dict with `model_path`, `img_size`, `stride`, `constants`; `cuda_id`), same
`__call__(image) -> dict` with the reference's keys, shapes, dtypes and side effects:

    "points"             int ndarray [68,2]     68 2-D landmarks re-adjusted to the input frame (predictor.py:147-152)
    "projected_vertices" Tensor [1,5023,2]      HeadMesh.reprojected_vertices                      (predictor.py:137)
    "3d_vertices"        Tensor [5023,3]        HeadMesh.vertices_3d(...)[0].squeeze()             (predictor.py:136)
    "3dmm_params"        Tensor [1,413]         scale/translation re-adjusted, tz zeroed           (predictor.py:154-176, head_mesh.py:41)

What is different underneath: the CNN runs on PyTorch-ROCm, its 413-vector never leaves HBM
(the reference does `.detach().cpu()`, predictor.py:104), the re-adjustment is a HIP kernel and the TWO
CPU decodes of predictor.py:136-137 are ONE fused HIP launch. `predict_batch` is the batched entry the
reference lacks. Third-party preprocessing (albumentations / cv2, absent here) is ONE HIP kernel for a batch of images
of any sizes (`dad3d_preprocess_images`, csrc/preprocess.hip): LongestMaxSize (cv2 INTER_LINEAR, the 8-bit fixed-point
path) -> PadIfNeeded(centre, 0) -> Normalize(imagenet) -> CHW (predictor.py:80-95,195-203).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .head_mesh import HeadMesh

# keys of the CNN's output dict (model_training/data/config.py:16-23: every constant's value is its own name)
OUTPUT_3DMM_PARAMS = "OUTPUT_3DMM_PARAMS"
OUTPUT_2D_LANDMARKS = "OUTPUT_2D_LANDMARKS"
OUTPUT_LANDMARKS_HEATMAP = "OUTPUT_LANDMARKS_HEATMAP"
_MEAN = (0.485, 0.456, 0.406)
_STD = (0.229, 0.224, 0.225)


def py3round(x: float) -> int:
    """albumentations.augmentations.geometric.py3round (imported at predictor.py:12)."""
    if abs(round(x) - x) == 0.5:
        return int(2.0 * round(x / 2.0))
    return int(round(x))


def calculate_paddings(orig_h: int, orig_w: int) -> List[int]:
    """model_training/model/utils.py:71-77 -> [top, bottom, left, right]."""
    m = max(orig_h, orig_w)
    top, left = int((m - orig_h) / 2), int((m - orig_w) / 2)
    return [top, m - orig_h - top, left, m - orig_w - left]


def find_3dmm_idx(key: str, consts: Dict[str, int]) -> int:
    idx = 0
    for k, v in consts.items():
        if k == key:
            break
        idx += v
    return idx


class FaceMeshPredictor:
    def __init__(self, config: Dict[str, Any], cuda_id: int = 0, model: Optional[Callable] = None,
                 flame_model: Any = None, flame_path: Optional[str] = None, landmarks: Optional[Sequence[int]] = None):
        self.cuda_id = cuda_id
        self.device = torch.device("cuda", cuda_id)
        self.flame_constants = config["constants"]
        if model is None:  # the reference's path: a TorchScript file under $HOME (predictor.py:72)
            path = os.path.join(os.path.expanduser("~"), config["model_path"])
            if not os.path.isfile(path):
                raise FileNotFoundError(
                    f"{path} not found. The reference downloads it on first use (predictor.py:29-65); there is no "
                    "network here -- place the file there or pass model=<module returning the output dict>.")
            model = torch.jit.load(path)
        self.model = model.to(self.device).eval() if hasattr(model, "to") else model
        self.head_mesh = HeadMesh(self.flame_constants, flame_model=flame_model, flame_path=flame_path,
                                  device=cuda_id, image_size=config["img_size"], landmarks=landmarks)
        self._img_size = config["img_size"]
        self._stride = config.get("stride", 2)
        self._lib = _lib.load()

    @classmethod
    def dad_3dnet(cls, **kwargs):
        from .config import load_default_config

        return cls(config=load_default_config(), **kwargs)

    @classmethod
    def random_init(cls, dtype: torch.dtype = torch.bfloat16, seed: int = 0, tune: bool = False, graph: bool = False, **kwargs):
        """DAD-3DNet architecture declared in `network.py` with seeded random weights (no checkpoint offline):
        the real compute and memory footprint of the front half for throughput and plumbing tests."""
        from .config import load_default_config
        from .network import DAD3DNet, GraphedNet, InferenceNet

        net = InferenceNet(DAD3DNet(seed=seed), dtype, tune=tune)
        device = torch.device("cuda", kwargs.get("cuda_id", 0))
        model = GraphedNet(net.to(device)) if graph else net  # graph=True: one hipGraph per input shape
        return cls(config=load_default_config(), model=model, **kwargs)

    # -- preprocess (predictor.py:86-95,195-203) ---------------------------------------------------------
    def _geometry(self, hw: Tuple[int, int]) -> Tuple[List[int], float, Tuple[int, int]]:
        h, w = hw
        scale = self._img_size / float(max(h, w))
        new_h, new_w = (py3round(d * scale) for d in (h, w))
        return calculate_paddings(new_h, new_w), scale, (new_h, new_w)

    def _preprocess_launch(self, sources: Sequence[Tuple[int, int, int, int]]) -> torch.Tensor:
        """sources: (device address, h, w, row stride in bytes) per image -> float32 [B,3,S,S] on the device, one launch."""
        rows = []
        for ptr, h, w, stride in sources:
            pads, _, (nh, nw) = self._geometry((h, w))
            rows.append([ptr, h, w, nh, nw, pads[0], pads[2], stride])
        descs = torch.tensor(rows, dtype=torch.int64).to(self.device, non_blocking=False)
        out = torch.empty((len(rows), 3, self._img_size, self._img_size), dtype=torch.float32, device=self.device)
        mean, std = (C.c_float * 3)(*_MEAN), (C.c_float * 3)(*_STD)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._lib.dad3d_preprocess_images(descs.data_ptr(), len(rows), self._img_size, mean, std, out.data_ptr(),
                                                     self.cuda_id, stream))
        return out

    def preprocess(self, x: np.ndarray, cache: Dict[str, Any]) -> torch.Tensor:
        cache["input_shape"] = x.shape[:2]
        if x.ndim != 3 or x.shape[2] != 3 or x.dtype != np.uint8:
            raise ValueError(f"expected a uint8 RGB image [H,W,3], got {x.dtype} {x.shape}")
        img = torch.from_numpy(np.ascontiguousarray(x)).to(self.device)
        out = self._preprocess_launch([(img.data_ptr(), x.shape[0], x.shape[1], x.shape[1] * 3)])
        cache["_staged"] = img  # keeps the upload alive until the launch has consumed it (stream order)
        return out

    def process(self, x: torch.Tensor) -> Dict[str, torch.Tensor]:
        with torch.no_grad():
            return self.model(x)

    # -- postprocess (predictor.py:102-176,188-193) -------------------------------------------------------
    def _landmarks_68(self, out: Dict[str, torch.Tensor]) -> Optional[np.ndarray]:
        if OUTPUT_2D_LANDMARKS in out:
            return out[OUTPUT_2D_LANDMARKS].detach().cpu().numpy() * 256.0
        if OUTPUT_LANDMARKS_HEATMAP in out:  # unravel_index(sigmoid(heatmap)).flip(-1) * stride (predictor.py:108-112)
            hm = torch.sigmoid(out[OUTPUT_LANDMARKS_HEATMAP]).detach()
            b, c, h, w = hm.shape
            flat = hm.view(b, c, -1).argmax(-1)
            yx = torch.stack((torch.div(flat, h, rounding_mode="trunc"), flat % h), dim=-1)
            return float(self._stride) * yx.flip(-1).cpu().numpy()
        return None

    def _readjust_and_decode(self, params: torch.Tensor, pads_scale: torch.Tensor, device_outputs: bool):
        b = params.shape[0]
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._lib.dad3d_flame_readjust_params(self.head_mesh.flame._handle, params.data_ptr(), b,
                                                         pads_scale.data_ptr(), 0.0, 0.0, 1.0, stream))
        out = self.head_mesh.decode(params, verts3d=True, proj=True, to_2d=True, landmarks=False, mutate=True)
        v3d, proj = out["verts3d"], out["proj"]
        if not device_outputs:
            v3d, proj, params = v3d.cpu(), proj.cpu(), params.cpu()
        return v3d, proj, params

    def __call__(self, x: Any) -> Dict[str, Any]:
        res = self.predict_batch([x])[0]
        return res

    def predict_tensor(self, images: torch.Tensor, landmarks_px: bool = True) -> Dict[str, torch.Tensor]:
        """Device-resident batch entry: uint8 RGB `[B, H, W, 3]` on this GPU, all images of one size -> device tensors
        `3dmm_params [B,413]`, `3d_vertices [B,5023,3]`, `projected_vertices [B,5023,2]`, `points [B,68,2]` int32
        and, when the head mesh was built with a landmark list, `landmarks [B,n,2]` int32. Nothing is copied to the
        host and nothing synchronises: normalisation, CNN, re-adjustment kernel and the fused decode are queued on
        the current stream (the reference does `.cpu()` after the CNN, predictor.py:104, then decodes twice on the CPU)."""
        assert images.ndim == 4 and images.shape[-1] == 3 and images.dtype == torch.uint8 and images.is_cuda
        b, h, w = images.shape[:3]
        pads, scale, (nh, nw) = self._geometry((h, w))
        images = images.contiguous()
        x = self._preprocess_launch([(images.data_ptr() + i * h * w * 3, h, w, w * 3) for i in range(b)])
        out = self.process(x)
        params = out[OUTPUT_3DMM_PARAMS].detach().to(torch.float32).contiguous().clone()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        # one (pad_left, pad_top, scale) triple for the whole batch: pads_scale == NULL
        _lib.check(self._lib.dad3d_flame_readjust_params(self.head_mesh.flame._handle, params.data_ptr(), b, None,
                                                         float(pads[2]), float(pads[0]), float(scale), stream))
        want_lmk = landmarks_px and self.head_mesh.flame.n_landmarks > 0
        dec = self.head_mesh.decode(params, verts3d=True, proj=True, to_2d=True, landmarks=False, landmarks_px=want_lmk,
                                    mutate=True)
        res = {"3dmm_params": params, "3d_vertices": dec["verts3d"], "projected_vertices": dec["proj"]}
        if want_lmk:
            res["landmarks"] = dec["lmk_px"]
        if OUTPUT_2D_LANDMARKS in out:  # predictor.py:147-152 on the device
            # numpy promotes to float64 at the integer pad subtraction; same here so `astype(int)` truncates alike
            pts = (out[OUTPUT_2D_LANDMARKS].detach().float() * 256.0).clamp_(0, self._img_size).double()
            pts = (pts - torch.tensor([pads[2], pads[0]], device=self.device, dtype=torch.float64)) / scale
            res["points"] = pts.to(torch.int32)
        return res

    def predict_batch(self, images: Sequence[np.ndarray], device_outputs: bool = False) -> List[Dict[str, Any]]:
        """Batched predictor: list of HxWx3 uint8 RGB arrays (any sizes) -> list of the reference's result dicts."""
        caches: List[Dict[str, Any]] = [{"input_shape": im.shape[:2]} for im in images]
        staged = []
        for im in images:
            if im.ndim != 3 or im.shape[2] != 3 or im.dtype != np.uint8:
                raise ValueError(f"expected uint8 RGB images [H,W,3], got {im.dtype} {im.shape}")
            staged.append(torch.from_numpy(np.ascontiguousarray(im)).to(self.device))
        batch = self._preprocess_launch([(t.data_ptr(), t.shape[0], t.shape[1], t.shape[1] * 3) for t in staged])
        out = self.process(batch)
        params = out[OUTPUT_3DMM_PARAMS].detach().to(self.device, torch.float32).contiguous().clone()
        geo = [self._geometry(c["input_shape"]) for c in caches]
        pads_scale = torch.tensor([[g[0][2], g[0][0], g[1]] for g in geo], dtype=torch.float32, device=self.device)
        lm = self._landmarks_68(out)
        if lm is None:  # `return {"3dmm_params": ...}` branch of predictor.py:144-145
            stream = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(self._lib.dad3d_flame_readjust_params(self.head_mesh.flame._handle, params.data_ptr(),
                                                             params.shape[0], pads_scale.data_ptr(), 0.0, 0.0, 1.0, stream))
            params = params if device_outputs else params.cpu()
            return [{"3dmm_params": params[i : i + 1]} for i in range(len(images))]
        v3d, proj, params = self._readjust_and_decode(params, pads_scale, device_outputs)
        results = []
        for i, (pads, scale, _) in enumerate(geo):
            pts = lm[i].clip(min=0, max=self._img_size) - np.array([[pads[2], pads[0]]])
            pts = (pts / scale).astype(int)  # predictor.py:147-152
            results.append({"points": np.reshape(pts, (-1, 2)), "projected_vertices": proj[i : i + 1],
                            "3d_vertices": v3d[i], "3dmm_params": params[i : i + 1]})
        return results
