"""Multi-GPU: images shard over ranks as independent batches; the only collective is the final gather.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on the MI355X node; "gloo" on CPU
for the tests). Nothing in the decode itself communicates: rank r owns the contiguous rows
`shard_range(n, r, world)` of the global batch, the FLAME constants are replicated at construction, and
`gather_rows` concatenates the per-rank results in rank order with ONE all-gather (ragged shards are padded
to the largest shard for the collective and trimmed afterwards).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Rows [lo, hi) of an n-row batch owned by `rank`: contiguous, sizes differ by at most one, the first
    `n % world` ranks get the extra row (BASELINE configs 4/5: 2048/8 = 256, 512/8 = 64)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(n: int, world: int) -> List[int]:
    return [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]


def gather_rows(local: torch.Tensor, n_total: int, group: Optional[dist.ProcessGroup] = None, direct=None,
                root: Optional[int] = None) -> Optional[torch.Tensor]:
    """Gather row-sharded `local` ([n_r, ...], n_r = this rank's shard of n_total rows) into the full `[n_total, ...]`
    tensor: on every rank (`root=None`, one all-gather) or on group rank `root` only (one gather; the other ranks get
    None and receive nothing -- SURVEY 8e's "`ncclGather`-style send/recv to rank 0"). One collective; works for any dtype
    the backend supports.
    `direct`: an `rccl.RcclAllGather` over the same ranks -- the collective is then queued on torch's current stream
    through RCCL's C API instead of the process group's own stream (no event hops; see rccl.py)."""
    if not dist.is_available() or not dist.is_initialized():
        if local.shape[0] != n_total:
            raise ValueError("single process: local must already hold every row")
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = shard_sizes(n_total, world)
    if local.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank} holds {local.shape[0]} rows, its shard of {n_total} is {sizes[rank]}")
    if root is not None and not (0 <= root < world):
        raise ValueError(f"root {root} outside world of {world}")
    receives = root is None or rank == root
    width = max(sizes)
    if width == 0:
        return local.new_empty((0,) + tuple(local.shape[1:])) if receives else None
    padded = local
    if local.shape[0] != width:  # ragged: pad to the widest shard for the collective
        padded = local.new_zeros((width,) + tuple(local.shape[1:]))
        padded[: local.shape[0]] = local
    out = local.new_empty((world * width,) + tuple(local.shape[1:])) if receives else None
    if root is None:
        if direct is not None:
            direct.all_gather(out, padded.contiguous())
        else:
            dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    elif direct is not None:
        direct.gather_to_root(out, padded.contiguous(), root=root)
    else:
        dst = dist.get_global_rank(group, root) if group is not None else root
        dist.gather(padded.contiguous(), list(out.view((world, width) + tuple(local.shape[1:])).unbind(0)) if receives else None,
                    dst=dst, group=group)
    if not receives:
        return None
    if all(s == width for s in sizes):
        return out
    return torch.cat([out[r * width : r * width + sizes[r]] for r in range(world)], dim=0)


class ShardedLandmarkDecoder:
    """Decode the rows this rank owns and gather the landmarks of the whole batch (BASELINE config 4)."""

    def __init__(self, head_mesh, group: Optional[dist.ProcessGroup] = None, direct_rccl: bool = False):
        """direct_rccl: gather through `rccl.RcclAllGather` (communicator created here, collective on the launch stream)
        instead of `torch.distributed.all_gather_into_tensor`; needs the "nccl" backend's hardware, i.e. one GPU per rank."""
        self.head_mesh = head_mesh
        self.group = group
        self.direct = None
        if direct_rccl and dist.is_available() and dist.is_initialized():
            from .rccl import RcclAllGather

            self.direct = RcclAllGather(group)

    def __call__(self, params_global: torch.Tensor) -> torch.Tensor:
        """params_global [B,P] (every rank passes the same host/device tensor or just needs its own rows valid)
        -> int32 landmarks [B, n_lmk, 2] on every rank."""
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        rank = dist.get_rank(self.group) if dist.is_initialized() else 0
        n = params_global.shape[0]
        lo, hi = shard_range(n, rank, world)
        dev = self.head_mesh.flame.torch_device
        mine = params_global[lo:hi].to(dev, torch.float32).contiguous()
        if hi > lo:
            # mutate=False: `mine` may be a view of the caller's tensor, and a landmark decode has no business zeroing
            # translation z in this rank's rows of it (head_mesh.py:41 is reprojected_vertices' side effect)
            px = self.head_mesh.decode(mine, verts3d=False, proj=False, landmarks=False, landmarks_px=True,
                                       mutate=False)["lmk_px"]
        else:
            px = torch.empty((0, self.head_mesh.flame.n_landmarks, 2), dtype=torch.int32, device=dev)
        return gather_rows(px, n, self.group, self.direct)


class ShardedRenderer:
    """BASELINE config 5: `head_mesh` decode + Sim3DR face-mesh render of the rows this rank owns, then ONE all-gather of
    the finished uint8 images (12.6 MB per rank at 64 x 256 x 256 x 3). Three launches per rank and
    batch -- fused decode (3-component projection, z flipped like `demo_utils.get_vertices_for_render`), the raster's
    geometry kernel (vertex normals + Phong light + triangle records), the tile kernel -- and no host copy."""

    def __init__(self, head_mesh, mesh, group: Optional[dist.ProcessGroup] = None, image_size: int = 256,
                 direct_rccl: bool = False, root: Optional[int] = None, **light):
        """root: group rank that receives the finished images (the others get None from `__call__`): every rank's 12.6 MB then
        crosses xGMI once, to one GPU. None = all-gather: every rank ends up with all `world x 12.6 MB`."""
        self.head_mesh, self.mesh, self.group, self.root = head_mesh, mesh, group, root
        self.direct = None
        if direct_rccl and dist.is_available() and dist.is_initialized():
            from .rccl import RcclAllGather

            self.direct = RcclAllGather(group)
        self.image_size = int(image_size)
        self.light = light
        self._dec, self._img, self._light_buf = {}, None, None

    def render_local(self, params_local: torch.Tensor) -> torch.Tensor:
        """[n_r, P] fp32 on this rank's device -> uint8 [n_r, h, w, 3] (black background), buffers reused across calls."""
        n, dev, s = params_local.shape[0], self.head_mesh.flame.torch_device, self.image_size
        if self._img is None or self._img.shape[0] != n:
            self._dec, self._img = {}, torch.empty((n, s, s, 3), dtype=torch.uint8, device=dev)
            self._light_buf = torch.empty((n, self.mesh.nver, 3), dtype=torch.float32, device=dev)
        if n == 0:
            return self._img
        verts = self.head_mesh.flame.decode(params_local, proj=True, to_2d=False, flip_z=True, out=self._dec)["proj"]
        return self.mesh.render(verts, self._img, light_out=self._light_buf, clear=True, **self.light)  # black background

    def __call__(self, params_global: torch.Tensor) -> Optional[torch.Tensor]:
        """params_global [B,P] (the same tensor on every rank, or at least this rank's rows valid) -> uint8 [B,h,w,3]
        on every rank (root=None) or on rank `root` alone (None elsewhere)."""
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        rank = dist.get_rank(self.group) if dist.is_initialized() else 0
        n = params_global.shape[0]
        lo, hi = shard_range(n, rank, world)
        dev = self.head_mesh.flame.torch_device
        mine = params_global[lo:hi].to(dev, torch.float32).contiguous()
        img = self.render_local(mine)
        if not (dist.is_available() and dist.is_initialized()):
            return img.clone()  # render_local's buffer is reused by the next call; the gathered tensor of the other path is fresh
        return gather_rows(img, n, self.group, self.direct, root=self.root)
