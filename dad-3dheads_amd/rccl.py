"""RCCL called directly (ctypes on the librccl.so PyTorch-ROCm ships): the job's ONE collective on the launch stream.

`torch.distributed.all_gather_into_tensor` on the "nccl" backend runs the collective on the process group's own stream:
an event record + stream wait in front, another behind, per call. For a job whose whole device time is a few hundred
microseconds (20 decode steps of 13 us) that plumbing was most of the timed region: ~180 us for a 228 KB gather in a
world of one. Here the communicator is created up front (`ncclCommInitRank`, its
unique id distributed through the existing torch.distributed group, whatever its backend) and `ncclAllGather` is queued on
the stream the decode launches use -- no second stream, no events, capturable into a hipGraph with the launches.

SURVEY section 8(e): the path shards by images and this all-gather is its only exchange; the reference has no counterpart
(its only parallel entry is Lightning DDP, `model_training/train/flame_lightning_model.py:182-186`).
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import threading
from typing import Optional

# dmabuf IPC only on this pool's host driver. Set when this module is imported -- before the first RCCL call of the process; a
# process that needs it for torch's own nccl backend has to export it before HIP initialises (bench.py's self-launch does).
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

_NCCL_CHAR = 0  # ncclInt8 / ncclChar


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_ubyte * 128)]  # not c_char: ctypes hands c_char arrays back as NUL-terminated bytes


_lib = None


def _load():
    global _lib
    if _lib is None:
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if not os.path.isfile(path):
            raise RuntimeError(f"librccl.so not found next to torch ({path}): the direct RCCL path needs PyTorch-ROCm")
        lib = C.CDLL(path)
        lib.ncclGetErrorString.restype = C.c_char_p
        lib.ncclGetErrorString.argtypes = [C.c_int]
        lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        lib.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        lib.ncclCommDestroy.argtypes = [C.c_void_p]
        lib.ncclSend.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.ncclRecv.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _lib = lib
    return _lib


_hip = None


def _device_copy(dst: int, src: int, nbytes: int, stream: int) -> None:
    """hipMemcpyAsync device-to-device on a raw stream (the runtime PyTorch-ROCm already loaded)."""
    global _hip
    if _hip is None:
        # the copy bundled with the PyTorch-ROCm wheel; a build linked against the system runtime has it on the loader path instead
        # (either way this is the library the process has already mapped: dlopen returns that handle)
        bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        hip = C.CDLL(bundled if os.path.isfile(bundled) else "libamdhip64.so")
        hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        _hip = hip
    status = _hip.hipMemcpyAsync(C.c_void_p(dst), C.c_void_p(src), nbytes, 3, C.c_void_p(stream))  # 3 = hipMemcpyDeviceToDevice
    if status != 0:
        raise RuntimeError(f"hipMemcpyAsync failed ({status})")


def _check(lib, status: int, what: str) -> None:
    if status != 0:
        raise RuntimeError(f"{what} failed: {lib.ncclGetErrorString(status).decode()} ({status})")


class RcclAllGather:
    """One RCCL communicator over the ranks of `group` (default: the world), one device per rank (the current one)."""

    def __init__(self, group=None, init_timeout_s: float = 120.0):
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("RcclAllGather needs an initialised torch.distributed group to hand out the unique id")
        if not torch.cuda.is_available():
            raise RuntimeError("RcclAllGather needs a GPU")
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.comm = C.c_void_p()
        # Every rank must enter ncclCommInitRank or none: a rank that cannot even load the library would leave the others
        # waiting in it for ever. Agree on that first, through the group that exists (object collectives: any backend).
        try:
            self.lib, err = _load(), None
        except (OSError, RuntimeError, AttributeError) as e:
            self.lib, err = None, f"rank {self.rank}: {e}"
        errs = [None] * self.world
        dist.all_gather_object(errs, err, group=group)
        if any(errs):
            raise RuntimeError("RcclAllGather: librccl.so unusable on " + "; ".join(e for e in errs if e))
        # The unique id travels as a (status, bytes) pair: a rank 0 whose ncclGetUniqueId failed still takes part in the broadcast,
        # so its failure reaches every rank instead of leaving them in the broadcast for ever.
        uid = _UniqueId()
        box = [None]
        if self.rank == 0:
            status = self.lib.ncclGetUniqueId(C.byref(uid))
            box = [(status, C.string_at(C.byref(uid), 128) if status == 0 else b"")]
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast_object_list(box, src=src, group=group)
        status, raw = box[0] if isinstance(box[0], tuple) and len(box[0]) == 2 else (-1, b"")
        if status != 0 or not isinstance(raw, bytes) or len(raw) != 128:
            raise RuntimeError(f"RcclAllGather: rank 0 could not produce a unique id (ncclGetUniqueId status {status})")
        C.memmove(C.byref(uid), raw, 128)
        self._rendezvous(uid, init_timeout_s)

    @classmethod
    def solo(cls, init_timeout_s: float = 120.0) -> "RcclAllGather":
        """A communicator of ONE rank on the current device, no torch.distributed needed: the N = 1 point of a scaling run then
        executes the same collective call as N = 8 (bench.py without a launcher)."""
        if not torch.cuda.is_available():
            raise RuntimeError("RcclAllGather needs a GPU")
        self = object.__new__(cls)
        self.world, self.rank, self.comm, self.lib = 1, 0, C.c_void_p(), _load()
        uid = _UniqueId()
        _check(self.lib, self.lib.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        self._rendezvous(uid, init_timeout_s)
        return self

    def _rendezvous(self, uid, init_timeout_s: float) -> None:
        # The rendezvous itself runs on a helper thread with a deadline (ctypes releases the GIL): a bootstrap that never
        # completes must surface as an error the caller can fall back from, not as a hung job. The thread needs the rank's
        # device selected -- the communicator binds to the calling thread's current device.
        device, done = torch.cuda.current_device(), {}

        def rendezvous():
            torch.cuda.set_device(device)
            comm = C.c_void_p()
            done["status"] = self.lib.ncclCommInitRank(C.byref(comm), self.world, uid, self.rank)
            done["comm"] = comm

        worker = threading.Thread(target=rendezvous, name="rccl-init", daemon=True)
        worker.start()
        worker.join(init_timeout_s)
        if "status" not in done:
            raise TimeoutError(f"ncclCommInitRank did not return within {init_timeout_s:.0f} s (rank {self.rank} of {self.world})")
        _check(self.lib, done["status"], "ncclCommInitRank")
        self.comm = done["comm"]

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor, stream: Optional[int] = None) -> torch.Tensor:
        """out[r * n : (r + 1) * n] = rank r's `inp` (n = inp.numel(), any dtype: moved as bytes), queued on `stream`
        (a raw hipStream_t; default: torch's current stream). Stream-ordered, no host synchronisation."""
        if not (out.is_cuda and inp.is_cuda and out.is_contiguous() and inp.is_contiguous()):
            raise ValueError("all_gather: contiguous device tensors only")
        nbytes = inp.numel() * inp.element_size()
        if out.dtype != inp.dtype or out.numel() * out.element_size() != self.world * nbytes:
            raise ValueError("all_gather: `out` must hold world x `inp` of the same dtype")
        if stream is None:
            stream = torch.cuda.current_stream(inp.device).cuda_stream
        _check(self.lib, self.lib.ncclAllGather(inp.data_ptr(), out.data_ptr(), nbytes, _NCCL_CHAR, self.comm, C.c_void_p(stream)),
               "ncclAllGather")
        return out

    def gather_to_root(self, out: Optional[torch.Tensor], inp: torch.Tensor, root: int = 0, stream: Optional[int] = None):
        """out[r * n : (r + 1) * n] = rank r's `inp` ON `root` ONLY (`out` is ignored elsewhere and may be None): one grouped
        ncclSend per non-root rank, world - 1 grouped ncclRecv on the root, the root's own rows by a device copy on the same
        stream. BASELINE configs[4] ends with the finished images in ONE place: an all-gather would push world x 12.6 MB into
        EVERY GPU (88 MB of xGMI ingress each at world 8) for copies nobody reads; this moves each rank's 12.6 MB once, over its
        own direct link to the root ("`ncclGather`-style send/recv to rank 0"). Stream-ordered, no host
        synchronisation. Returns `out` on the root, None elsewhere."""
        if not (0 <= root < self.world):
            raise ValueError(f"root {root} outside world of {self.world}")
        if not (inp.is_cuda and inp.is_contiguous()):
            raise ValueError("gather_to_root: contiguous device tensors only")
        nbytes = inp.numel() * inp.element_size()
        if stream is None:
            stream = torch.cuda.current_stream(inp.device).cuda_stream
        st = C.c_void_p(stream)
        if self.rank != root:
            _check(self.lib, self.lib.ncclSend(inp.data_ptr(), nbytes, _NCCL_CHAR, root, self.comm, st), "ncclSend")
            return None
        if out is None or not (out.is_cuda and out.is_contiguous()) or out.dtype != inp.dtype or \
                out.numel() * out.element_size() != self.world * nbytes:
            raise ValueError("gather_to_root: the root's `out` must be a contiguous device tensor holding world x `inp`")
        base = out.data_ptr()
        if self.world > 1:
            _check(self.lib, self.lib.ncclGroupStart(), "ncclGroupStart")
            try:
                for r in range(self.world):
                    if r != root:
                        _check(self.lib, self.lib.ncclRecv(base + r * nbytes, nbytes, _NCCL_CHAR, r, self.comm, st), "ncclRecv")
            finally:  # a group is never left open: a failing ncclRecv must not swallow every later call of this thread into it
                end_status = self.lib.ncclGroupEnd()
            _check(self.lib, end_status, "ncclGroupEnd")
        if base + root * nbytes != inp.data_ptr():
            _device_copy(base + root * nbytes, inp.data_ptr(), nbytes, stream)
        return out

    def destroy(self) -> None:
        if getattr(self, "comm", None) is not None and self.comm.value:
            self.lib.ncclCommDestroy(self.comm)
            self.comm = C.c_void_p()

    def __del__(self):
        if sys.is_finalizing():  # HIP / RCCL may already be gone: a native crash there is not an exception
            return
        try:
            self.destroy()
        except Exception:
            pass
