#!/usr/bin/env python3
"""Headline benchmark: images/sec of FLAME decode + 445-landmark projection, batch 64 @ 256^2 per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W]          # N > 1 re-launches itself under torch.distributed.run
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of 64 synthetic parameter rows per GPU, already
resident in HBM: ONE fused launch (pose role + blend-shape GEMM / skinning / projection role) producing, per image,
`3d_vertices [5023,3]`, `projected_vertices [5023,2]` and the 445 integer landmarks -- everything
`FaceMeshPredictor` + `draw_3d_landmarks` derive from one params row (predictor.py:136-137,
demo_utils.py:42-46; the reference decodes twice, this path once). BASELINE.json configs[1].

Timing. An untimed, disclosed clock-ramp phase (launches until `--prewarm-ms` have passed, default 50; reported as
`config.prewarm_ms`) precedes the W counted warm-up steps: the driver runs `--steps 20 --warmup 5`, and 25 launches after a
cold start measure the power state, not the kernel. The K timed steps sit between barrier + synchronize on both sides
(`wall_ms_per_step`, MAX over ranks). With one stream per rank (the contract line) two hipEvents on the launch stream
bracket exactly the K launches; `value` and `ms_per_step` are taken from them (MAX over ranks) -- the wall clock around a
0.3 ms region adds the host's launch latency and the final synchronize (~2 us per step at K = 20) to a 13 us step. Both
are printed; `config.value_from` says which one `value` is. `roofline.achieved` = algorithmic flops per launch / (event
time / K), the number `rocprofv3 --kernel-trace --stats` of the same command reports as the kernel's average duration.

`--streams S` issues the steps round-robin on S HIP streams (S batches of 64 in flight per GPU; kernels of different
streams overlap, so `value` is the wall clock and the kernel duration for the roofline object comes from one more pass of K
launches on ONE stream).

Multi-GPU: images shard over ranks (weak scaling, 64 per GPU per step, no data-path collective); the timed
region ends with the job's single RCCL all-gather of the last step's landmarks (north_star: "RCCL/xGMI
only for the final gather"). EVERY line -- the plain N = 1 run included, through a communicator of one rank -- executes that
collective behind its K steps and prints two per-step times from device events on the launch stream, MAX over ranks:
`ms_per_step_compute` ([first launch ... last launch]) and `ms_per_step_with_gather` ([first launch ... end of the all-gather]).
`value` is images / the COMPUTE time for the plain N = 1 run (the metric as BASELINE.json defines it: decode + landmark
projection; there is nobody to gather from) and images / the time WITH the gather under a process group (any N, N = 1 under
torchrun included); `config.value_definition` says which, and BOTH rates are printed at every N as `value_compute` / `value_with_gather` with the
gather's built-in share of the region in `config.scaling_note` -- read a scaling curve on one of them. `config.toolchain` = the compiler / HIP headers
the library was built with and the runtime it runs on. Rank 0 prints ONE JSON line.

Secondary legs (plain N = 1 decode run only, AFTER the contract region, none of them touches `value` / `ms_per_step` / `roofline`;
`--no-secondary` skips them): `long_region` = 2000 more launches of the same step between two hipEvents (the driver's K = 20 region
is 0.26 ms -- too short to mean much on its own); `secondary.decode_b256` = BASELINE configs[2] (batch 256, head_mesh path:
3d_vertices + 3-component projection + landmarks, 125 764 B per image) with its own MFMA / HBM fractions, every row checked against
reference-HeadMesh goldens, with `landmarks_only` (configs[3]'s per-GPU share and all 2048 rows: bit-equal to the whole-mesh launch);
`secondary.decode_b256_split` = the same configs[2] step on the GATED bf16x3 exact-product split (its own fractions against the bf16 peak; never the
headline); `secondary.e2e_b64` = the drop-in predictor end to end against the reference CPU predictor's call sequence in the same process (the north
star's 200x sentence; reported separately from the metric); `secondary.render_b64` = BASELINE configs[4]'s per-GPU share (decode -> normals + Phong -> raster, three
launches per batch of 64) with the timed images checked against the reference rasteriser, and `cpu_baseline_render` beside it.

`--workload render` (BASELINE configs[4], not the headline metric): per step and GPU 64 images of head_mesh decode ->
vertex normals + Phong light -> z-buffer raster of the 9976-triangle mesh onto 256 x 256 x 3 (three launches), the timed
region ends with one all-gather of the uint8 images of the last step.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import socket
import sys
import time

# dmabuf IPC only on this pool's host driver: must be in the environment before HIP initialises (i.e. before torch is imported)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 64
N_VERTS, N_LMK, N_PARAMS = 5023, 445, 413
# algorithmic work of ONE fused decode
FLOP_PER_IMAGE = 14.5e6
CONST_BYTES = 26_541_532  # basis + template + weights + regressor, read once per launch
BYTES_PER_IMAGE = 105_672  # params 1652 + verts3d 60276 + proj2d 40184 + landmarks 3560
BYTES_PER_IMAGE_3D = 125_764  # configs[2] (to_2d=False): params 1652 + verts3d 60276 + proj3d 60276 + landmarks 3560
B256_SEED = 104  # tests/golden/decode_b256_golden.npz
RASTER_BYTES_PER_IMAGE = 513_768  # rasterize; + 120 552 for the normals
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
REWARM_STEPS = int(os.environ.get("DAD3D_BENCH_REWARM", "8"))  # untimed steps between the opening barrier and the opening synchronize
GOLDEN_SEED = 102  # tests/golden/decode_golden.npz "b64": rank 0 / stream 0 decodes exactly these rows


def cpu_baseline(model, lmk_idx, budget_s: float = 15.0):
    """Reference CPU path, timed on this host: per image (B=1) readjust + vertices_3d + reprojected_vertices
    + astype(int) landmark gather, exactly the call sequence of predictor.py:125-145 / demo_utils.py:37-47,
    through the torch-CPU oracle (the reference's own arithmetic library). The reference never sets a
    thread count; on a many-core host torch's default (= all cores) is pathologically slow for these small
    ops, so a few thread counts are tried inside the time budget and the FASTEST is reported (`cores` = the
    threads it used) -- the most generous reading of the baseline."""
    from dad_3dheads_amd import synthetic
    from oracle import flame_ref

    consts = flame_ref.FlameConstants.from_model(model)
    ncpu = os.cpu_count() or 1
    params = torch.from_numpy(synthetic.synthetic_params(256, seed=4242))
    tried = {}
    candidates = sorted({t for t in (1, 8, 32, ncpu) if t <= ncpu}) or [1]  # os.cpu_count() threads as well
    with torch.no_grad():
        for threads in candidates:
            torch.set_num_threads(threads)
            for i in range(5):  # warm-up
                flame_ref.predictor_postprocess(consts, params[i : i + 1].clone(), lmk_idx)
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < budget_s / len(candidates):
                flame_ref.predictor_postprocess(consts, params[n % 256 : n % 256 + 1].clone(), lmk_idx)
                n += 1
            tried[threads] = (n, time.perf_counter() - t0)
    best = max(tried, key=lambda t: tried[t][0] / tried[t][1])
    n, dt = tried[best]
    return {
        "value": n / dt,
        "unit": "images/sec",
        "cores": best,
        "kind": "port",
        "sample": f"{n} images in {dt:.1f} s, one per call (B=1) like predictor.py: readjust + 2 FLAME decodes + "
                  f"445-landmark int gather each; torch {torch.__version__} CPU fp32 (oracle/flame_ref.py); host has "
                  f"{ncpu} logical cores; img/s by torch threads: "
                  + ", ".join(f"{t}: {tried[t][0] / tried[t][1]:.0f}" for t in tried),
    }


def cpu_baseline_render(verts0, faces, timed=None, budget_s: float = 10.0):
    """RenderPipeline of the reference on one host core (the reference's own Sim3DR C++ when oracle/_ref holds it,
    else the C port), one image per call like demo_utils.py:152-170. The same leg checks the buffers the TIMED launches
    wrote: `timed` = (vertices, per-vertex light, image) of the first image of every stream; the reference rasteriser given those
    vertices and that light must produce that image byte for byte (the light itself differs from numpy's by the host's powf,
    tests/render_checks.py)."""
    from oracle import sim3dr_ref

    kind = "reference" if sim3dr_ref.available("reference") else "port"
    orc = sim3dr_ref.Sim3DROracle(kind)
    fn = lambda: sim3dr_ref.render_pipeline_ref(orc, verts0.copy(), faces, np.zeros((256, 256, 3), np.uint8))  # noqa: E731
    fn()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        fn()
        n += 1
    dt = time.perf_counter() - t0
    match = None
    if timed:
        match = all(np.array_equal(orc.rasterize(v.copy(), faces, light.copy(), bg=np.zeros((256, 256, 3), np.uint8)), image)
                    for v, light, image in timed)
    return {"value": n / dt, "unit": "images/sec", "cores": 1, "kind": kind, "timed_images_match": match,
            "sample": f"{n} images in {dt:.1f} s: RenderPipeline (_get_normal + numpy Phong light + _rasterize) of one decoded mesh "
                      f"per call, Sim3DR C++ ({kind}) on one host core"}


def pmc_json(name, key):
    """A per-launch figure of the fused kernel from the committed rocprofv3 PMC passes (profiles/: FETCH_SIZE and WRITE_SIZE
    collected in separate runs, KB units, FETCH doubled per MI355X_MICROARCH.md; SQ_VALU_MFMA_BUSY_CYCLES). PMC cannot be
    collected inside this process: these are CONSTANTS READ FROM COMMITTED FILES (the file is named next to them), null when
    absent."""
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        try:
            with open(os.path.join(ROOT, "profiles", f"{rnd}_{name}.json")) as f:
                return json.load(f)[key], f"profiles/{rnd}_{name}.json"
        except Exception:
            continue
    return None, None


def self_launch(args) -> None:
    """`python bench.py --gpus N` with N > 1 and no launcher around it: become
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>`."""
    visible = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if visible < args.gpus:
        raise SystemExit(f"bench.py: {args.gpus} GPUs requested, {visible} visible on this node -- nothing was run "
                         f"(one rank per GPU; launch on a node with at least {args.gpus} MI355X)")
    with socket.socket() as s:  # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execvpe(sys.executable, cmd, env)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--streams", type=int, default=1, help="HIP streams the steps are issued on, round-robin")
    ap.add_argument("--prewarm-ms", type=float, default=50.0, help="untimed clock-ramp phase before the counted warm-up")
    ap.add_argument("--workload", choices=("decode", "render"), default="decode")
    ap.add_argument("--gather", choices=("root", "all"), default="root",
                    help="render workload: the finished images go to rank 0 only (grouped ncclSend/ncclRecv) or to every rank (ncclAllGather)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip long_region / secondary.* (plain N = 1 decode run only)")
    args = ap.parse_args()

    under_launcher = "RANK" in os.environ and "MASTER_ADDR" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not under_launcher:
        self_launch(args)  # does not return
    # stdout carries ONE JSON line and nothing else: file descriptor 1 is pointed at stderr for the rest of the process (RCCL
    # prints a version banner through C stdio when a communicator is created; it would land behind the JSON line at exit) and
    # the line goes to a private duplicate of the real stdout.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1")) if under_launcher else 1
    rank = int(os.environ.get("RANK", "0")) if under_launcher else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if under_launcher else 0
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # TEST HOOK (tests/test_gpu_sharding.py, a box with ONE GPU): DAD3D_BENCH_SHARE_GPU=1 lets every rank use device
    # (local_rank mod visible) and joins them over gloo with host-staged gathers -- the N > 1 code path (shard seeds, the
    # per-rank timed regions, MAX over ranks, the gather's placement and its check) then runs on real kernels. RCCL refuses
    # two ranks on one device, so this is NOT the multi-GPU measurement: the line says so in `config.parallelism`.
    share_gpu = os.environ.get("DAD3D_BENCH_SHARE_GPU") == "1" and under_launcher
    dev_index = local_rank % max(torch.cuda.device_count(), 1) if share_gpu else local_rank
    if dev_index >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank}, {torch.cuda.device_count()} visible")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if under_launcher:  # under torch.distributed.run, even N = 1: the RCCL path is the one that is timed
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share_gpu:
            dist.init_process_group("gloo")
            _stage_gathers_through_the_host(dist)
        else:
            dist.init_process_group("nccl", device_id=dev)

    from dad_3dheads_amd import _lib, landmarks, synthetic
    from dad_3dheads_amd.head_mesh import HeadMesh

    static = synthetic.load_static()
    model = synthetic.synthetic_flame_model(0, static)
    lmk_idx = landmarks.canonical("445", static)
    hm = HeadMesh(flame_model=model, landmarks=lmk_idx, static=static, device=dev_index)
    lib = _lib.load()
    run = run_render if args.workload == "render" else run_decode
    out = run(args, dist, dev, rank, world, hm, lib, static, model, lmk_idx)
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _stage_gathers_through_the_host(dist):
    """Test hook only: gloo has no device all_gather_into_tensor / all_reduce / gather / barrier for HIP tensors everywhere, so the
    collectives bench.py uses are wrapped to go through host copies. Never active on the measured (nccl) path."""
    real_gather, real_reduce = dist.all_gather_into_tensor, dist.all_reduce

    def gather(out, inp, group=None):
        torch.cuda.synchronize()
        o, i = out.cpu(), inp.cpu().contiguous()
        real_gather(o, i, group=group)
        out.copy_(o)

    def reduce(t, op=dist.ReduceOp.SUM, group=None):
        c = t.cpu()
        real_reduce(c, op=op, group=group)
        t.copy_(c)

    real_gather_root = dist.gather

    def gather_root(t, gather_list=None, dst=0, group=None):
        torch.cuda.synchronize()
        c = t.cpu().contiguous()
        staged = [torch.empty_like(c) for _ in gather_list] if gather_list is not None else None
        real_gather_root(c, staged, dst=dst, group=group)
        for o, h in zip(gather_list or [], staged or []):
            o.copy_(h)

    dist.all_gather_into_tensor, dist.all_reduce, dist.gather = gather, reduce, gather_root
    dist._dad3d_shared_gpu = True


def make_direct_gather(dist):
    """The RCCL communicator for the job's one collective, created before anything is timed: over the process group's ranks, or
    -- without a process group -- of this one rank, so that the N = 1 point runs the same collective call as N = 8. None under the
    shared-GPU test hook (gloo) or when RCCL cannot be used (the line then says so)."""
    from dad_3dheads_amd.rccl import RcclAllGather

    if dist is None:
        if os.environ.get("DAD3D_BENCH_NO_COLLECTIVE") == "1":  # diagnostics: a plain run without the world-1 communicator
            return None
        try:
            return RcclAllGather.solo()
        except Exception as e:
            print(f"bench.py: world-1 RCCL communicator unavailable ({type(e).__name__}: {e}); no collective in this run", file=sys.stderr)
            return None
    if getattr(dist, "_dad3d_shared_gpu", False) or dist.get_backend() != "nccl":
        return None

    try:
        ok, direct = 1, RcclAllGather()
    except Exception as e:  # never lose a multi-GPU run to the shortcut: the process group's own all-gather is always there
        print(f"bench.py: direct RCCL gather unavailable ({type(e).__name__}: {e}); using torch.distributed", file=sys.stderr)
        ok, direct = 0, None
    # every rank must take the same path (the collectives have to match): all-reduce the outcome
    flag = torch.tensor([ok], dtype=torch.int32, device=torch.device("cuda", torch.cuda.current_device()))
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        if direct is not None:
            direct.destroy()
        return None
    return direct


def time_gather(gather, dev, n: int = 20) -> float:
    """Device time of ONE final gather in microseconds: events around n back-to-back gathers on the stream they run on
    (after the timed region; its own figure so that a short scaling run can be read)."""
    s = gather()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(n):
        s = gather()
    e1.record(s)
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / n * 1e3


def observed_shader_clock_mhz(lib, handle, call, dev):
    """Shader clock the decode kernel actually ran at: ONE extra launch after the timed region with the kernel's own phase
    stamps switched on (dad3d_flame_debug_trace: s_memtime shader cycles and the 100 MHz wall clock at each wave's start and
    end); clock = cycles / wall time, median over the decode workgroups' first waves. The roofline peak assumes 2400 MHz; rocm-smi
    reports the DPM level (~2.39 GHz), not what a power-limited MFMA kernel sustains. None when the trace comes back empty."""
    from dad_3dheads_amd import _lib

    try:
        n_rows = int(lib.dad3d_flame_debug_trace_entries(handle, BATCH))  # what either decode kernel stamps for this batch
        trace = torch.zeros(n_rows, dtype=torch.int64, device=dev)
        for _ in range(50):  # clocks as in the timed region
            lib.dad3d_flame_decode(*call)
        _lib.check(lib.dad3d_flame_debug_trace(handle, trace.data_ptr(), n_rows))
        _lib.check(lib.dad3d_flame_decode(*call))
        torch.cuda.synchronize(dev)
        _lib.check(lib.dad3d_flame_debug_trace(handle, None, 0))
        t = trace.cpu().numpy().astype(np.float64)[: 240 * 8 * 32].reshape(240, 8, 32)[:, 0]  # first mma wave of every workgroup
        cyc, wall = t[:, 5] - t[:, 0], (t[:, 13] - t[:, 12]) / 100.0  # cycles, microseconds
        ok = (cyc > 0) & (wall > 0)
        return float(np.median(cyc[ok] / wall[ok])) if ok.any() else None
    except Exception:
        return None


def fence(dist, dev):
    """barrier + synchronize. The device is drained FIRST when a process group is up: the direct RCCL gather (its own
    communicator, on the launch stream) has then finished on this rank before the group's barrier kernel is queued -- two
    communicators are never in flight together on one GPU."""
    if dist is not None:
        torch.cuda.synchronize(dev)
        dist.barrier()
    torch.cuda.synchronize(dev)


def prewarm(step, ms: float, dev) -> float:
    """Untimed: issue steps until `ms` of wall clock have passed (clock ramp, instruction caches, RCCL-free)."""
    t0, k = time.perf_counter(), 0
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(16):
            step(k)
            k += 1
        torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) * 1e3


def max_over_ranks(dist, dev, x: float) -> float:
    if dist is None:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def min_over_ranks(dist, dev, x: float) -> float:
    if dist is None:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t.item())


class AlignedStart:
    """Device-side aligned start of a multi-rank timed region. The ranks leave the opening barrier + synchronize at host-jittered
    times (tens of microseconds); every rank's region ends behind a collective that waits for the LAST rank, so without alignment
    that host skew lands inside a region that is only K x 13 us long and reads as lost scaling. Here each rank queues one 4-byte
    collective of the direct RCCL communicator on its launch stream immediately BEFORE the region's opening event, with no host
    synchronisation between it and the K launches: the GPUs leave that collective together (to a few microseconds), whatever the
    hosts did. `start_skew_us` = how long this rank's GPU sat in it (~ how much earlier than the last rank it arrived), reported
    MAX and MIN over ranks -- the skew is visible instead of being folded into `value`.
    Under the shared-GPU test hook (gloo) the same call sequence runs through host-staged copies (a host barrier, labelled)."""

    def __init__(self, dist, direct, dev, world):
        self.dist, self.direct, self.dev = dist, direct, dev
        self.on = dist is not None and (direct is not None or getattr(dist, "_dad3d_shared_gpu", False))
        self.e_in = torch.cuda.Event(enable_timing=True)
        if self.on:
            self.token = torch.zeros(1, dtype=torch.int32, device=dev)
            self.tokens = torch.zeros(world, dtype=torch.int32, device=dev)

    def how(self):
        if not self.on:
            return "none" if self.dist is None else "none (no direct RCCL communicator)"
        return ("4-byte ncclAllGather on the launch stream in front of the opening event" if self.direct is not None
                else "TEST HOOK: host-staged gloo all-gather in front of the opening event")

    def queue(self, stream):
        """Queue the alignment collective on `stream` (warm it up once before the region with the same call)."""
        if not self.on:
            return
        self.e_in.record(stream)
        if self.direct is not None:
            self.direct.all_gather(self.tokens, self.token, stream=stream.cuda_stream)
        else:
            with torch.cuda.stream(stream):
                self.dist.all_gather_into_tensor(self.tokens, self.token)

    def skew_us(self, e_region_open):
        """(MAX, MIN) over ranks of the time between arriving at the alignment collective and the region's opening event."""
        if not self.on:
            return None
        mine = self.e_in.elapsed_time(e_region_open) * 1e3
        return {"max": max_over_ranks(self.dist, self.dev, mine), "min": min_over_ranks(self.dist, self.dev, mine)}


def verify_against_golden(sets0, lmk_idx, dev):
    """The buffers the TIMED launches wrote (rank 0, stream 0: params = the golden's own rows) against
    tests/golden/decode_golden.npz -- outputs of the reference's own HeadMesh on the same seeded model: a fixed subset of
    128 vertices (3-D within 5e-6, projection within 1e-3 px) and the 445 integer landmarks (equal; a pixel may differ by
    one only where the reference's float coordinate is within 1e-3 of an integer)."""
    path = os.path.join(ROOT, "tests", "golden", "decode_golden.npz")
    try:
        g = np.load(path)
    except Exception as e:  # a checkout without tests/: say so instead of claiming a check
        return {"checked": False, "why": f"{type(e).__name__}: {e}"}
    sub = g["b64_subset"]
    dv = float(np.abs(sets0["verts3d"].cpu().numpy()[:, sub] - g["b64_v3d_sub"]).max())
    dp = float(np.abs(sets0["proj"].cpu().numpy()[:, sub] - g["b64_proj_sub"]).max())
    diff = sets0["lmk_px"].cpu().numpy() != g["b64_lmk_px"]
    near_int = np.abs(g["b64_lmk_xy"] - np.round(g["b64_lmk_xy"])) < 1e-3
    idx_dev = torch.from_numpy(lmk_idx).to(dev)
    gather_exact = bool(torch.equal(sets0["lmk_px"], sets0["proj"][:, idx_dev, :].to(torch.int32)))
    ok = dv < 5e-6 and dp < 1e-3 and bool(near_int[diff].all()) and gather_exact
    return {"checked": True, "ok": bool(ok), "max_abs_3d": dv, "max_abs_px": dp, "landmark_px_differ": int(diff.sum()),
            "landmark_gather_exact": gather_exact, "golden": "tests/golden/decode_golden.npz b64_* (reference HeadMesh, seed 102)"}


def decode_kernel_label():
    """Which kernel the decode launches of this process take (the environment switch of csrc/capi.cpp's dispatch)."""
    env = os.environ.get("DAD3D_DECODE_KERNEL", "")
    if env.startswith("v1"):
        return "flame_decode_kernel (two-role kernel of rounds 1-3, forced by DAD3D_DECODE_KERNEL=v1)"
    if env in ("split", "split_f16"):
        return f"split_params_kernel + flame_decode_split_kernel ({'bf16x3' if env == 'split' else 'fp16x2'} split, forced by DAD3D_DECODE_KERNEL={env})"
    return "flame_decode_pipe_kernel<true> (single role, persistent tiles, one launch per step)"


def decode_dtype_label():
    """`dtype` of the line: the arithmetic the contraction runs in -- f32 unless the environment forces a gated split form (never the driver's run)."""
    return {"split": "bf16x3 split, f32 accumulate", "split_f16": "f16x2 split, f32 accumulate"}.get(os.environ.get("DAD3D_DECODE_KERNEL", ""), "f32")


def events_per_step(fn, steps: int, stream, dev, warmup: int = 0, settle: int = 0):
    """Seconds per call of `fn` from two hipEvents on `stream` around `steps` back-to-back calls, after `warmup` untimed calls and up to
    `settle` untimed PASSES of the same `steps` calls (stops early once two consecutive passes agree within 1 %). A leg that starts behind
    seconds of host-only work otherwise measures the clock ramp, not the kernel: after idle the shader clock needs ~30 ms of UNINTERRUPTED
    work to settle (B = 256: 47 -> 38.3 us per launch over the first 700 launches; short bursts with a synchronize in between do not
    ramp it). Returns (seconds per call of the timed pass, untimed passes run) -- the count is printed as `settle_passes`."""
    def one_pass():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        e0.record(stream)
        for _ in range(steps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) * 1e-3 / steps

    for _ in range(warmup):
        fn()
    prev, passes = None, 0
    while passes < settle:
        t = one_pass()
        passes += 1
        if prev is not None and abs(t - prev) <= 0.01 * prev:
            break
        prev = t
    return one_pass(), passes


def secondary_decode_b256(hm, lib, lmk_idx, dev, steps: int = 400, settle: int = 6):
    """BASELINE configs[2]: batch = 256, head_mesh (.obj vertices) path -- `vertices_3d` + `reprojected_vertices(to_2d=False)`
    (head_mesh.py:28-46) + the landmark gather, one fused launch; `steps` launches between two hipEvents on the launch stream.
    Every one of the 256 rows the timed launches wrote is held to tests/golden/decode_b256_golden.npz (reference HeadMesh, 40
    vertices per row)."""
    from dad_3dheads_amd import _lib, synthetic

    b = 256
    stream = torch.cuda.Stream(dev)
    params = torch.from_numpy(synthetic.synthetic_params(b, seed=B256_SEED)).to(dev)
    verts3d = torch.empty((b, N_VERTS, 3), dtype=torch.float32, device=dev)
    proj3 = torch.empty((b, N_VERTS, 3), dtype=torch.float32, device=dev)
    lmk_px = torch.empty((b, N_LMK, 2), dtype=torch.int32, device=dev)
    call = (hm.flame._handle, params.data_ptr(), b, _lib.MUTATE_PARAMS, verts3d.data_ptr(), proj3.data_ptr(), None, lmk_px.data_ptr(),
            stream.cuda_stream)

    def step():
        st = lib.dad3d_flame_decode(*call)
        if st:
            _lib.check(st)

    t, settle_passes = events_per_step(step, steps, stream, dev, warmup=50, settle=settle)
    flops, alg = FLOP_PER_IMAGE * b, CONST_BYTES + b * BYTES_PER_IMAGE_3D
    out = {"workload": "BASELINE configs[2]: batch=256 head_mesh path (3d_vertices + 3-component projected_vertices + 445 int landmarks), "
                       "one fused launch per step", "kernel": decode_kernel_label().replace("<true>", "<false>"),
           "steps": steps, "settle_passes": settle_passes, "ms_per_step": t * 1e3, "images_per_sec": b / t, "bound": "mfma", "achieved": flops / t / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS,
           "unit": "TFLOP/s", "frac": flops / t / 1e12 / PEAK_FP32_MFMA_TFLOPS, "algorithmic_bytes_per_launch": alg,
           "hbm_equiv_GBps": alg / t / 1e9, "hbm_frac": alg / t / 1e9 / PEAK_HBM_GBS}
    g = None
    try:
        g = np.load(os.path.join(ROOT, "tests", "golden", "decode_b256_golden.npz"))
        sub = g["subset"]
        dv = float(np.abs(verts3d[:, torch.from_numpy(sub).to(dev)].cpu().numpy() - g["v3d_sub"]).max())
        dp = float(np.abs(proj3[:, torch.from_numpy(sub).to(dev)].cpu().numpy() - g["proj3_sub"]).max())
        idx_dev = torch.from_numpy(lmk_idx).to(dev)
        gather_exact = bool(torch.equal(lmk_px, proj3[:, idx_dev, :2].to(torch.int32)))
        tz_ok = bool((params[:, 411] == 0).all()) and bool(np.array_equal(g["tz_after"], np.zeros(b, np.float32)))
        out["verification"] = {"checked": True, "rows": b, "max_abs_3d": dv, "max_abs_px": dp, "landmark_gather_exact": gather_exact,
                               "tz_zeroed_like_reference": tz_ok,
                               "golden": "tests/golden/decode_b256_golden.npz (reference HeadMesh, seed 104, every row, 40 vertices)"}
        out["outputs_verified"] = bool(dv < 5e-6 and dp < 1e-3 and gather_exact and tz_ok)
    except Exception as e:
        out["verification"] = {"checked": False, "why": f"{type(e).__name__}: {e}"}
        out["outputs_verified"] = None
    gold = g if out["verification"].get("checked") else None
    out["split"] = secondary_decode_b256_split(hm, lib, dev, params, gold, lmk_idx, stream, steps, settle, t)
    out["split_f16"] = secondary_decode_b256_split(hm, lib, dev, params, gold, lmk_idx, stream, steps, settle, t, form="split_f16")
    # BASELINE configs[3]'s per-GPU work (2048 rows over 8 GPUs = 256 each, only the 445 projected landmarks are gathered): the same rows,
    # landmark outputs only -> the C ABI runs the sub-model of the listed vertices (include/dad3d.h: dad3d_flame_num_landmark_vertices).
    lmk_only = torch.zeros_like(lmk_px)
    call_l = (hm.flame._handle, params.data_ptr(), b, _lib.TO_2D | _lib.MUTATE_PARAMS, None, None, None, lmk_only.data_ptr(), stream.cuda_stream)

    def step_l():
        st = lib.dad3d_flame_decode(*call_l)
        if st:
            _lib.check(st)

    tl, passes_l = events_per_step(step_l, steps, stream, dev, warmup=50, settle=settle)
    # one arithmetic: the sub-model runs the kernel of the whole-mesh launch above on the same basis values -> the same bits
    diff = lmk_only != lmk_px
    ok_l = int(diff.sum()) == 0 and int(lmk_only.abs().max()) > 0
    n_sub = int(lib.dad3d_flame_num_landmark_vertices(hm.flame._handle))
    flops_l = 2.0 * b * (3 * n_sub) * 416  # the blend-shape contraction of the listed vertices' columns (the rest is per-vertex epilogue)
    out["landmarks_only"] = {"workload": "BASELINE configs[3] per-GPU share: 256 rows -> 445 int landmarks only (what the final gather moves)",
                             "sub_model_vertices": n_sub, "steps": steps,
                             "settle_passes": passes_l, "ms_per_step": tl * 1e3, "images_per_sec": b / tl,
                             "gemm_flop_per_launch": flops_l, "achieved": flops_l / tl / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                             "frac": flops_l / tl / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                             "landmark_px_differ_from_whole_mesh_launch": int(diff.sum()), "outputs_verified": ok_l}
    # the whole of configs[3] on one GPU (2048 rows), same launch shape: the sub-model's tiles x batch chunks
    b2 = 2048
    params2 = torch.from_numpy(synthetic.synthetic_params(b2, seed=B256_SEED + 1)).to(dev)
    lmk2 = torch.zeros((b2, N_LMK, 2), dtype=torch.int32, device=dev)
    call_2 = (hm.flame._handle, params2.data_ptr(), b2, _lib.TO_2D | _lib.MUTATE_PARAMS, None, None, None, lmk2.data_ptr(), stream.cuda_stream)

    def step_2():
        st = lib.dad3d_flame_decode(*call_2)
        if st:
            _lib.check(st)

    t2, passes_2 = events_per_step(step_2, max(steps // 4, 20), stream, dev, warmup=10, settle=settle)
    f2 = 2.0 * b2 * (3 * n_sub) * 416
    out["landmarks_only"]["b2048"] = {"ms_per_step": t2 * 1e3, "images_per_sec": b2 / t2, "settle_passes": passes_2,
                                      "frac": f2 / t2 / 1e12 / PEAK_FP32_MFMA_TFLOPS, "nonzero": bool(int(lmk2.abs().max()) > 0)}
    return out


PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md; the 2:1-sparsity headline figure is twice this)


def secondary_decode_b256_split(hm, lib, dev, params, golden, lmk_idx, stream, steps, settle, t_fp32, form="split_bf16"):
    """The same configs[2] step on a GATED exact-product split of the contraction (csrc/flame_decode_split.hip;
    dad3d_flame_select_kernel(DAD3D_KERNEL_SPLIT_BF16): three bf16 planes, six products per K = 32; DAD3D_KERNEL_SPLIT_F16: two fp16
    planes, three products): a pre-pass + the tile kernel per step, both inside the timed launches; outputs held to the same goldens and
    bars as the fp32 leg. Not the default, not the headline: `value` / `dtype` / `roofline` stay on fp32."""
    from dad_3dheads_amd import _lib, synthetic

    b = params.shape[0]
    twin = hm.fork()
    twin.flame.select_kernel(form)
    n_prod, planes = (6, "bf16x3") if form == "split_bf16" else (3, "fp16x2")
    p2 = params.clone()
    verts3d = torch.empty((b, N_VERTS, 3), dtype=torch.float32, device=dev)
    proj3 = torch.empty((b, N_VERTS, 3), dtype=torch.float32, device=dev)
    lmk_px = torch.empty((b, N_LMK, 2), dtype=torch.int32, device=dev)
    call = (twin.flame._handle, p2.data_ptr(), b, _lib.MUTATE_PARAMS, verts3d.data_ptr(), proj3.data_ptr(), None, lmk_px.data_ptr(), stream.cuda_stream)

    def step():
        st = lib.dad3d_flame_decode(*call)
        if st:
            _lib.check(st)

    try:
        t, passes = events_per_step(step, steps, stream, dev, warmup=50, settle=settle)
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"}
    rows16 = (b + 15) // 16 * 16
    issued = n_prod * 2.0 * rows16 * 416 * 64 * ((N_VERTS + 19) // 20)  # six (three) MFMA products per (row, k, column) of every 64-column tile
    flops = FLOP_PER_IMAGE * b
    out = {"workload": f"BASELINE configs[2] on the gated {planes} split: the same 256 rows and outputs, pre-pass + tile kernel per step",
           "kernel": f"split_params_kernel<{planes}> + flame_decode_split_kernel<{planes}, false>",
           "dtype": f"{planes} exact-product split ({n_prod} products per K = 32), fp32 accumulate",
           "steps": steps, "settle_passes": passes, "ms_per_step": t * 1e3, "images_per_sec": b / t, "speedup_vs_fp32_leg": t_fp32 / t,
           "issued_bf16_flop_per_step": issued, "frac_bf16": issued / t / 1e12 / PEAK_BF16_MFMA_TFLOPS, "peak_bf16": PEAK_BF16_MFMA_TFLOPS,
           "peak_note": "dense 16-bit matrix peak: the same for bf16 and fp16 operands",
           "fp32_equivalent_TFLOPs": flops / t / 1e12, "fp32_equivalent_frac_of_fp32_mfma_peak": flops / t / 1e12 / PEAK_FP32_MFMA_TFLOPS,
           "error_vs_float64": "profiles/r06_split_error.md (both forms ~2x closer than the fp32 kernel on every 3-D line)"}
    if golden is not None:
        sub = torch.from_numpy(golden["subset"]).to(dev)
        dv = float(np.abs(verts3d[:, sub].cpu().numpy() - golden["v3d_sub"]).max())
        dp = float(np.abs(proj3[:, sub].cpu().numpy() - golden["proj3_sub"]).max())
        gather_exact = bool(torch.equal(lmk_px, proj3[:, torch.from_numpy(lmk_idx).to(dev), :2].to(torch.int32)))
        out["verification"] = {"rows": b, "max_abs_3d": dv, "max_abs_px": dp, "landmark_gather_exact": gather_exact,
                               "tz_zeroed_like_reference": bool((p2[:, 411] == 0).all())}
        out["outputs_verified"] = bool(dv < 5e-6 and dp < 1e-3 and gather_exact and out["verification"]["tz_zeroed_like_reference"])
    else:
        out["outputs_verified"] = None
    # the contract's own step (configs[1]: 64 rows, 2-D projection + 3-D + int landmarks) on this form, against the default kernel's outputs
    try:
        b64 = BATCH
        p64 = torch.from_numpy(synthetic.synthetic_params(b64, seed=GOLDEN_SEED)).to(dev)
        bufs = {k: (torch.empty((b64, N_VERTS, 3), device=dev), torch.empty((b64, N_VERTS, 2), device=dev),
                    torch.empty((b64, N_LMK, 2), dtype=torch.int32, device=dev)) for k in ("split", "default")}
        calls = {k: (hnd, p64.data_ptr(), b64, _lib.TO_2D | _lib.MUTATE_PARAMS, bufs[k][0].data_ptr(), bufs[k][1].data_ptr(), None, bufs[k][2].data_ptr(),
                     stream.cuda_stream) for k, hnd in (("split", twin.flame._handle), ("default", hm.flame._handle))}
        # (4000 launches per pass: a pass shorter than ~30 ms measures the clock ramp behind the host-side verification above, not the kernel)
        t64, _ = events_per_step(lambda: lib.dad3d_flame_decode(*calls["split"]), 4000, stream, dev, warmup=200, settle=settle)
        _lib.check(lib.dad3d_flame_decode(*calls["default"]))
        torch.cuda.synchronize(dev)
        out["contract_step_b64"] = {"ms_per_step": t64 * 1e3, "images_per_sec": b64 / t64,
                                    "max_abs_3d_vs_default_kernel": float((bufs["split"][0] - bufs["default"][0]).abs().max()),
                                    "max_abs_px_vs_default_kernel": float((bufs["split"][1] - bufs["default"][1]).abs().max())}
    except Exception as e:
        out["contract_step_b64"] = {"error": f"{type(e).__name__}: {e}"}
    # landmark outputs only on the same handle: the sub-model on the SAME form (phases dealt over workgroups), bit-equal to its whole-mesh launch
    try:
        lmk_only = torch.zeros_like(lmk_px)
        call_l = (twin.flame._handle, p2.data_ptr(), b, _lib.MUTATE_PARAMS, None, None, None, lmk_only.data_ptr(), stream.cuda_stream)
        tl, _ = events_per_step(lambda: lib.dad3d_flame_decode(*call_l), steps, stream, dev, warmup=20, settle=2)
        b2 = 2048
        params2 = torch.from_numpy(synthetic.synthetic_params(b2, seed=B256_SEED + 1)).to(dev)
        lmk2 = torch.zeros((b2, N_LMK, 2), dtype=torch.int32, device=dev)
        call_2 = (twin.flame._handle, params2.data_ptr(), b2, _lib.MUTATE_PARAMS, None, None, None, lmk2.data_ptr(), stream.cuda_stream)
        t2, _ = events_per_step(lambda: lib.dad3d_flame_decode(*call_2), max(steps // 4, 20), stream, dev, warmup=10, settle=2)
        same = bool(torch.equal(lmk_only, lmk_px))
        out["landmarks_only"] = {"ms_per_step": tl * 1e3, "b2048_ms_per_step": t2 * 1e3, "b2048_images_per_sec": b2 / t2,
                                 "bit_equal_to_this_handles_whole_mesh_launch": same, "b2048_nonzero": bool(int(lmk2.abs().max()) > 0)}
        if out["outputs_verified"] is not None:
            out["outputs_verified"] = bool(out["outputs_verified"] and same)
    except Exception as e:
        out["landmarks_only"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def secondary_render_b64(hm, static, dev, rank_seed: int, steps: int = 200, cpu_budget_s: float = 8.0, settle: int = 6):
    """BASELINE configs[4]'s per-GPU share on ONE stream: 64 images of decode (3-component projection, z flipped) -> vertex normals +
    Phong light + triangle records -> z-buffer raster onto 256 x 256 x 3 uint8, three launches per step; hipEvents around `steps`
    steps. Returns (the leg, cpu_baseline_render): the CPU leg also re-rasterises the first timed image with the reference's own
    rasteriser (Sim3DR/lighting.py:37-71 -> rasterize_kernel.cpp:219-292) and compares bytes."""
    from dad_3dheads_amd import synthetic
    from dad_3dheads_amd.Sim3DR import Mesh
    from dad_3dheads_amd.sharding import ShardedRenderer

    faces = static["faces"]
    stream = torch.cuda.Stream(dev)
    params = torch.from_numpy(synthetic.synthetic_params(BATCH, seed=rank_seed)).to(dev)
    with torch.cuda.stream(stream):
        renderer = ShardedRenderer(hm.fork(), Mesh(faces, N_VERTS, device=dev.index))
        t, settle_passes = events_per_step(lambda: renderer.render_local(params), steps, stream, dev, warmup=30, settle=settle)
    img = renderer._img
    covered = float((img.reshape(BATCH, -1).max(dim=1).values > 0).float().mean().item())
    alg = BATCH * (RASTER_BYTES_PER_IMAGE + 120_552) + CONST_BYTES + BATCH * (1652 + 60_276)
    leg = {"workload": "BASELINE configs[4] per-GPU share: batch=64 head_mesh (3-component projection, z flipped) + vertex normals + Phong light "
                       "+ z-buffer raster of 9976 triangles onto 256x256x3 uint8, three launches per step, one stream",
           "steps": steps, "settle_passes": settle_passes, "us_per_batch": t * 1e6, "images_per_sec": BATCH / t, "bound": "hbm", "algorithmic_bytes_per_step": alg,
           "achieved": alg / t / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": alg / t / 1e9 / PEAK_HBM_GBS,
           "images_with_coverage": covered}
    timed = [(np.ascontiguousarray(renderer._dec["proj"][i].cpu().numpy()), np.ascontiguousarray(renderer._light_buf[i].cpu().numpy()),
              img[i].cpu().numpy()) for i in (0, BATCH - 1)]
    # the serving shape of the same workload: TWO batches of 64 in flight (a forked decode handle, a mesh handle and buffers per stream,
    # steps alternate) -- one batch's latency-bound phases overlap the other's ALU-bound ones. Kernels of different streams overlap, so this
    # is the wall clock between synchronizes over `steps` steps (same settle rule); its first image of the second lane is checked as well.
    stream2 = torch.cuda.Stream(dev)
    params2 = torch.from_numpy(synthetic.synthetic_params(BATCH, seed=rank_seed + 1)).to(dev)
    with torch.cuda.stream(stream2):
        renderer2 = ShardedRenderer(hm.fork(), Mesh(faces, N_VERTS, device=dev.index))
        renderer2.render_local(params2)
    torch.cuda.synchronize(dev)
    lanes = ((renderer, stream, params), (renderer2, stream2, params2))

    def two_stream_pass():
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for k in range(steps):
            r, st, p = lanes[k & 1]
            with torch.cuda.stream(st):
                r.render_local(p)
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / steps

    prev, passes2 = None, 0
    while passes2 < settle:
        t2 = two_stream_pass()
        passes2 += 1
        if prev is not None and abs(t2 - prev) <= 0.01 * prev:
            break
        prev = t2
    t2 = two_stream_pass()
    leg["two_streams"] = {"us_per_batch": t2 * 1e6, "images_per_sec": BATCH / t2, "settle_passes": passes2, "steps": steps,
                          "value_from": "wall clock between synchronizes, two batches of 64 in flight (steps alternate between two streams)"}
    timed.append((np.ascontiguousarray(renderer2._dec["proj"][0].cpu().numpy()), np.ascontiguousarray(renderer2._light_buf[0].cpu().numpy()),
                  renderer2._img[0].cpu().numpy()))
    cpu = cpu_baseline_render(timed[0][0], faces, timed, budget_s=cpu_budget_s)
    leg["timed_images_match_reference_raster"] = cpu.pop("timed_images_match")
    leg["speedup_vs_cpu_baseline_render"] = leg["images_per_sec"] / cpu["value"]
    return leg, cpu


class E2EWarmup:
    """Builds the end-to-end predictor and runs its first batches on a thread of its own WHILE the decode-path CPU baseline runs (the GPU
    is idle then): a fresh box compiles / loads ~200 MIOpen kernels on the CNN's first call (~45 s), and the driver's whole run has to
    stay inside its budget. Joined before any GPU-timed secondary leg starts."""

    def __init__(self, model, lmk_idx, dev):
        import threading

        self.pred = self.images = self.error = None
        self.seconds = 0.0

        def work():
            t0 = time.perf_counter()
            try:
                from dad_3dheads_amd.predictor import FaceMeshPredictor

                torch.cuda.set_device(dev)
                with torch.cuda.stream(torch.cuda.Stream(dev)):
                    pred = FaceMeshPredictor.random_init(dtype=torch.bfloat16, tune=False, cuda_id=dev.index or 0, flame_model=model, landmarks=lmk_idx)
                    g = torch.Generator().manual_seed(0)
                    images = torch.randint(0, 255, (BATCH, 256, 256, 3), dtype=torch.uint8, generator=g).to(dev)
                    for _ in range(3):
                        pred.predict_tensor(images)
                    torch.cuda.current_stream(dev).synchronize()
                self.pred, self.images = pred, images
            except Exception as e:  # reported in the leg, never fatal for the metric's line
                self.error = f"{type(e).__name__}: {e}"
            self.seconds = time.perf_counter() - t0

        self.thread = threading.Thread(target=work, daemon=True)
        self.thread.start()

    def join(self):
        self.thread.join()
        return self


def secondary_e2e_b64(warm, model, lmk_idx, dev, batches: int = 20, cpu_images: int = 30):
    """The north star's literal sentence, reported SEPARATELY from the metric: the drop-in predictor end to end on
    the GPU -- uint8 frames resident in HBM -> preprocess -> DAD-3DNet (network.py declaration, random weights, PyTorch-ROCm bf16
    channels-last) -> re-adjust -> fused decode + 445 landmarks, batches of 64, no host copy in between -- against the reference CPU
    predictor's call sequence in the same process (predictor.py:97-145: fp32, one 256 x 256 image per call, torch.set_num_threads(8)).
    MIOpen is NOT tuned here (tuning takes minutes; tools/bench_e2e.py does it: +36 % on the CNN), so the GPU figure is the lower one."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))
    from cpu_predictor import cpu_reference_predictor  # imports the oracle: the CPU comparator, never the product

    if warm.error:
        return {"error": warm.error}
    t_start = time.perf_counter()
    pred, images = warm.pred, warm.images
    out = pred.predict_tensor(images)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(batches):
        out = pred.predict_tensor(images)
    torch.cuda.synchronize(dev)
    t_gpu = (time.perf_counter() - t0) / batches
    finite = all(bool(torch.isfinite(v.float()).all()) for v in out.values() if isinstance(v, torch.Tensor))
    cpu = cpu_reference_predictor(model, lmk_idx, cpu_images, threads=(8,), budget_s=25.0)
    torch.set_num_threads(os.cpu_count() or 1)
    gpu_ips = BATCH / t_gpu
    return {"workload": "FaceMeshPredictor end to end, batch 64 of 256 x 256 uint8 frames resident in HBM (bf16 CNN, random weights) vs the "
                        "reference CPU predictor's call sequence, one image per call (reported separately from the metric: BASELINE.md 3.7)",
            "gpu_images_per_sec": gpu_ips, "gpu_ms_per_batch": t_gpu * 1e3, "gpu_batches_timed": batches, "gpu_dtype": "bf16 CNN (MIOpen untuned), f32 decode",
            "gpu_outputs_finite": finite, "cpu_reference_predictor": cpu, "ratio": gpu_ips / cpu["images_per_s"], "north_star_target_ratio": 200,
            "meets_north_star_sentence": bool(gpu_ips / cpu["images_per_s"] >= 200), "leg_seconds": time.perf_counter() - t_start,
            "warmup_seconds_overlapped_with_cpu_baseline": warm.seconds}


def run_decode(args, dist, dev, rank, world, hm, lib, static, model, lmk_idx):
    from dad_3dheads_amd import _lib, synthetic

    n_streams = max(1, args.streams)
    meshes = [hm] + [hm.fork() for _ in range(n_streams - 1)]  # one handle per stream, constants shared in HBM
    streams = [torch.cuda.Stream(dev) for _ in range(n_streams)]
    flags = _lib.TO_2D | _lib.MUTATE_PARAMS
    sets = []
    for i in range(n_streams):  # per-rank seed = base + rank; further streams continue the sequence
        params = torch.from_numpy(synthetic.synthetic_params(BATCH, seed=GOLDEN_SEED + rank + world * i)).to(dev)
        verts3d = torch.empty((BATCH, N_VERTS, 3), dtype=torch.float32, device=dev)
        proj = torch.empty((BATCH, N_VERTS, 2), dtype=torch.float32, device=dev)
        lmk_px = torch.empty((BATCH, N_LMK, 2), dtype=torch.int32, device=dev)
        sets.append({"params": params, "verts3d": verts3d, "proj": proj, "lmk_px": lmk_px,
                     "call": (meshes[i].flame._handle, params.data_ptr(), BATCH, flags, verts3d.data_ptr(), proj.data_ptr(),
                              None, lmk_px.data_ptr(), streams[i].cuda_stream)})
    decode = lib.dad3d_flame_decode
    direct = make_direct_gather(dist)  # RCCL's C API on the launch stream (rccl.py); None under the shared-GPU test hook
    have_gather = dist is not None or direct is not None
    gathered = torch.empty((world * BATCH, N_LMK, 2), dtype=torch.int32, device=dev) if have_gather else None
    torch.cuda.synchronize(dev)

    def step(k):
        st = decode(*sets[k % n_streams]["call"])
        if st:
            _lib.check(st)

    def gather_last(k_last):  # the job's one collective: landmarks of the last step, ON that step's stream
        s_last = streams[k_last % n_streams]
        if direct is not None:
            direct.all_gather(gathered, sets[k_last % n_streams]["lmk_px"], stream=s_last.cuda_stream)
            return s_last
        torch.cuda.current_stream(dev).wait_stream(s_last)
        dist.all_gather_into_tensor(gathered, sets[k_last % n_streams]["lmk_px"])
        return torch.cuda.current_stream(dev)

    prewarm_ms = prewarm(step, args.prewarm_ms, dev)
    for k in range(args.warmup):
        step(k)
    align = AlignedStart(dist if n_streams == 1 else None, direct, dev, world)
    if have_gather:
        gather_last(max(args.warmup - 1, 0))  # RCCL communicator warm-up (untimed)
        align.queue(streams[0])

    handle, stream = sets[0]["call"][0], sets[0]["call"][-1]
    tot, cnt = C.c_double(), C.c_int()
    region0, region1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fence(dist, dev)
    if dist is not None and REWARM_STEPS:
        # the barrier parks this rank's GPU for as long as the slowest rank needs (an idle millisecond drops the shader clock):
        # a few untimed steps behind it, then the synchronize that opens the timed region (disclosed in config.rewarm_steps)
        for k in range(REWARM_STEPS):
            step(k)
        torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    if n_streams == 1:  # two hipEvents on the launch stream bracket the K launches of the timed region itself; no host
        align.queue(streams[0])  # synchronisation between the alignment collective, the K launches and the collective behind them
        region0.record(streams[0])
    for k in range(args.steps):
        step(k)
    gather_host_us = gather_region_us = None
    pre_gather = torch.cuda.Event(enable_timing=True)
    if n_streams == 1:
        pre_gather.record(streams[0])  # behind the K launches
    if have_gather:
        th0 = time.perf_counter()
        s_gather = gather_last(args.steps - 1)
        gather_host_us = (time.perf_counter() - th0) * 1e6
        region1.record(s_gather)  # behind the collective: the region's device time, host latency aside
    fence(dist, dev)
    wall = max_over_ranks(dist, dev, time.perf_counter() - t0)
    gather_us = time_gather(lambda: gather_last(args.steps - 1), dev) if have_gather else None
    # the two device-side definitions of the job's time, each MAX over ranks: [first launch .. last launch] and [.. end of the all-gather]
    compute_s = max_over_ranks(dist, dev, region0.elapsed_time(pre_gather) * 1e-3) if n_streams == 1 else None
    with_gather_s = max_over_ranks(dist, dev, region0.elapsed_time(region1) * 1e-3) if (have_gather and n_streams == 1) else None
    region_s = with_gather_s if dist is not None else None  # `value` under a process group
    if have_gather and n_streams == 1:
        gather_region_us = pre_gather.elapsed_time(region1) * 1e3  # the gather as it sat in the timed region (queued behind K launches)

    start_skew = align.skew_us(region0) if n_streams == 1 else None
    # dominant-kernel duration = hipEvent time of the K back-to-back launches / K (one kernel per step). With one
    # stream the events bracketed the timed region; with several, kernels of different streams overlap and a launch's
    # duration is no longer a property of the kernel, so the same K steps are run once more on ONE stream.
    if n_streams > 1:
        _lib.check(lib.dad3d_flame_profile_begin(handle, stream))
        for _ in range(args.steps):
            st = decode(*sets[0]["call"])
            if st:
                _lib.check(st)
        _lib.check(lib.dad3d_flame_profile_end(handle, stream, C.byref(tot), C.byref(cnt)))
        kern_s = tot.value / max(cnt.value, 1) * 1e-3
    else:
        kern_s = region0.elapsed_time(pre_gather) * 1e-3 / args.steps  # this rank's K launches on its stream
    clock_mhz = observed_shader_clock_mhz(lib, handle, sets[0]["call"], dev) if rank == 0 else None
    events_s = compute_s if compute_s is not None else max_over_ranks(dist, dev, tot.value * 1e-3)  # MAX over ranks

    # long_region: the same launch 2000 more times between two hipEvents (plain N = 1 run; the contract region above is untouched)
    secondary_on = world == 1 and dist is None and n_streams == 1 and not args.no_secondary
    long_s = events_per_step(lambda: step(0), 2000, streams[0], dev)[0] if secondary_on else None

    timeouts = C.c_uint()
    _lib.check(lib.dad3d_flame_handoff_timeouts(handle, C.byref(timeouts)))
    check = verify_against_golden(sets[0], lmk_idx, dev) if rank == 0 else None  # after EVERY launch of this process on these buffers
    ok_gather = True
    if have_gather:  # this rank's slice of the gathered landmarks is what its last timed step wrote
        mine = sets[(args.steps - 1) % n_streams]["lmk_px"]
        ok_gather = bool(torch.equal(gathered[rank * BATCH:(rank + 1) * BATCH], mine))
    if rank != 0:
        return None
    images = world * BATCH * args.steps
    flops = FLOP_PER_IMAGE * BATCH
    alg_bytes = CONST_BYTES + BATCH * BYTES_PER_IMAGE
    # one stream: device events ARE the timed region (K launches; plus the final all-gather under a process group); the wall
    # clock of a short region (the driver's 20 steps) mostly measures the host's launch latency
    from_events = n_streams == 1
    elapsed = (region_s if region_s is not None else events_s) if from_events else wall
    traffic, traffic_src = pmc_json("pmc_traffic", "traffic_bytes_per_launch")
    mfma_busy, mfma_src = pmc_json("pmc_sq", "mfma_busy_fraction_of_kernel_time_all_1024_simds")
    out = {
        "metric": "images/sec (FLAME decode + 445-lmk projection), batch 64 @ 256^2",
        "value": images / elapsed,
        "value_compute": images / compute_s if compute_s is not None else None,
        "value_with_gather": images / with_gather_s if with_gather_s is not None else None,
        "unit": "images/sec",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "ms_per_step_compute": compute_s / args.steps * 1e3 if compute_s is not None else None,
        "ms_per_step_with_gather": with_gather_s / args.steps * 1e3 if with_gather_s is not None else None,
        "wall_ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": decode_dtype_label(),
        "data": "synthetic",
        "config": {
            "value_definition": ("images / ms_per_step_with_gather (process group: the job ends behind its one all-gather)" if region_s is not None
                                 else "images / compute time (plain N = 1 run: the metric as BASELINE.json defines it; the same collective ran "
                                      "behind the K steps and is reported as ms_per_step_with_gather)") if from_events
                                else "wall clock (several streams)",
            "scaling_note": (None if not (have_gather and n_streams == 1 and with_gather_s) else
                             f"built-in gather share of the timed region: gather_in_region_us / region = {gather_region_us / (with_gather_s * 1e6):.3f} "
                             f"(K = {args.steps} steps; the share shrinks as K grows). `value_compute` and `value_with_gather` are printed at EVERY N: read a "
                             "scaling curve on ONE of them, not on `value` across the plain N = 1 line and the process-group lines. "
                             + ("Unmeasured on N > 1: this is a world of one." if world == 1 else "")),
            "toolchain": (lib.dad3d_build_info() or b"").decode("utf-8", "replace"),
            "workload": "BASELINE configs[1]: batch=64 synthetic 256x256 per GPU, 445_landmarks path "
                        "(3d_vertices + projected_vertices + 445 int landmarks per image), seeded synthetic "
                        "FLAME-shaped model (real flame.pkl not redistributed)",
            "batch_per_gpu": BATCH,
            "global_batch": world * BATCH,
            "parallelism": f"image-sharded x{world}, one final RCCL all-gather of landmarks"
                           + ((" (TEST HOOK: ranks share one GPU, gloo with host-staged gathers -- not a multi-GPU measurement)"
                               if getattr(dist, "_dad3d_shared_gpu", False) else " (process group: nccl)")
                              if dist is not None else " (no process group: plain N=1 run, communicator of one rank)"),
            "streams": n_streams,
            "prewarm_ms": prewarm_ms,
            "value_from": ("device events around the K timed launches AND the final all-gather (MAX over ranks)" if region_s is not None
                           else "hipEvents on the launch stream around the K timed launches") if from_events
                          else "wall clock between barrier+synchronize pairs (MAX over ranks)",
            "images_per_sec_wall": images / wall,
            "single_stream_ms_per_step": kern_s * 1e3,
            "outputs_verified": bool(check.get("ok")) and ok_gather if check and check.get("checked") else None,
            "verification": check,
            "handoff_timeouts": int(timeouts.value),
            "gather_us": gather_us,
            "gather_in_region_us": gather_region_us,
            "gather_host_call_us": gather_host_us,
            "rewarm_steps": REWARM_STEPS if dist is not None else 0,
            "start_alignment": align.how(),
            "start_skew_us": start_skew,
            "gather_via": None if not have_gather else ("ncclAllGather on the launch stream (dad_3dheads_amd/rccl.py)" if direct is not None
                                                       else "torch.distributed.all_gather_into_tensor"),
        },
        "roofline": {
            "kernel": decode_kernel_label() + "; duration = back-to-back launches on ONE stream",
            "bound": "mfma",
            "achieved": flops / kern_s / 1e12,
            "peak": PEAK_FP32_MFMA_TFLOPS,
            "unit": "TFLOP/s",
            "frac": flops / kern_s / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            "traffic": traffic,
            "traffic_source": traffic_src and traffic_src + " (committed PMC pass, NOT measured in this run)",
            "mfma_busy_pmc": mfma_busy,
            "mfma_busy_source": mfma_src and mfma_src + " (committed PMC pass, NOT measured in this run)",
            "kernel_us": kern_s * 1e6,
            "shader_clock_mhz": clock_mhz,
            "frac_at_observed_clock": (flops / kern_s / 1e12 / (PEAK_FP32_MFMA_TFLOPS * clock_mhz / 2400.0)) if clock_mhz else None,
            "algorithmic_flop_per_launch": flops,
            "algorithmic_bytes_per_launch": alg_bytes,
            "hbm_equiv_GBps": alg_bytes / kern_s / 1e9,
            "hbm_frac": alg_bytes / kern_s / 1e9 / PEAK_HBM_GBS,
        },
    }
    e2e_warm = E2EWarmup(model, lmk_idx, dev) if secondary_on else None  # (after every timed decode region; the GPU idles during the next leg)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(model, lmk_idx)
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    if e2e_warm is not None:
        e2e_warm.join()
    if secondary_on:
        out["long_region"] = {"steps": 2000, "ms_per_step": long_s * 1e3, "images_per_sec": BATCH / long_s,
                              "frac": flops / long_s / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                              "what": "the contract step 2000 more times between two hipEvents on the launch stream, same buffers; "
                                      "config.verification was taken after these launches too"}
        out["secondary"] = {"decode_b256": secondary_decode_b256(hm, lib, lmk_idx, dev)}
        render, cpu_render = secondary_render_b64(hm, static, dev, GOLDEN_SEED)
        out["secondary"]["render_b64"] = render
        out["cpu_baseline_render"] = cpu_render
        try:
            out["secondary"]["e2e_b64"] = secondary_e2e_b64(e2e_warm, model, lmk_idx, dev)
        except Exception as e:  # the metric's line must not depend on the CNN stack
            out["secondary"]["e2e_b64"] = {"error": f"{type(e).__name__}: {e}"}
        out["secondary"]["decode_b256_split"] = out["secondary"]["decode_b256"].pop("split")
        out["secondary"]["decode_b256_split_f16"] = out["secondary"]["decode_b256"].pop("split_f16")
        out["secondary"]["outputs_verified"] = bool(out["secondary"]["decode_b256"]["outputs_verified"]) and \
            bool(out["secondary"]["decode_b256_split"].get("outputs_verified")) and \
            bool(out["secondary"]["decode_b256_split_f16"].get("outputs_verified")) and \
            bool(out["secondary"]["decode_b256"]["landmarks_only"]["outputs_verified"]) and \
            bool(render["timed_images_match_reference_raster"]) and bool(out["config"]["outputs_verified"])
    return out


def run_render(args, dist, dev, rank, world, hm, lib, static, model, lmk_idx):
    """BASELINE configs[4] per-GPU share: decode (3-component projection, z flipped) -> normals + Phong + raster.
    `--streams 2`: two batches of 64 in flight per GPU (a forked decode handle, a mesh handle and a buffer set per stream; the
    steps alternate); `value` is then the wall clock between synchronizes, per-kernel times belong to the one-stream run."""
    from dad_3dheads_amd import synthetic
    from dad_3dheads_amd.Sim3DR import Mesh
    from dad_3dheads_amd.sharding import ShardedRenderer

    faces = static["faces"]
    n_streams = max(1, args.streams)
    lanes = []
    for i in range(n_streams):
        lanes.append({"renderer": ShardedRenderer(hm if i == 0 else hm.fork(), Mesh(faces, N_VERTS, device=dev.index)),
                      "stream": torch.cuda.current_stream(dev) if n_streams == 1 else torch.cuda.Stream(dev),
                      "params": torch.from_numpy(synthetic.synthetic_params(BATCH, seed=GOLDEN_SEED + rank + world * i)).to(dev)})
    renderer = lanes[0]["renderer"]
    direct = make_direct_gather(dist) if dist is not None else None
    to_root = args.gather == "root"
    receives = rank == 0 or not to_root
    gathered = torch.empty((world * BATCH, 256, 256, 3), dtype=torch.uint8, device=dev) if (dist is not None and receives) else None
    torch.cuda.synchronize(dev)

    def step(k):
        ln = lanes[k % n_streams]
        if n_streams == 1:
            ln["renderer"].render_local(ln["params"])
        else:
            with torch.cuda.stream(ln["stream"]):
                ln["renderer"].render_local(ln["params"])

    def gather_last():  # the images the last step rendered, on that step's stream (ordered behind it)
        ln = lanes[(args.steps - 1) % n_streams]
        img = ln["renderer"]._img
        with torch.cuda.stream(ln["stream"]):
            if to_root and direct is not None:  # every rank's 12.6 MB crosses xGMI once, to rank 0
                direct.gather_to_root(gathered, img, root=0)
            elif to_root:
                dist.gather(img, list(gathered.view(world, BATCH, 256, 256, 3).unbind(0)) if rank == 0 else None, dst=0)
            elif direct is not None:
                direct.all_gather(gathered, img)
            else:
                dist.all_gather_into_tensor(gathered, img)
        return ln["stream"]

    prewarm_ms = prewarm(step, args.prewarm_ms, dev)
    for k in range(args.warmup):
        step(k)
    s0 = lanes[0]["stream"]
    align = AlignedStart(dist if n_streams == 1 else None, direct, dev, world)
    if dist is not None:
        gather_last()
        align.queue(s0)
    e0, e_steps, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    fence(dist, dev)
    t0 = time.perf_counter()
    align.queue(s0)  # device-side aligned start (AlignedStart): no host synchronisation between it and the K steps
    e0.record(s0)
    for k in range(args.steps):
        step(k)
    e_steps.record(s0)
    s_end = s0
    if dist is not None:  # under a process group the region ends behind the all-gather
        s_end = gather_last()
    e1.record(s_end)
    fence(dist, dev)
    wall = max_over_ranks(dist, dev, time.perf_counter() - t0)
    ev_s = max_over_ranks(dist, dev, e0.elapsed_time(e1) * 1e-3)
    gather_us = time_gather(gather_last, dev, n=5) if dist is not None else None
    steps_s = e0.elapsed_time(e_steps) * 1e-3  # one stream: this rank's K steps alone, the per-step duration behind `roofline`
    start_skew = align.skew_us(e0)
    if rank != 0:
        return None
    img = lanes[(args.steps - 1) % n_streams]["renderer"]._img
    covered = min(float((ln["renderer"]._img.reshape(BATCH, -1).max(dim=1).values > 0).float().mean().item()) for ln in lanes)
    ok_gather = True if dist is None else bool(torch.equal(gathered[:BATCH], img))
    images = world * BATCH * args.steps
    # one stream: device events around the K steps (plus the final all-gather under a process group), MAX over ranks; several
    # streams: kernels of different streams overlap and only the wall clock between the synchronizes brackets them all
    elapsed = ev_s if n_streams == 1 else wall
    per_step = (steps_s if n_streams == 1 else wall) / args.steps
    alg = BATCH * (RASTER_BYTES_PER_IMAGE + 120_552) + CONST_BYTES + BATCH * (1652 + 60_276)
    out = {
        "metric": "images/sec (head_mesh decode + Sim3DR face-mesh render), batch 64 @ 256^2 per GPU",
        "value": images / elapsed, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "wall_ms_per_step": wall / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[4] per-GPU share: batch=64 head_mesh (3-component projection) + vertex normals + "
                               "Phong light + z-buffer raster of 9976 triangles onto 256x256x3 uint8, three launches per step; "
                               + ("one gather of the uint8 images to rank 0 ends the job" if to_root else "one all-gather of the uint8 images ends the job"),
                   "batch_per_gpu": BATCH, "global_batch": world * BATCH, "streams": n_streams,
                   "parallelism": f"image-sharded x{world}, one final " + ("gather to rank 0 (grouped ncclSend / ncclRecv)" if to_root else "RCCL all-gather")
                                  + " of [64,256,256,3] uint8 per rank",
                   "gather_mode": args.gather, "start_alignment": align.how(), "start_skew_us": start_skew,
                   "prewarm_ms": prewarm_ms, "images_with_coverage": covered, "gather_verified": ok_gather, "gather_us": gather_us,
                   "value_from": ("device events around the K timed steps" + (" AND the final all-gather (MAX over ranks)" if dist is not None else ""))
                                 if n_streams == 1 else f"wall clock between synchronizes, {n_streams} batches in flight (steps alternate between the streams)"},
        "roofline": {"kernel": "decode + tri_geometry(+normals+light) + raster_kernel, three launches"
                               + ("" if n_streams == 1 else f"; {n_streams} streams: per-step time = wall clock / K, kernels of different streams overlap"),
                     "bound": "hbm", "achieved": alg / per_step / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": alg / per_step / 1e9 / PEAK_HBM_GBS, "traffic": None, "algorithmic_bytes_per_step": alg},
    }
    if world == 1 and not args.no_cpu_baseline:
        timed = [(np.ascontiguousarray(ln["renderer"]._dec["proj"][0].cpu().numpy()), np.ascontiguousarray(ln["renderer"]._light_buf[0].cpu().numpy()),
                  ln["renderer"]._img[0].cpu().numpy()) for ln in lanes]
        out["cpu_baseline"] = cpu_baseline_render(timed[0][0], faces, timed)
        out["config"]["timed_images_match_reference_raster"] = out["cpu_baseline"].pop("timed_images_match")
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    return out


if __name__ == "__main__":
    main()
