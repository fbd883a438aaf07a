"""CPU, 2 processes over gloo: the multi-GPU path's partitioning + the one collective (`gather_rows`).
Each rank produces the oracle landmarks of ITS shard only; the gathered result must equal the single-process
answer row for row, including ragged shards and a batch smaller than the world."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dad_3dheads_amd import sharding, synthetic


def test_shard_ranges_cover_and_balance():
    for n in (0, 1, 5, 64, 512, 2048, 2051):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == sharding.shard_sizes(n, world)
    assert sharding.shard_range(2048, 3, 8) == (768, 1024)  # BASELINE config 4: 256 per GPU
    assert sharding.shard_range(512, 7, 8) == (448, 512)    # BASELINE config 5: 64 per GPU
    with pytest.raises(ValueError):
        sharding.shard_range(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_list, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import flame_ref

        static = synthetic.load_static()
        consts = flame_ref.FlameConstants.from_model(synthetic.synthetic_flame_model(0, static))
        lmk = static["lmk_445"]
        torch.set_num_threads(2)
        for n in n_list:
            params = torch.from_numpy(synthetic.synthetic_params(max(n, 1), seed=77)[:n])
            lo, hi = sharding.shard_range(n, rank, world)
            if hi > lo:
                proj = flame_ref.reprojected_vertices(consts, params[lo:hi].clone(), to_2d=True)
                local = torch.from_numpy(flame_ref.gather_landmarks_int(proj, lmk).astype(np.int32))
            else:
                local = torch.empty((0, len(lmk), 2), dtype=torch.int32)
            full = sharding.gather_rows(local, n)
            np.save(os.path.join(out_dir, f"r{rank}_n{n}.npy"), full.numpy())
        with pytest.raises(ValueError):
            sharding.gather_rows(torch.zeros((3, 2)), 100)
    finally:
        dist.destroy_process_group()


def test_gather_rows_two_ranks_matches_single_process(tmp_path, flame_consts, static):
    from oracle import flame_ref

    n_list = [8, 7, 1]  # even, ragged, smaller than the world (rank 1 holds nothing)
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_list, str(tmp_path)), nprocs=world, join=True)
    for n in n_list:
        params = torch.from_numpy(synthetic.synthetic_params(max(n, 1), seed=77)[:n])
        proj = flame_ref.reprojected_vertices(flame_consts, params.clone(), to_2d=True)
        ref = flame_ref.gather_landmarks_int(proj, static["lmk_445"]).astype(np.int32)
        for r in range(world):
            got = np.load(tmp_path / f"r{r}_n{n}.npy")
            assert got.shape == ref.shape and np.array_equal(got, ref), (n, r)


def test_gather_rows_single_process_passthrough():
    x = torch.arange(12).reshape(4, 3)
    assert sharding.gather_rows(x, 4) is x
    with pytest.raises(ValueError):
        sharding.gather_rows(x, 5)


def _image_worker(rank, world, port, n_list, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        for n in n_list:
            lo, hi = sharding.shard_range(n, rank, world)
            rows = torch.arange(lo, hi, dtype=torch.int64)  # image i is filled with a pattern derived from i
            local = ((rows[:, None, None, None] * 7 + torch.arange(3)[None, None, None, :] * 11 +
                      torch.arange(8)[None, :, None, None] * 3 + torch.arange(6)[None, None, :, None]) % 251).to(torch.uint8)
            full = sharding.gather_rows(local.contiguous(), n)
            np.save(os.path.join(out_dir, f"img_r{rank}_n{n}.npy"), full.numpy())
            for root in (0, 1):  # gather-to-root: only `root` receives, the other rank gets None
                at_root = sharding.gather_rows(local.contiguous(), n, root=root)
                assert (at_root is None) == (rank != root)
                if at_root is not None:
                    np.save(os.path.join(out_dir, f"img_root{root}_n{n}.npy"), at_root.numpy())
    finally:
        dist.destroy_process_group()


def test_gather_rows_uint8_images_two_ranks(tmp_path):
    """BASELINE config 5's collective: [n_r, h, w, 3] uint8 images per rank -> the whole batch on every rank."""
    n_list, world = [6, 5], 2
    mp.spawn(_image_worker, args=(world, _free_port(), n_list, str(tmp_path)), nprocs=world, join=True)
    for n in n_list:
        rows = np.arange(n)[:, None, None, None]
        ref = ((rows * 7 + np.arange(3)[None, None, None, :] * 11 + np.arange(8)[None, :, None, None] * 3 +
                np.arange(6)[None, None, :, None]) % 251).astype(np.uint8)
        for r in range(world):
            got = np.load(tmp_path / f"img_r{r}_n{n}.npy")
            assert got.dtype == np.uint8 and np.array_equal(got, ref), (n, r)
            at_root = np.load(tmp_path / f"img_root{r}_n{n}.npy")  # the default of the render workload (bench.py --gather root)
            assert at_root.dtype == np.uint8 and np.array_equal(at_root, ref), (n, r)


def _pattern_rows(lo, hi):
    """int32 landmark-shaped rows [hi - lo, 445, 2] whose content is a function of the GLOBAL row index."""
    rows = torch.arange(lo, hi, dtype=torch.int64)
    return ((rows[:, None, None] * 131 + torch.arange(445)[None, :, None] * 7 + torch.arange(2)[None, None, :]) % 65521).to(torch.int32)


def _world8_worker(rank, world, port, n_list, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        for n in n_list:
            lo, hi = sharding.shard_range(n, rank, world)
            full = sharding.gather_rows(_pattern_rows(lo, hi).contiguous(), n)
            ok = full.shape == (n, 445, 2) and bool(torch.equal(full, _pattern_rows(0, n)))
            img = ((torch.arange(lo, hi)[:, None, None, None] * 5 + torch.arange(4)[None, :, None, None] * 3 +
                    torch.arange(4)[None, None, :, None] + torch.arange(3)[None, None, None, :] * 2) % 251).to(torch.uint8)
            full_img = sharding.gather_rows(img.contiguous(), n)
            ref_img = ((torch.arange(0, n)[:, None, None, None] * 5 + torch.arange(4)[None, :, None, None] * 3 +
                        torch.arange(4)[None, None, :, None] + torch.arange(3)[None, None, None, :] * 2) % 251).to(torch.uint8)
            ok = ok and full_img.dtype == torch.uint8 and bool(torch.equal(full_img, ref_img))
            at_root = sharding.gather_rows(img.contiguous(), n, root=0)  # configs[4]'s default: the images end up on rank 0 only
            ok = ok and ((at_root is None) if rank != 0 else bool(torch.equal(at_root, ref_img)))
            with open(os.path.join(out_dir, f"w8_r{rank}_n{n}.txt"), "w") as f:
                f.write("ok" if ok else "MISMATCH")
        dist.barrier()
    finally:
        dist.destroy_process_group()  # clean teardown at world 8


def test_gather_rows_world_of_eight_even_ragged_and_tiny(tmp_path):
    """BASELINE configs[3] shape: 2048 rows over 8 ranks (256 each) -- and 2048 + 3 (ragged), 11 and 5 (fewer rows than ranks: three
    ranks hold nothing). Landmark rows (int32) and uint8 images; every rank must end up with the whole batch in row order."""
    n_list, world = [2048, 2051, 11, 5], 8
    mp.spawn(_world8_worker, args=(world, _free_port(), n_list, str(tmp_path)), nprocs=world, join=True)
    for n in n_list:
        for r in range(world):
            assert (tmp_path / f"w8_r{r}_n{n}.txt").read_text() == "ok", (n, r)


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus N` is what the driver runs: with fewer devices than N it must say so (not print a usage
    error, not hang in a rendezvous). This container has no GPU at all, so N = 2 is refused the same way."""
    import subprocess
    import sys

    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible: the refusal path is not reachable here")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0
    assert "2 GPUs requested" in p.stderr and "visible" in p.stderr, p.stderr[-500:]
