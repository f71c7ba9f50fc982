"""The N>1 path on CPU: two processes, gloo, each renders its own super-tiles (with the oracle
standing in for the GPU kernel -- tests/ may use it as the checker), the slabs are gathered on
rank 0 and assembled by pbrt_amd.dist exactly as bench.py does with RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pbrt_amd
        from oracle import binding as ob
        from pbrt_amd import dist as pdist, scenes
        sd = scenes.cornell_scene(200, 136)  # 4 x 3 super-tiles, ragged edges, odd split over 2 ranks
        part, _ = ob.OracleScene(sd).render(max_depth=3, spp=(2, 1), seed=5, rank=rank, world_size=world, n_threads=2)
        # this rank's film -> its tile-major slab (the layout the HIP kernel writes)
        idx = pbrt_amd.slab_pixel_index(sd.xres, sd.yres, sd.crop, rank, world)
        slab = np.zeros((len(idx), 4), np.float32)
        slab[idx >= 0] = part.reshape(-1, 4)[idx[idx >= 0]]
        film = pdist.gather_film(torch.from_numpy(slab), sd.xres, sd.yres, sd.crop, rank, world)
        if rank == 0:
            np.save(out_path, film.numpy())
        else:
            assert film is None
    finally:
        dist.destroy_process_group()


def _worker_wide(rank, world, port, out_path):
    """a box filter radius other than 0.5: the exchange is one integer sum reduction (pbrt_amd.dist.reduce_accumulators)"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pbrt_amd
        from oracle import binding as ob
        from pbrt_amd import dist as pdist, scenes
        sd = scenes.cornell_scene(200, 136)
        acc, _ = ob.OracleScene(sd).render_acc((1.5, 1.0), max_depth=3, spp=(2, 1), seed=5, rank=rank, world_size=world, n_threads=2)
        total = pdist.reduce_accumulators(torch.from_numpy(acc.reshape(-1, 4)), rank, world)
        if rank == 0:
            np.save(out_path, pbrt_amd.film_from_acc(total.numpy().reshape(acc.shape)))
        else:
            assert total is None
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_reduce_of_a_wide_filter_film(tmp_path, oracle):
    out = str(tmp_path / "film.npy")
    mp.spawn(_worker_wide, args=(2, _free_port(), out), nprocs=2, join=True)
    from pbrt_amd import dist as pdist, scenes
    sd = scenes.cornell_scene(200, 136)
    want, _ = oracle.OracleScene(sd).render(max_depth=3, spp=(2, 1), seed=5, filter_width=(1.5, 1.0))
    got = np.load(out)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert pdist.is_wide_filter((1.5, 1.0)) and not pdist.is_wide_filter((0.5, 0.0)) and not pdist.is_wide_filter(None)


@pytest.mark.timeout(300)
def test_two_rank_gather_assembles_the_single_rank_film(tmp_path, oracle):
    out = str(tmp_path / "film.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    from pbrt_amd import scenes
    sd = scenes.cornell_scene(200, 136)
    want, _ = oracle.OracleScene(sd).render(max_depth=3, spp=(2, 1), seed=5)
    got = np.load(out)
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_assemble_single_rank_and_padding():
    sys.path.insert(0, ROOT)
    import pbrt_amd
    from pbrt_amd import dist as pdist
    xres, yres, crop = 100, 70, (0.0, 1.0, 0.0, 1.0)
    world = 3
    film = np.arange(xres * yres * 4, dtype=np.float32).reshape(yres, xres, 4)
    slabs = []
    n = pdist.max_slab_slots(xres, yres, crop, world)
    for r in range(world):
        idx = pbrt_amd.slab_pixel_index(xres, yres, crop, r, world)
        s = np.full((n, 4), -7, np.float32)  # padding garbage must be ignored
        s[: len(idx)][idx >= 0] = film.reshape(-1, 4)[idx[idx >= 0]]
        slabs.append(torch.from_numpy(s))
    got = pdist.assemble_film(slabs, xres, yres, crop, world).numpy()
    assert np.array_equal(got, film)


def _worker_stats(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pbrt_amd import dist as pdist
        per_rank = pdist.rank_kernel_stats([10.0 + rank, 12.0 + 3 * rank], 1000 * (rank + 1), world)
        if rank == 0:
            import json
            pr, exchange = pdist.step_breakdown(per_rank, 20.0)
            json.dump({"per_rank_kernel_ms": pr, "exchange_ms": exchange}, open(out_path, "w"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_bench_line_carries_every_ranks_kernel_time(tmp_path):
    """bench.py in ranks mode gathers EVERY rank's kernel time and sample count (pbrt_amd.dist.rank_kernel_stats, the
    function bench.py calls) and reports exchange_ms = ms_per_step - the slowest rank's kernel -- the keys --single-process prints too."""
    import json
    out = str(tmp_path / "stats.json")
    mp.spawn(_worker_stats, args=(2, _free_port(), out), nprocs=2, join=True)
    got = json.load(open(out))
    pr = got["per_rank_kernel_ms"]
    assert pr["mean_per_rank"] == [11.0, 13.0] and pr["max_step_per_rank"] == [12.0, 15.0] and pr["samples_per_rank"] == [1000, 2000]
    assert pr["max"] == 13.0 and pr["min"] == 11.0 and abs(pr["imbalance"] - 13.0 / 12.0) < 1e-12
    assert abs(got["exchange_ms"] - 7.0) < 1e-12
