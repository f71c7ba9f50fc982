"""Multi-GPU film assembly: one process per GPU, 64x64 super-tiles dealt round-robin
(super-tile t belongs to rank t % world_size), scene replicated, no communication while
rendering, and ONE gather of the per-rank slabs to rank 0 (RCCL over xGMI when the process group
backend is "nccl"; "gloo" in the CPU tests).  The box filter's radius 0.5 (reference
src/filters/box.rs:57-61) keeps every sample inside its own pixel, so tiles never overlap and the
assembly is a pure scatter -- the multi-process form of Film::merge_film_tile
(src/core/film.rs:313-326).  Any other radius (film.rs:264-273: tiles overlap) is accumulated in fixed
point and the exchange is one integer sum reduction (reduce_accumulators).

torch is used for device buffers, streams and torch.distributed only.
"""
import numpy as np

from . import api


_index_cache = {}


def slab_index_tensors(xres, yres, crop, world_size, device="cpu"):
    """Per rank: (int64 index tensor of the slab's pixels inside the film, boolean mask of the slab slots that are
    inside) -- built once per (film geometry, world size, device) and kept: the geometry of a job never changes, and
    ADVICE r01 found the per-step rebuild + host-to-device copy inside bench.py's timed region."""
    import torch
    key = (xres, yres, tuple(float(c) for c in crop), world_size, str(device))
    if key not in _index_cache:
        per_rank = []
        for r in range(world_size):
            idx = torch.from_numpy(api.slab_pixel_index(xres, yres, crop, r, world_size))
            keep = idx >= 0
            per_rank.append((idx[keep].to(device), keep.to(device)))
        _index_cache[key] = per_rank
    return _index_cache[key]


def max_slab_slots(xres, yres, crop, world_size):
    """slab slots of the rank that owns the most super-tiles (rank 0): the gather's common size"""
    import ctypes as C
    from ._lib import lib
    return int(lib().pbrt_hip_slab_floats(xres, yres, (C.c_float * 4)(*crop), 0, world_size)) // 4


def assemble_film(slabs, xres, yres, crop, world_size, scene=None):
    """Scatter the gathered slabs (list indexed by rank, each [slots, 4], possibly padded at the end) into the
    row-major film [h, w, 4].  Device slabs of a `scene` go through the library's scatter kernel
    (pbrt_hip_film_assemble_device); host slabs (the gloo tests) through cached index tensors."""
    import torch
    b = api.film_cropped_bounds(xres, yres, crop)
    w, h = max(b[2] - b[0], 0), max(b[3] - b[1], 0)
    film = torch.zeros(h * w, 4, dtype=torch.float32, device=slabs[0].device)
    if scene is not None and film.is_cuda:
        stream = torch.cuda.current_stream().cuda_stream
        for r in range(world_size):
            scene.film_assemble_device(slabs[r].data_ptr(), r, world_size, film.data_ptr(), stream)
        return film.view(h, w, 4)
    for r, (idx, keep) in enumerate(slab_index_tensors(xres, yres, crop, world_size, film.device)):
        film[idx] = slabs[r][: keep.numel()][keep]
    return film.view(h, w, 4)


def gather_film(local_slab, xres, yres, crop, rank, world_size, group=None, scene=None):
    """Gather every rank's slab on rank 0 and assemble the film there (None elsewhere).
    `local_slab`: [slots_of_this_rank, 4] float32 tensor on the rank's device."""
    import torch
    import torch.distributed as dist
    n = max_slab_slots(xres, yres, crop, world_size)
    dev = local_slab.device
    if world_size > 1 and dist.get_backend(group) == "gloo":
        dev = torch.device("cpu")  # gloo cannot gather device tensors: stage through the host
    if local_slab.shape[0] == n and local_slab.device == dev:
        send = local_slab  # (rank 0's slab has the common size already)
    else:
        send = torch.zeros(n, 4, dtype=torch.float32, device=dev)
        send[: local_slab.shape[0]] = local_slab.to(dev)
    if world_size == 1 and not (dist.is_available() and dist.is_initialized()):
        return assemble_film([send], xres, yres, crop, 1, scene)
    # (a one-rank process group still gathers: under torch.distributed.run the RCCL path runs on single-GPU boxes too)
    recv = [torch.empty_like(send) for _ in range(world_size)] if rank == 0 else None
    dist.gather(send, recv, dst=0, group=group)
    if rank != 0:
        return None
    return assemble_film(recv, xres, yres, crop, world_size, scene)


def is_wide_filter(filter_width):
    """a box filter radius other than the default 0.5 (0 stands for the default): DESIGN.md 3.11"""
    fw = filter_width or (0.0, 0.0)
    return any(float(v) not in (0.0, 0.5) for v in fw)


def reduce_accumulators(local_acc, rank, world_size, group=None):
    """A box filter radius other than 0.5: every rank holds fixed-point accumulators of the WHOLE cropped window
    ([h * w, 4] int64; its own samples, which reach into its neighbours' tiles) -- integer sums are exact in any order, so
    the film is ONE sum reduction to rank 0 (RCCL ncclReduce over xGMI under "nccl") instead of the gather.  Returns the
    summed accumulators on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist
    if world_size == 1:
        return local_acc  # one rank: its accumulators ARE the sum (an initialised one-rank group has nothing to add)
    t = local_acc
    if dist.get_backend(group) == "gloo" and t.is_cuda:
        t = t.cpu()  # gloo cannot reduce device tensors: stage through the host
    dist.reduce(t, dst=0, op=dist.ReduceOp.SUM, group=group)
    return t if rank == 0 else None


def render_sharded(scene, rank, world_size, group=None, **render_kw):
    """Render this rank's super-tiles on its GPU (asynchronously on torch's current stream),
    gather, and return (film on rank 0 or None, stats of the local kernel)."""
    import torch
    sd = scene.sd
    if is_wide_filter(render_kw.get("filter_width")):
        w, h = sd.crop_size()
        acc = torch.empty(max(h * w, 1), 4, dtype=torch.int64, device="cuda")  # (render_device zeroes it)
        stream = torch.cuda.current_stream().cuda_stream
        scene.render_device(acc.data_ptr(), stream, rank=rank, world_size=world_size, **render_kw)
        total = reduce_accumulators(acc, rank, world_size, group)
        film = None
        if total is not None:
            total = total.to("cuda")
            film = torch.empty(h, w, 4, dtype=torch.float32, device="cuda")
            scene.film_from_acc_device(total.data_ptr(), film.data_ptr(), stream)
        return film, scene.render_wait()
    n_floats = scene.slab_floats(rank, world_size)
    slab = torch.empty(max(n_floats // 4, 1), 4, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    scene.render_device(slab.data_ptr(), stream, rank=rank, world_size=world_size, **render_kw)
    film = gather_film(slab[: n_floats // 4], sd.xres, sd.yres, sd.crop, rank, world_size, group, scene)
    stats = scene.render_wait()
    return film, stats


def rank_kernel_stats(kernel_ms, samples, world_size, device="cpu", group=None):
    """Every rank's kernel time over its timed steps (mean and slowest step) and sample count, on EVERY rank: what makes an N-GPU bench
    record diagnosable -- imbalance between the ranks' shares, and (step_breakdown) what a step costs beyond its slowest kernel.
    `kernel_ms`: this rank's per-step kernel times.  One all_gather of three doubles (RCCL under "nccl", host tensors under gloo)."""
    import torch
    import torch.distributed as dist
    mine = torch.tensor([sum(kernel_ms) / max(len(kernel_ms), 1), max(kernel_ms) if kernel_ms else 0.0, float(samples)], dtype=torch.float64, device=device)
    if world_size > 1 or (dist.is_available() and dist.is_initialized()):
        got = [torch.zeros_like(mine) for _ in range(world_size)]
        dist.all_gather(got, mine, group=group)
    else:
        got = [mine]
    return {"mean_per_rank": [round(float(g[0]), 3) for g in got], "max_step_per_rank": [round(float(g[1]), 3) for g in got],
            "samples_per_rank": [int(g[2]) for g in got]}


def step_breakdown(per_rank, ms_per_step):
    """-> (per_rank with max / min / imbalance added, exchange_ms): exchange_ms = a step's wall time beyond its SLOWEST rank's kernel
    (the gather or reduce, the assembly, launch and host overhead); imbalance = slowest / mean of the ranks' kernel times (1.0 = even)."""
    m = per_rank["mean_per_rank"]
    out = dict(per_rank, max=max(m), min=min(m), imbalance=(max(m) / (sum(m) / len(m)) if sum(m) > 0 else None))
    return out, ms_per_step - max(m)
