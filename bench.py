#!/usr/bin/env python3
"""bench.py -- Msamples/s of the BVH-traversal + path-tracing hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c1|c4|big] [--single-process]

A "step" is one pass of the hot path over the whole workload: every rank renders its 64x64
super-tiles (one launch of render_kernel + the merge of the partial sums) and the film is gathered on
rank 0 (one RCCL gather for N>1).  Default workload = BASELINE.json configs[3], the one the
metric/target is quoted on: 1M random triangles, 2048x2048, 512 spp (32x16 strata), path integrator
maxdepth 8; it fits one GPU, so N=1 runs all of it and N>1 shards the same frame (strong scaling).
Scene build / upload and image writing are outside the timed region; inputs are resident in HBM when
timing starts.  N>1: one process per GPU under torch.distributed.run (the driver's contract), or
`--single-process`: all N GPUs from this one process through pbrt_hip_multi_* (one host thread per GPU,
ncclGather inside the library).

`python bench.py --gpus N` with N > 1 and NO torch.distributed environment (no WORLD_SIZE) takes the in-library path by
itself -- decided from argv and the environment before anything touches a GPU, no re-exec -- so the command produces a line
however it is launched.

Rank 0 prints ONE JSON line.  Every figure of `roofline` (dominant kernel = render_kernel) follows from three inputs --
P = profiles/pmc_<workload>.json (the committed rocprofv3 --pmc passes of THIS kernel's ISA and builder), R = rays_per_launch
(counted live by the exact-counter instantiation) and T = kernel_ms (HIP events on the launch stream, average of the timed
steps) -- by one formula each:
  busy              = P.valu_issue_quadcycles_per_ray x R / T                     [G vector-issue quad-cycles/s: `achieved_busy`]
  peak              = CUs x 4 SIMDs x 2.4 GHz / 4                                  (614.4 G for 256 CUs)
  frac_busy         = busy / peak                 the share of all SIMD quad-cycles in which a vector instruction issued: how often
                                                  the issue port was OCCUPIED, whatever the lanes did
  lane_utilisation  = P.lane_utilisation          active lanes per issued vector instruction / 64
  achieved          = busy x lane_utilisation     [G lane-weighted vector-issue quad-cycles/s]
  frac              = achieved / peak = frac_busy x lane_utilisation   THE roofline fraction (round 6): the share of
                                                  the chip's lane-issue slots that did work for a ray.  (Until round 5 `frac` was the busy
                                                  fraction, half of it on idle lanes; `frac_useful` is kept as an alias of `frac`.)
  frac_at_measured_clock, frac_busy_at_measured_clock = the same against CUs x 4 x clock_ghz_measured / 4
  l1.frac           = P.l1_accesses_per_ray x R / T / 693.6 G   (16-byte gathers against tools/ubench/gather_wide.hip's roof)
  l2_miss.requests_per_s = (hbm.kernel_fetches_per_ray + hbm.kernel_tris_per_ray) x R / T x (1 - P.l2_hit_rate)
                                                  64-byte records that miss the XCD's L2, against l2_miss.peak = 58.1 G records/s
                                                  (the same microbenchmark on a 112 MB table: past L2, inside the Infinity Cache)
  hbm.frac          = (32 x nodes_visited + 48 x tris_tested of the CANONICAL walk + 28 + 16/spp per sample) x samples / T / 8 TB/s
                                                  SURVEY 8(d)'s contract figure (exceeds 1 when hbm.cache_resident: the hot working
                                                  set -- quad nodes + triangle records -- then fits the 256 MiB Infinity Cache and
                                                  the production walk moves fewer bytes, kernel_*)
  traffic           = P.FETCH_SIZE_KB_per_launch x 1024 x P.fetch_size_calibration.factor + P.WRITE_SIZE_KB_per_launch x 1024
                                                  memory-side bytes per launch; the factor is MEASURED for this access shape (64-byte
                                                  records gathered at random: tools/fetch_size_calibration.sh, profiles/r06_fetch_size_calibration.txt)
  hbm_counter_frac  = traffic / P.avg_ms / 8 TB/s                                  north_star's "rocprof achieved HBM GB/s"
The per-ray counters of P belong to ONE tree: beside the kernel's machine code (P.kernel_isa_id) and the builder's name the profile
records the production walk's fetches and triangle tests per ray on that tree (P.tree), and everything that rests on P is withheld when
the live counting pass differs by more than 0.5 % (a builder edit changes the tree and leaves the render kernel's ISA alone: ADVICE r05).
`cpu_baseline` (N=1): the CPU oracle timed on this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured stream)
CLOCK_GHZ = 2.4         # max shader clock (same guide); the clock a kernel holds is lower, so frac is conservative
SIMDS_PER_CU = 4
# what the CUs' vector L1s deliver to random 16-byte-per-lane gathers of 64-byte records that hit in L2: 173.4 G records/s x 4 accesses,
# measured with tools/ubench/gather_wide.hip (profiles/r03end_gather_wide_48B.txt); 1.13 accesses per CU and clock at 2.4 GHz
L1_GATHER_ROOF_G_ACCESSES = 173.4 * 4
# 64-byte records gathered at random from a 112 MB table (past the 4 MiB L2s, inside the Infinity Cache), same file
L2_MISS_ROOF_G_RECORDS = 58.08
INFINITY_CACHE_BYTES = 256 * 2 ** 20


def launch_mode(gpus, single_process, env):
    """How `--gpus N` is driven, decided from the arguments and the environment alone (before any GPU call; nothing is re-executed):
    -> ("ranks", world) one process per GPU under torch.distributed.run (the driver's contract for N > 1), ("in-process", 1) all N
    GPUs from this one process through pbrt_hip_multi_* (RCCL inside the library) -- also what a bare `python bench.py --gpus N`
    gets --, ("single", 1) one GPU; or raises SystemExit for a contradictory launch."""
    world = int(env.get("WORLD_SIZE", "1"))
    if gpus < 1:
        raise SystemExit(f"--gpus {gpus}")
    if single_process and gpus > 1:
        if world > 1:  # (every rank of a launcher would drive all the GPUs)
            raise SystemExit(f"--single-process under a launcher with WORLD_SIZE={world}: start it as ONE process")
        return "in-process", 1
    if world == gpus:
        return ("ranks", world) if ("RANK" in env or world > 1) else ("single", 1)
    if world == 1 and "RANK" not in env and gpus > 1:
        return "in-process", 1  # a bare `python bench.py --gpus N`: no launcher, so the library drives the N GPUs itself
    raise SystemExit(f"--gpus {gpus} but WORLD_SIZE={world}")

WORKLOADS = {
    # name: (scene factory args, integrator, maxdepth, (spp_x, spp_y), description)
    "c3": (("mesh", 1_000_000, 2048), 0, 8, (32, 16), "C3: 1M random triangles in a box, 2048x2048, 512 spp, path maxdepth 8"),
    "c2": (("mesh", 100_000, 1024), 0, 8, (16, 16), "C2: 100k random triangles in a box, 1024x1024, 256 spp, path maxdepth 8"),
    "c1": (("sphere", 0, 1024), 1, 5, (8, 8), "C1: analytic sphere + point light, 1024x1024, 64 spp, direct lighting"),
    "c4": (("cornell", 0, 4096), 0, 16, (64, 64), "C4: Cornell-style box, 4096x4096, 4096 spp, path maxdepth 16"),
    # not a BASELINE config: the out-of-cache regime (0.9 GB of nodes + triangles against the 256 MiB Infinity Cache)
    "big": (("mesh", 12_000_000, 2048), 0, 8, (8, 8), "BIG (out of cache): 12M random triangles in a box, 2048x2048, 64 spp, path maxdepth 8"),
}


def make_scene_data(kind, n, res, crop=(0.0, 1.0, 0.0, 1.0)):
    from pbrt_amd import scenes
    if kind == "mesh":
        return scenes.random_mesh_scene(n, res, res, crop=crop)
    if kind == "sphere":
        return scenes.sphere_scene(res, res, crop=crop)
    return scenes.cornell_scene(res, res, crop=crop)


def algorithmic_bytes_per_sample(st, samples, spp):
    """sum over rays of (32 B per node visited + 48 B per triangle tested)
    + 28 B camera ray + 16/spp B film store, per camera sample."""
    return (32.0 * st["nodes_visited"] + 48.0 * st["tris_tested"]) / samples + 28.0 + 16.0 / spp


def usable_cores():
    """Host cores this process may actually use: the affinity mask capped by the cgroup CPU quota
    (the GPU box shows 256 CPUs but its cgroup grants cpu.max = 16 CPUs)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(kind, n, res, integrator, depth, spp, target_s=15.0, gpu_film=None):
    """The CPU oracle ("port"), rebuilt -march=native on this box, all host cores, on a bounded
    sample of the same workload: every `world`-th 64x64 super-tile of the frame (spread over the
    whole image, at least 8 work tiles per host thread) at a reduced sample count chosen by a
    calibration pass so that the timed pass does about `target_s` seconds of CPU work
    (throughput is spp-independent: every sample is an independent path)."""
    from oracle import binding as ob
    ob.build(native=True)
    sd = make_scene_data(kind, n, res)
    sc = ob.OracleScene(sd, native=True)
    cores = usable_cores()
    n_super = ((res + 63) // 64) ** 2
    want_super = min(n_super, max(1, (8 * cores + 15) // 16))
    world = max(1, n_super // want_super)
    if world > 1 and world % 2 == 0:
        world += 1  # odd stride: the sample does not fall on one column of super-tiles
    kw = dict(integrator=integrator, max_depth=depth, seed=0, rank=0, world_size=world, n_threads=cores)
    film, st = sc.render(spp=(1, 1), **kw)  # calibration
    pixels = int((film[..., 3] > 0).sum())
    rate = pixels / max(st["seconds"], 1e-6)
    want_spp = max(1, min(spp[0] * spp[1], int(target_s * rate / pixels)))
    sx = 1
    while sx * sx * 4 <= want_spp and sx * 2 <= spp[0]:
        sx *= 2
    sy = max(1, min(spp[1], want_spp // sx))
    film, st = sc.render(spp=(sx, sy), **kw)
    samples = pixels * sx * sy
    rays = st["camera_rays"] + st["bounce_rays"] + st["shadow_rays"]
    # the metric's quality half ("PSNR vs CPU reference"): a 48x48 window of the frame at the frame's FULL sample
    # count on the oracle against the same pixels of the film the GPU has just rendered (linear RGB clamped to [0,1])
    parity = None
    if gpu_film is not None:
        import numpy as np
        import pbrt_amd
        w = 48
        x0, y0 = (res // 2 // w) * w, (res // 3 // w) * w
        crop = (x0 / res, (x0 + w) / res, y0 / res, (y0 + w) / res)
        ref, pst = ob.OracleScene(make_scene_data(kind, n, res, crop=crop), native=True).render(
            integrator=integrator, max_depth=depth, seed=0, spp=spp, n_threads=cores)
        got = gpu_film[y0:y0 + w, x0:x0 + w]
        a = np.clip(pbrt_amd.film_to_rgb(np.ascontiguousarray(got)).astype(np.float64), 0, 1)
        b = np.clip(pbrt_amd.film_to_rgb(np.ascontiguousarray(ref)).astype(np.float64), 0, 1)
        mse = float(((a - b) ** 2).mean())
        parity = {"window": [x0, y0, w, w], "spp": spp[0] * spp[1], "bit_equal": bool((got.view(np.uint32) == ref.view(np.uint32)).all()),
                  "psnr_db": None if mse == 0 else 10 * np.log10(1.0 / mse), "psnr_note": "null = identical images (infinite PSNR)",
                  "oracle_seconds": round(pst["seconds"], 2)}
    return {
        "parity_window": parity,
        "value": samples / st["seconds"] / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
        "sample": f"CPU oracle (ours; the reference has no renderer), {samples} samples = super-tiles t%{world}==0 "
                  f"({pixels} pixels) of the {res}x{res} frame at {sx}x{sy} spp, {st['seconds']:.1f} s on {cores} threads",
        "mrays_per_s": rays / st["seconds"] / 1e6,
        "bytes_per_sample": algorithmic_bytes_per_sample(st, samples, sx * sy),
    }


class ClockSampler:
    """The shader clock the GPU holds while the timed region runs, sampled from a host thread: the `*` line of the device's
    /sys/class/drm/card*/device/pp_dpm_sclk (found through its PCI address).  Reading sysfs starts no process and costs
    microseconds; where it cannot be read the clock is simply not reported (an earlier version fell back to spawning
    `rocm-smi` every 0.25 s INSIDE the timed region: 80 ms per sample, and a child exec'd from a GPU-initialised process).
    roofline.frac prices the kernel against the NOMINAL 2.4 GHz, so the same kernel reads 0.84 on a box that holds 2.32 GHz and 0.86
    on one that holds 2.38; frac_at_measured_clock takes the droop out."""

    def __init__(self, device_index, period_s=0.25):
        import glob
        import threading
        self.device_index, self.period_s, self.mhz, self.sysfs = device_index, period_s, [], None
        try:
            import torch
            p = torch.cuda.get_device_properties(device_index)
            addr = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
            for dev in glob.glob("/sys/class/drm/card*/device"):
                if os.path.basename(os.path.realpath(dev)) == addr and os.access(os.path.join(dev, "pp_dpm_sclk"), os.R_OK):
                    self.sysfs = os.path.join(dev, "pp_dpm_sclk")
        except Exception:  # no such attributes, no sysfs: no clock in the line
            pass
        self.source = f"{self.sysfs} (the level marked *), sampled from a host thread during the timed steps" if self.sysfs else None
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _sample(self):
        import re
        if not self.sysfs:
            return None
        try:
            for line in open(self.sysfs):
                if "*" in line:
                    m = re.search(r"(\d+)\s*Mhz", line, re.I)
                    return int(m.group(1)) if m else None
        except OSError:
            pass
        return None

    def _run(self):
        while not self._stop.is_set():
            v = self._sample()
            if v is not None:
                self.mhz.append(v)
            self._stop.wait(self.period_s)

    def start(self):
        if self.sysfs:
            self._thread.start()

    def stop(self):
        self._stop.set()
        if self.sysfs:
            self._thread.join(timeout=10)
        busy = sorted(v for v in self.mhz if v >= 1000)  # (samples between launches can catch a sleeping clock)
        if not busy:
            return None
        return {"ghz_median": busy[len(busy) // 2] / 1e3, "ghz_min": busy[0] / 1e3, "ghz_max": busy[-1] / 1e3, "samples": len(busy), "source": self.source}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, nargs=2, default=None, help="override strata (diagnostics; invalid as a headline)")
    ap.add_argument("--builder", default="gpu", choices=["host", "gpu", "gpu-plain", "host-optimized"],
                    help="accelerator builder: the device builder (the product's default: binned SAH + parallel re-insertion, about 0.13 s "
                         "for 1M triangles), the same without the re-insertion passes (gpu-plain: A-B runs), the host's binned SAH (one core, "
                         "about a second for 1M triangles) or the host's tree optimised by the same pass run on one host core (seconds for 1M "
                         "triangles); same film either way")
    ap.add_argument("--sampler", default="stratified", choices=["stratified", "sobol"])
    ap.add_argument("--filter", type=float, nargs=2, default=None, metavar=("XW", "YW"),
                    help="box filter radii (diagnostics; the BASELINE configs use the default 0.5): other radii take the fixed-point film "
                         "path, whose multi-GPU exchange is a sum reduction instead of the gather")
    ap.add_argument("--single-process", action="store_true",
                    help="N GPUs from ONE process through pbrt_hip_multi_* (ncclGather inside the library) instead of one rank per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-counters", action="store_true", help="skip the untimed counting passes (roofline.achieved = null)")
    ap.add_argument("--profiles", default=os.path.join(ROOT, "profiles"), help="directory of pmc_<workload>.json (tests point it at a copy)")
    args = ap.parse_args()

    mode, world = launch_mode(args.gpus, args.single_process, os.environ)  # (before torch / HIP are even imported)
    in_process = mode == "in-process"
    # The identity of the profiled kernel in the library on disk -- file parsing, done HERE, before torch or HIP are imported: a profile
    # that carries the kernel's mangled symbol needs no demangler at all, an older one runs c++filt now, while no GPU is initialised
    # (no child process from a GPU-initialised process; under rocprofv3 every child inherits the profiler's preload).
    from pbrt_amd import isa_id
    from pbrt_amd.build import LIB_PATH
    pmc_path = os.path.join(args.profiles, f"pmc_{args.workload}.json")
    pmc = json.load(open(pmc_path)) if os.path.exists(pmc_path) else None
    kernel_id_loaded = isa_id.kernel_id(LIB_PATH, pmc["kernel"], pmc.get("kernel_symbol")) if pmc is not None and pmc.get("kernel") else None

    import torch
    import torch.distributed as dist

    import pbrt_amd
    from pbrt_amd import dist as pdist

    rank = int(os.environ.get("RANK", "0")) if mode == "ranks" else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if mode == "ranks" else 0
    if not torch.cuda.is_available() or pbrt_amd.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: pbrt_amd has no CPU fallback")
    # one rank per GPU; PBRT_DIST_BACKEND=gloo lets several ranks share one GPU (test boxes with a single device)
    backend = os.environ.get("PBRT_DIST_BACKEND", "nccl")
    device_index = local_rank % torch.cuda.device_count()
    if backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit(f"{world} ranks but {torch.cuda.device_count()} GPU(s): RCCL needs one GPU per rank")
    torch.cuda.set_device(device_index)
    # under torch.distributed.run the process group is ALWAYS initialised, also for one rank: the RCCL path
    # (init with device_id, barrier, all_reduce, gather) then runs on single-GPU boxes as well
    use_pg = mode == "ranks"
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    (kind, n, res), integrator, depth, spp, descr = WORKLOADS[args.workload]
    if args.spp:
        spp = tuple(args.spp)
    t0 = time.time()
    sd = make_scene_data(kind, n, res)
    if in_process:
        # (PBRT_HIP_MULTI_LOOPBACK, a debug knob of multi_gpu.cpp: more ranks than devices, the exchange made of device-to-device copies --
        # the N-rank code path on a one-GPU box; its line says so in "dist" and is no scaling measurement)
        loopback = os.environ.get("PBRT_HIP_DEBUG_KNOBS", "0") not in ("", "0") and os.environ.get("PBRT_HIP_MULTI_LOOPBACK", "0") not in ("", "0")
        if pbrt_amd.device_count() < args.gpus and not loopback:
            raise SystemExit(f"--gpus {args.gpus} but {pbrt_amd.device_count()} HIP device(s) visible")
        scene = pbrt_amd.MultiScene(sd, args.gpus, builder=args.builder)
        args.no_counters = True
        info = {"n_nodes": None, "depth": None, "device_bytes": None}
    else:
        scene = pbrt_amd.Scene(sd, device=device_index, builder=args.builder)
        info = scene.info()
    build_s = time.time() - t0  # scene data + accelerator build + upload (outside the timed region)
    kw = dict(integrator=integrator, max_depth=depth, spp=spp, seed=0, sampler=args.sampler)
    wide = pdist.is_wide_filter(args.filter)
    if wide:
        kw["filter_width"] = tuple(args.filter)
        args.no_counters = True  # (the counter flags need the default filter)

    def barrier():
        torch.cuda.synchronize()
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    per_gpu_ms = []  # --single-process: every GPU's kernel time of every timed step (the first real 8-GPU run shows its imbalance)

    def step():
        if in_process:
            film, sts = scene.render(host_film=False, **kw)
            per_gpu_ms.append([s["kernel_ms"] for s in sts])
            return None, {"kernel_ms": max(s["kernel_ms"] for s in sts), "samples": sum(s["samples"] for s in sts)}
        return pdist.render_sharded(scene, rank, world, **kw)

    for _ in range(args.warmup):
        step()
    per_gpu_ms.clear()
    clocks = ClockSampler(device_index) if rank == 0 else None
    if clocks:
        clocks.start()  # (before the barrier: nothing of the sampler's start-up falls into the timed region)
    barrier()
    t_start = time.perf_counter()
    kernel_ms, local_samples = [], 0
    film = None
    for _ in range(args.steps):
        film, st = step()
        kernel_ms.append(st["kernel_ms"])
        local_samples = st["samples"]
    barrier()
    elapsed = time.perf_counter() - t_start
    clock = clocks.stop() if clocks else None
    # what the process group really was (a SCALE record must show that RCCL saw N ranks on N GPUs)
    dist_info = {"backend": "none (one process, no process group)", "world_size": 1, "ranks_on_distinct_gpus": 1}
    if in_process:
        dist_info = {"backend": "rccl inside the library (pbrt_hip_multi_*: ncclCommInitAll, one group call per frame)", "world_size": 1,
                     "gpus_in_process": scene.n_gpus, "ranks_on_distinct_gpus": min(scene.n_gpus, pbrt_amd.device_count())}
        if dist_info["ranks_on_distinct_gpus"] < scene.n_gpus:
            dist_info["backend"] = "loopback inside the library (PBRT_HIP_MULTI_LOOPBACK: ranks share devices, device-to-device copies instead of RCCL) -- NOT a scaling measurement"
    if use_pg:
        dist_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks_on_distinct_gpus": None,
                     "nccl_is_rccl": bool(getattr(torch.version, "hip", None))}
        try:  # (reporting only: whatever goes wrong here must not cost the bench line)
            ids = [None] * world
            props = torch.cuda.get_device_properties(device_index)
            dist.all_gather_object(ids, (os.uname().nodename, getattr(props, "uuid", None) and str(props.uuid), getattr(props, "pci_bus_id", None), device_index))
            dist_info["ranks_on_distinct_gpus"] = len(set(ids))
        except Exception as e:  # noqa: BLE001
            dist_info["ranks_on_distinct_gpus_error"] = repr(e)[:200]
    per_rank = None  # every rank's kernel time (mean of its timed steps) and sample count: what makes an N-GPU record diagnosable
    if use_pg:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        # (in ranks mode the line used to carry rank 0's kernel time only -- imbalance between the ranks' shares and the
        # cost of the gather would have been invisible in the first real 8-GPU record)
        per_rank = pdist.rank_kernel_stats(kernel_ms, local_samples, world, "cuda" if backend == "nccl" else "cpu")
    total_samples = res * res * spp[0] * spp[1]
    value = total_samples * args.steps / elapsed / 1e6
    if in_process:  # the host copy of the film for the checks below (outside the timed region)
        film_np, _ = scene.render(host_film=True, **kw)
        film = torch.from_numpy(film_np)

    n_simd = torch.cuda.get_device_properties(device_index).multi_processor_count * SIMDS_PER_CU
    peak = n_simd * CLOCK_GHZ  # G issue-cycles/s
    roof = {"bound": "valu", "achieved": None, "peak": peak / 4.0, "unit": "G vector-issue quad-cycles/s", "frac": None, "traffic": None,
            "kernel": "render_kernel"}
    avg_kernel_ms = sum(kernel_ms) / max(len(kernel_ms), 1)
    roof["kernel_ms"] = avg_kernel_ms
    # A profile prices the KERNEL it was taken on and no other: the committed counters carry the id of that kernel's machine code
    # (pbrt_amd/isa_id.py: a hash of the production instantiation's .text bytes and kernel descriptor in the gfx950 code object);
    # when the loaded library's kernel of that name hashes differently -- a changed source line, flag or compiler that reaches
    # its ISA -- every figure that rests on the counters is withheld ("frac is silently stale").  Edits that do not
    # reach that kernel (the parser, another instantiation, the builder's host code) leave the profile valid (VERDICT r04 item 7;
    # the tree the rays walk is checked separately, by builder).  A profile without an ISA id falls back to the source hash.
    lib_id = pbrt_amd.build_id()
    if pmc is not None and pmc.get("kernel_isa_id"):
        pmc_stale = pmc["kernel_isa_id"] != kernel_id_loaded
    else:
        pmc_stale = pmc is not None and pmc.get("build_id") != lib_id
    if pmc is not None and not pmc_stale and pmc.get("builder", "gpu") != args.builder:
        # the committed per-ray counters (L1 accesses, issue quad-cycles) are those of ONE tree's walk; another builder's rays do a
        # different amount of both, so pricing them with these counters would misstate every fraction: withheld
        roof["profile_note"] = (f"{os.path.relpath(pmc_path, ROOT)} was taken with --builder {pmc.get('builder', 'gpu')}: roofline.frac / valu / l1 / traffic "
                                f"are withheld for --builder {args.builder} (a different tree: different work per ray)")
        pmc = None
    if pmc_stale:
        roof["stale_profile"] = (f"{os.path.relpath(pmc_path, ROOT)} was taken on kernel {pmc.get('kernel_isa_id') or 'of library build ' + str(pmc.get('build_id', '(none recorded)'))} "
                                 f"({isa_id.normalise(pmc.get('kernel', '?'))}), the loaded library's is {kernel_id_loaded or lib_id}: "
                                 "roofline.frac / valu / traffic withheld -- rerun tools/measure_round.sh")
        pmc = None
    elif pmc is not None:
        roof["profile_kernel"] = {"name": isa_id.normalise(pmc.get("kernel", "?")), "isa_id": kernel_id_loaded, "round": pmc.get("round")}
    if not args.no_counters:
        # untimed counting pass: exact nodes-visited / triangles-tested of THIS rank's share (the counting instantiation
        # of the same kernel; equal to the oracle's counters, tests/test_gpu_parity.py) and the rays of the launch
        slab = torch.empty(max(scene.render_buffer_bytes(rank=rank, world_size=world, **kw) // 16, 1), 4, device="cuda")
        scene.render_device(slab.data_ptr(), torch.cuda.current_stream().cuda_stream, rank=rank, world_size=world,
                            counters=True, **kw)
        cst = scene.render_wait()
        info = scene.info()  # (a device-built scene has its canonical tree -- node count, depth -- only now)
        bps = algorithmic_bytes_per_sample(cst, cst["samples"], spp[0] * spp[1])
        alg_bytes = bps * local_samples  # per launch of this rank's kernel
        ach = alg_bytes / (avg_kernel_ms * 1e-3) / 1e9
        rays = cst["camera_rays"] + cst["bounce_rays"] + cst["shadow_rays"]
        rays_per_s = rays / (avg_kernel_ms * 1e-3)
        roof.update({"rays_per_launch": rays, "rays_per_sample": rays / cst["samples"], "mrays_per_s": rays_per_s / 1e6})
        # what the production kernel itself fetches: 64-byte quad nodes + 48-byte triangles + the path-state records
        # (3 x 16 B read + written per ray, + 2 x 16 B of light-sample / partial-sum records now and then: ~112 B), from its own counters
        scene.render_device(slab.data_ptr(), torch.cuda.current_stream().cuda_stream, rank=rank, world_size=world,
                            counters="walk", **kw)
        wst = scene.render_wait()
        # (64-byte quad nodes, 64-byte triangle records: 16 x kTriStride since r02j)
        kbytes = (64.0 * wst["nodes_visited"] + 64.0 * wst["tris_tested"] + 112.0 * rays) / wst["samples"] + 28.0 + 128.0 / (spp[0] * spp[1])
        # the hot loop's working set: the production walk's quad nodes + the triangle records (64 B each).  device_bytes also counts
        # the uploaded vertex / index buffers and, after the counting pass above, the canonical arrays -- none of which a timed step reads
        hot_bytes = 64 * (info["quad_nodes"] + int(sd.idx.shape[0]))
        roof["hbm"] = {
            "note": "SURVEY 8(d) contract figure: algorithmic bytes of the CANONICAL binary-BVH walk (exact counters, equal to the "
                    "oracle's) / HIP-event kernel time, against the 8 TB/s HBM spec.  It is not a roof of this workload when "
                    f"cache_resident: the hot working set ({hot_bytes / 1e6:.0f} MB of quad nodes + triangle records) then sits in the 256 MiB "
                    "Infinity Cache and the production kernel walks a quantised 4-wide form of the tree that moves fewer bytes (kernel_*)",
            "cache_resident": bool(hot_bytes < INFINITY_CACHE_BYTES), "hot_working_set_bytes": hot_bytes,
            "achieved_gbps": ach, "peak_gbps": HBM_PEAK_GBS, "frac": ach / HBM_PEAK_GBS, "bytes_per_sample": bps,
            "algorithmic_bytes_per_launch": alg_bytes, "nodes_per_ray": cst["nodes_visited"] / rays, "tris_per_ray": cst["tris_tested"] / rays,
            "kernel_bytes_per_sample": kbytes, "kernel_fetches_per_ray": wst["nodes_visited"] / rays,
            "kernel_tris_per_ray": wst["tris_tested"] / rays, "kernel_gbps": kbytes * local_samples / (avg_kernel_ms * 1e-3) / 1e9,
        }
        tree = (pmc or {}).get("tree") or {}
        if pmc and tree.get("kernel_fetches_per_ray"):
            # the committed per-ray counters are those of ONE tree's walk.  The render kernel's ISA does not change when the BUILDER is
            # edited (bvh_gpu.hip, reinsert_core.hpp, the collapse), the tree does: the profile records the production walk's work per
            # ray on its tree, and a live walk that differs by more than 0.5 % is another tree
            live = (wst["nodes_visited"] / rays, wst["tris_tested"] / rays)
            rel = 0.0
            if tree.get("spp") in (None, spp[0] * spp[1]) and world == 1:  # (per-ray work is compared on the same frame: other sample counts, or one rank's share of the frame, trace other rays)
                rel = max(abs(live[0] / tree["kernel_fetches_per_ray"] - 1.0), abs(live[1] / tree["kernel_tris_per_ray"] - 1.0) if tree.get("kernel_tris_per_ray") else 0.0)
            if tree.get("quad_nodes") not in (None, info["quad_nodes"]):  # (the builders are deterministic: another node count IS another tree)
                rel = max(rel, 1.0)
            if rel > 0.005:
                roof["stale_profile"] = (f"{os.path.relpath(pmc_path, ROOT)} was taken on a tree that costs {tree['kernel_fetches_per_ray']:.3f} fetches + "
                                         f"{tree.get('kernel_tris_per_ray', float('nan')):.3f} triangle tests per ray; this build's tree costs {live[0]:.3f} + {live[1]:.3f} "
                                         "(the builder changed): roofline.frac / valu / l1 / traffic withheld -- rerun tools/measure_round.sh")
                roof.pop("profile_kernel", None)
                pmc = None
        if pmc and pmc.get("valu_issue_quadcycles_per_ray"):
            # VALU roof, MEASURED: the SQ counts the quad-cycles in which a SIMD issued a vector instruction (one, or two of
            # the full-rate class: profiles/r03c_issue_counter_calibration.txt); per ray from the committed profile OF THIS
            # BUILD (checked above), x the rays this launch traced, counted live, / the kernel's time; the roof is every
            # quad-cycle of every SIMD at the nominal clock.  No instruction census and no per-class prices are involved.
            # frac (round 6): the USEFUL share -- busy x the lanes that were active when the port issued; frac_busy says how often the
            # port was occupied, half of it on idle lanes
            lanes = pmc.get("lane_utilisation")
            roof["unit"] = "G lane-weighted vector-issue quad-cycles/s (quad-cycles in which a SIMD issued a vector instruction x active lanes / 64)"
            roof["peak"] = peak = n_simd * CLOCK_GHZ / 4.0
            roof["achieved_busy"] = pmc["valu_issue_quadcycles_per_ray"] * rays_per_s / 1e9
            roof["frac_busy"] = roof["achieved_busy"] / peak
            if lanes is not None:
                roof["lane_utilisation"] = lanes
                roof["achieved"] = roof["achieved_busy"] * lanes
                roof["frac"] = roof["frac_useful"] = roof["achieved"] / peak
            if clock:  # the same against the clock this box held during the timed steps
                roof["frac_busy_at_measured_clock"] = roof["achieved_busy"] / (n_simd * clock["ghz_median"] / 4.0)
                if lanes is not None:
                    roof["frac_at_measured_clock"] = roof["achieved"] / (n_simd * clock["ghz_median"] / 4.0)
            if pmc.get("l2_hit_rate") is not None:
                # the third roof the kernel sits under: 64-byte records (quad nodes, triangle records) that miss the XCD's L2
                recs = (wst["nodes_visited"] + wst["tris_tested"]) / rays * rays_per_s / 1e9
                roof["l2_miss"] = {"requests_per_s_G": recs * (1.0 - pmc["l2_hit_rate"]), "records_per_s_G": recs, "l2_hit_rate": pmc["l2_hit_rate"],
                                   "peak_G": L2_MISS_ROOF_G_RECORDS, "frac": recs * (1.0 - pmc["l2_hit_rate"]) / L2_MISS_ROOF_G_RECORDS,
                                   "peak_source": "tools/ubench/gather_wide.hip, 112 MB table (past L2, inside the Infinity Cache), 64-byte records: "
                                                  "58.08 G records/s (profiles/r03end_gather_wide_48B.txt)"}
                if pmc.get("FETCH_SIZE_KB_per_launch") and pmc.get("avg_ms"):
                    roof["l2_miss"]["fetch_size_64B_requests_per_s_G"] = pmc["FETCH_SIZE_KB_per_launch"] * 1024 / 64 / (pmc["avg_ms"] * 1e-3) / 1e9
            m = pmc.get("valu_issue_busy_measured")
            roof["valu"] = {k: pmc.get(k) for k in ("valu_issue_busy_measured", "valu_issue_quadcycles_per_ray", "valu_dual_issue_share_of_instructions",
                                                    "valu_instructions_per_ray", "lane_utilisation", "l2_hit_rate", "wave_wait_frac", "wave_issue_wait_frac", "wave_issuing_frac",
                                                    "clock_ghz_in_profile", "round", "build_id")}
            if pmc.get("l1_accesses_per_cu_clock") is not None:
                # the other shared resource, from the same profile: the CU's vector L1.  A lane's 16-byte
                # gather is one cache access; the address unit waits for the L1 `ta_stalled_by_l1_frac` of the cycles.
                roof["l1"] = {k: pmc.get(k) for k in ("l1_accesses_per_cu_clock", "l1_accesses_per_ray", "ta_stalled_by_l1_frac", "l1_tag_conflict_stall_frac")}
                roof["l1"]["achieved_G_accesses_per_s"] = pmc["l1_accesses_per_ray"] * rays_per_s / 1e9
                # the roof this is measured against: random 64-byte records gathered as 4 x dwordx4 per lane from an L2-resident
                # table, tools/ubench/gather_wide.hip on this chip (profiles/r03end_gather_wide_48B.txt: 173.4 G records/s x 4)
                roof["l1"]["peak_G_accesses_per_s"] = L1_GATHER_ROOF_G_ACCESSES
                roof["l1"]["frac"] = roof["l1"]["achieved_G_accesses_per_s"] / L1_GATHER_ROOF_G_ACCESSES
                roof["l1"]["peak_source"] = "tools/ubench/gather_wide.hip, 1-2 MB table, 64-byte records: 173.4 G records/s (profiles/r03end_gather_wide_48B.txt)"
            roof["valu"]["valu_busy_bracket"] = [m, m]  # a measurement, not a model: the r02 bracket [0.72, 1.07] is gone
            roof["valu"]["class_model_r02"] = {k: pmc.get(k) for k in ("valu_busy_frac_at_profile_clock", "valu_issue_cycles_per_ray", "mean_issue_cycles_per_instruction")}
            roof["valu"]["source"] = f"{os.path.relpath(pmc_path, ROOT)} (rocprofv3 --pmc passes of tools/measure_round.sh; counters calibrated in profiles/r03c_issue_counter_calibration.txt)"
    # HBM-side traffic cannot be read in-process: it comes from the separate rocprofv3 --pmc passes of this same
    # workload whose summary is committed under profiles/; null when there is none.
    # (per FRAME of the workload's own sample count: an overridden --spp gets it only from a profile that says it was taken at that count)
    if pmc and world == 1 and "traffic_bytes_raw" in pmc and (not args.spp or (pmc.get("tree") or {}).get("spp") == spp[0] * spp[1]):
        cal = pmc.get("fetch_size_calibration") or {}
        factor = cal.get("factor")
        if factor and pmc.get("FETCH_SIZE_KB_per_launch") is not None:
            roof["traffic"] = (pmc["FETCH_SIZE_KB_per_launch"] * factor + pmc.get("WRITE_SIZE_KB_per_launch", 0.0)) * 1024
            roof["fetch_size_calibration"] = cal
        else:
            roof["traffic"] = pmc["traffic_bytes_raw"]
        roof["traffic_bytes_raw"] = pmc["traffic_bytes_raw"]
        roof["hbm_counter_frac"] = roof["traffic"] / (pmc["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        roof["traffic_note"] = (f"(FETCH_SIZE x factor + WRITE_SIZE) x 1024 per frame from {os.path.relpath(pmc_path, ROOT)} "
                                f"(round {pmc.get('round', '?')} kernel, {pmc['avg_ms']:.0f} ms per frame); FETCH_SIZE counts L2-miss requests "
                                "incl. Infinity-Cache hits; factor = known bytes / FETCH_SIZE bytes MEASURED for 64-byte records gathered at random "
                                + (f"({cal.get('source', '?')})" if factor else "(no calibration in the profile: raw)")
                                + "; hbm_counter_frac = that traffic / that time / 8 TB/s")
    roof["clock_ghz_measured"] = clock["ghz_median"] if clock else None
    roof["clock"] = clock
    out = {
        "metric": "Msamples/sec (rays/sec) at 1/2/4/8 GPUs; PSNR vs CPU reference",
        "value": value, "unit": "Msamples/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": descr, "triangles": int(sd.idx.shape[0]), "resolution": [res, res], "spp": spp[0] * spp[1],
                   "maxdepth": depth, "sampler": args.sampler, "box_filter_radius": list(args.filter) if wide else [0.5, 0.5],
                   "sharding": (f"64x64 super-tiles round-robin over {args.gpus} GPU(s), one gather"
                                + (" (one process, ncclGather inside the library)" if in_process else " (one rank per GPU)")),
                   "bvh_nodes": info["n_nodes"], "bvh_depth": info["depth"], "scene_bytes": info["device_bytes"],
                   "scene_build_s": round(build_s, 2), "library_build_id": lib_id,
                   # build_ms: the builder's own time (device: HIP events); canonical_tree_host_build_ms: the host builder's time
                   # for the oracle's tree, which only the untimed counting pass uses
                   "accelerator": dict(scene.build_info(), builder=args.builder) if not in_process else {"builder": args.builder}},
        "roofline": roof,
        "dist": dist_info,
    }
    if in_process and per_gpu_ms:
        import statistics
        cols = list(zip(*per_gpu_ms))
        out["per_gpu_kernel_ms"] = {"mean_per_gpu": [round(statistics.mean(c), 3) for c in cols],
                                    "max": round(max(map(max, cols)), 3), "min": round(min(map(min, cols)), 3)}
        per_rank = {"mean_per_rank": out["per_gpu_kernel_ms"]["mean_per_gpu"], "max_step_per_rank": [round(max(c), 3) for c in cols]}
    if per_rank is not None:
        # the same keys however the N GPUs are driven: exchange_ms = what a step costs beyond its slowest rank's kernel (the gather /
        # reduce, the assembly, launch and host overhead); imbalance = slowest / mean of the ranks' kernel times
        out["per_rank_kernel_ms"], out["exchange_ms"] = pdist.step_breakdown(per_rank, out["ms_per_step"])
        out["scaling_note"] = ("unmeasured on hardware until a SCALE run with N > 1 distinct GPUs exists (dist.ranks_on_distinct_gpus); "
                               "roofline.kernel_ms is rank 0's, per_rank_kernel_ms every rank's")
    if rank == 0:
        if world == 1 and not in_process and not args.no_cpu_baseline and args.workload != "big":
            out["cpu_baseline"] = cpu_baseline(kind, n, res, integrator, depth, spp,
                                               gpu_film=film.cpu().numpy() if film is not None else None)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        if film is not None:
            import numpy as np
            f = film.cpu().numpy()
            out["film_check"] = {"weight_ok": bool((f[..., 3] == spp[0] * spp[1]).all()) if not wide else None, "finite": bool(np.isfinite(f).all()),
                                 "mean_Y": float(f[..., 1].mean() / (spp[0] * spp[1])), "mean_weight": float(f[..., 3].mean())}
        print(json.dumps(out), flush=True)
    scene.close()
    if use_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
