#!/usr/bin/env python3
"""Tiny driver for rocprofv3 --pmc passes: builds a BASELINE scene and renders it once at a low
sample count (no torch, no baseline leg).  usage: pmc_probe.py [c2|c3|c4|big] [spp_x spp_y]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrt_amd  # noqa: E402
from pbrt_amd import scenes  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
spp = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2, 2)
n, res = {"c3": (1_000_000, 2048), "c2": (100_000, 1024), "big": (12_000_000, 2048), "c4": (0, 4096)}[wl]
depth = 16 if wl == "c4" else 8
sd = scenes.cornell_scene(res, res) if wl == "c4" else scenes.random_mesh_scene(n, res, res)
with pbrt_amd.Scene(sd, builder=os.environ.get("PROBE_BUILDER")) as sc:
    print("accelerator:", sc.build_info(), sc.info())
    world = int(os.environ.get("PROBE_WORLD", "1"))  # PROBE_WORLD=8: rank 0's share of an 8-GPU job (strong scaling)
    film, st = sc.render(max_depth=depth, spp=spp, seed=0, world_size=world, rank=int(os.environ.get("PROBE_RANK", "0")))
    print(wl, spp, "world", world, "kernel_ms", st["kernel_ms"], "Msamples/s", st["samples"] / st["kernel_ms"] / 1e3)
    if os.environ.get("PROBE_COUNTERS"):
        _, ex = sc.render(max_depth=depth, spp=spp, seed=0, counters=True)
        _, wk = sc.render(max_depth=depth, spp=spp, seed=0, counters="walk")
        rays = ex["camera_rays"] + ex["bounce_rays"] + ex["shadow_rays"]
        print("exact: nodes/ray %.1f tris/ray %.2f | production walk: fetches/ray %.1f tris/ray %.2f" % (
            ex["nodes_visited"] / rays, ex["tris_tested"] / rays, wk["nodes_visited"] / rays, wk["tris_tested"] / rays))
        print("RAYS %d SAMPLES %d KERNEL_MS %.4f" % (rays, ex["samples"], st["kernel_ms"]))
        # the TREE the per-ray counters of this profile belong to (bench.py withholds a profile whose tree is not the live one: a builder
        # edit changes the tree and leaves the render kernel's machine code alone) and the launch's dynamic LDS (render_stack_plan)
        import ctypes as C
        i = sc.info()
        rows, waves, ovf = C.c_uint32(), C.c_uint32(), C.c_uint32()
        pbrt_amd.api.lib().pbrt_hip_render_stack_plan(i["quad_stack_need"], C.byref(rows), C.byref(waves), C.byref(ovf))
        print("TREE FETCHES_PER_RAY %.6f TRIS_PER_RAY %.6f QUAD_NODES %d STACK_NEED %d LDS_ROWS %d WAVES_PER_CU %d SPP %d" % (
            wk["nodes_visited"] / rays, wk["tris_tested"] / rays, i["quad_nodes"], i["quad_stack_need"], rows.value, waves.value, spp[0] * spp[1]))
