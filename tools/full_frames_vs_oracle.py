#!/usr/bin/env python3
"""WHOLE frames of the BASELINE scenes from the HIP path (pbrt_hip_scene_create: the tree built and optimised on the device, the quantised
4-wide production walk) against the CPU oracle's (canonical binary tree), EVERY pixel -- the check behind profiles/r05zz_full_frames_vs_oracle.txt,
as a tool since round 6 (the own-box rule changed films: re-run).  Minutes of oracle time on the GPU box's host cores.
usage (on the GPU box): python3 tools/full_frames_vs_oracle.py [c3_spp c2_spp c4_spp]   default 16 32 2"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import pbrt_amd  # noqa: E402
from oracle import binding as ob  # noqa: E402
from pbrt_amd import INTEGRATOR_DIRECT, INTEGRATOR_PATH, INTEGRATOR_PATH_MIS, scenes  # noqa: E402


def strata(spp):
    sx = 1
    while sx * sx * 2 <= spp:
        sx *= 2
    return sx, spp // sx


def frame(name, sd, **kw):
    ob.build(native=True)
    threads = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            threads = min(threads, max(1, -(-int(q) // int(p))))
    except (OSError, ValueError):
        pass
    with pbrt_amd.Scene(sd) as sc:
        assert sc.build_info()["gpu_built"] or sc.n_prims < 2
        sc.render(**kw)
        film, st = sc.render(**kw)
    t0 = time.time()
    ref, rst = ob.OracleScene(sd, native=True).render(n_threads=threads, **kw)
    dt = time.time() - t0
    bad = int((film.view(np.uint32) != ref.view(np.uint32)).any(axis=-1).sum())
    rays = rst["camera_rays"] + rst["bounce_rays"] + rst["shadow_rays"]
    print(f"{name}: HIP {st['kernel_ms']:.1f} ms, oracle {dt:.0f} s ({threads} threads): {bad} of {film.shape[0] * film.shape[1]} pixels differ "
          f"({st['samples']} samples, {rays} rays)", flush=True)
    return bad


def main():
    a = [int(x) for x in sys.argv[1:4]] + [16, 32, 2][len(sys.argv[1:4]):]
    bad = 0
    bad += frame(f"C3's scene (1 000 014 triangles, 2048 x 2048, path depth 8) at {a[0]} spp", scenes.random_mesh_scene(1_000_000, 2048, 2048),
                 integrator=INTEGRATOR_PATH, max_depth=8, spp=strata(a[0]), seed=0)
    bad += frame(f"C2's scene (100 014 triangles, 1024 x 1024, path depth 8) at {a[1]} spp", scenes.random_mesh_scene(100_000, 1024, 1024),
                 integrator=INTEGRATOR_PATH, max_depth=8, spp=strata(a[1]), seed=0)
    bad += frame(f"C4's scene (Cornell-style, 4096 x 4096, path depth 16) at {a[2]} spp", scenes.cornell_scene(4096, 4096),
                 integrator=INTEGRATOR_PATH, max_depth=16, spp=strata(a[2]), seed=0)
    bad += frame("C1 as BASELINE states it (sphere + point light, 1024 x 1024, direct, 64 spp)", scenes.sphere_scene(1024, 1024),
                 integrator=INTEGRATOR_DIRECT, max_depth=5, spp=(8, 8), seed=0)
    from util import sphere_cloud_scene
    bad += frame("20 000 spheres + 64 triangles (512 x 512, path depth 8, 16 spp): spheres as primitives of the tree", sphere_cloud_scene(20_000, 512, 512),
                 integrator=INTEGRATOR_PATH, max_depth=8, spp=(4, 4), seed=0)
    bad += frame("C3's scene, whole frame, Halton sampler + integrator 2 (MIS), 8 spp", scenes.random_mesh_scene(1_000_000, 2048, 2048),
                 integrator=INTEGRATOR_PATH_MIS, max_depth=8, spp=(4, 2), seed=0, sampler="halton")
    print("every pixel of every frame bit-equal:", bad == 0)
    return 0 if bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
