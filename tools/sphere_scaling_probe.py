#!/usr/bin/env python3
"""Round 6: what a scene of SPHERES costs beside a scene of triangles now that spheres are primitives of the tree --
production-walk fetches and primitive tests per ray, frame time; 256 x 256, 16 spp, path depth 8.  usage (GPU box): python3 tools/sphere_scaling_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import pbrt_amd  # noqa: E402
from pbrt_amd import scenes  # noqa: E402
from util import sphere_cloud_scene  # noqa: E402

kw = dict(max_depth=8, spp=(4, 4), seed=1)
for name, sd in (("1 sphere (C1's scene)", scenes.sphere_scene(256, 256)), ("100 spheres", sphere_cloud_scene(100, 256, 256, n_tris=0)),
                 ("1 000 spheres", sphere_cloud_scene(1000, 256, 256, n_tris=0)), ("10 000 spheres", sphere_cloud_scene(10_000, 256, 256, n_tris=0)),
                 ("100 000 spheres", sphere_cloud_scene(100_000, 256, 256, n_tris=0)), ("1 000 000 spheres", sphere_cloud_scene(1_000_000, 256, 256, n_tris=0)),
                 ("20 000 triangles", scenes.random_mesh_scene(20_000, 256, 256)), ("200 000 triangles", scenes.random_mesh_scene(200_000, 256, 256))):
    with pbrt_amd.Scene(sd) as sc:
        bi = sc.build_info()
        sc.render(**kw)
        _, st = sc.render(**kw)
        _, wk = sc.render(counters="walk", **kw)
    rays = max(wk["camera_rays"] + wk["bounce_rays"] + wk["shadow_rays"], 1)
    print(f"{name:22s} build {bi['build_ms']:7.1f} ms ({'device' if bi['gpu_built'] else 'host'})  kernel {st['kernel_ms']:8.2f} ms  {st['samples'] / st['kernel_ms'] / 1e3:7.1f} Msamples/s  "
          f"fetches/ray {wk['nodes_visited'] / rays:6.2f}  tests/ray {wk['tris_tested'] / rays:5.2f}  rays/sample {rays / st['samples']:.2f}", flush=True)
