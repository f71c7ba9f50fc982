#!/usr/bin/env python3
"""Round 6: is the production walk's work per ray the same from build to build?  (bench.py withholds a counter profile whose tree costs the live
walk more than 0.5 % other work per ray: a build-to-build variation of that size would make the guard fire at random.)  Builds BASELINE C2's and
C3's scenes several times with the default (device) builder and prints quad nodes, fetches and triangle tests per ray of a 1-spp frame.
usage (GPU box): python3 tools/tree_determinism_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrt_amd  # noqa: E402
from pbrt_amd import scenes  # noqa: E402

for name, n, res in (("c2", 100_000, 1024), ("c3", 1_000_000, 2048)):
    sd = scenes.random_mesh_scene(n, res, res)
    seen = []
    for i in range(5):
        with pbrt_amd.Scene(sd) as sc:
            _, wk = sc.render(max_depth=8, spp=(1, 1), seed=0, counters="walk")
            q = sc.info()["quad_nodes"]
        rays = wk["camera_rays"] + wk["bounce_rays"] + wk["shadow_rays"]
        seen.append((q, wk["nodes_visited"], wk["tris_tested"], rays))
        print(name, "build", i, "quad nodes", q, "fetches/ray %.6f" % (wk["nodes_visited"] / rays), "tests/ray %.6f" % (wk["tris_tested"] / rays), flush=True)
    f = [s[1] / s[3] for s in seen]
    print(name, "spread of fetches/ray over 5 builds: %.4f %%" % (100 * (max(f) - min(f)) / min(f)), "identical counters:", len(set(seen)) == 1)
