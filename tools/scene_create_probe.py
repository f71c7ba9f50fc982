#!/usr/bin/env python3
"""Where does `scene_build_s` of the bench line go?  Creates the C3 scene several times in one process (the first call also pays
for the HIP context, the code object and the first hipMalloc) with both builders and prints wall time and the library's
own build_ms.   usage: scene_create_probe.py [c3|c2|big]"""
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import pbrt_amd  # noqa: E402
from pbrt_amd import scenes  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
sd = scenes.random_mesh_scene({"c3": 1_000_000, "c2": 100_000}[wl], 256, 256).normalized()
print(f"{wl}: {len(sd.idx)} triangles")
for builder in ("gpu", "gpu", "gpu", "host", "gpu"):
    t0 = time.time()
    sc = pbrt_amd.Scene(sd, builder=builder)
    t1 = time.time()
    info = sc.build_info()
    sc.close()
    print(f"builder {builder:4s}: Scene() {1e3 * (t1 - t0):8.1f} ms wall, library build_ms {info.get('build_ms', float('nan')):8.1f}, close {1e3 * (time.time() - t1):6.1f} ms")
