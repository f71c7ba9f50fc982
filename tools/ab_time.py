#!/usr/bin/env python3
"""Frame time of one library build on a BASELINE scene, for A-B jobs (tools/ab_variants.sh builds the variants; select one with
PBRT_HIP_LIB_DIR): usage ab_time.py [c2|c3|big] spp_x spp_y [frames] -> the frames' kernel_ms and the film's CRC (equal films = equal CRC)."""
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrt_amd  # noqa: E402
from pbrt_amd import scenes  # noqa: E402

wl = sys.argv[1]
spp = (int(sys.argv[2]), int(sys.argv[3]))
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 3
n, res = {"c3": (1_000_000, 2048), "c2": (100_000, 1024), "big": (12_000_000, 2048)}[wl]
sd = scenes.random_mesh_scene(n, res, res)
with pbrt_amd.Scene(sd) as sc:
    ms = []
    for _ in range(frames):
        film, st = sc.render(max_depth=8, spp=spp, seed=0)
        ms.append(st["kernel_ms"])
print(os.path.basename(os.environ.get("PBRT_HIP_LIB_DIR", "lib")), wl, spp, "kernel_ms", " ".join(f"{m:.1f}" for m in ms), "min %.1f" % min(ms), "crc %08x" % zlib.crc32(film.tobytes()))
