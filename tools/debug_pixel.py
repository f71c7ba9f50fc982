#!/usr/bin/env python3
"""Parity debugging: per-sample radiance of one pixel, HIP (library built with -DPBRT_DEBUG_PIXEL_X=x
-DPBRT_DEBUG_PIXEL_Y=y, which prints SAMPLE lines) against the oracle's pixel_samples().
usage: debug_pixel.py <small-scene-name> <x> <y> <depth> <sx> <sy> <seed>"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
name, x, y, depth, sx, sy, seed = sys.argv[1], *map(int, sys.argv[2:8])
if os.environ.get("DEBUG_PIXEL_CHILD"):
    import pbrt_amd
    from util import SMALL_SCENES
    with pbrt_amd.Scene(SMALL_SCENES[name]()) as sc:
        sc.render(max_depth=depth, spp=(sx, sy), seed=seed, counters=bool(int(os.environ.get("DEBUG_EXACT", "1"))))
    sys.exit(0)
out = subprocess.run([sys.executable] + sys.argv, env=dict(os.environ, DEBUG_PIXEL_CHILD="1"), capture_output=True, text=True)
got = {}
trace = [l for l in out.stdout.splitlines() if l.startswith("HIP")]
for line in out.stdout.splitlines():
    if line.startswith("SAMPLE"):
        f = line.split()
        got[int(f[1])] = [int(v, 16) for v in f[2:5]]
from oracle import binding as ob
from util import SMALL_SCENES
ref = ob.OracleScene(SMALL_SCENES[name]()).pixel_samples(x, y, max_depth=depth, spp=(sx, sy), seed=seed).reshape(-1, 3)
rb = ref.view(np.uint32)
print("samples from the kernel:", len(got), "oracle:", len(rb))
bad = []
for i in range(len(rb)):
    g = got.get(i)
    if g is None or list(rb[i]) != g:
        print("sample", i, "kernel", g, "oracle", list(rb[i]), ref[i])
        bad.append(i)
for i in bad:
    print("---- kernel trace of sample", i)
    print("\n".join(l for l in trace if l.startswith(f"HIP s {i} ")))
if bad:
    print("---- oracle trace (all samples of the pixel; count 'sample begins' lines)")
    os.environ["ORC_DEBUG_LI"] = "1"
    sys.stderr.flush()
    ob.OracleScene(SMALL_SCENES[name]()).pixel_samples(x, y, max_depth=depth, spp=(sx, sy), seed=seed)
