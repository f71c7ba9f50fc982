#!/usr/bin/env python3
"""Tuning aid: throughput of the traversal kernel alone on incoherent rays through a BASELINE mesh.
usage: PBRT_HIP_TIME_INTERSECT=1 ray_probe.py [n_tris] [n_rays]   (kernel time goes to stderr)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrt_amd  # noqa: E402
from pbrt_amd import scenes  # noqa: E402

n_tris = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8_000_000
sd = scenes.random_mesh_scene(n_tris, 64, 64)
lo, hi = sd.P.min(0), sd.P.max(0)
rng = np.random.default_rng(1)
o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
d = rng.normal(size=(n, 3))
d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
tmax = np.full(n, np.inf, np.float32)
with pbrt_amd.Scene(sd) as sc:
    for _ in range(3):
        r = sc.intersect(o, d, tmax)
    print("hits", int((r[1] != 0xFFFFFFFF).sum()), "of", n)
