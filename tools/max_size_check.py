#!/usr/bin/env python3
"""The largest scene pbrt_hip_scene_create takes -- 2^24 triangles (a leaf reference holds a 24-bit slot) -- built on the device, rendered and
intersected, against the CPU oracle on the same arrays: film of a small frame and 20 000 hit records, bit for bit.  One more triangle must
be refused.  Run on the GPU box (about two minutes, most of it the oracle's own tree):  python3 tools/max_size_check.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pbrt_amd  # noqa: E402
from pbrt_amd import _lib, scenes  # noqa: E402
from oracle import binding as ob  # noqa: E402
from util import random_rays  # noqa: E402

n = (1 << 24) - 14  # + the box's 12 and the light's 2
t = time.time()
sd = scenes.random_mesh_scene(n, 64, 48)
assert sd.idx.shape[0] == 1 << 24
print(f"scene: {sd.idx.shape[0]} triangles ({time.time() - t:.1f} s)")
kw = dict(max_depth=6, spp=(2, 2), seed=7)
o, d, tmax = random_rays(20000, 3, inside=1.9)
t = time.time()
t0 = time.time()
with pbrt_amd.Scene(sd) as sc:
    info = sc.build_info()
    print(f"device build: {info.get('build_ms', 0):.0f} ms, optimisation {info.get('reinsert_ms', 0):.0f} ms, {sc.info()['device_bytes'] / 2**30:.2f} GiB on the device ({time.time() - t:.1f} s with the upload)")
    film, st = sc.render(**kw)
    hit = sc.intersect(o, d, tmax)
    occ = sc.occluded(o, d, tmax)
    print(f"render: {st['kernel_ms']:.1f} ms for {st['samples']} samples")
print(f"GPU side in all: {time.time() - t0:.1f} s")
t = time.time()
ref = ob.OracleScene(sd)
print(f"oracle tree: {time.time() - t:.1f} s")
t = time.time()
rfilm, _ = ref.render(**kw)
rhit = ref.intersect(o, d, tmax)
rocc = ref.occluded(o, d, tmax)
print(f"oracle film + hits: {time.time() - t:.1f} s")
ok = np.array_equal(film.view(np.uint32), rfilm.view(np.uint32))
ok_hit = all(np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32)) for a, b in zip(hit[:4], rhit[:4]))
ok_occ = np.array_equal(occ != 0, rocc != 0)
print("film bit-equal:", ok, "| hit records bit-equal:", ok_hit, f"({(rhit[1] != 0xffffffff).mean():.3f} of the rays hit)", "| occlusion equal:", ok_occ)
t = time.time()
one_more = scenes.random_mesh_scene(64, 16, 16)
one_more.P = np.zeros((3, 3), np.float32)
one_more.idx = np.zeros(((1 << 24) + 1, 3), np.uint32)
one_more.mat_id = np.zeros((1 << 24) + 1, np.uint16)
one_more.tri_uv = np.zeros((0, 6), np.float32)
try:
    pbrt_amd.Scene(one_more.normalized()).close()
    refused = False
except _lib.PbrtHipError as e:
    refused = e.code == -4 and "2^24" in str(e)
print("2^24 + 1 triangles refused (PBRT_HIP_ERR_LIMIT):", refused, f"({time.time() - t:.1f} s)")
sys.exit(0 if ok and ok_hit and ok_occ and refused else 1)
