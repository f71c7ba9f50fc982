#!/usr/bin/env python3
"""Quality of the production walk's tree as it sits in HBM (pbrt_hip_scene_export_quads): node count, depth, and the
SAH-style cost sum of decoded child-box areas / root area (interior children: one node step each; leaf children: one
triangle test each), for the host-built and the device-built tree of the same scene.
usage: quad_quality.py [n_tris]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrt_amd  # noqa: E402
from pbrt_amd import scenes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
sd = scenes.random_mesh_scene(n, 256, 256)
for builder in ("host", "gpu"):
    with pbrt_amd.Scene(sd, builder=builder) as sc:
        quads, order = sc.export_quads()
        info = sc.info()
        film, st = sc.render(max_depth=8, spp=(2, 2), seed=0, counters="walk")
    f = quads.view(np.float32)
    origin = f[:, 0:3].astype(np.float64)
    cell = np.stack([f[:, 3], f[:, 10], f[:, 11]], axis=1).astype(np.float64)  # powers of two, as f32
    qlo = quads[:, 4:7]
    qhi = np.stack([quads[:, 7], quads[:, 8], quads[:, 9]], axis=1)
    refs = quads[:, 12:16]
    area_int = area_leaf = 0.0
    kids = 0
    for k in range(4):
        lo = origin + ((qlo >> (8 * k)) & 0xFF) * cell
        hi = origin + ((qhi >> (8 * k)) & 0xFF) * cell
        d = np.maximum(hi - lo, 0)
        a = d[:, 0] * d[:, 1] + d[:, 0] * d[:, 2] + d[:, 1] * d[:, 2]
        used = refs[:, k] != 0x80000000
        leaf = used & ((refs[:, k] & 0x80000000) != 0)
        area_int += a[used & ~leaf].sum()
        area_leaf += a[leaf].sum()
        kids += used.sum()
    d0 = (f[0, 0:3] * 0)  # root box = union of the root's children
    lo0 = np.min([origin[0] + ((qlo[0] >> (8 * k)) & 0xFF) * cell[0] for k in range(4) if refs[0, k] != 0x80000000], axis=0)
    hi0 = np.max([origin[0] + ((qhi[0] >> (8 * k)) & 0xFF) * cell[0] for k in range(4) if refs[0, k] != 0x80000000], axis=0)
    dr = hi0 - lo0
    root = dr[0] * dr[1] + dr[0] * dr[2] + dr[1] * dr[2]
    rays = st["camera_rays"] + st["bounce_rays"] + st["shadow_rays"]
    print(f"{builder}: quads {len(quads)} children/node {kids / len(quads):.2f} stack_need {info['quad_stack_need']} "
          f"cost: interior {area_int / root:.1f} + leaves {area_leaf / root:.1f} | walk: fetches/ray {st['nodes_visited'] / rays:.1f} tris/ray {st['tris_tested'] / rays:.2f} "
          f"kernel_ms {st['kernel_ms']:.2f} build_ms {sc.build_info()['build_ms'] if False else 0}")
