#!/usr/bin/env python3
"""Experiment: how fast is the traversal kernel ALONE on the rays a real frame traces?
Step 1 (library built with -DPBRT_RAY_LOG): render C3 at low spp, every launched ray is written to /tmp/raylog.bin.
Step 2 (any library): run pbrt_hip_intersect / pbrt_hip_occluded over those rays in launch order with
PBRT_HIP_TIME_INTERSECT=1 (kernel time goes to stderr).
Step 3 (round 5, `sorted`): the same rays in SORTED orders -- by the Morton code of the origin (10 bits per axis) within a direction
octant, and by the Morton code alone -- against the launch order and a random shuffle: what a wavefront design with sorted ray queues
could hope to gain in its traversal stage.
usage: raylog_probe.py log | raylog_probe.py trace | raylog_probe.py sorted"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrt_amd  # noqa: E402
from pbrt_amd import scenes  # noqa: E402

sd = scenes.random_mesh_scene(1_000_000, 2048, 2048)
with pbrt_amd.Scene(sd) as sc:
    if sys.argv[1] == "log":
        film, st = sc.render(max_depth=8, spp=(2, 2), seed=0)
        print("rendered", st["samples"], "samples in", st["kernel_ms"], "ms (with logging)")
    elif sys.argv[1] == "sorted":
        r = np.fromfile("/tmp/raylog.bin", np.float32).reshape(-1, 8)
        anyhit = r[:, 7] != 0

        def spread(v):  # 10 bits -> every third bit
            v = v.astype(np.uint64)
            v = (v | (v << 16)) & 0x030000FF
            v = (v | (v << 8)) & 0x0300F00F
            v = (v | (v << 4)) & 0x030C30C3
            v = (v | (v << 2)) & 0x09249249
            return v
        for name, sel in (("closest", ~anyhit), ("shadow", anyhit)):
            q = r[sel][:24_000_000]
            lo, hi = q[:, 0:3].min(0), q[:, 0:3].max(0)
            g = np.clip((q[:, 0:3] - lo) / np.maximum(hi - lo, 1e-9) * 1024, 0, 1023).astype(np.uint32)
            morton = (spread(g[:, 0]) << 2) | (spread(g[:, 1]) << 1) | spread(g[:, 2])
            octant = ((q[:, 4] < 0).astype(np.uint64) << 2) | ((q[:, 5] < 0).astype(np.uint64) << 1) | (q[:, 6] < 0).astype(np.uint64)
            orders = {"launch order": np.arange(len(q)), "random shuffle": np.random.default_rng(1).permutation(len(q)),
                      "Morton(origin)": np.argsort(morton, kind="stable"), "octant, then Morton(origin)": np.argsort((octant << 30) | morton, kind="stable")}
            for oname, perm in orders.items():
                p = q[perm]
                o, d, tmax = np.ascontiguousarray(p[:, 0:3]), np.ascontiguousarray(p[:, 4:7]), np.ascontiguousarray(p[:, 3])
                print(f"{name} rays, {len(p)}, {oname}:", flush=True)
                for _ in range(2):
                    (sc.occluded if name == "shadow" else sc.intersect)(o, d, tmax)
    else:
        r = np.fromfile("/tmp/raylog.bin", np.float32).reshape(-1, 8)
        anyhit = r[:, 7] != 0
        print("rays", len(r), "closest", int((~anyhit).sum()), "shadow", int(anyhit.sum()))
        for name, sel in (("closest", ~anyhit), ("shadow", anyhit)):
            q = r[sel][:48_000_000]
            o, d, tmax = np.ascontiguousarray(q[:, 0:3]), np.ascontiguousarray(q[:, 4:7]), np.ascontiguousarray(q[:, 3])
            print(name, len(q), "rays:", flush=True)
            for _ in range(2):
                (sc.occluded if name == "shadow" else sc.intersect)(o, d, tmax)
