#!/usr/bin/env python3
"""Experiment: how fast is the traversal kernel ALONE on the rays a real frame traces?
Step 1 (library built with -DPBRT_RAY_LOG): render C3 at low spp, every launched ray is written to /tmp/raylog.bin.
Step 2 (any library): run pbrt_hip_intersect / pbrt_hip_occluded over those rays in launch order with
PBRT_HIP_TIME_INTERSECT=1 (kernel time goes to stderr).
usage: raylog_probe.py log | raylog_probe.py trace"""
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
