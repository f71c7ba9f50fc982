#!/usr/bin/env python3
"""Writes the mesh and the ray mix of a path-traced frame (tools/walk_sim.py path_rays, with the oracle's hits) for wide_sim.cpp,
builds it and runs it.  usage: run.py [n_tris]"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import binding as ob  # noqa: E402
from pbrt_amd import scenes  # noqa: E402
from walk_sim import path_rays  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
out = f"/tmp/wide_sim_{n}"
os.makedirs(out, exist_ok=True)
sd = scenes.random_mesh_scene(n, 256, 256).normalized()
osc = ob.OracleScene(sd)
(co, cd, ct), (so, sdd, stm) = path_rays(sd, osc)
rt, rprim, _, _, _ = osc.intersect(co, cd, ct)
rocc = osc.occluded(so, sdd, stm)
P = np.ascontiguousarray(sd.P, np.float32).reshape(-1, 3)
with open(f"{out}/mesh.bin", "wb") as f:
    f.write(np.array([sd.idx.shape[0], P.shape[0]], np.uint32).tobytes() + P.tobytes() + np.ascontiguousarray(sd.idx, np.uint32).tobytes())
with open(f"{out}/closest.bin", "wb") as f:
    f.write(np.array([len(co)], np.uint32).tobytes() + co.tobytes() + cd.tobytes() + ct.tobytes() + rt.tobytes() + rprim.astype(np.uint32).tobytes())
with open(f"{out}/shadow.bin", "wb") as f:
    f.write(np.array([len(so)], np.uint32).tobytes() + so.tobytes() + sdd.tobytes() + stm.tobytes() + (rocc != 0).astype(np.uint8).tobytes())
exe = "/tmp/wide_sim"
subprocess.run(["g++", "-O2", "-std=c++17", f"-I{ROOT}/pbrt_amd/csrc", f"{ROOT}/tools/experiments/r04_wide8/wide_sim.cpp", f"{ROOT}/pbrt_amd/csrc/bvh_build.cpp",
                f"{ROOT}/pbrt_amd/csrc/reinsert_batch.cpp", "-o", exe], check=True)
subprocess.run([exe, out], check=True)
