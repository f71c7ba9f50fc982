#!/usr/bin/env python3
"""Round 5, VERDICT r04 task 2b: the treetop variant of the render kernel (PBRT_HIP_TREETOP = nodes in the workgroup's LDS,
PBRT_HIP_TREETOP_WAVES = waves per workgroup) against the product's one-wave-workgroup kernel, IN ONE JOB on one box: same scene
handle, same device-built tree, interleaved runs; films must be identical bit for bit.  Also checks that the device builder numbers
the tree level by level (what makes "nodes 0 .. K-1" the treetop): the share of the frame's node steps that the K first nodes
take, from the walk counters of two renders (nodes fetched through L1 = all fetches - LDS ones is not counted separately; the
simulator over the EXPORTED tree gives the share).

usage (GPU box): r05_treetop_ab.py [c3|c2] [spp_x spp_y]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
os.environ["PBRT_HIP_DEBUG_KNOBS"] = "1"
import pbrt_amd  # noqa: E402
from pbrt_amd import scenes  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
spp = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (8, 8)
n, res = {"c3": (1_000_000, 2048), "c2": (100_000, 1024)}[wl]
sd = scenes.random_mesh_scene(n, res, res)
variants = [(0, 0)] + [(k, w) for w in (4, 2, 1, 10) for k in (64, 128, 192)] + [(0, 0)]
with pbrt_amd.Scene(sd) as sc:
    print("accelerator:", sc.build_info(), sc.info())
    # is the device-built tree numbered top first?  level of every node by a breadth-first walk over the exported nodes
    quads, _ = sc.export_quads()
    refs = quads[:, 12:16]
    level = np.full(len(quads), -1, np.int32)
    level[0] = 0
    frontier = [0]
    while frontier:
        nxt = []
        for i in frontier:
            for r in refs[i]:
                if not (r & 0x80000000):
                    level[int(r) // 64] = level[i] + 1
                    nxt.append(int(r) // 64)
        frontier = nxt
    print("levels of nodes 0..15:", level[:16].tolist(), "| monotone in the node number:", bool((np.diff(level) >= 0).all()),
          "| nodes per level:", np.bincount(level[level >= 0]).tolist()[:10])
    ref_film = None
    for rep in range(2):
        for k, w in variants:
            os.environ["PBRT_HIP_TREETOP"] = str(k)
            os.environ["PBRT_HIP_TREETOP_WAVES"] = str(w or 10)
            film, st = sc.render(max_depth=8, spp=spp, seed=0)
            if ref_film is None:
                ref_film = film
            same = bool((film.view(np.uint32) == ref_film.view(np.uint32)).all())
            print(f"{wl} {spp[0]}x{spp[1]} spp  treetop {k:5d} nodes x {w:2d} waves/workgroup: kernel {st['kernel_ms']:9.2f} ms  "
                  f"{st['samples'] / st['kernel_ms'] / 1e3:7.1f} Msamples/s  film {'identical' if same else 'DIFFERS'}", flush=True)
    os.environ["PBRT_HIP_TREETOP"] = "128"
    _, wk = sc.render(max_depth=8, spp=spp, seed=0, counters="walk")
    os.environ["PBRT_HIP_TREETOP"] = "0"
    _, wk0 = sc.render(max_depth=8, spp=spp, seed=0, counters="walk")
    print("walk counters, treetop 128 vs none:", wk["nodes_visited"], wk0["nodes_visited"], wk["tris_tested"], wk0["tris_tested"])
