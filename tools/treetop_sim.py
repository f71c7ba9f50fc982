#!/usr/bin/env python3
"""How much of the production walk happens at the TOP of the tree?  (VERDICT r04 task 2a; no GPU needed.)

Builds the product's 4-wide tree of a BASELINE mesh scene on the host (the re-insertion-optimised tree, as the device builder
ships), walks the path-tracing ray mix of tools/walk_sim.py through it with the kernel's node step restated
(oracle/quad_walk.cpp) counting the steps PER NODE, and prints which share of all node steps lands in the K most-visited nodes
(the best any K-node resident set could do), in the first K nodes of a breadth-first numbering, and in the K nodes of largest
surface area (what a builder can pick without rays), and in the nodes numbered 0 .. K-1 as the tree stands -- for K = 64 ... 4096, i.e. 4 ... 256 KB of 64-byte nodes.  That share is
the ceiling of what a copy of the treetop in LDS can take off the vector L1.

usage: treetop_sim.py [n_tris] [sah|reinsert]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import binding as ob  # noqa: E402
from pbrt_amd import scenes  # noqa: E402
from pbrt_amd.api import quad_build_host_ex  # noqa: E402
from walk_sim import path_rays  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    tree = sys.argv[2] if len(sys.argv) > 2 else "reinsert"
    sd = scenes.random_mesh_scene(n, 256, 256).normalized()
    osc = ob.OracleScene(sd)
    (co, cd, ct), (so, sdd, stm) = path_rays(sd, osc)
    q = quad_build_host_ex(sd.P, sd.idx, tree=tree)
    quads = q["quads"].reshape(-1, 16)
    nq = len(quads)
    visits = np.zeros(nq, np.uint64)
    ob.quad_walk_count_visits(visits)
    c = ob.quad_walk(quads, q["root_box"], sd.P, sd.idx, q["order"], co, cd, ct)
    s = ob.quad_walk(quads, q["root_box"], sd.P, sd.idx, q["order"], so, sdd, stm, any_hit=True)
    ob.quad_walk_count_visits(None)
    rays = len(co) + len(so)
    total = int(visits.sum())
    assert total == int(c["steps"].sum() + s["steps"].sum())
    print(f"{n} triangles, tree '{tree}': {nq} quad nodes ({nq * 64 / 1e6:.1f} MB), {rays} rays, {total / rays:.2f} node steps per ray")
    # breadth-first level and order of every node; a node's box area from its quantisation grid (origin + 255 cells bound it)
    refs = quads[:, 12:16]
    level = np.full(nq, -1, np.int32)
    level[0] = 0
    order, frontier = [0], [0]
    while frontier:
        nxt = []
        for i in frontier:
            for r in refs[i]:
                if not (r & 0x80000000):
                    j = int(r) // 64
                    level[j] = level[i] + 1
                    nxt.append(j)
        order += nxt
        frontier = nxt
    order = np.array(order)
    # the area of a node = the area of its children's union, decoded from the node's own grid
    cell = np.stack([quads[:, 3].view(np.float32), quads[:, 10].view(np.float32), quads[:, 11].view(np.float32)], 1).astype(np.float64)
    def planes(w):  # (nq, 4) bytes of one plane word
        return np.stack([(quads[:, w] >> (8 * k)) & 0xFF for k in range(4)], 1).astype(np.float64)
    used = (refs != 0x80000000)
    ext = []
    for lo_w, hi_w, a in ((4, 7, 0), (5, 8, 1), (6, 9, 2)):
        lo, hi = planes(lo_w), planes(hi_w)
        lo = np.where(used, lo, 1e9).min(1)
        hi = np.where(used, hi, -1e9).max(1)
        ext.append(np.maximum(hi - lo, 0) * cell[:, a])
    area = 2 * (ext[0] * ext[1] + ext[0] * ext[2] + ext[1] * ext[2])
    by_visits = np.argsort(-visits.astype(np.int64), kind="stable")
    by_area = np.argsort(-area, kind="stable")
    per_level = [(int((level == l).sum()), float(visits[level == l].sum()) / rays) for l in range(level.max() + 1)]
    print("level: nodes, steps per ray:  " + "  ".join(f"{l}: {a}, {b:.2f}" for l, (a, b) in enumerate(per_level)))
    print(f"{'K':>6} {'KB':>6} | share of node steps in: the K most-visited | the first K breadth-first | the K of largest area | nodes 0 .. K-1 as numbered")
    for K in (64, 128, 256, 384, 512, 640, 768, 1024, 1280, 1536, 2048, 4096):
        if K > nq:
            break
        row = [float(visits[sel[:K]].sum()) / total for sel in (by_visits, order, by_area, np.arange(nq))]
        print(f"{K:6d} {K * 64 // 1024:6d} | {row[0]:22.3f} | {row[1]:25.3f} | {row[2]:20.3f} | {row[3]:.3f}")


if __name__ == "__main__":
    main()
