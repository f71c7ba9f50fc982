#!/usr/bin/env python3
"""Node steps / triangle tests per ray of the production walk's tree, on the CPU (no GPU needed): builds the 4-wide tree of
a BASELINE mesh scene with each binary-tree builder (`sah` = canonical binned SAH, `reinsert` = the same optimised by the device
builder's parallel re-insertion pass run on the host: PBRT_HIP_SCENE_OPTIMIZED_TREE's tree; PBRT_HIP_REINSERT_VERBOSE=1 prints the
passes and the collapse's expected-work figure G, which tracks steps per ray), walks a path-tracing-like set of rays through it with oracle/quad_walk.cpp (the kernel's step restated),
checks every hit against the oracle's own BVH, and prints the per-ray work.

usage: walk_sim.py [n_tris] [trees...]      e.g.  PBRT_HIP_DEBUG_KNOBS=1 PBRT_HIP_REINSERT=4 walk_sim.py 100000 sah reinsert
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrt_amd  # noqa: E402
from oracle import binding as ob  # noqa: E402
from pbrt_amd import scenes  # noqa: E402
from pbrt_amd.api import quad_build_host_ex  # noqa: E402


def path_rays(sd, osc, res=96, bounces=5, seed=1):
    """camera rays of a res x res grid, then `bounces` generations of cosine-distributed bounce rays and shadow rays towards
    the ceiling light from the hit points: the ray mix of a path-traced frame (closest-hit and any-hit sets)."""
    rng = np.random.default_rng(seed)
    w, h = sd.xres, sd.yres
    xs = (np.arange(res) + 0.5) * w / res
    fx, fy = np.meshgrid(xs, (np.arange(res) + 0.5) * h / res)
    o = np.zeros((res * res, 3), np.float32)
    d = np.zeros((res * res, 3), np.float32)
    for i, (x, y) in enumerate(zip(fx.ravel(), fy.ravel())):
        o[i], d[i] = osc.camera_ray(float(x), float(y))
    closest = [(o, d, np.full(len(o), np.inf, np.float32))]
    shadow = []
    P = sd.P.reshape(-1, 3)
    for _ in range(bounces):
        o, d, tm = closest[-1]
        t, prim, b1, b2, _ = osc.intersect(o, d, tm)
        hit = prim < sd.idx.shape[0]
        o, d, t, prim = o[hit], d[hit], t[hit], prim[hit]
        if len(o) == 0:
            break
        p = o + d * t[:, None]
        tri = sd.idx[prim]
        n = np.cross(P[tri[:, 1]] - P[tri[:, 0]], P[tri[:, 2]] - P[tri[:, 0]])
        n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-30)
        n[(n * d).sum(1) > 0] *= -1
        po = (p + n * 1e-4).astype(np.float32)
        # shadow rays to a point of the 1 x 1 ceiling light at z = 1.99
        lp = np.stack([rng.random(len(po)) - 0.5, rng.random(len(po)) - 0.5, np.full(len(po), 1.99)], 1)
        dv = lp - po
        dist = np.linalg.norm(dv, axis=1)
        shadow.append((po, (dv / dist[:, None]).astype(np.float32), (dist * 0.9999).astype(np.float32)))
        # cosine-distributed bounce
        u1, u2 = rng.random(len(po)), rng.random(len(po))
        r, phi = np.sqrt(u1), 2 * np.pi * u2
        a = np.where(np.abs(n[:, :1]) > 0.9, [[0, 1, 0]], [[1, 0, 0]])
        tx = np.cross(n, a)
        tx /= np.linalg.norm(tx, axis=1, keepdims=True)
        ty = np.cross(n, tx)
        nd = tx * (r * np.cos(phi))[:, None] + ty * (r * np.sin(phi))[:, None] + n * np.sqrt(np.maximum(0, 1 - u1))[:, None]
        closest.append((po, nd.astype(np.float32), np.full(len(po), np.inf, np.float32)))
    cat = lambda sets: tuple(np.concatenate([s[k] for s in sets]) for k in range(3))
    return cat(closest), cat(shadow)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    trees = sys.argv[2:] or ["sah", "reinsert"]
    sd = scenes.random_mesh_scene(n, 256, 256).normalized()
    osc = ob.OracleScene(sd)
    (co, cd, ct), (so, sdd, stm) = path_rays(sd, osc)
    rt, rprim, rb1, rb2, _ = osc.intersect(co, cd, ct)
    rocc = osc.occluded(so, sdd, stm)
    print(f"{n} triangles; {len(co)} closest-hit rays, {len(so)} shadow rays")
    for tree in trees:
        t0 = time.time()
        q = quad_build_host_ex(sd.P, sd.idx, tree=tree)
        dt = time.time() - t0
        ex = q["exact_boxes"] if os.environ.get("WALK_SIM_EXACT_BOXES") else None  # experiment: unquantised child boxes
        c = ob.quad_walk(q["quads"], q["root_box"], sd.P, sd.idx, q["order"], co, cd, ct, exact_boxes=ex)
        s = ob.quad_walk(q["quads"], q["root_box"], sd.P, sd.idx, q["order"], so, sdd, stm, any_hit=True, exact_boxes=ex)
        ok = (np.array_equal(c["t"].view(np.uint32), rt.view(np.uint32)) and np.array_equal(c["prim"], rprim)
              and np.array_equal(c["b1"].view(np.uint32), rb1.view(np.uint32)) and np.array_equal(s["occluded"], rocc))
        steps = (c["steps"].sum() + s["steps"].sum()) / (len(co) + len(so))
        tris = (c["tris"].sum() + s["tris"].sum()) / (len(co) + len(so))
        print(f"{tree:8s} quads {len(q['quads']):8d} ({len(q['quads']) * 64 / 1e6:.1f} MB) refs {q['n_refs']:8d} stack_need {q['stack_need']:3d} "
              f"max_stack {max(c['max_stack'], s['max_stack']):3d} | steps/ray {steps:6.2f} (closest {c['steps'].mean():6.2f}, shadow {s['steps'].mean():6.2f}) "
              f"tris/ray {tris:5.2f} | hits {'EQUAL to the oracle' if ok else 'DIFFER'} | build {dt:.1f} s")


if __name__ == "__main__":
    main()
