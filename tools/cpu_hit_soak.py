#!/usr/bin/env python3
"""CPU only: "a hit is a function of (ray, primitive) alone"  over the random scenes of the GPU soak (tests/test_gpu_parity.py's
_random_scene: 0 ... 20 000 triangles, duplicates, degenerate triangles, sphere clouds) instead of the suite's five fixed ones: for each seed
adversarial rays (util.adversarial_rays: at vertices, edge points, along edges and axes, tmax on the target) and random rays -- the oracle's
BVH against its brute force over all primitives (hit record and occlusion flag, bit for bit), and for scenes without spheres the production walk
restated (oracle/quad_walk.cpp) over both product builders' quantised trees.
python3 tools/cpu_hit_soak.py N [FIRST]       (profiles/r06p_cpu_hit_soak.txt)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import binding as oracle  # noqa: E402
from pbrt_amd.api import quad_build_host_ex  # noqa: E402
from util import random_scene as _random_scene  # noqa: E402
from util import adversarial_rays, random_rays  # noqa: E402,F811


def main():
    n, first = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0, rays, hits, walked = time.time(), 0, 0, 0
    for seed in range(first, first + n):
        sd, _ = _random_scene(seed)
        nt = sd.idx.shape[0]
        sc = oracle.OracleScene(sd)
        m = 4000 if nt <= 2500 else 600   # (the brute force is O(primitives) per ray)
        sets = [random_rays(m, seed, inside=2.5)]
        if nt > 0:
            sets.append(adversarial_rays(sd, m, seed))
        for o, d, tmax in sets:
            t, prim, b1, b2, _ = sc.intersect(o, d, tmax)
            bt, bprim, bb1, bb2, _ = sc.intersect(o, d, tmax, brute_force=True)
            same = (np.array_equal(prim, bprim) and np.array_equal(t.view(np.uint32), bt.view(np.uint32))
                    and np.array_equal(b1.view(np.uint32), bb1.view(np.uint32)) and np.array_equal(b2.view(np.uint32), bb2.view(np.uint32))
                    and np.array_equal(sc.occluded(o, d, tmax), sc.occluded(o, d, tmax, brute_force=True)))
            if not same:
                print("DIFFER: seed", seed, "BVH vs brute force on", int((prim != bprim).sum()), "rays", flush=True)
                return 1
            rays += len(o)
            hits += int((prim != 0xffffffff).sum())
            if sd.spheres.shape[0] == 0 and nt > 0:
                n_walked = 0
                for tree in ("sah", "reinsert"):
                    q = quad_build_host_ex(sd.P, sd.idx, tree=tree)
                    if len(q["quads"]) == 0:  # (a scene that fits one leaf: the product enters it through a leaf ref this entry point does not return)
                        continue
                    got = oracle.quad_walk(q["quads"], q["root_box"], sd.P, sd.idx, q["order"], o, d, tmax)
                    if not (np.array_equal(got["prim"], prim) and np.array_equal(got["t"].view(np.uint32), t.view(np.uint32))):
                        print("DIFFER: seed", seed, "production walk over the", tree, "tree", flush=True)
                        return 1
                    n_walked = len(o)
                walked += n_walked
        sc.close()
        if (seed - first) % 500 == 499:
            print(f"seeds {first} ... {seed}: {rays} rays ({hits} hits; {walked} also through the production walk x 2 trees), all equal, {time.time() - t0:.0f} s", flush=True)
    print(f"{n} scenes, {rays} rays ({hits} hits; {walked} also through the production walk x 2 trees): every record equal bit for bit, {time.time() - t0:.0f} s")
    return 0


if __name__ == "__main__":
    sys.exit(main())
