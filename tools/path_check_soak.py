#!/usr/bin/env python3
"""The own-box rule's promise checked node by node over many seeds, on the CPU: for every adversarial (ray, triangle) pair the
rule accepts, every node test from the root of a product builder's quantised tree ("sah", "reinsert") to the triangle's leaf slot passes
(oracle/quad_walk.cpp: orc_quad_path_check) -- tests/test_oracle_selfcheck.py's test_every_walk_reaches_what_the_own_box_rule_accepts with
seeds 1000 ... 1000 + N.  python3 tools/path_check_soak.py 6000: 105 M pairs in six minutes (profiles/r06p_path_check_soak.txt)."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import binding as oracle
from pbrt_amd.api import quad_build_host_ex
from util import adversarial_rays, random_rays, SMALL_SCENES
t0 = time.time(); total = 0
for name in ("mesh1k", "cornell", "ties", "deep"):
    if name not in SMALL_SCENES: continue
    sd = SMALL_SCENES[name]().normalized()
    sc = oracle.OracleScene(sd)
    trees = {t: quad_build_host_ex(sd.P, sd.idx, tree=t) for t in ("sah", "reinsert")}
    for seed in range(1000, 1000 + int(sys.argv[1])):
        o, d, tmax, tri = adversarial_rays(sd, 40_000, seed, with_targets=True)
        ok, th = sc.tri_accepts(o, d, tmax, tri)
        keep = ok != 0
        total += int(keep.sum())
        for tname, q in trees.items():
            fails = oracle.quad_path_check(q["quads"], q["root_box"], q["order"], o[keep], d[keep], tri[keep], th[keep])
            if fails.any(): print("FAIL", name, tname, seed, int((fails != 0).sum())); sys.exit(1)
    print(name, "ok", total, "pairs", round(time.time() - t0), "s", flush=True)
print("all ok", total)
