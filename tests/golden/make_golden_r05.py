"""Generates tests/golden/render_golden_r05.npz with the CPU oracle: regression pins of what round 5 ADDED to the spec beside the
default path (self-generated, like render_golden.npz: the reference has no renderer) -- sampler 2 on a maxdepth-16 path (its requests
17 .. 61 on their own Sobol' dimensions), sampler 3 (Halton), integrator 2 (MIS), a checkerboard Kd on triangles and on a sphere.  render_golden.npz itself is
NOT regenerated: the oracle still reproducing it is the evidence that round 5 changed nothing of the default path's arithmetic.
Run from the repo root:  python tests/golden/make_golden_r05.py
Key format: scene-integrator-maxdepth-sppx-sppy-seed-sampler."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import binding as ob  # noqa: E402
from util import SMALL_SCENES, checker_plane_scene, checker_sphere_scene  # noqa: E402

SCENES = dict(SMALL_SCENES, checker=lambda: checker_plane_scene(40)[0], checkersphere=lambda: checker_sphere_scene(48, 40))
CASES = [("cornell", 0, 16, 4, 4, 21, 2), ("cornell", 0, 6, 4, 4, 22, 3), ("check_sphere", 0, 5, 3, 2, 23, 3), ("mesh1k", 0, 8, 2, 2, 24, 3),
         ("cornell", 2, 8, 4, 4, 25, 0), ("check_sphere", 2, 5, 2, 2, 26, 1), ("mesh1k", 2, 8, 2, 2, 27, 3), ("checker", 0, 4, 2, 2, 28, 0),
         ("checker", 2, 4, 2, 1, 29, 3), ("checkersphere", 0, 3, 2, 2, 30, 0), ("checkersphere", 2, 3, 2, 2, 31, 3)]
if __name__ == "__main__":
    out = {}
    for name, integ, depth, sx, sy, seed, sampler in CASES:
        film, _ = ob.OracleScene(SCENES[name]()).render(integrator=integ, max_depth=depth, spp=(sx, sy), seed=seed, sampler=sampler)
        out[f"{name}-{integ}-{depth}-{sx}-{sy}-{seed}-{sampler}"] = film
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "render_golden_r05.npz"), **out)
    print({k: v.shape for k, v in out.items()})
