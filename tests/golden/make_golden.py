"""Generates tests/golden/render_golden.npz with the CPU oracle (self-generated fixtures: the
reference has no renderer to generate them from, SURVEY.md section 8c "parity unpinned").
Run from the repo root:  python tests/golden/make_golden.py
Key format: scene-integrator-maxdepth-sppx-sppy-seed-sampler (sampler: 0 stratified, 1 sobol).
Regenerated in round 2 (r02f): a pixel's samples run in K = sample_chunks(spp) chunks with their own RNG streams."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import binding as ob  # noqa: E402
from util import SMALL_SCENES  # noqa: E402

CASES = [("mesh1k", 0, 8, 2, 2, 11, 0), ("cornell", 0, 16, 5, 3, 12, 0), ("check_sphere", 0, 5, 2, 1, 13, 0),
         ("sphere", 1, 5, 4, 4, 14, 0), ("cornell", 0, 6, 4, 4, 15, 1), ("mesh1k", 0, 8, 3, 2, 16, 1)]
out = {}
for name, integ, depth, sx, sy, seed, sampler in CASES:
    film, _ = ob.OracleScene(SMALL_SCENES[name]()).render(integrator=integ, max_depth=depth, spp=(sx, sy), seed=seed, sampler=sampler)
    out[f"{name}-{integ}-{depth}-{sx}-{sy}-{seed}-{sampler}"] = film
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "render_golden.npz"), **out)
print({k: v.shape for k, v in out.items()})
