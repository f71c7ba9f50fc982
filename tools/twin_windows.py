#!/usr/bin/env python3
"""Windows of the BASELINE scenes at THEIR sizes from three programs: tests/independent_twin.py (float64 numpy, the spec of DESIGN.md section 3
with the same random numbers, EVERY ray against EVERY triangle -- a million of them for C3), the CPU oracle (its BVH) and, where a GPU is
present, the HIP path (the tree built and optimised on the device).  No GPU needed for the first two.  usage: tools/twin_windows.py [--gpu]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import independent_twin as tw  # noqa: E402
from oracle import binding as ob  # noqa: E402
from pbrt_amd import scenes  # noqa: E402


def main():
    gpu = "--gpu" in sys.argv
    cases = [("C3's scene (1 000 014 triangles, 2048 x 2048)", lambda crop: scenes.random_mesh_scene(1_000_000, 2048, 2048, crop=crop), 2048, (1000, 700, 3, 3), dict(integrator=0, max_depth=8, spp=(4, 4), seed=0)),
             ("C2's scene (100 014 triangles, 1024 x 1024)", lambda crop: scenes.random_mesh_scene(100_000, 1024, 1024, crop=crop), 1024, (500, 340, 4, 4), dict(integrator=0, max_depth=8, spp=(8, 8), seed=0)),
             ("C4's scene (Cornell-style, 4096 x 4096), depth 16", lambda crop: scenes.cornell_scene(4096, 4096, crop=crop), 4096, (2000, 1500, 16, 16), dict(integrator=0, max_depth=16, spp=(16, 16), seed=0))]
    for name, make, res, (x0, y0, w, h), kw in cases:
        crop = (x0 / res, (x0 + w) / res, y0 / res, (y0 + h) / res)
        t0 = time.time()
        twin = tw.render(make((0.0, 1.0, 0.0, 1.0)), window=(x0, y0, w, h), **kw)
        t1 = time.time()
        sd = make(crop)
        film, _ = ob.OracleScene(sd).render(**kw)
        rel = np.abs(twin[..., :3] - film[..., :3]) / np.maximum(np.abs(film[..., :3]), 1e-3 * film[..., :3].max())
        line = (f"{name}: window {w} x {h} at ({x0}, {y0}), {kw['spp'][0] * kw['spp'][1]} spp: twin {t1 - t0:.0f} s; oracle vs twin PSNR {tw.psnr_db(twin, film):.1f} dB, "
                f"{(rel.max(-1) < 1e-4).mean() * 100:.1f} % of the pixels to 1e-4")
        if gpu:
            import pbrt_amd
            with pbrt_amd.Scene(sd) as sc:
                hip, _ = sc.render(**kw)
            line += f"; HIP vs twin PSNR {tw.psnr_db(twin, hip):.1f} dB; HIP == oracle bit for bit: {np.array_equal(hip.view(np.uint32), film.view(np.uint32))}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
