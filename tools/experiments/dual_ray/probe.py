#!/usr/bin/env python3
"""EXPERIMENT: two rays per lane in a traversal-only kernel.  Builds dual_ray.hip into its own
library, makes a path-tracing-like ray set of a BASELINE mesh scene (camera rays, cosine bounces, shadow rays to the ceiling
light; hits from the product's own intersect), and times the product's intersect_kernel (mode 0), the experiment's kernel with
one ray per lane (mode 1) and with two (mode 2) on the same rays; results must be identical.
usage (GPU box): probe.py [n_tris] [camera_res]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
import pbrt_amd  # noqa: E402
from oracle import binding as ob  # noqa: E402
from pbrt_amd import scenes  # noqa: E402

LIB = os.path.join(HERE, "libdual_ray.so")


def build():
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-mllvm",
           "-amdgpu-sdwa-peephole=0", "--offload-arch=gfx950", '-DPBRT_HIP_BUILD_ID="exp"', "-x", "hip", os.path.join(HERE, "dual_ray.hip"), "-o", LIB]
    subprocess.run(cmd, check=True)


def rays_of(sd, sc, res, bounces=5, seed=1):
    rng = np.random.default_rng(seed)
    osc = ob.OracleScene(sd)
    o00, d00 = osc.camera_ray(0.5, 0.5)
    _, d10 = osc.camera_ray(sd.xres - 0.5, 0.5)
    _, d01 = osc.camera_ray(0.5, sd.yres - 0.5)
    _, d11 = osc.camera_ray(sd.xres - 0.5, sd.yres - 0.5)
    u, v = np.meshgrid((np.arange(res) + 0.5) / res, (np.arange(res) + 0.5) / res)
    u, v = u.ravel()[:, None], v.ravel()[:, None]
    d = (1 - u) * (1 - v) * d00 + u * (1 - v) * d10 + (1 - u) * v * d01 + u * v * d11
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    o = np.broadcast_to(np.asarray(o00, np.float32), d.shape).copy()
    closest = [(o, d, np.full(len(o), np.inf, np.float32))]
    shadow = []
    P = sd.P.reshape(-1, 3)
    for _ in range(bounces):
        o, d, tm = closest[-1]
        t, prim, b1, b2 = sc.intersect(o, d, tm)[:4]
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
        lp = np.stack([rng.random(len(po)) - 0.5, rng.random(len(po)) - 0.5, np.full(len(po), 1.99)], 1)
        dv = lp - po
        dist = np.linalg.norm(dv, axis=1)
        shadow.append((po, (dv / dist[:, None]).astype(np.float32), (dist * 0.9999).astype(np.float32)))
        u1, u2 = rng.random(len(po)), rng.random(len(po))
        r, phi = np.sqrt(u1), 2 * np.pi * u2
        a = np.where(np.abs(n[:, :1]) > 0.9, [[0, 1, 0]], [[1, 0, 0]])
        tx = np.cross(n, a)
        tx /= np.linalg.norm(tx, axis=1, keepdims=True)
        ty = np.cross(n, tx)
        nd = tx * (r * np.cos(phi))[:, None] + ty * (r * np.sin(phi))[:, None] + n * np.sqrt(np.maximum(0, 1 - u1))[:, None]
        closest.append((po, nd.astype(np.float32), np.full(len(po), np.inf, np.float32)))
    cat = lambda sets: tuple(np.ascontiguousarray(np.concatenate([s[k] for s in sets])) for k in range(3))
    return cat(closest), cat(shadow)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    res = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    newest = max(os.path.getmtime(os.path.join(HERE, "dual_ray.hip")), os.path.getmtime(os.path.join(ROOT, "pbrt_amd", "csrc", "kernels.hip")),
                 os.path.getmtime(os.path.join(ROOT, "pbrt_amd", "csrc", "device_types.h")))
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < newest:
        build()
    lib = C.CDLL(LIB)
    fp, up = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    lib.exp_dual_intersect.argtypes = [C.c_void_p, C.c_int64, fp, fp, fp, fp, up, fp, fp, C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int, C.c_uint32,
                                       C.c_uint32, C.c_uint32, fp, C.POINTER(C.c_uint64)]
    sd = scenes.random_mesh_scene(n, 256, 256).normalized()
    with pbrt_amd.Scene(sd, builder="gpu") as sc:
        (co, cd, ct), (so, sdd, stm) = rays_of(sd, sc, res)
        print(f"{n} triangles; {len(co)} closest-hit rays, {len(so)} shadow rays; stack bound {sc.info()['quad_stack_need']}")

        def run(o, d, tm, any_hit, mode, steps=3, min_done=16, min_walkers=36, min_parked=16):
            m = len(o)
            t, prim, b1, b2 = np.zeros(m, np.float32), np.zeros(m, np.uint32), np.zeros(m, np.float32), np.zeros(m, np.float32)
            occ = np.zeros(m, np.uint8)
            ms = C.c_float()
            probe = (C.c_uint64 * 8)()
            f = lambda a: a.ctypes.data_as(fp)
            rc = lib.exp_dual_intersect(sc._h, m, f(o), f(d), f(tm), f(t), prim.ctypes.data_as(up), f(b1), f(b2), occ.ctypes.data_as(C.POINTER(C.c_uint8)),
                                        int(any_hit), mode, steps, min_done, min_walkers, min_parked, C.byref(ms), probe)
            assert rc == 0, rc
            return (occ if any_hit else (t, prim, b1, b2)), ms.value, list(probe)[:5]

        for name, (o, d, tm), any_hit in (("closest-hit", (co, cd, ct), False), ("shadow", (so, sdd, stm), True)):
            ref, ms0, _ = run(o, d, tm, any_hit, 0)
            print(f"{name}: product intersect_kernel                  {ms0:8.2f} ms  {len(o) / ms0 / 1e3:7.1f} Mrays/s")
            for mode, label in ((5, "one ray per lane, 5 waves per SIMD"), (3, "one ray per lane, 6 waves per SIMD"), (4, "one ray per lane, 8 waves per SIMD")):
                out, ms, pr = run(o, d, tm, any_hit, mode, 3, 1, 36)
                same = np.array_equal(out, ref) if any_hit else all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(out, ref))
                print(f"{name}: experiment, {label}: {ms:8.2f} ms  {len(o) / ms / 1e3:7.1f} Mrays/s  ({'identical' if same else 'RESULTS DIFFER'}; node-step passes {pr[0]} with {pr[1] / max(pr[0], 1):.1f} lanes)")
            for mode, label in ((1, "one ray per lane"), (2, "TWO rays per lane")):
                for steps in (3,):
                    for min_done, min_walkers in ((1, 36),):
                        out, ms, pr = run(o, d, tm, any_hit, mode, steps, min_done, min_walkers)
                        same = np.array_equal(out, ref) if any_hit else all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(out, ref))
                        lanes = pr[1] / max(pr[0], 1)
                        leaf = pr[3] / max(pr[2], 1)
                        print(f"{name}: experiment, {label}, {steps} steps/check, min_done {min_done:2d} min_walkers {min_walkers}: {ms:8.2f} ms  {len(o) / ms / 1e3:7.1f} Mrays/s  "
                              f"({'identical' if same else 'RESULTS DIFFER'}; node-step passes {pr[0]} with {lanes:.1f} lanes, leaf passes {pr[2]} with {leaf:.1f}, swap blocks {pr[4]})")


if __name__ == "__main__":
    main()
