#!/usr/bin/env python3
"""EXPERIMENT (round 4): a pool of rays per wave, dealt to the lanes per pass, in a traversal-only
kernel.  Builds ray_pool.hip into its own library, makes the path-tracing-like ray set of tools/experiments/dual_ray/probe.py on a
BASELINE mesh scene, and times the product's intersect_kernel (mode 0) against the pooled kernel's variants on the same rays;
results must be identical.   usage (GPU box): probe.py [n_tris] [camera_res]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "dual_ray"))
import pbrt_amd  # noqa: E402
from pbrt_amd import scenes  # noqa: E402
from probe import rays_of  # noqa: E402  (tools/experiments/dual_ray/probe.py)

LIB = os.path.join(HERE, "libray_pool.so")


def build():
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-mllvm",
           "-amdgpu-sdwa-peephole=0", "--offload-arch=gfx950", '-DPBRT_HIP_BUILD_ID="exp"', "-x", "hip", os.path.join(HERE, "ray_pool.hip"), "-o", LIB]
    subprocess.run(cmd, check=True)


MODES = ((1, "pool of 128 rays, 12 LDS stack rows, 3 steps per deal"), (2, "pool of 128, 12 rows, 2 steps per deal"), (4, "pool of 128, 12 rows, 4 steps per deal"),
         (3, "pool of 192, 8 rows, 3 steps per deal"), (5, "pool of 64 = one ray per lane, dealt (control for the dealing's cost)"))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    res = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    if "--build" in sys.argv or not os.path.exists(LIB):
        build()
        if "--build" in sys.argv:
            return
    lib = C.CDLL(LIB)
    fp, up = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    lib.exp_pool_intersect.argtypes = [C.c_void_p, C.c_int64, fp, fp, fp, fp, up, fp, fp, C.POINTER(C.c_uint8), C.c_int, C.c_int, fp, C.POINTER(C.c_uint64)]
    sd = scenes.random_mesh_scene(n, 256, 256).normalized()
    with pbrt_amd.Scene(sd, builder="gpu") as sc:
        (co, cd, ct), (so, sdd, stm) = rays_of(sd, sc, res)
        print(f"{n} triangles; {len(co)} closest-hit rays, {len(so)} shadow rays; stack bound {sc.info()['quad_stack_need']}")

        def run(o, d, tm, any_hit, mode):
            m = len(o)
            t, prim, b1, b2 = np.zeros(m, np.float32), np.zeros(m, np.uint32), np.zeros(m, np.float32), np.zeros(m, np.float32)
            occ = np.zeros(m, np.uint8)
            ms = C.c_float()
            probe = (C.c_uint64 * 8)()
            f = lambda a: a.ctypes.data_as(fp)
            rc = lib.exp_pool_intersect(sc._h, m, f(o), f(d), f(tm), f(t), prim.ctypes.data_as(up), f(b1), f(b2), occ.ctypes.data_as(C.POINTER(C.c_uint8)),
                                        int(any_hit), mode, C.byref(ms), probe)
            assert rc == 0, rc
            return (occ if any_hit else (t, prim, b1, b2)), ms.value, list(probe)[:5]

        for name, (o, d, tm), any_hit in (("closest-hit", (co, cd, ct), False), ("shadow", (so, sdd, stm), True)):
            ref, ms0, _ = run(o, d, tm, any_hit, 0)
            print(f"{name}: product intersect_kernel (8 waves per SIMD): {ms0:8.2f} ms  {len(o) / ms0 / 1e3:7.1f} Mrays/s")
            for mode, label in MODES:
                out, ms, pr = run(o, d, tm, any_hit, mode)
                same = np.array_equal(out, ref) if any_hit else all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(out, ref))
                print(f"{name}: {label}: {ms:8.2f} ms  {len(o) / ms / 1e3:7.1f} Mrays/s  ({'identical' if same else 'RESULTS DIFFER'}; "
                      f"node-step passes {pr[0]} with {pr[1] / max(pr[0], 1):.1f} lanes, triangle passes {pr[2]} with {pr[3] / max(pr[2], 1):.1f}, deals {pr[4]})")


if __name__ == "__main__":
    main()
