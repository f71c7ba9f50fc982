#!/usr/bin/env python3
"""Looks for performance cliffs and parity breaks off the BASELINE configs' beaten track (round 5 found one this way: rays parallel to an axis,
DESIGN.md 3.4).  Part 1: the 1 M-triangle scene under different lights, cameras, materials, integrators, samplers -- frame times side by side.
Part 2: geometry that stresses the tree (flat grids, stacked floors, needles spanning the scene, a tiny cluster in a corner, coincident
triangles, concentric shells) -- build time, frame time, and a 16 x 16 window against the CPU oracle.  Slow rows that are inherent to a BVH
of boxes without spatial splits (needles, coincident triangles) are expected; anything else that is orders of magnitude off is a finding.
Run on the GPU box:  python3 tools/torture_probe.py  (about a minute)"""
import dataclasses
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pbrt_amd  # noqa: E402
from pbrt_amd import INTEGRATOR_DIRECT, INTEGRATOR_PATH_MIS, LIGHT_DISTANT, LIGHT_INFINITE, LIGHT_POINT, scenes  # noqa: E402
from oracle import binding as ob  # noqa: E402

N = 1_000_000


def variant_scene(lights=None, mats=None, cam=None, n=N, res=512):
    """the n-triangle BASELINE mesh scene with other lights / materials / camera"""
    sd = scenes.random_mesh_scene(n, res, res)
    if lights is not None:
        sd.lights = np.array(lights, np.float32).reshape(-1, 7)
    if mats is not None:
        sd.materials = mats(sd.materials.copy())
    if cam is not None:
        sd.cam_to_world = pbrt_amd.look_at(*cam)[1]
    return sd.normalized()


def walk_work(sc, **k):
    """(64-byte fetches, triangle tests) per ray of the PRODUCTION walk (PBRT_HIP_FLAG_WALK_COUNTERS): box-independent, unlike a frame time;
    None for the variants that have no counting instantiation (MIS, the table samplers, a wide filter)"""
    if k.get("integrator") == INTEGRATOR_PATH_MIS or k.get("sampler", "stratified") != "stratified" or k.get("filter_width"):
        return None
    _, wk = sc.render(counters="walk", **k)  # (the counting instantiation of the production walk counts its rays too)
    rays = max(wk["camera_rays"] + wk["bounce_rays"] + wk["shadow_rays"], 1)
    return wk["nodes_visited"] / rays, wk["tris_tested"] / rays


def variant(name, lights=None, mats=None, cam=None, **kw):
    sd = variant_scene(lights, mats, cam)
    k = dict(max_depth=8, spp=(2, 2), seed=1)
    k.update(kw)
    with pbrt_amd.Scene(sd) as sc:
        sc.render(**k)
        _, st = sc.render(**k)
        w = walk_work(sc, **k)
    print(f"{name:48s} kernel {st['kernel_ms']:8.2f} ms  {st['samples'] / st['kernel_ms'] / 1e3:7.1f} Msamples/s"
          + (f"  fetches/ray {w[0]:7.2f} tris/ray {w[1]:5.2f}" if w else ""), flush=True)


def all_mirrors(m):
    m[:, 0] = 1
    return m


# part 1: (name, scene changes, render arguments)
VARIANTS = [
    ("default (area light)", {}, {}),
    ("point light at the origin", dict(lights=[[LIGHT_POINT, 0, 0, 0, 5, 5, 5]]), {}),
    ("point light on round coordinates (0.5, 0.25, 1.5)", dict(lights=[[LIGHT_POINT, 0.5, 0.25, 1.5, 5, 5, 5]]), {}),
    ("sun straight overhead (0, 0, 1)", dict(lights=[[LIGHT_DISTANT, 0, 0, 1, 3, 3, 3]]), {}),
    ("sun along -x", dict(lights=[[LIGHT_DISTANT, -1, 0, 0, 3, 3, 3]]), {}),
    ("sun along (1, 1, 0) / sqrt 2", dict(lights=[[LIGHT_DISTANT, 0.70710678, 0.70710678, 0, 3, 3, 3]]), {}),
    ("sky", dict(lights=[[LIGHT_INFINITE, 0, 0, 0, 1, 1, 1]]), {}),
    ("all mirrors", dict(mats=all_mirrors), {}),
    ("camera looking exactly along +y", dict(cam=((0, -1.95, 0), (0, 1, 0), (0, 0, 1))), {}),
    ("camera looking exactly along -z", dict(cam=((0, 0, 1.9), (0, 0, 0), (0, 1, 0))), {}),
    ("direct lighting", {}, dict(integrator=INTEGRATOR_DIRECT)),
    ("MIS", {}, dict(integrator=INTEGRATOR_PATH_MIS)),
    ("halton", {}, dict(sampler="halton")),
    ("depth 64", {}, dict(max_depth=64)),
    ("depth 0", {}, dict(max_depth=0)),
    ("box filter 2.5", {}, dict(filter_width=(2.5, 2.5))),
]


def with_mesh(P, idx, res=256):
    sd = scenes.random_mesh_scene(64, res, res)  # its box, light, camera and materials; its 64 random triangles are dropped
    base_idx, base_mat, nv = sd.idx[-14:], sd.mat_id[-14:], sd.P.shape[0]
    sd.P = np.concatenate([sd.P, P.astype(np.float32)])
    sd.idx = np.concatenate([base_idx, idx.astype(np.uint32) + nv])
    sd.mat_id = np.concatenate([base_mat, (np.arange(len(idx)) % sd.materials.shape[0]).astype(np.uint16)]).astype(np.uint16)
    sd.tri_uv = np.zeros((0, 6), np.float32)
    return sd.normalized()


def geometry(name, sd):
    kw = dict(max_depth=8, spp=(2, 2), seed=1)
    with pbrt_amd.Scene(sd) as sc:
        bi = sc.build_info()
        sc.render(**kw)
        film, st = sc.render(**kw)
        w = walk_work(sc, **kw)
    x0, y0 = sd.xres // 2, sd.yres // 2
    crop = (0.5, 0.5 + 16 / sd.xres, 0.5, 0.5 + 16 / sd.yres)
    ref, _ = ob.OracleScene(dataclasses.replace(sd, crop=crop).normalized()).render(**kw)
    ok = np.array_equal(film[y0:y0 + 16, x0:x0 + 16].view(np.uint32), ref.view(np.uint32))
    print(f"{name:48s} {sd.idx.shape[0]:8d} tris  build {bi['build_ms']:7.1f} ms  kernel {st['kernel_ms']:8.2f} ms  "
          f"{st['samples'] / st['kernel_ms'] / 1e3:7.1f} Msamples/s  fetches/ray {w[0]:9.2f} tris/ray {w[1]:8.2f}  window {'bit-equal' if ok else 'DIFFERS'}", flush=True)
    return ok


def grid(m, z, base=0):
    g = np.linspace(-1, 1, m + 1)
    X, Y = np.meshgrid(g, g)
    P = np.stack([X.ravel(), Y.ravel(), np.full(X.size, z)], 1)
    q = np.arange(m * m)
    i0 = (q // m) * (m + 1) + q % m
    return P, np.concatenate([np.stack([i0, i0 + 1, i0 + m + 2], 1), np.stack([i0, i0 + m + 2, i0 + m + 1], 1)]) + base


def _soup(n=200_000):
    rng = np.random.default_rng(3)
    c = rng.uniform(-1, 1, (n, 1, 3))
    return (c + rng.uniform(-1, 1, (n, 3, 3)) * n ** (-1 / 3)).reshape(-1, 3), np.arange(3 * n).reshape(n, 3)


def _floors():
    parts = [grid(70, -0.95 + 0.1 * k, k * 71 * 71) for k in range(20)]
    return np.concatenate([p for p, _ in parts]), np.concatenate([i for _, i in parts])


def _needles(n=200_000):
    rng = np.random.default_rng(4)
    a, b = rng.uniform(-1, 1, (n, 3)), rng.uniform(-1, 1, (n, 3))
    return np.stack([a, b, a + rng.normal(size=(n, 3)) * 1e-3], 1).reshape(-1, 3), np.arange(3 * n).reshape(n, 3)


def _cluster(n=200_000):
    rng = np.random.default_rng(5)
    c = rng.uniform(0.899, 0.9, (n, 1, 3))
    return (c + rng.uniform(-1, 1, (n, 3, 3)) * 1e-5).reshape(-1, 3), np.arange(3 * n).reshape(n, 3)


def _coincident(n=20_000):
    return np.tile(np.array([[-0.5, -0.5, 0.1], [0.5, -0.5, 0.1], [0, 0.5, 0.1]]), (n, 1)), np.arange(3 * n).reshape(n, 3)


def _shells():
    Ps, Is = [], []
    for k in range(40):
        r = 0.2 + 0.02 * k
        T, Ph = np.meshgrid(np.linspace(0, np.pi, 36), np.linspace(0, 2 * np.pi, 72))
        V = np.stack([r * np.sin(T) * np.cos(Ph), r * np.sin(T) * np.sin(Ph), r * np.cos(T)], -1).reshape(-1, 3)
        q = np.arange(71 * 35)
        i0 = (q // 35) * 36 + q % 35
        Ps.append(V)
        Is.append(np.concatenate([np.stack([i0, i0 + 1, i0 + 37], 1), np.stack([i0, i0 + 37, i0 + 36], 1)]) + k * V.shape[0])
    return np.concatenate(Ps), np.concatenate(Is)


# part 2: (name, () -> (P, idx)); "(inherent)" = slow by the nature of a BVH of boxes without spatial splits
GEOMETRIES = [
    ("random soup", _soup),
    ("flat grid of quads in z = 0", lambda: grid(316, 0.0)),
    ("20 stacked floors of quads", _floors),
    ("needles spanning the scene (inherent)", _needles),
    ("a cluster of 1e-5 triangles in a corner", _cluster),
    ("20 000 coincident triangles (inherent)", _coincident),
    ("40 concentric spherical shells", _shells),
]


def main():
    print("-- part 1: the 1 M-triangle scene (512 x 512, 4 spp, depth 8 unless said)")
    for name, scene_kw, render_kw in VARIANTS:
        variant(name, **scene_kw, **render_kw)
    print("-- part 2: geometry (256 x 256, 4 spp, depth 8), each with a 16 x 16 window against the oracle")
    ok = True
    for name, make in GEOMETRIES:
        ok &= geometry(name, with_mesh(*make()))
    print("all windows bit-equal:", bool(ok))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
