"""Shared helpers for the parity tests."""
import numpy as np

from pbrt_amd import scenes


def random_rays(n, seed, sd=None, inside=2.0):
    """n rays: origins uniform in [-inside, inside]^3, directions uniform on the sphere; a few
    axis-aligned and degenerate ones appended (zero components -> infinite inverse direction)."""
    rng = np.random.default_rng(seed)
    o = rng.uniform(-inside, inside, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    tmax = np.full(n, np.inf, np.float32)
    tmax[::7] = rng.uniform(0.05, 3.0, len(tmax[::7])).astype(np.float32)
    special_d = np.array([[1, 0, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1], [1, 1, 0], [0, -0.0, 1]], np.float32)
    special_o = np.array([[-1.5, 0.1, 0.2], [0.3, 1.5, -0.2], [0.1, 0.2, -1.9], [0.0, 0.0, 1.0], [-1, -1, 0.5],
                          [0.25, 0.25, -1]], np.float32)
    o = np.concatenate([o, special_o])
    d = np.concatenate([d, special_d])
    tmax = np.concatenate([tmax, np.full(len(special_d), np.inf, np.float32)])
    return o, d, tmax


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def assert_bit_equal(a, b, what=""):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    if a.dtype.kind == "f":
        same = (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))
    else:
        same = a == b
    if not same.all():
        bad = np.argwhere(~same)
        i = tuple(bad[0])
        raise AssertionError(f"{what}: {len(bad)} of {a.size} elements differ; first at {i}: {a[i]!r} vs {b[i]!r}")


SMALL_SCENES = {
    "mesh1k": lambda: scenes.random_mesh_scene(1000, 48, 40),
    "mesh20k": lambda: scenes.random_mesh_scene(20000, 64, 64),
    "cornell": lambda: scenes.cornell_scene(64, 64),
    "sphere": lambda: scenes.sphere_scene(64, 64),
    "check_sphere": lambda: scenes.check_sphere_scene(64, 48),
}


def tie_scene(xres=48, yres=48):
    """Geometry built to produce EXACT ties and degenerate cases: every triangle of a small random
    mesh duplicated (same vertices, different primitive id and material), two coplanar overlapping
    quads, zero-area and needle triangles, a triangle lying in an axis-aligned bounding plane."""
    from pbrt_amd.api import MATTE, MIRROR, SceneData
    base = scenes.random_mesh_scene(300, xres, yres)
    n = base.idx.shape[0]
    P = [base.P]
    idx = [base.idx, base.idx.copy()]                    # exact duplicates: equal t, lower id must win
    mat = [base.mat_id, ((base.mat_id.astype(np.int64) + 7) % 250).astype(np.uint16)]
    extra_v = np.array([
        [-0.8, -0.8, -0.5], [0.8, -0.8, -0.5], [0.8, 0.8, -0.5], [-0.8, 0.8, -0.5],   # quad A at z=-0.5
        [-0.4, -0.4, -0.5], [1.2, -0.4, -0.5], [1.2, 1.2, -0.5], [-0.4, 1.2, -0.5],   # quad B, coplanar, overlapping
        [0.1, 0.1, 0.3], [0.1, 0.1, 0.3], [0.1, 0.1, 0.3],                             # zero-area (a point)
        [0.0, 0.0, 0.6], [0.5, 0.5, 0.6], [1.0, 1.0, 0.6],                             # zero-area (collinear)
        [-1.0, 0.2, 0.9], [1.0, 0.2, 0.9], [0.0, 0.2000001, 0.9],                      # needle
    ], np.float32)
    b = base.P.shape[0]
    extra_i = np.array([[b, b + 1, b + 2], [b, b + 2, b + 3], [b + 4, b + 5, b + 6], [b + 4, b + 6, b + 7],
                        [b + 8, b + 9, b + 10], [b + 11, b + 12, b + 13], [b + 14, b + 15, b + 16]], np.uint32)
    extra_m = np.array([1, 1, 5, 5, 2, 3, 4], np.uint16)  # quad A matte, quad B mirror (id 5 is a mirror)
    return SceneData(P=np.concatenate(P + [extra_v]), idx=np.concatenate(idx + [extra_i]),
                     mat_id=np.concatenate(mat + [extra_m]), materials=base.materials, cam_to_world=base.cam_to_world,
                     fov=base.fov, xres=xres, yres=yres).normalized()


SMALL_SCENES["ties"] = tie_scene


def deep_tree_scene(xres=32, yres=32, k=400):
    """Triangles whose positions AND sizes shrink geometrically (x_i = 2^(-i/2)): the 16-bucket SAH
    peels a handful off per level, so the tree runs into the depth guard (39 levels for k = 400) --
    the case that needs the 64-entry LDS stack of the exact walk and the HBM overflow area of the
    production walk's stack (3 entries per quad level > 40 LDS entries)."""
    from pbrt_amd.api import LIGHT_INFINITE, MATTE, SceneData, look_at
    i = np.arange(k)
    x = (0.5 ** (i / 2.0)).astype(np.float32)
    z = (i / k * 0.5).astype(np.float32)
    P = np.concatenate([np.stack([x, x * 0, z], 1), np.stack([x * 1.2, x * 0, z], 1), np.stack([x, x * 0.2, z], 1)]).astype(np.float32)
    idx = np.stack([i, k + i, 2 * k + i], 1).astype(np.uint32)
    return SceneData(P=P, idx=idx, mat_id=np.zeros(k, np.uint16), materials=np.array([[MATTE, .6, .5, .4, 0, 0, 0]], np.float32),
                     lights=np.array([[LIGHT_INFINITE, 0, 0, 0, 1, 1, 1]], np.float32),
                     cam_to_world=look_at((0.3, 0.05, 2.0), (0.3, 0.05, 0), (0, 1, 0))[1], fov=40, xres=xres, yres=yres).normalized()


SMALL_SCENES["deep"] = deep_tree_scene
