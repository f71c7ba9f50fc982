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
