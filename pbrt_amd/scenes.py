"""Synthetic inputs of BASELINE.json's configs, seeded with PCG32 streams
(the reference's generator, src/core/rng.rs:46-93, vectorised over independent sequences).

  sphere_scene        C1: analytic sphere + point light, direct lighting
  check_sphere_scene  C0-like: scenes/check-sphere.pbrt's mirror sphere over a ground plane
  random_mesh_scene   C2 / C3: N random triangles in a box with a ceiling area light
  cornell_scene       C4: Cornell-style box
"""
import numpy as np

from .api import LIGHT_DISTANT, LIGHT_INFINITE, LIGHT_POINT, MATTE, MIRROR, SceneData, look_at

MESH_SEED = 0x5EED0001

_MULT = np.uint64(0x5851F42D4C957F2D)
_DEFAULT_STATE = np.uint64(0x853C49E6748FEA9B)


class PcgStreams:
    """n independent PCG32 generators, one per sequence index (Rng::new(seq), rng.rs:46-59)."""

    def __init__(self, seqs):
        seqs = np.asarray(seqs, np.uint64)
        self.inc = (seqs << np.uint64(1)) | np.uint64(1)
        self.state = np.zeros_like(seqs)
        self.u32()
        with np.errstate(over="ignore"):
            self.state = self.state + _DEFAULT_STATE
        self.u32()

    def u32(self):  # rng.rs:62-76
        old = self.state
        with np.errstate(over="ignore"):
            self.state = old * _MULT + self.inc
        xs = (((old >> np.uint64(18)) ^ old) >> np.uint64(27)).astype(np.uint32)
        rot = (old >> np.uint64(59)).astype(np.uint32)
        return (xs >> rot) | (xs << ((~rot + np.uint32(1)) & np.uint32(31)))

    def uniform(self):  # rng.rs:91-93
        f = self.u32().astype(np.float32) * np.float32(2.3283064365386963e-10)
        return np.minimum(f, np.float32(1.0) - np.finfo(np.float32).eps)


def _mat(t, k, le=(0, 0, 0)):
    return [t, k[0], k[1], k[2], le[0], le[1], le[2]]


def _quad(p0, p1, p2, p3):
    """two triangles (p0 p1 p2), (p0 p2 p3)"""
    return [p0, p1, p2, p3], [[0, 1, 2], [0, 2, 3]]


def _camera(eye, look, up):
    return look_at(eye, look, up)[1]


def sphere_scene(xres=1024, yres=1024, crop=(0.0, 1.0, 0.0, 1.0)):
    """C1: sphere r=1 at the origin, matte Kd=.5; point light I=10 at (2,2,3); camera of
    scenes/check-sphere.pbrt:1-4."""
    return SceneData(
        materials=np.array([_mat(MATTE, (0.5, 0.5, 0.5))], np.float32),
        lights=np.array([[LIGHT_POINT, 2, 2, 3, 10, 10, 10]], np.float32),
        spheres=np.array([[0, 0, 0, 1, 0]], np.float32),
        cam_to_world=_camera((3, 4, 1.5), (0.5, 0.5, 0), (0, 0, 1)), fov=45.0, xres=xres, yres=yres, crop=crop,
    ).normalized()


def check_sphere_scene(xres=256, yres=256, crop=(0.0, 1.0, 0.0, 1.0)):
    """scenes/check-sphere.pbrt with the pieces this path covers: mirror sphere, matte ground
    quad at z=-1 (constant Kd .45 in place of the checkerboard texture, which is out of scope),
    constant infinite light (.4 .45 .5) and a distant light from (-30, 40, 100) (the file's
    blackbody 3000K x 1.5 replaced by its approximate RGB)."""
    verts, tris = _quad((-20, -20, -1), (20, -20, -1), (20, 20, -1), (-20, 20, -1))
    d = np.array([-30, 40, 100], np.float64)
    d = (d / np.linalg.norm(d)).astype(np.float32)
    return SceneData(
        P=np.array(verts, np.float32), idx=np.array(tris, np.uint32), mat_id=np.array([1, 1], np.uint16),
        materials=np.array([_mat(MIRROR, (0.9, 0.9, 0.9)), _mat(MATTE, (0.45, 0.45, 0.45))], np.float32),
        lights=np.array([[LIGHT_INFINITE, 0, 0, 0, 0.4, 0.45, 0.5],
                         [LIGHT_DISTANT, d[0], d[1], d[2], 1.5 * 1.0, 1.5 * 0.55, 1.5 * 0.25]], np.float32),
        spheres=np.array([[0, 0, 0, 1, 0]], np.float32),
        cam_to_world=_camera((3, 4, 1.5), (0.5, 0.5, 0), (0, 0, 1)), fov=45.0, xres=xres, yres=yres, crop=crop,
    ).normalized()


def random_mesh_scene(n_tris=100_000, xres=1024, yres=1024, crop=(0.0, 1.0, 0.0, 1.0), seed=MESH_SEED):
    """C2 / C3: n random triangles (centre ~ U[-1,1]^3, vertices centre + U[-s,s]^3, s = n^(-1/3))
    inside the box [-2,2]^3 (12 matte triangles, Kd .7) lit by a 1x1 ceiling area light at z=1.99
    (2 emissive triangles, Le 20, facing down).  Triangle i uses material i % 250: every fifth one
    is a mirror (Kr .9), the others matte with Kd ~ U[.2,.8]^3.  Camera at (0,-1.95,0) looking at the
    origin, up +z, fov 60."""
    n = int(n_tris)
    s = np.float32(float(n) ** (-1.0 / 3.0)) if n > 0 else np.float32(1)
    rng = PcgStreams(np.uint64(seed) + np.arange(n, dtype=np.uint64))
    centre = np.stack([rng.uniform() for _ in range(3)], axis=1) * np.float32(2) - np.float32(1)
    offs = np.stack([rng.uniform() for _ in range(9)], axis=1).reshape(n, 3, 3)
    offs = (offs * np.float32(2) - np.float32(1)) * s
    verts = (centre[:, None, :] + offs).astype(np.float32).reshape(-1, 3)
    idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    mat_id = (np.arange(n) % 250).astype(np.uint16)
    mrng = PcgStreams(np.uint64(seed) + np.uint64(1 << 40) + np.arange(250, dtype=np.uint64))
    kd = np.stack([mrng.uniform() for _ in range(3)], axis=1) * np.float32(0.6) + np.float32(0.2)
    mats = [_mat(MIRROR, (0.9, 0.9, 0.9)) if i % 5 == 0 else _mat(MATTE, kd[i]) for i in range(250)]
    BOX, LIGHT = 250, 251
    mats.append(_mat(MATTE, (0.7, 0.7, 0.7)))
    mats.append(_mat(MATTE, (0.0, 0.0, 0.0), (20, 20, 20)))
    b = 2.0
    c = [(-b, -b, -b), (b, -b, -b), (b, b, -b), (-b, b, -b), (-b, -b, b), (b, -b, b), (b, b, b), (-b, b, b)]
    faces = [(0, 1, 2, 3), (4, 5, 6, 7), (0, 1, 5, 4), (3, 2, 6, 7), (0, 3, 7, 4), (1, 2, 6, 5)]
    extra_v, extra_i, extra_m = [], [], []
    for f in faces:
        base = 3 * n + len(extra_v)
        v, t = _quad(*[c[i] for i in f])
        extra_v += v
        extra_i += [[base + a for a in tri] for tri in t]
        extra_m += [BOX, BOX]
    base = 3 * n + len(extra_v)
    v, t = _quad((-0.5, -0.5, 1.99), (-0.5, 0.5, 1.99), (0.5, 0.5, 1.99), (0.5, -0.5, 1.99))  # normal -z
    extra_v += v
    extra_i += [[base + a for a in tri] for tri in t]
    extra_m += [LIGHT, LIGHT]
    return SceneData(
        P=np.concatenate([verts, np.array(extra_v, np.float32)]),
        idx=np.concatenate([idx, np.array(extra_i, np.uint32)]),
        mat_id=np.concatenate([mat_id, np.array(extra_m, np.uint16)]),
        materials=np.array(mats, np.float32),
        cam_to_world=_camera((0, -1.95, 0), (0, 0, 0), (0, 0, 1)), fov=60.0, xres=xres, yres=yres, crop=crop,
    ).normalized()


def cornell_scene(xres=4096, yres=4096, crop=(0.0, 1.0, 0.0, 1.0)):
    """C4: Cornell-style box [-1,1]^3 open towards the camera: white floor / ceiling / back wall, red
    left and green right walls, two white boxes, a 0.5 x 0.5 ceiling light with Le (17, 12, 4)."""
    WHITE, RED, GREEN, LIGHT = 0, 1, 2, 3
    mats = [_mat(MATTE, (0.73, 0.73, 0.73)), _mat(MATTE, (0.65, 0.05, 0.05)), _mat(MATTE, (0.12, 0.45, 0.15)),
            _mat(MATTE, (0, 0, 0), (17, 12, 4))]
    V, I, M = [], [], []

    def add(q, m):
        base = len(V)
        v, t = _quad(*q)
        V.extend(v)
        I.extend([[base + a for a in tri] for tri in t])
        M.extend([m, m])

    add(((-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1)), WHITE)  # floor
    add(((-1, -1, 1), (-1, 1, 1), (1, 1, 1), (1, -1, 1)), WHITE)      # ceiling
    add(((-1, 1, -1), (1, 1, -1), (1, 1, 1), (-1, 1, 1)), WHITE)      # back wall (y = +1)
    add(((-1, -1, -1), (-1, 1, -1), (-1, 1, 1), (-1, -1, 1)), RED)    # left
    add(((1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1)), GREEN)      # right
    add(((-0.25, -0.25, 0.999), (-0.25, 0.25, 0.999), (0.25, 0.25, 0.999), (0.25, -0.25, 0.999)), LIGHT)  # normal -z

    def box(cx, cy, hx, hy, z0, z1, ang):
        ca, sa = np.cos(ang), np.sin(ang)
        cor = [(cx + ca * dx - sa * dy, cy + sa * dx + ca * dy) for dx, dy in ((-hx, -hy), (hx, -hy), (hx, hy), (-hx, hy))]
        lo = [(x, y, z0) for x, y in cor]
        hi = [(x, y, z1) for x, y in cor]
        add((hi[0], hi[1], hi[2], hi[3]), WHITE)
        for k in range(4):
            add((lo[k], lo[(k + 1) % 4], hi[(k + 1) % 4], hi[k]), WHITE)

    box(-0.35, 0.3, 0.3, 0.3, -1.0, 0.2, 0.3)
    box(0.35, -0.25, 0.3, 0.3, -1.0, -0.4, -0.3)
    return SceneData(
        P=np.array(V, np.float32), idx=np.array(I, np.uint32), mat_id=np.array(M, np.uint16),
        materials=np.array(mats, np.float32),
        cam_to_world=_camera((0, -3.6, 0), (0, 0, 0), (0, 0, 1)), fov=40.0, xres=xres, yres=yres, crop=crop,
    ).normalized()


def big_mesh_scene(xres=2048, yres=2048, crop=(0.0, 1.0, 0.0, 1.0), n_tris=12_000_000):
    """The OUT-OF-CACHE workload of bench.py (`--workload big`): the C2 / C3 scene with 12M triangles -- 0.31 GB of
    quantised nodes + 0.58 GB of triangle records, well past the 256 MiB Infinity Cache -- the one regime in which HBM
    bandwidth is a real roof for this path."""
    return random_mesh_scene(n_tris, xres, yres, crop=crop)
