"""ctypes binding of the CPU ORACLE (oracle/pbrt_oracle.h).

TEST INFRASTRUCTURE ONLY: import this from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from pbrt_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "liboracle.so")


class Material(C.Structure):
    _fields_ = [("type", C.c_uint32), ("k", C.c_float * 3), ("le", C.c_float * 3), ("kd_tex", C.c_uint32)]


class Texture(C.Structure):
    _fields_ = [("type", C.c_uint32), ("tex1", C.c_float * 3), ("tex2", C.c_float * 3), ("su", C.c_float), ("sv", C.c_float),
                ("du", C.c_float), ("dv", C.c_float), ("pad", C.c_uint32 * 5)]


class Light(C.Structure):
    _fields_ = [("type", C.c_uint32), ("p", C.c_float * 3), ("c", C.c_float * 3), ("pad", C.c_float)]


class Sphere(C.Structure):
    _fields_ = [("c", C.c_float * 3), ("r", C.c_float), ("mat", C.c_uint32), ("pad", C.c_uint32 * 3)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("P", C.POINTER(C.c_float)), ("idx", C.POINTER(C.c_uint32)), ("mat_id", C.POINTER(C.c_uint16)),
        ("mats", C.POINTER(Material)), ("lights", C.POINTER(Light)), ("spheres", C.POINTER(Sphere)),
        ("n_verts", C.c_uint32), ("n_tris", C.c_uint32), ("n_mats", C.c_uint32), ("n_lights", C.c_uint32),
        ("n_spheres", C.c_uint32),
        ("cam_to_world", C.c_float * 16), ("fov", C.c_float), ("xres", C.c_int32), ("yres", C.c_int32),
        ("crop", C.c_float * 4),
        ("tri_uv", C.POINTER(C.c_float)), ("textures", C.POINTER(Texture)), ("n_textures", C.c_uint32),
    ]


class RenderDesc(C.Structure):
    _fields_ = [
        ("integrator", C.c_uint32), ("max_depth", C.c_uint32), ("spp_x", C.c_uint32), ("spp_y", C.c_uint32),
        ("seed", C.c_uint64), ("rank", C.c_uint32), ("world_size", C.c_uint32), ("flags", C.c_uint32),
        ("sampler", C.c_uint32), ("filter_xwidth", C.c_float), ("filter_ywidth", C.c_float), ("max_sample_luminance", C.c_float),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("camera_rays", C.c_uint64), ("bounce_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
        ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64), ("seconds", C.c_double),
    ]


def build(native=False, force=False):
    """Compile the oracle with g++ (oracle/Makefile).  native=True: -march=native, into a separate
    file, for timing on the machine it is built on."""
    out = LIB_PATH if not native else os.path.join(HERE, "_build", "liboracle_native.so")
    cmd = ["make", "-C", HERE, f"OUT={os.path.relpath(out, HERE)}"]
    if native:
        cmd.append("ARCH=-march=native")
    if force:
        cmd.append("-B")
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL)
    return out


_libs = {}


def lib(native=False):
    if native not in _libs:
        path = LIB_PATH if not native else os.path.join(HERE, "_build", "liboracle_native.so")
        if not os.path.exists(path):
            build(native=native)
        l = C.CDLL(path)
        l.orc_scene_create.restype = C.c_void_p
        l.orc_scene_create.argtypes = [C.POINTER(SceneDesc)]
        l.orc_scene_destroy.argtypes = [C.c_void_p]
        for n in ("orc_bvh_node_count", "orc_bvh_depth", "orc_light_count"):
            getattr(l, n).restype = C.c_uint32
            getattr(l, n).argtypes = [C.c_void_p]
        l.orc_bvh_export.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        l.orc_intersect.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 8 + [C.c_int]
        l.orc_occluded.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 4 + [C.c_int]
        l.orc_tri_accepts.restype = None
        l.orc_tri_accepts.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 6
        l.orc_quad_path_check.restype = None
        l.orc_quad_path_check.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int64] + [C.c_void_p] * 5
        l.orc_camera_ray.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        l.orc_pixel_samples.argtypes = [C.c_void_p, C.POINTER(RenderDesc), C.c_int, C.c_int, C.c_void_p]
        l.orc_render.restype = C.c_int
        l.orc_render.argtypes = [C.c_void_p, C.POINTER(RenderDesc), C.c_void_p, C.POINTER(Stats), C.c_int]
        l.orc_render_acc.restype = C.c_int
        l.orc_render_acc.argtypes = [C.c_void_p, C.POINTER(RenderDesc), C.c_void_p, C.POINTER(Stats), C.c_int]
        l.orc_film_from_acc.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        l.orc_quadratic.argtypes = [C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        l.orc_gamma_correct.restype = C.c_float
        l.orc_gamma_correct.argtypes = [C.c_float]
        l.orc_to_byte.restype = C.c_uint8
        l.orc_to_byte.argtypes = [C.c_float]
        l.orc_film_write_rgb.argtypes = [C.c_void_p, C.c_int64, C.c_float, C.c_void_p]
        l.orc_film_sample_bounds.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        l.orc_film_tile_bounds.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        l.orc_film_physical_extent.argtypes = [C.c_int, C.c_int, C.c_float, C.c_void_p]
        l.orc_film_cropped_bounds.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        l.orc_rng_default_threshold.argtypes = [C.c_uint32, C.c_void_p, C.c_int]
        l.orc_rng_seq_u32.argtypes = [C.c_uint64, C.c_void_p, C.c_int]
        l.orc_rng_seq_float.argtypes = [C.c_uint64, C.c_void_p, C.c_int]
        l.orc_sobol_dims.restype = C.c_int
        l.orc_sobol_matrix.argtypes = [C.c_int, C.c_void_p]
        l.orc_sobol_points.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
        l.orc_halton_points.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        l.orc_halton_points.restype = C.c_uint32
        l.orc_quad_walk.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32] + [C.c_void_p] * 4 + [C.c_int64] + [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 8 + [C.c_int, C.c_void_p]
        _libs[native] = l
    return _libs[native]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def rng_default_u32(n):
    out = np.zeros(n, np.uint32); lib().orc_rng_default_u32(_p(out), n); return out


def rng_default_float(n):
    out = np.zeros(n, np.float32); lib().orc_rng_default_float(_p(out), n); return out


def rng_default_threshold(b, n):
    out = np.zeros(n, np.uint32); lib().orc_rng_default_threshold(b, _p(out), n); return out


def rng_seq_u32(seq, n):
    out = np.zeros(n, np.uint32); lib().orc_rng_seq_u32(seq, _p(out), n); return out


def sobol_matrices():
    """(dims, 52) uint32: the generator matrices the oracle builds from the Joe-Kuo direction numbers."""
    n = lib().orc_sobol_dims()
    out = np.zeros((n, 52), np.uint32)
    for d in range(n):
        lib().orc_sobol_matrix(d, _p(out[d]))
    return out


def halton_points(d, key, n, spp_mask=None):
    """(u float32[n], head uint32[n], b^D): the Halton sampler's dimension d (base b = the d-th prime) at point indices 0 .. n - 1 under
    `key`, in a frame whose largest sample index is spp_mask (default: n rounded up to 2^k, minus 1)"""
    if spp_mask is None:
        spp_mask = (1 << max(n - 1, 0).bit_length()) - 1
    u = np.zeros(n, np.float32); v = np.zeros(n, np.uint32); pw = lib().orc_halton_points(d, key, spp_mask, n, _p(u), _p(v)); return u, v, pw


def sobol_points(n):
    out = np.zeros((n, 2), np.float32); lib().orc_sobol_points(0, n, _p(out)); return out


def rng_seq_float(seq, n):
    out = np.zeros(n, np.float32); lib().orc_rng_seq_float(seq, _p(out), n); return out


def film_cropped_bounds(xres, yres, crop):
    out = np.zeros(4, np.int32); c = np.asarray(crop, np.float32)
    lib().orc_film_cropped_bounds(xres, yres, _p(c), _p(out)); return tuple(int(v) for v in out)


def film_sample_bounds(xres, yres, crop, radius):
    out = np.zeros(4, np.int32); c = np.asarray(crop, np.float32)
    lib().orc_film_sample_bounds(xres, yres, _p(c), radius[0], radius[1], _p(out)); return tuple(int(v) for v in out)


def film_tile_bounds(xres, yres, crop, radius, sb):
    out = np.zeros(4, np.int32); c = np.asarray(crop, np.float32); s = np.asarray(sb, np.int32)
    lib().orc_film_tile_bounds(xres, yres, _p(c), radius[0], radius[1], _p(s), _p(out)); return tuple(int(v) for v in out)


def film_physical_extent(xres, yres, diagonal_mm):
    out = np.zeros(4, np.float32); lib().orc_film_physical_extent(xres, yres, diagonal_mm, _p(out)); return out


def rgb_to_xyz(rgb):
    a = np.asarray(rgb, np.float32); out = np.zeros(3, np.float32); lib().orc_rgb_to_xyz(_p(a), _p(out)); return out


def xyz_to_rgb(xyz):
    a = np.asarray(xyz, np.float32); out = np.zeros(3, np.float32); lib().orc_xyz_to_rgb(_p(a), _p(out)); return out


def film_write_rgb(film_xyzw, scale=1.0):
    f = np.ascontiguousarray(film_xyzw, np.float32)
    out = np.zeros(f.shape[:-1] + (3,), np.float32)
    lib().orc_film_write_rgb(_p(f), f.size // 4, scale, _p(out)); return out


def look_at(pos, look, up):
    a = [np.asarray(v, np.float32) for v in (pos, look, up)]
    m = np.zeros(16, np.float32); mi = np.zeros(16, np.float32)
    lib().orc_look_at(_p(a[0]), _p(a[1]), _p(a[2]), _p(m), _p(mi)); return m.reshape(4, 4), mi.reshape(4, 4)


def matrix_transpose(m):
    a = np.ascontiguousarray(m, np.float32).reshape(16); out = np.zeros(16, np.float32)
    lib().orc_matrix_transpose(_p(a), _p(out)); return out.reshape(4, 4)


def matrix_inverse(m):
    a = np.ascontiguousarray(m, np.float32); out = np.zeros(16, np.float32)
    lib().orc_matrix_inverse(_p(a), _p(out)); return out.reshape(4, 4)


def matrix_mul(a, b):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32); out = np.zeros(16, np.float32)
    lib().orc_matrix_mul(_p(a), _p(b), _p(out)); return out.reshape(4, 4)


def clamp(v, lo, hi):
    """lib.rs:115-127, for floats and for integers"""
    l = lib()
    if isinstance(v, int) and isinstance(lo, int) and isinstance(hi, int):
        l.orc_clamp_i.restype = C.c_long
        l.orc_clamp_i.argtypes = [C.c_long] * 3
        return int(l.orc_clamp_i(v, lo, hi))
    l.orc_clamp_f.restype = C.c_float
    l.orc_clamp_f.argtypes = [C.c_float] * 3
    return float(l.orc_clamp_f(v, lo, hi))


def lerp(t, v1, v2):
    """lib.rs:139-141"""
    l = lib()
    l.orc_lerp.restype = C.c_float
    l.orc_lerp.argtypes = [C.c_float] * 3
    return float(l.orc_lerp(t, v1, v2))


def solve_linear_system_2x2(a, b):
    """transform.rs:59-71 -> [x0, x1] or None"""
    l = lib()
    A = (C.c_float * 4)(a[0][0], a[0][1], a[1][0], a[1][1])
    B = (C.c_float * 2)(*b)
    X = (C.c_float * 2)()
    l.orc_solve_2x2.restype = C.c_int
    return [X[0], X[1]] if l.orc_solve_2x2(A, B, X) else None


def quadratic(a, b, c):
    t0 = C.c_float(); t1 = C.c_float()
    ok = lib().orc_quadratic(a, b, c, C.byref(t0), C.byref(t1))
    return (t0.value, t1.value) if ok else None


def gamma_correct(v):
    return lib().orc_gamma_correct(v)


def to_byte(v):
    return lib().orc_to_byte(v)


def debug_own_box_rule(on):
    """tests only: False = Triangle::Intersect without the own-box rule (the spec until round 5); True = the default"""
    lib().orc_debug_own_box_rule.argtypes = [C.c_int]
    lib().orc_debug_own_box_rule(1 if on else 0)


def quad_path_check(quads, root_box, order, o, d, tri, th):
    """fails[i] = node tests on the way from the root of a product tree to triangle tri[i]'s leaf slot that do not pass for ray i with
    tfar = th[i] (the production step's arithmetic): 0 = every walk reaches the triangle while its best hit is still >= th[i]"""
    quads = np.ascontiguousarray(quads, np.uint32).reshape(-1, 16); order = np.ascontiguousarray(order, np.uint32)
    o = np.ascontiguousarray(o, np.float32).reshape(-1, 3); d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
    tri = np.ascontiguousarray(tri, np.uint32); th = np.ascontiguousarray(th, np.float32); box = np.ascontiguousarray(root_box, np.float32)
    fails = np.zeros(len(o), np.uint32)
    lib().orc_quad_path_check(_p(quads), len(quads), _p(box), _p(order), len(order), len(o), _p(o), _p(d), _p(tri), _p(th), _p(fails))
    return fails


def quad_walk_count_visits(per_node):
    """per_node: a uint64 array with one word per quad node that the following quad_walk calls add their node steps to, or None"""
    lib().orc_quad_walk_count_visits.argtypes = [C.c_void_p]
    lib().orc_quad_walk_count_visits(_p(per_node) if per_node is not None else None)


def quad_walk(quads, root_box, P, idx, order, o, d, tmax, any_hit=False, root_ref=None, n_threads=None, exact_boxes=None):
    """The production walk of the render kernel restated on the CPU (oracle/quad_walk.cpp) over a 4-wide tree a product
    builder emitted: dict(t, prim, b1, b2, occluded, steps, tris, max_stack).  root_ref: None = quad 0 (or no tree when
    there are no quads)."""
    quads = np.ascontiguousarray(quads, np.uint32).reshape(-1, 16)
    P = np.ascontiguousarray(P, np.float32).reshape(-1, 3); idx = np.ascontiguousarray(idx, np.uint32).reshape(-1, 3)
    order = np.ascontiguousarray(order, np.uint32); box = np.ascontiguousarray(root_box, np.float32)
    o = np.ascontiguousarray(o, np.float32).reshape(-1, 3); d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
    tmax = np.ascontiguousarray(tmax, np.float32).reshape(-1); n = o.shape[0]
    if root_ref is None:
        root_ref = 0 if len(quads) else 0xFFFFFFFF
    out = {"t": np.zeros(n, np.float32), "prim": np.zeros(n, np.uint32), "b1": np.zeros(n, np.float32), "b2": np.zeros(n, np.float32),
           "occluded": np.zeros(n, np.uint8), "steps": np.zeros(n, np.uint32), "tris": np.zeros(n, np.uint32)}
    ms = C.c_uint32()
    lib().orc_quad_walk(_p(quads), len(quads), root_ref, _p(box), _p(P), _p(idx), _p(order), n, _p(o), _p(d), _p(tmax), int(any_hit),
                        _p(out["t"]), _p(out["prim"]), _p(out["b1"]), _p(out["b2"]), _p(out["occluded"]), _p(out["steps"]),
                        _p(out["tris"]), C.byref(ms), n_threads or os.cpu_count(),
                        _p(np.ascontiguousarray(exact_boxes, np.float32)) if exact_boxes is not None else None)
    out["max_stack"] = ms.value
    return out


class OracleScene:
    """The oracle's own scene (own BVH build) from the same SceneData arrays the product gets."""

    def __init__(self, sd, native=False):
        from pbrt_amd.api import fill_desc  # layout helper only (struct filling, no computation)
        self.l = lib(native)
        self.sd = sd.normalized()
        desc = SceneDesc()
        keep = fill_desc(desc, self.sd, Material, Light, Sphere, Texture)
        self.h = self.l.orc_scene_create(C.byref(desc))
        del keep

    def close(self):
        if getattr(self, "h", None):
            self.l.orc_scene_destroy(self.h); self.h = None

    __del__ = close

    def bvh(self):
        n = self.l.orc_bvh_node_count(self.h)
        n_prims = int(self.sd.idx.shape[0]) + int(self.sd.spheres.shape[0])  # the triangles, then the spheres (primitive n_tris + s)
        nodes = np.zeros((max(n, 1), 8), np.uint32); order = np.zeros(max(n_prims, 1), np.uint32)
        self.l.orc_bvh_export(self.h, _p(nodes), _p(order))
        return nodes[:n], order[:n_prims], self.l.orc_bvh_depth(self.h)

    def tri_accepts(self, o, d, tmax, tri):
        """Triangle::Intersect (Moeller-Trumbore + the own-box rule) of ray i against triangle tri[i] alone -> (ok[n] uint8, t[n])"""
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3); d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        tmax = np.ascontiguousarray(tmax, np.float32).reshape(-1); tri = np.ascontiguousarray(tri, np.uint32)
        ok = np.zeros(len(o), np.uint8); t = np.zeros(len(o), np.float32)
        self.l.orc_tri_accepts(self.h, len(o), _p(o), _p(d), _p(tmax), _p(tri), _p(ok), _p(t))
        return ok, t

    def light_count(self):
        return self.l.orc_light_count(self.h)

    def intersect(self, o, d, tmax, brute_force=False):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3); d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        tmax = np.ascontiguousarray(tmax, np.float32).reshape(-1); n = o.shape[0]
        t = np.zeros(n, np.float32); prim = np.zeros(n, np.uint32); b1 = np.zeros(n, np.float32); b2 = np.zeros(n, np.float32)
        cnt = np.zeros(2, np.uint64)
        self.l.orc_intersect(self.h, n, _p(o), _p(d), _p(tmax), _p(t), _p(prim), _p(b1), _p(b2), _p(cnt), int(brute_force))
        return t, prim, b1, b2, (int(cnt[0]), int(cnt[1]))

    def occluded(self, o, d, tmax, brute_force=False):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3); d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        tmax = np.ascontiguousarray(tmax, np.float32).reshape(-1); n = o.shape[0]
        hit = np.zeros(n, np.uint8)
        self.l.orc_occluded(self.h, n, _p(o), _p(d), _p(tmax), _p(hit), int(brute_force))
        return hit

    def camera_ray(self, fx, fy):
        o = np.zeros(3, np.float32); d = np.zeros(3, np.float32)
        self.l.orc_camera_ray(self.h, fx, fy, _p(o), _p(d)); return o, d

    def _rd(self, integrator, max_depth, spp, seed, rank, world_size, sampler=0, filter_width=None, max_sample_luminance=0.0):
        r = RenderDesc(); r.integrator = integrator; r.max_depth = max_depth; r.spp_x, r.spp_y = spp
        r.seed = seed; r.rank = rank; r.world_size = world_size
        r.sampler = {"stratified": 0, "sobol": 1, "sobol_nd": 2, "halton": 3}.get(sampler, sampler)
        if filter_width is not None:
            r.filter_xwidth, r.filter_ywidth = filter_width
        r.max_sample_luminance = max_sample_luminance
        return r

    def pixel_samples(self, x, y, integrator=0, max_depth=5, spp=(1, 1), seed=0, sampler=0, filter_width=None, max_sample_luminance=0.0):
        r = self._rd(integrator, max_depth, spp, seed, 0, 1, sampler, filter_width, max_sample_luminance)
        out = np.zeros((spp[0] * spp[1], 3), np.float32)
        self.l.orc_pixel_samples(self.h, C.byref(r), x, y, _p(out)); return out

    def render(self, integrator=0, max_depth=5, spp=(1, 1), seed=0, rank=0, world_size=1, n_threads=None, sampler=0,
               filter_width=None, max_sample_luminance=0.0):
        """sampler: "stratified" / "sobol" (or 0 / 1).  filter_width: box filter radii (None / (0.5, 0.5): the default;
        others: DESIGN.md 3.11 -- with world_size > 1 the film then holds this rank's samples only and ranks combine by
        adding ACCUMULATORS, render_acc)."""
        r = self._rd(integrator, max_depth, spp, seed, rank, world_size, sampler, filter_width, max_sample_luminance)
        w, h = self.sd.crop_size()
        film = np.zeros((h, w, 4), np.float32); st = Stats()
        rc = self.l.orc_render(self.h, C.byref(r), _p(film), C.byref(st), n_threads or os.cpu_count())
        if rc != 0:
            raise ValueError("orc_render: bad render description")
        return film, {k: getattr(st, k) for k, _ in Stats._fields_}

    def render_acc(self, filter_width, integrator=0, max_depth=5, spp=(1, 1), seed=0, rank=0, world_size=1, n_threads=None, sampler=0,
                   max_sample_luminance=0.0):
        """The fixed-point film accumulators (h, w, 4) int64 {r, g, b, samples} of this rank's samples."""
        r = self._rd(integrator, max_depth, spp, seed, rank, world_size, sampler, filter_width, max_sample_luminance)
        w, h = self.sd.crop_size()
        acc = np.zeros((h, w, 4), np.int64); st = Stats()
        rc = self.l.orc_render_acc(self.h, C.byref(r), _p(acc), C.byref(st), n_threads or os.cpu_count())
        if rc != 0:
            raise ValueError("orc_render_acc: bad render description")
        return acc, {k: getattr(st, k) for k, _ in Stats._fields_}


def film_from_acc(acc):
    a = np.ascontiguousarray(acc, np.int64)
    film = np.zeros(a.shape[:-1] + (4,), np.float32)
    lib().orc_film_from_acc(_p(a), a.size // 4, _p(film))
    return film
