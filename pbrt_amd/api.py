"""Host-side mirror of the render call: what `PbrtAPI::world_end` (reference src/core/api.rs:432-473)
would do with its RenderOptions (api.rs:201-224) -- build the scene on the device and render it.

Everything that computes goes through the C ABI of include/pbrt_hip.h into the HIP kernels.
"""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from ._lib import Light, Material, RenderDesc, SceneDesc, Sphere, Stats, Texture, check, lib

MATTE, MIRROR = 0, 1
LIGHT_POINT, LIGHT_DISTANT, LIGHT_INFINITE = 0, 1, 2
INTEGRATOR_PATH, INTEGRATOR_DIRECT, INTEGRATOR_PATH_MIS = 0, 1, 2  # 2: the path integrator with MIS
FLAG_COUNTERS = 1
FLAG_WALK_COUNTERS = 2
SCENE_GPU_BUILD = 1  # pbrt_hip_scene_create_ex flags
SCENE_OPTIMIZED_TREE = 2
SCENE_PLAIN_TREE = 4
SCENE_HOST_BUILD = 8
BUILDERS = {"host": SCENE_HOST_BUILD, "gpu": SCENE_GPU_BUILD, "gpu-plain": SCENE_GPU_BUILD | SCENE_PLAIN_TREE, "host-optimized": SCENE_OPTIMIZED_TREE}


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


@dataclass
class SceneData:
    """Plain arrays describing a scene: RenderOptions + the geometry the reference never stores
    (api.rs:220-223).  `materials` rows: (type, kr, kg, kb, ler, leg, leb); `lights` rows:
    (type, px, py, pz, cr, cg, cb); `spheres` rows: (cx, cy, cz, r, material).  Textured materials: `mat_tex[i]` =
    0 or 1 + the row of `textures` that is material i's Kd, `textures` rows: (type 0 = checkerboard, tex1 rgb, tex2 rgb, su, sv, du,
    dv), `tri_uv` rows: (u0, v0, u1, v1, u2, v2) per triangle (needed when a triangle's material is textured)."""
    P: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.float32))
    idx: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.uint32))
    mat_id: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.uint16))
    materials: np.ndarray = field(default_factory=lambda: np.zeros((0, 7), np.float32))
    lights: np.ndarray = field(default_factory=lambda: np.zeros((0, 7), np.float32))
    spheres: np.ndarray = field(default_factory=lambda: np.zeros((0, 5), np.float32))
    cam_to_world: np.ndarray = field(default_factory=lambda: np.eye(4, dtype=np.float32))
    fov: float = 45.0
    xres: int = 64
    yres: int = 64
    crop: tuple = (0.0, 1.0, 0.0, 1.0)
    mat_tex: np.ndarray = field(default_factory=lambda: np.zeros((0,), np.uint32))
    textures: np.ndarray = field(default_factory=lambda: np.zeros((0, 11), np.float32))
    tri_uv: np.ndarray = field(default_factory=lambda: np.zeros((0, 6), np.float32))

    def normalized(self):
        self.P = np.ascontiguousarray(self.P, np.float32).reshape(-1, 3)
        self.idx = np.ascontiguousarray(self.idx, np.uint32).reshape(-1, 3)
        self.mat_id = np.ascontiguousarray(self.mat_id, np.uint16).reshape(-1)
        self.materials = np.ascontiguousarray(self.materials, np.float32).reshape(-1, 7)
        self.lights = np.ascontiguousarray(self.lights, np.float32).reshape(-1, 7)
        self.spheres = np.ascontiguousarray(self.spheres, np.float32).reshape(-1, 5)
        self.cam_to_world = np.ascontiguousarray(self.cam_to_world, np.float32).reshape(4, 4)
        self.mat_tex = np.ascontiguousarray(self.mat_tex, np.uint32).reshape(-1)
        if self.mat_tex.shape[0] != self.materials.shape[0]:
            assert self.mat_tex.shape[0] == 0
            self.mat_tex = np.zeros(self.materials.shape[0], np.uint32)
        self.textures = np.ascontiguousarray(self.textures, np.float32).reshape(-1, 11)
        self.tri_uv = np.ascontiguousarray(self.tri_uv, np.float32).reshape(-1, 6)
        assert self.idx.shape[0] == self.mat_id.shape[0] and self.tri_uv.shape[0] in (0, self.idx.shape[0])
        return self

    def crop_size(self):
        b = film_cropped_bounds(self.xres, self.yres, self.crop)
        return max(b[2] - b[0], 0), max(b[3] - b[1], 0)


def fill_desc(desc, sd, mat_t, light_t, sphere_t, tex_t=None):
    """Fill a SceneDesc-shaped ctypes struct from a SceneData; returns the keep-alive list."""
    sd.normalized()
    mats = (mat_t * max(len(sd.materials), 1))()
    for i, m in enumerate(sd.materials):
        mats[i].type = int(m[0])
        mats[i].k[:] = [float(x) for x in m[1:4]]
        mats[i].le[:] = [float(x) for x in m[4:7]]
        mats[i].kd_tex = int(sd.mat_tex[i])
    texs = None
    if len(sd.textures):
        tex_t = tex_t or dict(desc._fields_)["textures"]._type_  # (the Texture class of the caller's own struct mirror)
        texs = (tex_t * len(sd.textures))()
        for i, t in enumerate(sd.textures):
            texs[i].type = int(t[0])
            texs[i].tex1[:] = [float(x) for x in t[1:4]]
            texs[i].tex2[:] = [float(x) for x in t[4:7]]
            texs[i].su, texs[i].sv, texs[i].du, texs[i].dv = (float(x) for x in t[7:11])
        desc.textures = texs
    desc.n_textures = len(sd.textures)
    if sd.tri_uv.shape[0]:
        desc.tri_uv = _fp(sd.tri_uv)
    lights = (light_t * max(len(sd.lights), 1))()
    for i, l in enumerate(sd.lights):
        lights[i].type = int(l[0])
        lights[i].p[:] = [float(x) for x in l[1:4]]
        lights[i].c[:] = [float(x) for x in l[4:7]]
    spheres = (sphere_t * max(len(sd.spheres), 1))()
    for i, s in enumerate(sd.spheres):
        spheres[i].c[:] = [float(x) for x in s[0:3]]
        spheres[i].r = float(s[3])
        spheres[i].mat = int(s[4])
    desc.P = _fp(sd.P)
    desc.idx = _u32p(sd.idx)
    desc.mat_id = sd.mat_id.ctypes.data_as(C.POINTER(C.c_uint16))
    desc.mats = mats
    desc.lights = lights
    desc.spheres = spheres
    desc.n_verts = sd.P.shape[0]
    desc.n_tris = sd.idx.shape[0]
    desc.n_mats = len(sd.materials)
    desc.n_lights = len(sd.lights)
    desc.n_spheres = len(sd.spheres)
    desc.cam_to_world[:] = [float(x) for x in sd.cam_to_world.reshape(-1)]
    desc.fov = float(sd.fov)
    desc.xres = int(sd.xres)
    desc.yres = int(sd.yres)
    desc.crop[:] = [float(x) for x in sd.crop]
    return [mats, lights, spheres, texs, sd]


SAMPLERS = {"stratified": 0, "sobol": 1, "sobol_nd": 2, "halton": 3}  # "sobol": the padded (0,2)-sequence sampler (3.10); "sobol_nd": Sobol' proper (3.12); "halton": 3.13


def make_render_desc(desc_t, integrator=INTEGRATOR_PATH, max_depth=5, spp=(1, 1), seed=0, rank=0, world_size=1,
                     flags=0, sampler="stratified", filter_width=(0.0, 0.0), max_sample_luminance=0.0):
    r = desc_t()
    r.sampler = SAMPLERS[sampler] if isinstance(sampler, str) else int(sampler)
    filter_width = filter_width or (0.0, 0.0)
    r.filter_xwidth, r.filter_ywidth = float(filter_width[0]), float(filter_width[1])
    r.max_sample_luminance = float(max_sample_luminance)
    r.integrator = integrator
    r.max_depth = max_depth
    r.spp_x, r.spp_y = int(spp[0]), int(spp[1])
    r.seed = seed
    r.rank = rank
    r.world_size = world_size
    r.flags = flags
    return r


def device_count():
    return lib().pbrt_hip_device_count()


def build_id():
    """pbrt_hip_build_id: hash of the sources and flags the loaded library was built from"""
    return lib().pbrt_hip_build_id().decode()


def look_at(pos, look, up):
    """Transform::look_at (transform.rs:485-520) -> (world_to_camera, camera_to_world) 4x4 float32."""
    m = np.zeros(16, np.float32)
    mi = np.zeros(16, np.float32)
    a = [np.asarray(v, np.float32) for v in (pos, look, up)]
    lib().pbrt_hip_look_at(_fp(a[0]), _fp(a[1]), _fp(a[2]), _fp(m), _fp(mi))
    return m.reshape(4, 4), mi.reshape(4, 4)


def film_cropped_bounds(xres, yres, crop):
    out = (C.c_int32 * 4)()
    lib().pbrt_hip_film_cropped_bounds(xres, yres, (C.c_float * 4)(*crop), out)
    return tuple(out)


def film_sample_bounds(xres, yres, crop, radius):
    out = (C.c_int32 * 4)()
    lib().pbrt_hip_film_sample_bounds(xres, yres, (C.c_float * 4)(*crop), radius[0], radius[1], out)
    return tuple(out)


def film_tile_bounds(xres, yres, crop, radius, sample_bounds):
    out = (C.c_int32 * 4)()
    lib().pbrt_hip_film_tile_bounds(xres, yres, (C.c_float * 4)(*crop), radius[0], radius[1],
                                    (C.c_int32 * 4)(*sample_bounds), out)
    return tuple(out)


def film_to_rgb(film_xyzw, scale=1.0):
    """Film::write_image's pixel arithmetic (film.rs:340-372): (h, w, 4) XYZW -> (h, w, 3) linear RGB."""
    f = np.ascontiguousarray(film_xyzw, np.float32)
    rgb = np.zeros(f.shape[:-1] + (3,), np.float32)
    lib().pbrt_hip_film_to_rgb(_fp(f), f.size // 4, scale, _fp(rgb))
    return rgb


def write_image(name, rgb):
    rgb = np.ascontiguousarray(rgb, np.float32)
    h, w = rgb.shape[:2]
    check(lib().pbrt_hip_write_image(str(name).encode(), _fp(rgb), w, h), "pbrt_hip_write_image")


def read_image(name):
    """imageio::read_image (imageio.rs:87-184): -> (h, w, 3) float32."""
    w, h = C.c_int32(0), C.c_int32(0)
    check(lib().pbrt_hip_read_image(str(name).encode(), None, C.byref(w), C.byref(h)), "pbrt_hip_read_image")
    rgb = np.zeros((h.value, w.value, 3), np.float32)
    check(lib().pbrt_hip_read_image(str(name).encode(), _fp(rgb), C.byref(w), C.byref(h)), "pbrt_hip_read_image")
    return rgb


def bvh_build_host(P, idx):
    """The host BVH builder alone (no device): returns (nodes[n,8] uint32 view, order, depth)."""
    P = np.ascontiguousarray(P, np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(idx, np.uint32).reshape(-1, 3)
    nt = idx.shape[0]
    nodes = np.zeros((max(2 * nt, 1), 8), np.uint32)
    order = np.zeros(max(nt, 1), np.uint32)
    nn, depth = C.c_uint32(), C.c_uint32()
    check(lib().pbrt_hip_bvh_build_host(_fp(P), P.shape[0], _u32p(idx), nt, _u32p(nodes), _u32p(order),
                                        C.byref(nn), C.byref(depth)), "pbrt_hip_bvh_build_host")
    return nodes[:nn.value].copy(), order[:nt].copy(), depth.value


def quad_build_host(P, idx, split_leaves=True):
    """The production walk's quantised 4-wide tree alone (no device): (quads[n, 16] uint32, stack_need)."""
    P = np.ascontiguousarray(P, np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(idx, np.uint32).reshape(-1, 3)
    cap = 2 * idx.shape[0] + 4
    quads = np.zeros((cap, 16), np.uint32)
    nq, need = C.c_uint32(), C.c_uint32()
    check(lib().pbrt_hip_quad_build_host(_fp(P), P.shape[0], _u32p(idx), idx.shape[0], int(split_leaves), _u32p(quads), cap,
                                         C.byref(nq), C.byref(need)), "pbrt_hip_quad_build_host")
    return quads[:nq.value].copy(), need.value


TREES = {"sah": 0, "reinsert": 2, "default": 0xFFFFFFFF}


def quad_build_host_ex(P, idx, tree="default", split_leaves=True):
    """pbrt_hip_quad_build_host_ex (no device): dict(quads[n, 16] uint32, stack_need, order = leaf slot -> triangle,
    root_box[6], n_refs) of the production walk's tree collapsed from the binary tree `tree` ("sah": canonical binned SAH; "reinsert": the same
    optimised by the device builder's re-insertion pass run on the host; "default")."""
    P = np.ascontiguousarray(P, np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(idx, np.uint32).reshape(-1, 3)
    nt = idx.shape[0]
    cap = 4 * nt + 4
    quads = np.zeros((cap, 16), np.uint32)
    order = np.zeros(max(nt, 1), np.uint32)
    box = np.zeros(6, np.float32)
    nq, need, nrefs = C.c_uint32(), C.c_uint32(), C.c_uint32()
    exact = np.zeros((cap, 24), np.float32)
    check(lib().pbrt_hip_quad_build_host_ex(_fp(P), P.shape[0], _u32p(idx), nt, int(split_leaves), TREES[tree], _u32p(quads), cap,
                                            C.byref(nq), C.byref(need), _u32p(order), _fp(box), C.byref(nrefs), _fp(exact)),
          "pbrt_hip_quad_build_host_ex")
    return {"quads": quads[:nq.value].copy(), "stack_need": need.value, "order": order[:nt].copy(), "root_box": box, "n_refs": nrefs.value,
            "exact_boxes": exact[:nq.value].copy()}


def film_from_acc(acc):
    """pbrt_hip_film_from_acc: (..., 4) int64 accumulators -> (..., 4) float32 film pixels {X, Y, Z, weight}."""
    a = np.ascontiguousarray(acc, np.int64)
    film = np.zeros(a.shape[:-1] + (4,), np.float32)
    lib().pbrt_hip_film_from_acc(a.ctypes.data_as(C.POINTER(C.c_int64)), a.size // 4, _fp(film))
    return film


def sobol_matrices():
    """pbrt_hip_sobol_matrices: (128, 32) uint32 generator matrices of sampler 2 (host only)"""
    out = np.zeros((128, 32), np.uint32)
    lib().pbrt_hip_sobol_matrices(_u32p(out))
    return out


def slab_pixel_index(xres, yres, crop, rank, world_size):
    n = lib().pbrt_hip_slab_floats(xres, yres, (C.c_float * 4)(*crop), rank, world_size)
    if n < 0:
        raise ValueError("bad sharding arguments")
    out = np.zeros(n // 4, np.int64)
    if n:
        check(lib().pbrt_hip_slab_pixel_index(xres, yres, (C.c_float * 4)(*crop), rank, world_size,
                                              out.ctypes.data_as(C.POINTER(C.c_int64))), "pbrt_hip_slab_pixel_index")
    return out


class MultiScene:
    """A scene replicated on n GPUs of this node inside ONE process (pbrt_hip_multi_*): GPU g renders the super-tiles
    t % n == g from its own host thread and stream, one RCCL gather assembles the film on GPU 0."""

    def __init__(self, sd, n_gpus=0, builder=None):
        """builder: as Scene (None = the library's default, the device builder)."""
        self._h = None
        self.sd = sd.normalized()
        desc = SceneDesc()
        keep = fill_desc(desc, self.sd, Material, Light, Sphere, Texture)
        h = C.c_void_p()
        check(lib().pbrt_hip_multi_create(C.byref(desc), int(n_gpus), BUILDERS[builder] if builder else 0, C.byref(h)),
              "pbrt_hip_multi_create")
        del keep
        self._h = h
        self.n_gpus = lib().pbrt_hip_multi_gpus(h)

    def render(self, host_film=True, **kw):
        """-> (film[h, w, 4] or None, [stats dict per GPU]).  kw as Scene.render (rank / world_size are set by the library)."""
        r = make_render_desc(RenderDesc, **kw)
        w, h = self.sd.crop_size()
        film = np.zeros((h, w, 4), np.float32) if host_film else None
        st = (Stats * self.n_gpus)()
        check(lib().pbrt_hip_multi_render(self._h, C.byref(r), _fp(film) if host_film else None, st), "pbrt_hip_multi_render")
        return film, [{k: getattr(s, k) for k, _ in Stats._fields_} for s in st]

    def film_device_ptr(self):
        p = C.c_void_p()
        check(lib().pbrt_hip_multi_film_device(self._h, C.byref(p)), "pbrt_hip_multi_film_device")
        return p.value

    def close(self):
        if self._h:
            lib().pbrt_hip_multi_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        self.close()


def render_multi(sd, n_gpus=0, **kw):
    """pbrt_hip_render_multi: create + render on n GPUs + destroy in one call -> (film, [stats per GPU])."""
    sd = sd.normalized()
    desc = SceneDesc()
    keep = fill_desc(desc, sd, Material, Light, Sphere, Texture)
    r = make_render_desc(RenderDesc, **kw)
    n = n_gpus if n_gpus > 0 else device_count()
    w, h = sd.crop_size()
    film = np.zeros((h, w, 4), np.float32)
    st = (Stats * max(n, 1))()
    check(lib().pbrt_hip_render_multi(C.byref(desc), C.byref(r), int(n_gpus), _fp(film), st), "pbrt_hip_render_multi")
    del keep
    return film, [{k: getattr(s, k) for k, _ in Stats._fields_} for s in st]


class Scene:
    """A scene resident in HBM (flattened BVH + leaf-ordered triangles + tables)."""

    def __init__(self, sd, device=-1, builder=None):
        """builder: None (the library's one default: pbrt_hip_scene_create = "gpu" unless PBRT_HIP_BUILDER=host), "host" (the host's
        binned-SAH builder, the canonical tree), "gpu" (accelerator built AND optimised by parallel
        re-insertion on the device: tens of milliseconds), "gpu-plain" (the device's tree as built,
        PBRT_HIP_SCENE_PLAIN_TREE: A-B runs) or "host-optimized" (PBRT_HIP_SCENE_OPTIMIZED_TREE: the host builder
        followed by the same re-insertion pass run on one host core, seconds per million triangles).  Same film and hit records whichever
        is used."""
        self.sd = sd.normalized()
        desc = SceneDesc()
        keep = fill_desc(desc, self.sd, Material, Light, Sphere, Texture)
        h = C.c_void_p()
        if builder is None:
            check(lib().pbrt_hip_scene_create(C.byref(desc), device, C.byref(h)), "pbrt_hip_scene_create")
        else:
            flags = BUILDERS[builder]
            check(lib().pbrt_hip_scene_create_ex(C.byref(desc), device, flags, C.byref(h)), "pbrt_hip_scene_create_ex")
        del keep
        self._h = h

    def build_info(self):
        g, ms = C.c_uint32(), C.c_double()
        check(lib().pbrt_hip_scene_build_info(self._h, C.byref(g), C.byref(ms)), "pbrt_hip_scene_build_info")
        r, cms = C.c_uint32(), C.c_double()
        check(lib().pbrt_hip_scene_canonical_info(self._h, C.byref(r), C.byref(cms)), "pbrt_hip_scene_canonical_info")
        op, om, oms = C.c_uint32(), C.c_uint32(), C.c_double()
        check(lib().pbrt_hip_scene_optimize_info(self._h, C.byref(op), C.byref(om), C.byref(oms)), "pbrt_hip_scene_optimize_info")
        cb, ca, un = C.c_double(), C.c_double(), C.c_uint32()
        check(lib().pbrt_hip_scene_optimize_cost(self._h, C.byref(cb), C.byref(ca), C.byref(un)), "pbrt_hip_scene_optimize_cost")
        return {"gpu_built": bool(g.value), "build_ms": ms.value, "canonical_tree_ready": bool(r.value), "canonical_tree_host_build_ms": cms.value,
                "reinsert_passes": op.value, "reinsert_moves": om.value, "reinsert_ms": oms.value,
                "reinsert_area_before": cb.value, "reinsert_area_after": ca.value, "reinsert_pass_undone": bool(un.value)}

    def export_quads(self):
        """(quads[n, 16] uint32, order[n_tris] uint32): the production walk's tree as it sits in HBM."""
        n = C.c_uint32()
        check(lib().pbrt_hip_scene_export_quads(self._h, None, 0, C.byref(n), None), "pbrt_hip_scene_export_quads")
        quads = np.zeros((max(n.value, 1), 16), np.uint32)
        order = np.zeros(max(self.n_prims, 1), np.uint32)
        check(lib().pbrt_hip_scene_export_quads(self._h, _u32p(quads), quads.shape[0], C.byref(n), _u32p(order)),
              "pbrt_hip_scene_export_quads")
        return quads[:n.value], order[:self.n_prims]

    def close(self):
        if getattr(self, "_h", None):
            lib().pbrt_hip_scene_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def n_prims(self):
        """primitives of the tree: the triangles, then the spheres (primitive n_tris + s)"""
        return int(self.sd.idx.shape[0]) + int(self.sd.spheres.shape[0])

    def info(self):
        nn, depth, nl, nb = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint64()
        check(lib().pbrt_hip_scene_info(self._h, C.byref(nn), C.byref(depth), C.byref(nl), C.byref(nb)),
              "pbrt_hip_scene_info")
        qn, need = C.c_uint32(), C.c_uint32()
        check(lib().pbrt_hip_scene_walk_info(self._h, C.byref(qn), C.byref(need)), "pbrt_hip_scene_walk_info")
        return {"n_nodes": nn.value, "depth": depth.value, "n_lights": nl.value, "device_bytes": nb.value,
                "quad_nodes": qn.value, "quad_stack_need": need.value}

    def export_bvh(self):
        i = self.info()
        nodes = np.zeros((max(i["n_nodes"], 1), 8), np.uint32)
        order = np.zeros(max(self.n_prims, 1), np.uint32)
        check(lib().pbrt_hip_scene_export_bvh(self._h, _u32p(nodes), _u32p(order)), "pbrt_hip_scene_export_bvh")
        return nodes[:i["n_nodes"]], order[:self.n_prims]

    def render(self, integrator=INTEGRATOR_PATH, max_depth=5, spp=(1, 1), seed=0, rank=0, world_size=1,
               counters=False, **kw):
        """-> (film[h, w, 4] float32 {X, Y, Z, weight}, stats dict).  counters: False, True (canonical
        walk, equal to the oracle's counters) or "walk" (what the production kernel itself fetches / tests).
        kw: sampler="stratified"|"sobol", filter_width=(xw, yw)."""
        r = make_render_desc(RenderDesc, integrator, max_depth, spp, seed, rank, world_size,
                             FLAG_WALK_COUNTERS if counters == "walk" else (FLAG_COUNTERS if counters else 0), **kw)
        w, h = self.sd.crop_size()
        film = np.zeros((h, w, 4), np.float32)
        st = Stats()
        check(lib().pbrt_hip_render(self._h, C.byref(r), _fp(film), C.byref(st)), "pbrt_hip_render")
        return film, {k: getattr(st, k) for k, _ in Stats._fields_}

    def render_acc(self, filter_width, **kw):
        """A box filter radius other than 0.5: this rank's fixed-point film accumulators (h, w, 4) int64 {r, g, b, samples}
        (pbrt_hip_render_acc); accumulators of ranks add exactly, film_from_acc converts.  kw as render."""
        r = make_render_desc(RenderDesc, filter_width=filter_width, **kw)
        w, h = self.sd.crop_size()
        acc = np.zeros((h, w, 4), np.int64)
        st = Stats()
        check(lib().pbrt_hip_render_acc(self._h, C.byref(r), acc.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(st)), "pbrt_hip_render_acc")
        return acc, {k: getattr(st, k) for k, _ in Stats._fields_}

    def render_buffer_bytes(self, **kw):
        """bytes render_device writes for this render description (slab of the rank, or the accumulators of a wide filter)"""
        kw.pop("counters", None)
        r = make_render_desc(RenderDesc, **kw)
        return lib().pbrt_hip_render_buffer_bytes(self._h, C.byref(r))

    def film_from_acc_device(self, d_acc_ptr, d_film_ptr, stream_ptr=None):
        check(lib().pbrt_hip_film_from_acc_device(self._h, C.c_void_p(d_acc_ptr), C.c_void_p(d_film_ptr), C.c_void_p(stream_ptr or 0)),
              "pbrt_hip_film_from_acc_device")

    def render_device(self, d_slab_ptr, stream_ptr=None, **kw):
        """Asynchronous render into a device slab (e.g. a torch tensor's data_ptr())."""
        counters = kw.pop("counters", False)
        r = make_render_desc(RenderDesc, flags=FLAG_WALK_COUNTERS if counters == "walk" else (FLAG_COUNTERS if counters else 0), **kw)
        check(lib().pbrt_hip_render_device(self._h, C.byref(r), C.c_void_p(d_slab_ptr), C.c_void_p(stream_ptr or 0)),
              "pbrt_hip_render_device")

    def render_prepare(self, **kw):
        """pbrt_hip_render_prepare: allocate the scratch a render with these arguments needs, launch nothing."""
        kw.pop("counters", None)
        r = make_render_desc(RenderDesc, **kw)
        check(lib().pbrt_hip_render_prepare(self._h, C.byref(r)), "pbrt_hip_render_prepare")

    def render_wait(self):
        st = Stats()
        check(lib().pbrt_hip_render_wait(self._h, C.byref(st)), "pbrt_hip_render_wait")
        return {k: getattr(st, k) for k, _ in Stats._fields_}

    def film_assemble_device(self, d_slab_ptr, rank, world_size, d_film_ptr, stream_ptr=None):
        check(lib().pbrt_hip_film_assemble_device(self._h, C.c_void_p(d_slab_ptr), rank, world_size,
                                                  C.c_void_p(d_film_ptr), C.c_void_p(stream_ptr or 0)),
              "pbrt_hip_film_assemble_device")

    def slab_floats(self, rank=0, world_size=1):
        return lib().pbrt_hip_slab_floats(self.sd.xres, self.sd.yres, (C.c_float * 4)(*self.sd.crop), rank, world_size)

    def intersect(self, o, d, tmax, counters=False):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        tmax = np.ascontiguousarray(tmax, np.float32).reshape(-1)
        n = o.shape[0]
        t = np.zeros(n, np.float32)
        prim = np.zeros(n, np.uint32)
        b1 = np.zeros(n, np.float32)
        b2 = np.zeros(n, np.float32)
        cnt = (C.c_uint64 * 2)()
        check(lib().pbrt_hip_intersect(self._h, n, _fp(o), _fp(d), _fp(tmax), _fp(t), _u32p(prim), _fp(b1), _fp(b2),
                                       cnt if counters else None), "pbrt_hip_intersect")
        return t, prim, b1, b2, (cnt[0], cnt[1])

    def occluded(self, o, d, tmax):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        tmax = np.ascontiguousarray(tmax, np.float32).reshape(-1)
        n = o.shape[0]
        hit = np.zeros(n, np.uint8)
        check(lib().pbrt_hip_occluded(self._h, n, _fp(o), _fp(d), _fp(tmax),
                                      hit.ctypes.data_as(C.POINTER(C.c_uint8))), "pbrt_hip_occluded")
        return hit
