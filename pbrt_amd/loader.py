"""`.pbrt` scene ingestion through the C ABI (pbrt_hip_load_file / pbrt_hip_load_string): the
reference's tokenizer + parameter lists + API state machine (src/core/parser.rs, src/core/api.rs),
completed in C++ (csrc/scene_parser.cpp) for the directives this path covers."""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from ._lib import RenderDesc, SceneDesc, check, lib
from .api import SceneData


@dataclass
class LoadedScene:
    scene: SceneData
    integrator: int
    max_depth: int
    spp: tuple
    filename: str
    warnings: list = field(default_factory=list)
    ctm: np.ndarray = None      # CTM when parsing stopped
    names: dict = None          # camera / sampler / integrator / filter / accelerator / film names as given
    film_scale: float = 1.0     # Film "float scale" (film.rs:368-371): pass to film_to_rgb
    sampler: int = 0            # PBRT_HIP_SAMPLER_*
    filter_width: tuple = (0.5, 0.5)
    max_sample_luminance: float = 0.0  # Film "float maxsampleluminance" (film.rs:75,279); 0 = none

    def render_kwargs(self):
        return dict(integrator=self.integrator, max_depth=self.max_depth, spp=self.spp, sampler=self.sampler,
                    filter_width=self.filter_width, max_sample_luminance=self.max_sample_luminance)


def _collect(h):
    l = lib()
    try:
        d, r = SceneDesc(), RenderDesc()
        fn = C.create_string_buffer(4096)
        check(l.pbrt_hip_loaded_get(h, C.byref(d), C.byref(r), fn, len(fn)), "pbrt_hip_loaded_get")

        def arr(ptr, n, dt):
            return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt).copy() if n else np.zeros(0, dt)

        sd = SceneData(
            P=arr(d.P, 3 * d.n_verts, np.float32).reshape(-1, 3), idx=arr(d.idx, 3 * d.n_tris, np.uint32).reshape(-1, 3),
            mat_id=arr(d.mat_id, d.n_tris, np.uint16),
            materials=np.array([[d.mats[i].type, *d.mats[i].k, *d.mats[i].le] for i in range(d.n_mats)], np.float32).reshape(-1, 7),
            lights=np.array([[d.lights[i].type, *d.lights[i].p, *d.lights[i].c] for i in range(d.n_lights)], np.float32).reshape(-1, 7),
            spheres=np.array([[*d.spheres[i].c, d.spheres[i].r, d.spheres[i].mat] for i in range(d.n_spheres)], np.float32).reshape(-1, 5),
            cam_to_world=np.array(list(d.cam_to_world), np.float32).reshape(4, 4), fov=d.fov, xres=d.xres, yres=d.yres,
            crop=tuple(d.crop),
            mat_tex=np.array([d.mats[i].kd_tex for i in range(d.n_mats)], np.uint32),
            textures=np.array([[d.textures[i].type, *d.textures[i].tex1, *d.textures[i].tex2, d.textures[i].su, d.textures[i].sv, d.textures[i].du,
                                d.textures[i].dv] for i in range(d.n_textures)], np.float32).reshape(-1, 11),
            tri_uv=(arr(d.tri_uv, 6 * d.n_tris, np.float32).reshape(-1, 6) if d.tri_uv else np.zeros((0, 6), np.float32))).normalized()
        wbuf = C.create_string_buffer(1 << 16)
        nw = l.pbrt_hip_loaded_warnings(h, wbuf, len(wbuf))
        ctm = np.zeros(16, np.float32)
        names = C.create_string_buffer(1024)
        check(l.pbrt_hip_loaded_state(h, ctm.ctypes.data_as(C.POINTER(C.c_float)), names, len(names)), "pbrt_hip_loaded_state")
        keys = ("camera", "sampler", "integrator", "filter", "accelerator", "film")
        return LoadedScene(sd, r.integrator, r.max_depth, (r.spp_x, r.spp_y), fn.value.decode(),
                           [w for w in wbuf.value.decode().split("\n") if w][:nw], ctm.reshape(4, 4),
                           dict(zip(keys, names.value.decode().split(" "))),
                           film_scale=float(l.pbrt_hip_loaded_film_scale(h)), sampler=int(r.sampler),
                           filter_width=(float(r.filter_xwidth), float(r.filter_ywidth)),
                           max_sample_luminance=float(r.max_sample_luminance))
    finally:
        l.pbrt_hip_loaded_free(h)


def load_file(path):
    h = C.c_void_p()
    check(lib().pbrt_hip_load_file(str(path).encode(), C.byref(h)), "pbrt_hip_load_file")
    return _collect(h)


def load_string(text, base_dir=None):
    b = text.encode() if isinstance(text, str) else text
    h = C.c_void_p()
    check(lib().pbrt_hip_load_string(b, len(b), base_dir.encode() if base_dir else None, C.byref(h)), "pbrt_hip_load_string")
    return _collect(h)


def tokenize(text):
    """parser.rs Tokenizer: -> (tokens, ok).  ok is False when the stream ends in an error."""
    b = text.encode() if isinstance(text, str) else text
    buf = C.create_string_buffer(len(b) * 2 + 16)
    n = lib().pbrt_hip_tokenize(b, len(b), buf, len(buf))
    toks = [t for t in buf.value.decode().split("\n") if t]
    return toks, n >= 0
