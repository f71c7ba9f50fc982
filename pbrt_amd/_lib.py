"""ctypes binding of include/pbrt_hip.h.  Loading fails loudly when the HIP library has not been
built: there is no CPU fallback anywhere in this package."""
import ctypes as C
import os

from .build import LIB_PATH


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
        ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64), ("kernel_ms", C.c_double),
        ("samples", C.c_uint64),
    ]


# every symbol include/pbrt_hip.h declares: (restype, argtypes)
_f, _u32, _i32, _i64, _u64, _vp = C.c_float, C.c_uint32, C.c_int32, C.c_int64, C.c_uint64, C.c_void_p
_pf, _pu32, _pi32, _pi64, _pu64, _pu8 = (C.POINTER(t) for t in (_f, _u32, _i32, _i64, _u64, C.c_uint8))
SYMBOLS = {
    "pbrt_hip_device_count": (C.c_int, []),
    "pbrt_hip_last_error": (C.c_char_p, []),
    "pbrt_hip_version": (C.c_char_p, []),
    "pbrt_hip_build_id": (C.c_char_p, []),
    "pbrt_hip_scene_create": (C.c_int, [C.POINTER(SceneDesc), C.c_int, C.POINTER(_vp)]),
    "pbrt_hip_scene_create_ex": (C.c_int, [C.POINTER(SceneDesc), C.c_int, _u32, C.POINTER(_vp)]),
    "pbrt_hip_scene_build_info": (C.c_int, [_vp, _pu32, C.POINTER(C.c_double)]),
    "pbrt_hip_scene_canonical_info": (C.c_int, [_vp, _pu32, C.POINTER(C.c_double)]),
    "pbrt_hip_scene_optimize_info": (C.c_int, [_vp, _pu32, _pu32, C.POINTER(C.c_double)]),
    "pbrt_hip_rccl_library": (C.c_int, [C.c_char_p, C.c_size_t]),
    "pbrt_hip_scene_optimize_cost": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), _pu32]),
    "pbrt_hip_scene_export_quads": (C.c_int, [_vp, _pu32, _u32, _pu32, _pu32]),
    "pbrt_hip_scene_destroy": (None, [_vp]),
    "pbrt_hip_scene_info": (C.c_int, [_vp, _pu32, _pu32, _pu32, _pu64]),
    "pbrt_hip_render_stack_plan": (C.c_int, [C.c_uint32, _pu32, _pu32, _pu32]),
    "pbrt_hip_scene_export_bvh": (C.c_int, [_vp, _pu32, _pu32]),
    "pbrt_hip_scene_walk_info": (C.c_int, [_vp, _pu32, _pu32]),
    "pbrt_hip_bvh_build_host": (C.c_int, [_pf, _u32, _pu32, _u32, _pu32, _pu32, _pu32, _pu32]),
    "pbrt_hip_quad_build_host": (C.c_int, [_pf, _u32, _pu32, _u32, C.c_int, _pu32, _u32, _pu32, _pu32]),
    "pbrt_hip_quad_build_host_ex": (C.c_int, [_pf, _u32, _pu32, _u32, C.c_int, _u32, _pu32, _u32, _pu32, _pu32, _pu32, _pf, _pu32, _pf]),
    "pbrt_hip_render": (C.c_int, [_vp, C.POINTER(RenderDesc), _pf, C.POINTER(Stats)]),
    "pbrt_hip_render_device": (C.c_int, [_vp, C.POINTER(RenderDesc), _vp, _vp]),
    "pbrt_hip_render_prepare": (C.c_int, [_vp, C.POINTER(RenderDesc)]),
    "pbrt_hip_render_wait": (C.c_int, [_vp, C.POINTER(Stats)]),
    "pbrt_hip_film_assemble_device": (C.c_int, [_vp, _vp, _u32, _u32, _vp, _vp]),
    "pbrt_hip_render_buffer_bytes": (_i64, [_vp, C.POINTER(RenderDesc)]),
    "pbrt_hip_render_acc": (C.c_int, [_vp, C.POINTER(RenderDesc), _pi64, C.POINTER(Stats)]),
    "pbrt_hip_film_from_acc_device": (C.c_int, [_vp, _vp, _vp, _vp]),
    "pbrt_hip_film_from_acc": (None, [_pi64, _i64, _pf]),
    "pbrt_hip_sobol_matrices": (None, [_pu32]),
    "pbrt_hip_slab_floats": (_i64, [_i32, _i32, _pf, _u32, _u32]),
    "pbrt_hip_slab_pixel_index": (C.c_int, [_i32, _i32, _pf, _u32, _u32, _pi64]),
    "pbrt_hip_multi_create": (C.c_int, [C.POINTER(SceneDesc), C.c_int, _u32, C.POINTER(_vp)]),
    "pbrt_hip_multi_gpus": (C.c_int, [_vp]),
    "pbrt_hip_multi_render": (C.c_int, [_vp, C.POINTER(RenderDesc), _pf, C.POINTER(Stats)]),
    "pbrt_hip_multi_film_device": (C.c_int, [_vp, C.POINTER(_vp)]),
    "pbrt_hip_multi_destroy": (None, [_vp]),
    "pbrt_hip_render_multi": (C.c_int, [C.POINTER(SceneDesc), C.POINTER(RenderDesc), C.c_int, _pf, C.POINTER(Stats)]),
    "pbrt_hip_intersect": (C.c_int, [_vp, _i64, _pf, _pf, _pf, _pf, _pu32, _pf, _pf, _pu64]),
    "pbrt_hip_occluded": (C.c_int, [_vp, _i64, _pf, _pf, _pf, _pu8]),
    "pbrt_hip_film_cropped_bounds": (None, [_i32, _i32, _pf, _pi32]),
    "pbrt_hip_film_sample_bounds": (None, [_i32, _i32, _pf, _f, _f, _pi32]),
    "pbrt_hip_film_tile_bounds": (None, [_i32, _i32, _pf, _f, _f, _pi32, _pi32]),
    "pbrt_hip_film_to_rgb": (None, [_pf, _i64, _f, _pf]),
    "pbrt_hip_write_image": (C.c_int, [C.c_char_p, _pf, _i32, _i32]),
    "pbrt_hip_read_image": (C.c_int, [C.c_char_p, _pf, _pi32, _pi32]),
    "pbrt_hip_look_at": (None, [_pf, _pf, _pf, _pf, _pf]),
    "pbrt_hip_load_file": (C.c_int, [C.c_char_p, C.POINTER(_vp)]),
    "pbrt_hip_load_string": (C.c_int, [C.c_char_p, C.c_size_t, C.c_char_p, C.POINTER(_vp)]),
    "pbrt_hip_loaded_free": (None, [_vp]),
    "pbrt_hip_loaded_get": (C.c_int, [_vp, C.POINTER(SceneDesc), C.POINTER(RenderDesc), C.c_char_p, C.c_size_t]),
    "pbrt_hip_loaded_film_scale": (_f, [_vp]),
    "pbrt_hip_loaded_warnings": (C.c_int, [_vp, C.c_char_p, C.c_size_t]),
    "pbrt_hip_loaded_state": (C.c_int, [_vp, _pf, C.c_char_p, C.c_size_t]),
    "pbrt_hip_tokenize": (C.c_int, [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]),
}

_lib = None


def _share_torch_hip_runtime():
    """One HIP / HSA runtime per process.  The PyTorch-ROCm wheel carries its own libamdhip64.so / libhsa-runtime64.so (torch/lib, sonames
    libamdhip64.so.7 / libhsa-runtime64.so.1); libpbrt_hip.so names the same sonames and finds /opt/rocm's.  Whichever copy is loaded first
    serves both -- unless this library comes first and torch is imported afterwards: then torch's own HSA runtime opens the device a second
    time and torch reports "No HIP GPUs are available" (seen on the MI355X box when a test that uses torch ran before anything imported
    it).  So when torch is installed but not imported yet, its copy of the runtime is loaded before libpbrt_hip.so -- by path, without
    importing torch (seconds) -- and a later `import torch` finds the runtime it expects.  PBRT_HIP_NO_TORCH_RUNTIME=1 skips this."""
    import sys
    if "torch" in sys.modules or os.environ.get("PBRT_HIP_NO_TORCH_RUNTIME"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)  # (its RPATH $ORIGIN brings torch's libhsa-runtime64.so with it)
    except (OSError, ImportError, ValueError):
        pass  # no torch, or a torch without a bundled runtime: /opt/rocm's serves this library alone


def lib():
    """The loaded libpbrt_hip.so; raises if it was not built (run `python -m pbrt_amd.build`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with __graft_entry__.build() or `python pbrt_amd/build.py`. "
                "pbrt_amd has no CPU fallback.")
        _share_torch_hip_runtime()
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)  # AttributeError here = header and library disagree
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


class PbrtHipError(RuntimeError):
    def __init__(self, code, where):
        msg = lib().pbrt_hip_last_error().decode("utf-8", "replace")
        super().__init__(f"{where} failed with status {code}: {msg}")
        self.code = code


def check(code, where):
    if code != 0:
        raise PbrtHipError(code, where)
