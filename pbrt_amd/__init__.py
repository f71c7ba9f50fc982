"""pbrt_amd -- MI355X-native BVH-traversal + path-tracing hot path behind the C ABI of
include/pbrt_hip.h (what `PbrtAPI::world_end`, reference src/core/api.rs:432-473, would call).

`csrc/` holds the HIP kernels and the C-ABI library; this package is the thin host mirror used by
tests, bench.py and the multi-GPU launcher.  There is no CPU fallback: importing is cheap, but any
call that computes needs lib/libpbrt_hip.so (built by `__graft_entry__.build()`) and a GPU.
"""
from .api import (FLAG_COUNTERS, INTEGRATOR_DIRECT, INTEGRATOR_PATH, INTEGRATOR_PATH_MIS, LIGHT_DISTANT, LIGHT_INFINITE, LIGHT_POINT, MATTE,
                  MIRROR, MultiScene, Scene, SceneData, render_multi, build_id, bvh_build_host, device_count, film_cropped_bounds, film_sample_bounds,
                  film_from_acc, film_tile_bounds, film_to_rgb, look_at, quad_build_host, quad_build_host_ex, read_image, slab_pixel_index,
                  write_image)

__all__ = [n for n in dir() if not n.startswith("_")]
