// capi_internal.hpp -- what the translation units behind the C ABI share: error reporting, device buffers and the scene
// object (capi.cpp owns single-GPU scenes, multi_gpu.cpp replicates them across the devices of a node).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/pbrt_hip.h"
#include "../../include/pbrt_hip_debug.h"
#include "bvh_build.hpp"
#include "device_types.h"

namespace pbrt_hip {
// sets the thread-local message behind pbrt_hip_last_error() and returns `code`
int fail(int code, const std::string &msg);
const char *last_error_message();

template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  hipError_t alloc(size_t count) {
    n = count;
    if (count == 0) return hipSuccess;
    return hipMalloc((void **)&p, count * sizeof(T));
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
};
}  // namespace pbrt_hip

#define HIP_TRY(expr)                                                                                     \
  do {                                                                                                    \
    hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess)                                                                                 \
      return pbrt_hip::fail(PBRT_HIP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));         \
  } while (0)

struct pbrt_hip_scene {
  int device = 0;
  uint32_t n_cu = 256;  // hipDeviceProp_t::multiProcessorCount of `device`
  pbrt_hip_scene_desc desc{};  // scalar fields only; pointers are cleared
  pbrt_hip::Bvh bvh;
  uint32_t n_lights = 0;
  pbrt_hip::DevScene dev{};
  // device allocations
  pbrt_hip::DevBuf<float> d_P;
  pbrt_hip::DevBuf<uint32_t> d_idx, d_order;
  pbrt_hip::DevBuf<uint16_t> d_mat_id;
  pbrt_hip::DevBuf<uint4> d_nodes, d_quads;
  pbrt_hip::DevBuf<uint32_t> d_stack_overflow;  // per-lane spill area of the quad walk's stack beyond its LDS part
  pbrt_hip::DevBuf<float4> d_tris, d_mats, d_lights, d_spheres;
  pbrt_hip::DevBuf<float4> d_slab, d_film;          // scratch of pbrt_hip_render()
  pbrt_hip::DevBuf<float4> d_lane_state;            // per-lane path state records of the render kernel
  pbrt_hip::DevBuf<float4> d_partials;              // partial film sums of the work items (8 chunks per slab pixel)
  pbrt_hip::DevBuf<unsigned long long> d_counters;  // 5
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipStream_t stream = nullptr;  // own stream of pbrt_hip_render()
  bool pending = false;
  bool pending_counters = false;
  uint32_t n_quads_gpu = 0;
  uint32_t n_prims = 0;  // primitives of the tree: dev.n_tris triangles + the spheres (whose places in d_P / d_idx are proxy triangles: capi.cpp)
  bool gpu_built = false;  // accelerator built on the device: the canonical tree (counter flags) is made on first use
  // The canonical walk's view of a device-built scene (pbrt_hip::ensure_canonical): the oracle's binary tree, built on the
  // host from the vertex / index buffers read back from the device, and triangle records in ITS leaf order.  For a
  // host-built scene the production arrays serve both walks and dev_exact == dev.
  pbrt_hip::DevScene dev_exact{};
  bool canonical_ready = false;
  pbrt_hip::DevBuf<uint32_t> d_sobol;  // generator matrices of sampler 2 (uploaded at its first use)
  pbrt_hip::DevBuf<uint32_t> d_halton; // per-dimension table of sampler 3 (likewise)
  pbrt_hip::DevBuf<float> d_tri_uv_in;  // textured scenes: corner (u, v) as uploaded (triangle order) ...
  pbrt_hip::DevBuf<float2> d_tri_uv;    // ... and in leaf-slot order (3 per slot)
  pbrt_hip::DevBuf<float4> d_textures;
  bool textured = false;                // some triangle's material has kd_tex != 0: the TEX instantiations render it
  pbrt_hip::DevBuf<float4> d_tris_exact;
  pbrt_hip::DevBuf<uint32_t> d_order_exact;
  double canonical_build_ms = 0.0;
  double build_ms = 0.0;
  uint32_t reinsert_passes = 0, reinsert_moves = 0;  // the device build's tree optimisation (GpuBuildInfo)
  double reinsert_ms = 0.0;
  double reinsert_cost_before = 0.0, reinsert_cost_after = 0.0;  // summed half surface area of the interior nodes (GpuBuildInfo)
  uint32_t reinsert_undone = 0;
  uint64_t pending_samples = 0;
  uint64_t device_bytes = 0;

  ~pbrt_hip_scene() {
    d_P.release(); d_idx.release(); d_order.release(); d_mat_id.release(); d_nodes.release(); d_quads.release(); d_stack_overflow.release();
    d_tris.release(); d_mats.release(); d_lights.release(); d_spheres.release();
    d_slab.release(); d_film.release(); d_counters.release(); d_lane_state.release(); d_partials.release();
    d_tris_exact.release(); d_order_exact.release(); d_sobol.release(); d_halton.release(); d_tri_uv_in.release(); d_tri_uv.release(); d_textures.release();
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

