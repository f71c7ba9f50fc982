// multi_gpu.cpp -- the render path on every GPU of a node behind ONE call of the C ABI (include/pbrt_hip.h,
// pbrt_hip_multi_* / pbrt_hip_render_multi): what a single-process host such as the reference's `world_end`
// (/root/reference/src/core/api.rs:432-473, reached from src/bin/pbrt.rs:72-83) needs to use 8 MI355X.
//
//   * the scene is built once and REPLICATED device to device (hipMemcpyPeerAsync: xGMI, not PCIe);
//   * one host thread, one stream and one scene replica per GPU; GPU g renders the 64x64 super-tiles t with
//     t % n == g -- no communication while rendering;
//   * one RCCL gather (ncclGather, /opt/rocm/include/rccl/rccl.h:745; communicators from ncclCommInitAll, one process)
//     brings the slabs to GPU 0 over its 7 xGMI links, where assemble_kernel scatters them into the film.
//
// RCCL is loaded with dlopen on first use: the single-GPU entry points of the library do not depend on it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pbrt_hip.h"
#include "capi_internal.hpp"
#include "host_math.hpp"

using namespace pbrt_hip;

namespace {

// the few RCCL entry points this file uses (signatures of rccl.h)
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;    // ncclSuccess == 0
constexpr int kNcclFloat = 7;  // ncclFloat32 (rccl.h ncclDataType_t)
struct Rccl {
  void *so = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Gather)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string why;
};
Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.so) break;
    }
    if (!r.so) { r.why = std::string("cannot load librccl: ") + dlerror(); return; }
    r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.so, "ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.so, "ncclCommDestroy");
    r.Gather = (decltype(r.Gather))dlsym(r.so, "ncclGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.so, "ncclGetErrorString");
    if (!r.CommInitAll || !r.CommDestroy || !r.Gather || !r.GetErrorString) r.why = "librccl lacks ncclCommInitAll / ncclGather";
  });
  return r;
}

size_t shard_pixels(int32_t xres, int32_t yres, const float crop[4], uint32_t rank, uint32_t world) {
  const int64_t f = pbrt_hip_slab_floats(xres, yres, crop, rank, world);
  return f > 0 ? (size_t)f / 4 : 0;
}

}  // namespace

struct pbrt_hip_multi {
  int n = 0;
  std::vector<pbrt_hip_scene *> scenes;  // [g] lives on device g; [0] is the one that was built, the others are copies
  std::vector<hipStream_t> streams;
  std::vector<DevBuf<float4>> slabs;     // per device: room for the largest shard (rank 0's), the gather's common count
  DevBuf<float4> gathered;               // device 0: n x max_slab
  DevBuf<float4> film;                   // device 0: the assembled film
  std::vector<ncclComm_t> comms;
  size_t max_slab = 0, n_px = 0;
  int32_t w = 0, h = 0;

  ~pbrt_hip_multi() {
    for (int g = 0; g < (int)comms.size(); g++)
      if (comms[g]) { (void)hipSetDevice(scenes[g] ? scenes[g]->device : g); (void)rccl().CommDestroy(comms[g]); }
    for (int g = 0; g < (int)scenes.size(); g++) {
      if (!scenes[g]) continue;
      (void)hipSetDevice(scenes[g]->device);
      if (g < (int)slabs.size()) slabs[g].release();
      if (g < (int)streams.size() && streams[g]) (void)hipStreamDestroy(streams[g]);
      if (g == 0) { gathered.release(); film.release(); }
      delete scenes[g];
    }
  }
};

namespace {

// a copy of `src` (any device) on `device`: every array of the scene travels device to device
int clone_scene(const pbrt_hip_scene *src, int device, pbrt_hip_scene **out) {
  HIP_TRY(hipSetDevice(device));
  std::unique_ptr<pbrt_hip_scene> s(new pbrt_hip_scene());
  s->device = device;
  {
    int cus = 0;
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    s->n_cu = cus > 0 ? (uint32_t)cus : 1u;
  }
  s->desc = src->desc;
  s->bvh.depth = src->bvh.depth;  // (the host copy of the tree stays with the original: export / info use that one)
  s->n_lights = src->n_lights;
  s->gpu_built = src->gpu_built;
  s->n_quads_gpu = src->n_quads_gpu;
  s->build_ms = src->build_ms;
  s->device_bytes = src->device_bytes;
  HIP_TRY(hipStreamCreate(&s->stream));
  HIP_TRY(hipEventCreate(&s->ev0));
  HIP_TRY(hipEventCreate(&s->ev1));
  int can = 0;
  if (hipDeviceCanAccessPeer(&can, device, src->device) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(src->device, 0);  // (already enabled is fine)
  (void)hipGetLastError();
#define CLONE(field)                                                                                                      \
  do {                                                                                                                    \
    HIP_TRY(s->field.alloc(src->field.n));                                                                                \
    if (src->field.n)                                                                                                     \
      HIP_TRY(hipMemcpyPeerAsync(s->field.p, device, src->field.p, src->device, src->field.n * sizeof(*src->field.p), s->stream)); \
  } while (0)
  CLONE(d_P); CLONE(d_idx); CLONE(d_order); CLONE(d_mat_id); CLONE(d_nodes); CLONE(d_quads);
  CLONE(d_tris); CLONE(d_mats); CLONE(d_lights); CLONE(d_spheres);
#undef CLONE
  HIP_TRY(s->d_counters.alloc(80));
  HIP_TRY(hipStreamSynchronize(s->stream));
  s->dev = src->dev;
  s->dev.nodes = s->d_nodes.p;
  s->dev.quads = s->d_quads.p;
  s->dev.tris = s->d_tris.p;
  s->dev.mats = s->d_mats.p;
  s->dev.lights = s->d_lights.p;
  s->dev.spheres = s->d_spheres.p;
  *out = s.release();
  return PBRT_HIP_OK;
}

}  // namespace

extern "C" {

int pbrt_hip_multi_create(const pbrt_hip_scene_desc *d, int n_gpus, uint32_t flags, pbrt_hip_multi **out) {
  if (!d || !out) return fail(PBRT_HIP_ERR_INVALID, "multi_create: null argument");
  *out = nullptr;
  try {
    const int ndev = pbrt_hip_device_count();
    if (ndev <= 0) return fail(PBRT_HIP_ERR_NO_DEVICE, "multi_create: no HIP device (there is no CPU fallback)");
    if (n_gpus <= 0) n_gpus = ndev;
    if (n_gpus > ndev) return fail(PBRT_HIP_ERR_INVALID, "multi_create: " + std::to_string(n_gpus) + " GPUs asked for, " + std::to_string(ndev) + " visible");
    Rccl &rc = rccl();
    if (!rc.why.empty()) return fail(PBRT_HIP_ERR_INTERNAL, "multi_create: " + rc.why);
    std::unique_ptr<pbrt_hip_multi> m(new pbrt_hip_multi());
    m->n = n_gpus;
    m->scenes.assign(n_gpus, nullptr);
    m->streams.assign(n_gpus, nullptr);
    m->slabs.resize(n_gpus);
    int rcode = pbrt_hip_scene_create_ex(d, 0, flags, &m->scenes[0]);
    if (rcode) return rcode;
    for (int g = 1; g < n_gpus; g++) {
      rcode = clone_scene(m->scenes[0], g, &m->scenes[g]);
      if (rcode) return rcode;
    }
    int32_t b[4];
    film_cropped_bounds(d->xres, d->yres, d->crop, b);
    m->w = b[2] > b[0] ? b[2] - b[0] : 0;
    m->h = b[3] > b[1] ? b[3] - b[1] : 0;
    m->n_px = (size_t)m->w * (size_t)m->h;
    m->max_slab = shard_pixels(d->xres, d->yres, d->crop, 0, (uint32_t)n_gpus);  // rank 0 owns the most super-tiles
    for (int g = 0; g < n_gpus; g++) {
      HIP_TRY(hipSetDevice(g));
      HIP_TRY(hipStreamCreate(&m->streams[g]));
      HIP_TRY(m->slabs[g].alloc(m->max_slab ? m->max_slab : 1));
      HIP_TRY(hipMemsetAsync(m->slabs[g].p, 0, (m->max_slab ? m->max_slab : 1) * sizeof(float4), m->streams[g]));
    }
    HIP_TRY(hipSetDevice(0));
    HIP_TRY(m->gathered.alloc((size_t)n_gpus * (m->max_slab ? m->max_slab : 1)));
    HIP_TRY(m->film.alloc(m->n_px ? m->n_px : 1));
    std::vector<int> devs(n_gpus);
    for (int g = 0; g < n_gpus; g++) devs[g] = g;
    m->comms.assign(n_gpus, nullptr);
    const ncclResult_t nr = rc.CommInitAll(m->comms.data(), n_gpus, devs.data());
    if (nr != 0) return fail(PBRT_HIP_ERR_HIP, std::string("ncclCommInitAll: ") + rc.GetErrorString(nr));
    for (int g = 0; g < n_gpus; g++) { HIP_TRY(hipSetDevice(g)); HIP_TRY(hipStreamSynchronize(m->streams[g])); }
    *out = m.release();
    return PBRT_HIP_OK;
  } catch (const std::exception &e) {
    return fail(PBRT_HIP_ERR_INTERNAL, e.what());
  }
}

int pbrt_hip_multi_gpus(const pbrt_hip_multi *m) { return m ? m->n : 0; }

void pbrt_hip_multi_destroy(pbrt_hip_multi *m) { delete m; }

int pbrt_hip_multi_render(pbrt_hip_multi *m, const pbrt_hip_render_desc *r, float *film, pbrt_hip_stats *per_gpu) {
  if (!m || !r) return fail(PBRT_HIP_ERR_INVALID, "multi_render: null argument");
  try {
    const int n = m->n;
    std::vector<int> rcs(n, PBRT_HIP_OK);
    std::vector<std::string> errs(n);
    std::vector<pbrt_hip_stats> stats(n);
    Rccl &rc = rccl();
    // one host thread per GPU: render its shard, join the gather; thread 0 also assembles the film on GPU 0
    auto worker = [&](int g) {
      auto bad = [&](int code, const std::string &what) { rcs[g] = code; errs[g] = "GPU " + std::to_string(g) + ": " + what; };
      if (hipSetDevice(g) != hipSuccess) return bad(PBRT_HIP_ERR_HIP, "hipSetDevice");
      pbrt_hip_render_desc rd = *r;
      rd.rank = (uint32_t)g;
      rd.world_size = (uint32_t)n;
      int code = pbrt_hip_render_device(m->scenes[g], &rd, m->slabs[g].p, m->streams[g]);
      // (a rank whose render could not start still has to join the collective, or the others would wait for ever)
      if (code) bad(code, pbrt_hip_last_error());
      const ncclResult_t nr = rc.Gather(m->slabs[g].p, g == 0 ? m->gathered.p : nullptr, 4 * (m->max_slab ? m->max_slab : 1), kNcclFloat, 0,
                                         m->comms[g], m->streams[g]);
      if (nr != 0 && !code) bad(PBRT_HIP_ERR_HIP, std::string("ncclGather: ") + rc.GetErrorString(nr));
      if (g == 0 && nr == 0) {
        hipError_t e = m->n_px ? hipMemsetAsync(m->film.p, 0, m->n_px * sizeof(float4), m->streams[0]) : hipSuccess;
        for (int s = 0; s < n && e == hipSuccess && !rcs[0]; s++) {
          const int c2 = pbrt_hip_film_assemble_device(m->scenes[0], m->gathered.p + (size_t)s * (m->max_slab ? m->max_slab : 1), (uint32_t)s, (uint32_t)n,
                                                       m->film.p, m->streams[0]);
          if (c2) bad(c2, pbrt_hip_last_error());
        }
        if (e == hipSuccess && film && m->n_px) e = hipMemcpyAsync(film, m->film.p, m->n_px * sizeof(float4), hipMemcpyDeviceToHost, m->streams[0]);
        if (e != hipSuccess && !rcs[0]) bad(PBRT_HIP_ERR_HIP, hipGetErrorString(e));
      }
      const hipError_t es = hipStreamSynchronize(m->streams[g]);
      if (es != hipSuccess && !rcs[g]) bad(PBRT_HIP_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(es));
      if (!code) {
        const int c3 = pbrt_hip_render_wait(m->scenes[g], &stats[g]);
        if (c3 && !rcs[g]) bad(c3, pbrt_hip_last_error());
      }
    };
    std::vector<std::thread> th;
    for (int g = 1; g < n; g++) th.emplace_back(worker, g);
    worker(0);
    for (auto &t : th) t.join();
    for (int g = 0; g < n; g++)
      if (rcs[g]) return fail(rcs[g], "multi_render: " + errs[g]);
    if (per_gpu) std::memcpy(per_gpu, stats.data(), (size_t)n * sizeof(pbrt_hip_stats));
    return PBRT_HIP_OK;
  } catch (const std::exception &e) {
    return fail(PBRT_HIP_ERR_INTERNAL, e.what());
  }
}

int pbrt_hip_multi_film_device(pbrt_hip_multi *m, void **d_film) {
  if (!m || !d_film) return fail(PBRT_HIP_ERR_INVALID, "multi_film_device: null argument");
  *d_film = m->film.p;
  return PBRT_HIP_OK;
}

int pbrt_hip_render_multi(const pbrt_hip_scene_desc *d, const pbrt_hip_render_desc *r, int n_gpus, float *film, pbrt_hip_stats *per_gpu) {
  pbrt_hip_multi *m = nullptr;
  int rc = pbrt_hip_multi_create(d, n_gpus, 0u, &m);
  if (rc) return rc;
  rc = pbrt_hip_multi_render(m, r, film, per_gpu);
  const std::string keep = rc ? pbrt_hip_last_error() : "";
  pbrt_hip_multi_destroy(m);
  if (rc) return fail(rc, keep);
  return PBRT_HIP_OK;
}

}  // extern "C"
