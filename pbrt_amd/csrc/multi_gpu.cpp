// multi_gpu.cpp -- the render path on every GPU of a node behind ONE call of the C ABI (include/pbrt_hip.h,
// pbrt_hip_multi_* / pbrt_hip_render_multi): what a single-process host such as the reference's `world_end`
// (/root/reference/src/core/api.rs:432-473, reached from src/bin/pbrt.rs:72-83) needs to use 8 MI355X.
//
//   * the scene is built once and REPLICATED device to device (hipMemcpyPeerAsync: xGMI, not PCIe);
//   * one stream and one scene replica per GPU, all driven from the calling thread: GPU g renders the 64x64 super-tiles
//     t with t % n == g -- the n launches are asynchronous, nothing is exchanged while rendering;
//   * one RCCL collective per frame, enqueued on the n streams behind the render kernels as ONE group call
//     (ncclGroupStart / ncclGroupEnd: the single-process form, rccl.h:904-923): ncclGather of the slabs to GPU 0 over its 7
//     xGMI links, where assemble_kernel scatters them into the film -- or, for a box filter radius other than 0.5
//     (DESIGN.md 3.11), ncclReduce(sum, int64) of the fixed-point accumulators, which add exactly;
//   * a launch that fails is known BEFORE the collective is enqueued, so no rank is ever left waiting in it; if the
//     collective itself fails the communicators are aborted and the handle refuses further renders.
//
// One GPU needs none of this: n == 1 renders through the single-GPU entry points, RCCL is not even loaded (dlopen on
// first use with n > 1).  The caller's current HIP device is restored on every exit.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/pbrt_hip.h"
#include "capi_internal.hpp"
#include "host_math.hpp"

using namespace pbrt_hip;

namespace {

// the few RCCL entry points this file uses (signatures of rccl.h)
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;      // ncclSuccess == 0
constexpr int kNcclInt64 = 4;  // rccl.h ncclDataType_t
constexpr int kNcclFloat = 7;  // ncclFloat32
constexpr int kNcclSum = 0;    // ncclRedOp_t
struct Rccl {
  void *so = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Gather)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Reduce)(const void *, void *, size_t, int, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string why;
};
Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // The RCCL that belongs to the HIP runtime this process runs on: a PyTorch-ROCm process carries its own libamdhip64 AND its own
    // librccl (torch/lib), a C++ host uses /opt/rocm's pair -- so the first candidates sit beside the libamdhip64 that is loaded (an
    // RCCL of another ROCm release against this runtime is the mismatch to avoid), then the loader's search path.
    std::vector<std::string> names;
    if (void *sym = dlsym(RTLD_DEFAULT, "hipGetDeviceCount")) {
      Dl_info info;
      if (dladdr(sym, &info) && info.dli_fname) {
        const std::string f = info.dli_fname;
        const size_t slash = f.rfind('/');
        if (slash != std::string::npos) {
          names.push_back(f.substr(0, slash) + "/librccl.so.1");
          names.push_back(f.substr(0, slash) + "/librccl.so");
        }
      }
    }
    for (const char *n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) names.push_back(n);
    for (const std::string &name : names) {
      r.so = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (r.so) break;
    }
    if (!r.so) { r.why = std::string("cannot load librccl: ") + dlerror(); return; }
    r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.so, "ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.so, "ncclCommDestroy");
    r.CommAbort = (decltype(r.CommAbort))dlsym(r.so, "ncclCommAbort");
    r.GroupStart = (decltype(r.GroupStart))dlsym(r.so, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.so, "ncclGroupEnd");
    r.Gather = (decltype(r.Gather))dlsym(r.so, "ncclGather");
    r.Reduce = (decltype(r.Reduce))dlsym(r.so, "ncclReduce");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.so, "ncclGetErrorString");
    if (!r.CommInitAll || !r.CommDestroy || !r.CommAbort || !r.GroupStart || !r.GroupEnd || !r.Gather || !r.Reduce || !r.GetErrorString)
      r.why = "librccl lacks ncclCommInitAll / ncclGroupStart / ncclGather / ncclReduce";
  });
  return r;
}

// restores the calling thread's current device on scope exit (a host such as torch must not find itself retargeted)
struct DeviceRestore {
  int dev = -1;
  DeviceRestore() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
  ~DeviceRestore() { if (dev >= 0) (void)hipSetDevice(dev); }
};

size_t shard_pixels(int32_t xres, int32_t yres, const float crop[4], uint32_t rank, uint32_t world) {
  const int64_t f = pbrt_hip_slab_floats(xres, yres, crop, rank, world);
  return f > 0 ? (size_t)f / 4 : 0;
}

}  // namespace

struct pbrt_hip_multi {
  int n = 0;
  std::vector<pbrt_hip_scene *> scenes;  // [g] lives on device g; [0] is the one that was built, the others are copies
  std::vector<hipStream_t> streams;
  std::vector<DevBuf<float4>> slabs;     // per device: what pbrt_hip_render_device fills (slab, or the accumulators of a wide filter)
  DevBuf<float4> gathered;               // device 0: n x max_slab (default filter) or the summed accumulators (wide filter)
  DevBuf<float4> film;                   // device 0: the assembled film
  std::vector<ncclComm_t> comms;         // n > 1 only
  size_t max_slab = 0, n_px = 0;
  int32_t w = 0, h = 0;
  bool broken = false;                   // a collective failed and the communicators were aborted
  // PBRT_HIP_MULTI_LOOPBACK (a debug knob, for boxes with fewer GPUs than ranks -- the test pool has one): rank g lives on device
  // g mod <visible devices>, and since RCCL refuses two ranks on one device the frame's one exchange is made of device-to-device copies
  // (gather) or an adding kernel (wide filter) ordered by events.  Everything else -- replication, the ranks' shares, the launches on
  // their own streams, the gathered layout, the assembly, the statistics, the error paths -- is the code N real GPUs run.
  bool loopback = false;
  std::vector<int> dev;                  // [g]: the device rank g lives on (g itself unless loopback)
  std::vector<hipEvent_t> done;          // loopback: rank g's slab is in place

  ~pbrt_hip_multi() {
    DeviceRestore keep;
    for (int g = 0; g < (int)comms.size(); g++)
      if (comms[g]) { (void)hipSetDevice(scenes[g] ? scenes[g]->device : g); (void)(broken ? rccl().CommAbort(comms[g]) : rccl().CommDestroy(comms[g])); }
    for (int g = 0; g < (int)scenes.size(); g++) {
      if (!scenes[g]) continue;
      (void)hipSetDevice(scenes[g]->device);
      if (g < (int)slabs.size()) slabs[g].release();
      if (g < (int)streams.size() && streams[g]) (void)hipStreamDestroy(streams[g]);
      if (g == 0) { gathered.release(); film.release(); }
      if (g < (int)done.size() && done[g]) (void)hipEventDestroy(done[g]);
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
  s->n_prims = src->n_prims;
  s->build_ms = src->build_ms;
  s->reinsert_passes = src->reinsert_passes;
  s->reinsert_moves = src->reinsert_moves;
  s->reinsert_ms = src->reinsert_ms;
  s->reinsert_cost_before = src->reinsert_cost_before;
  s->reinsert_cost_after = src->reinsert_cost_after;
  s->reinsert_undone = src->reinsert_undone;
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
  CLONE(d_tris); CLONE(d_mats); CLONE(d_lights); CLONE(d_spheres); CLONE(d_tri_uv); CLONE(d_textures);
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
  s->textured = src->textured;
  *out = s.release();
  return PBRT_HIP_OK;
}

int multi_create(const pbrt_hip_scene_desc *d, int n_gpus, uint32_t flags, pbrt_hip_multi **out) {
  const int ndev = pbrt_hip_device_count();
  if (ndev <= 0) return fail(PBRT_HIP_ERR_NO_DEVICE, "multi_create: no HIP device (there is no CPU fallback)");
  if (d->xres <= 0 || d->yres <= 0) return fail(PBRT_HIP_ERR_INVALID, "multi_create: resolution must be positive");
  int32_t b[4];
  film_cropped_bounds(d->xres, d->yres, d->crop, b);
  const int32_t w = b[2] > b[0] ? b[2] - b[0] : 0, h = b[3] > b[1] ? b[3] - b[1] : 0;
  if (n_gpus <= 0) {
    // all visible devices -- but no more than there are 64x64 super-tiles to deal out (a 64x64 film on eight GPUs would
    // replicate the scene seven times for ranks that own nothing)
    const int64_t tiles = (int64_t)((w + 63) / 64) * (int64_t)((h + 63) / 64);
    n_gpus = (int)std::max<int64_t>(1, std::min<int64_t>(ndev, tiles));
  }
  const char *lb = debug_knob("PBRT_HIP_MULTI_LOOPBACK");
  const bool loopback = lb && *lb && !(lb[0] == '0' && !lb[1]);
  if (n_gpus > ndev && !loopback) return fail(PBRT_HIP_ERR_INVALID, "multi_create: " + std::to_string(n_gpus) + " GPUs asked for, " + std::to_string(ndev) + " visible");
  if (n_gpus > 64) return fail(PBRT_HIP_ERR_INVALID, "multi_create: more than 64 ranks");
  if (n_gpus > 1 && !loopback) {  // (one GPU: no collective, RCCL is not needed and not loaded)
    Rccl &rc = rccl();
    if (!rc.why.empty()) return fail(PBRT_HIP_ERR_INTERNAL, "multi_create: " + rc.why);
  }
  std::unique_ptr<pbrt_hip_multi> m(new pbrt_hip_multi());
  m->n = n_gpus;
  m->loopback = loopback && n_gpus > 1;
  m->dev.resize(n_gpus);
  for (int g = 0; g < n_gpus; g++) m->dev[g] = loopback ? g % ndev : g;
  m->scenes.assign(n_gpus, nullptr);
  m->streams.assign(n_gpus, nullptr);
  m->slabs.resize(n_gpus);
  int rcode = pbrt_hip_scene_create_ex(d, 0, flags, &m->scenes[0]);
  if (rcode) return rcode;
  for (int g = 1; g < n_gpus; g++) {
    rcode = clone_scene(m->scenes[0], m->dev[g], &m->scenes[g]);
    if (rcode) return rcode;
  }
  m->w = w;
  m->h = h;
  m->n_px = (size_t)w * (size_t)h;
  m->max_slab = shard_pixels(d->xres, d->yres, d->crop, 0, (uint32_t)n_gpus);  // rank 0 owns the most super-tiles
  if (m->loopback) m->done.assign(n_gpus, nullptr);
  for (int g = 0; g < n_gpus; g++) {
    HIP_TRY(hipSetDevice(m->dev[g]));
    HIP_TRY(hipStreamCreate(&m->streams[g]));
    if (m->loopback) HIP_TRY(hipEventCreateWithFlags(&m->done[g], hipEventDisableTiming));
  }
  HIP_TRY(hipSetDevice(0));
  HIP_TRY(m->film.alloc(m->n_px ? m->n_px : 1));
  if (n_gpus > 1 && !m->loopback) {
    std::vector<int> devs(n_gpus);
    for (int g = 0; g < n_gpus; g++) devs[g] = g;
    m->comms.assign(n_gpus, nullptr);
    const ncclResult_t nr = rccl().CommInitAll(m->comms.data(), n_gpus, devs.data());
    if (nr != 0) return fail(PBRT_HIP_ERR_HIP, std::string("ncclCommInitAll: ") + rccl().GetErrorString(nr));
  }
  *out = m.release();
  return PBRT_HIP_OK;
}

// room for `count` float4 on every device (and n x count gathered on device 0 when `gather`)
int ensure_buffers(pbrt_hip_multi *m, size_t count, bool gather) {
  count = count ? count : 1;
  for (int g = 0; g < m->n; g++) {
    if (m->slabs[g].n >= count) continue;
    HIP_TRY(hipSetDevice(m->dev[g]));
    m->slabs[g].release();
    HIP_TRY(m->slabs[g].alloc(count));
    HIP_TRY(hipMemsetAsync(m->slabs[g].p, 0, count * sizeof(float4), m->streams[g]));  // (a rank without tiles sends zeros)
  }
  const size_t need = gather ? (size_t)m->n * count : count;
  if (m->n > 1 && m->gathered.n < need) {
    HIP_TRY(hipSetDevice(0));
    m->gathered.release();
    HIP_TRY(m->gathered.alloc(need));
  }
  return PBRT_HIP_OK;
}

int multi_render(pbrt_hip_multi *m, const pbrt_hip_render_desc *r, float *film, pbrt_hip_stats *per_gpu) {
  if (m->broken) return fail(PBRT_HIP_ERR_INTERNAL, "multi_render: an earlier collective failed and the communicators were aborted; create a new handle");
  const int n = m->n;
  const float fx = r->filter_xwidth == 0.f ? 0.5f : r->filter_xwidth, fy = r->filter_ywidth == 0.f ? 0.5f : r->filter_ywidth;
  const bool wide = fx != 0.5f || fy != 0.5f;  // accumulators of the whole window, summed instead of gathered
  const size_t per_gpu_f4 = wide ? 2 * m->n_px : m->max_slab;
  int code = ensure_buffers(m, per_gpu_f4, !wide);
  if (code) return code;
  // ---- every GPU's scratch first (hipMalloc synchronises: none may sit between the launches below; no-ops from the second frame on) ----
  for (int g = 0; g < n; g++) {
    pbrt_hip_render_desc rd = *r;
    rd.rank = (uint32_t)g;
    rd.world_size = (uint32_t)n;
    code = pbrt_hip_render_prepare(m->scenes[g], &rd);
    if (code) return fail(code, "multi_render: GPU " + std::to_string(g) + ": " + pbrt_hip_last_error());
  }
  // ---- every GPU's shard, asynchronously on its own stream ----
  std::vector<bool> started(n, false);
  std::string err;
  for (int g = 0; g < n && !code; g++) {
    pbrt_hip_render_desc rd = *r;
    rd.rank = (uint32_t)g;
    rd.world_size = (uint32_t)n;
    const char *inject = debug_knob("PBRT_HIP_MULTI_FAIL_RANK");  // tests: a rank whose launch fails (the others must not be left waiting)
    if (inject && std::atoi(inject) == g) code = fail(PBRT_HIP_ERR_INTERNAL, "injected launch failure (PBRT_HIP_MULTI_FAIL_RANK)");
    else code = pbrt_hip_render_device(m->scenes[g], &rd, m->slabs[g].p, m->streams[g]);
    if (code) err = "GPU " + std::to_string(g) + ": " + pbrt_hip_last_error();
    else started[g] = true;
  }
  auto drain = [&]() {  // leave no render "in flight" behind an error
    for (int g = 0; g < n; g++)
      if (started[g]) { (void)hipSetDevice(m->dev[g]); (void)hipStreamSynchronize(m->streams[g]); (void)pbrt_hip_render_wait(m->scenes[g], nullptr); }
  };
  if (code) { drain(); return fail(code, "multi_render: " + err); }  // (known before any collective is enqueued: nobody waits in one)
  // ---- the frame's one exchange, then the film on GPU 0 ----
  const float4 *result = m->slabs[0].p;  // n == 1: GPU 0's own slab / accumulators
  if (n > 1 && m->loopback) {
    // the exchange without RCCL (ranks share devices): rank g's slab to its place in `gathered` on its own stream -- for a wide filter rank
    // 0's accumulators, the others added to them on stream 0 once they are complete (integers: any order gives the same sums)
    const size_t slab_f4 = m->max_slab ? m->max_slab : 1;
    for (int g = 0; g < n; g++) {
      HIP_TRY(hipSetDevice(m->dev[g]));
      if (!wide) HIP_TRY(hipMemcpyPeerAsync(m->gathered.p + (size_t)g * slab_f4, 0, m->slabs[g].p, m->dev[g], slab_f4 * sizeof(float4), m->streams[g]));
      else if (g == 0) HIP_TRY(hipMemcpyAsync(m->gathered.p, m->slabs[0].p, 2 * m->n_px * sizeof(float4), hipMemcpyDeviceToDevice, m->streams[0]));
      HIP_TRY(hipEventRecord(m->done[g], m->streams[g]));
    }
    HIP_TRY(hipSetDevice(0));
    for (int g = 1; g < n; g++) {
      HIP_TRY(hipStreamWaitEvent(m->streams[0], m->done[g], 0));
      if (wide) HIP_TRY(launch_acc_add((unsigned long long *)m->gathered.p, (const unsigned long long *)m->slabs[g].p, 4 * m->n_px, m->streams[0]));
    }
    result = m->gathered.p;
  } else if (n > 1) {
    Rccl &rc = rccl();
    ncclResult_t nr = rc.GroupStart();
    for (int g = 0; g < n && nr == 0; g++) {
      if (hipSetDevice(m->dev[g]) != hipSuccess) { nr = -1; break; }
      nr = wide ? rc.Reduce(m->slabs[g].p, g == 0 ? m->gathered.p : nullptr, 4 * m->n_px, kNcclInt64, kNcclSum, 0, m->comms[g], m->streams[g])
                : rc.Gather(m->slabs[g].p, g == 0 ? m->gathered.p : nullptr, 4 * (m->max_slab ? m->max_slab : 1), kNcclFloat, 0, m->comms[g], m->streams[g]);
    }
    const ncclResult_t ne = rc.GroupEnd();
    if (nr == 0) nr = ne;
    if (nr != 0) {
      // some ranks may sit in a collective the others never joined: abort every communicator (that releases them)
      for (int g = 0; g < n; g++) { (void)hipSetDevice(m->dev[g]); (void)rc.CommAbort(m->comms[g]); m->comms[g] = nullptr; }
      m->broken = true;
      drain();
      return fail(PBRT_HIP_ERR_HIP, std::string("multi_render: RCCL ") + (wide ? "reduce: " : "gather: ") + (nr > 0 ? rc.GetErrorString(nr) : "hipSetDevice failed"));
    }
    result = m->gathered.p;
  }
  HIP_TRY(hipSetDevice(0));
  if (wide) {
    code = pbrt_hip_film_from_acc_device(m->scenes[0], result, m->film.p, m->streams[0]);
  } else {
    if (m->n_px) HIP_TRY(hipMemsetAsync(m->film.p, 0, m->n_px * sizeof(float4), m->streams[0]));
    for (int s = 0; s < n && !code; s++)
      code = pbrt_hip_film_assemble_device(m->scenes[0], result + (size_t)s * (m->max_slab ? m->max_slab : 1), (uint32_t)s, (uint32_t)n, m->film.p, m->streams[0]);
  }
  hipError_t e = hipSuccess;
  if (!code && film && m->n_px) e = hipMemcpyAsync(film, m->film.p, m->n_px * sizeof(float4), hipMemcpyDeviceToHost, m->streams[0]);
  std::vector<pbrt_hip_stats> stats(n);
  for (int g = 0; g < n; g++) {
    (void)hipSetDevice(m->dev[g]);
    const hipError_t es = hipStreamSynchronize(m->streams[g]);
    if (es != hipSuccess && e == hipSuccess) e = es;
    const int c3 = pbrt_hip_render_wait(m->scenes[g], &stats[g]);
    if (c3 && !code) { code = c3; err = "GPU " + std::to_string(g) + ": " + pbrt_hip_last_error(); }
  }
  if (code) return fail(code, "multi_render: " + (err.empty() ? std::string(pbrt_hip_last_error()) : err));
  if (e != hipSuccess) return fail(PBRT_HIP_ERR_HIP, std::string("multi_render: ") + hipGetErrorString(e));
  if (per_gpu) std::memcpy(per_gpu, stats.data(), (size_t)n * sizeof(pbrt_hip_stats));
  return PBRT_HIP_OK;
}

}  // namespace

extern "C" {

int pbrt_hip_multi_create(const pbrt_hip_scene_desc *d, int n_gpus, uint32_t flags, pbrt_hip_multi **out) {
  if (!d || !out) return fail(PBRT_HIP_ERR_INVALID, "multi_create: null argument");
  *out = nullptr;
  try {
    DeviceRestore keep;
    return multi_create(d, n_gpus, flags, out);
  } catch (const std::exception &e) {
    return fail(PBRT_HIP_ERR_INTERNAL, e.what());
  }
}

int pbrt_hip_multi_gpus(const pbrt_hip_multi *m) { return m ? m->n : 0; }

int pbrt_hip_rccl_library(char *path, size_t cap) {
  Rccl &rc = rccl();
  if (!rc.why.empty()) return fail(PBRT_HIP_ERR_INTERNAL, rc.why);
  Dl_info info;
  if (!dladdr((void *)rc.Gather, &info) || !info.dli_fname) return fail(PBRT_HIP_ERR_INTERNAL, "rccl_library: dladdr failed");
  if (path && cap) { std::strncpy(path, info.dli_fname, cap - 1); path[cap - 1] = 0; }
  return PBRT_HIP_OK;
}

void pbrt_hip_multi_destroy(pbrt_hip_multi *m) { delete m; }

int pbrt_hip_multi_render(pbrt_hip_multi *m, const pbrt_hip_render_desc *r, float *film, pbrt_hip_stats *per_gpu) {
  if (!m || !r) return fail(PBRT_HIP_ERR_INVALID, "multi_render: null argument");
  try {
    DeviceRestore keep;
    return multi_render(m, r, film, per_gpu);
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
  // world_end has one frame to render: the accelerator is built on the device (milliseconds)
  int rc = pbrt_hip_multi_create(d, n_gpus, 0u /* the one default: built and optimised on the device */, &m);
  if (rc) return rc;
  rc = pbrt_hip_multi_render(m, r, film, per_gpu);
  const std::string keep = rc ? pbrt_hip_last_error() : "";
  pbrt_hip_multi_destroy(m);
  if (rc) return fail(rc, keep);
  return PBRT_HIP_OK;
}

}  // extern "C"
