// BVH construction on the device: Morton codes -> radix sort -> binary radix tree
// (Karras 2012) -> bottom-up box fit -> greedy collapse into the quantised 4-wide nodes the production walk
// reads.  The reference has no accelerator at all (core/api.rs:237 is a name), so there is
// nothing to conform to but the RESULT: by the tie rule of DESIGN.md 3.4 a ray's hit does not depend on the shape
// of the tree, so a scene built here renders the same film, bit for bit, as one built by the host's SAH builder
// (tests/test_gpu_parity.py::test_gpu_built_scene_*).  What differs is speed: an LBVH is built in milliseconds
// and walks slower than the SAH tree.  The canonical counters (oracle order) exist only for the host-built tree.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

#include <hipcub/hipcub.hpp>

#include "device_types.h"

namespace pbrt_hip {
namespace {

constexpr uint32_t kLeafRef = 0x80000000u;  // in quad refs (kernels.hip) and in the radix tree's child words
constexpr uint32_t kNone = 0xffffffffu;

__device__ __forceinline__ uint32_t f2ord(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

__device__ __forceinline__ void tri_box(const float *P, const uint32_t *idx, uint32_t t, float lo[3], float hi[3]) {
  const uint32_t i0 = idx[3 * (size_t)t], i1 = idx[3 * (size_t)t + 1], i2 = idx[3 * (size_t)t + 2];
  for (int a = 0; a < 3; a++) {
    const float v0 = P[3 * (size_t)i0 + a], v1 = P[3 * (size_t)i1 + a], v2 = P[3 * (size_t)i2 + a];
    lo[a] = fminf(v0, fminf(v1, v2));
    hi[a] = fmaxf(v0, fmaxf(v1, v2));
  }
}

// A node's box as three 64-bit words {lo.x lo.y}{lo.z hi.x}{hi.y hi.z}: the bottom-up fit reads a sibling's box with
// three device-scope atomic loads (a box another CU has just written must not come from this CU's L1).
__device__ __forceinline__ unsigned long long pack2(float a, float b) {
  return (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32);
}
__device__ __forceinline__ void box_store(unsigned long long *bx, uint32_t id, const float lo[3], const float hi[3]) {
  bx[3 * (size_t)id] = pack2(lo[0], lo[1]);
  bx[3 * (size_t)id + 1] = pack2(lo[2], hi[0]);
  bx[3 * (size_t)id + 2] = pack2(hi[1], hi[2]);
}
template <bool COHERENT>
__device__ __forceinline__ void box_load(const unsigned long long *bx, uint32_t id, float lo[3], float hi[3]) {
  unsigned long long w[3];
  for (int k = 0; k < 3; k++)
    w[k] = COHERENT ? __hip_atomic_load(&bx[3 * (size_t)id + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : bx[3 * (size_t)id + k];
  lo[0] = __uint_as_float((uint32_t)w[0]); lo[1] = __uint_as_float((uint32_t)(w[0] >> 32));
  lo[2] = __uint_as_float((uint32_t)w[1]); hi[0] = __uint_as_float((uint32_t)(w[1] >> 32));
  hi[1] = __uint_as_float((uint32_t)w[2]); hi[2] = __uint_as_float((uint32_t)(w[2] >> 32));
}

// bounds[0..2] = min, bounds[3..5] = max of the triangle-box centres, as order-preserving integers
__global__ void centroid_bounds_kernel(const float *P, const uint32_t *idx, uint32_t n, uint32_t *bounds) {
  __shared__ uint32_t red[6];
  if (threadIdx.x < 6) red[threadIdx.x] = threadIdx.x < 3 ? 0xffffffffu : 0u;
  __syncthreads();
  // a few triangles per thread, then wave, then block: 6 global atomics per 2048 triangles
  float c[3];
  uint32_t mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
  for (uint32_t t = blockIdx.x * blockDim.x * 8u + threadIdx.x, k = 0; k < 8u; k++, t += blockDim.x) {
    if (t >= n) break;
    float lo[3], hi[3];
    tri_box(P, idx, t, lo, hi);
    for (int a = 0; a < 3; a++) {
      c[a] = 0.5f * lo[a] + 0.5f * hi[a];
      mn[a] = min(mn[a], f2ord(c[a]));
      mx[a] = max(mx[a], f2ord(c[a]));
    }
  }
  for (int a = 0; a < 3; a++) {
    for (int off = 32; off > 0; off >>= 1) {
      mn[a] = min(mn[a], (uint32_t)__shfl_down(mn[a], off, 64));
      mx[a] = max(mx[a], (uint32_t)__shfl_down(mx[a], off, 64));
    }
    if ((threadIdx.x & 63u) == 0u) {
      atomicMin(&red[a], mn[a]);
      atomicMax(&red[3 + a], mx[a]);
    }
  }
  __syncthreads();
  if (threadIdx.x < 3) atomicMin(&bounds[threadIdx.x], red[threadIdx.x]);
  else if (threadIdx.x < 6) atomicMax(&bounds[threadIdx.x], red[threadIdx.x]);
}

__device__ __forceinline__ uint32_t spread10(uint32_t v) {  // 10 bits -> every third bit
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

__global__ void morton_kernel(const float *P, const uint32_t *idx, uint32_t n, const uint32_t *bounds, uint32_t *keys, uint32_t *vals) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  float lo[3], hi[3];
  tri_box(P, idx, t, lo, hi);
  uint32_t q[3];
  for (int a = 0; a < 3; a++) {
    const float mn = ord2f(bounds[a]), mx = ord2f(bounds[3 + a]);
    const float c = 0.5f * lo[a] + 0.5f * hi[a];
    const float ext = mx - mn;
    float u = ext > 0.f ? (c - mn) / ext : 0.f;
    u = fminf(fmaxf(u * 1024.f, 0.f), 1023.f);
    q[a] = (uint32_t)u;
  }
  keys[t] = (spread10(q[0]) << 2) | (spread10(q[1]) << 1) | spread10(q[2]);
  vals[t] = t;
}

// length of the common prefix of the keys of sorted positions i and j (ties broken by position), -1 out of range
__device__ __forceinline__ int delta(const uint32_t *keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const uint32_t a = keys[i], b = keys[j];
  if (a == b) return 32 + __clz((uint32_t)i ^ (uint32_t)j);
  return __clz(a ^ b);
}

// Karras 2012, "Maximizing parallelism in the construction of BVHs, octrees and k-d trees", section 4: internal
// node i of the binary radix tree over n sorted keys.  child[2i], child[2i+1]: internal index, or kLeafRef | leaf.
__global__ void radix_tree_kernel(const uint32_t *keys, int n, uint32_t *child, uint32_t *parent_internal, uint32_t *parent_leaf) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n - 1) return;
  const int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
  const int dmin = delta(keys, n, i, i - d);
  int lmax = 2;
  while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
  int l = 0;
  for (int t = lmax / 2; t >= 1; t /= 2)
    if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = delta(keys, n, i, j);
  int s = 0, t = l;
  do {
    t = (t + 1) / 2;
    if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
  } while (t > 1);
  const int gamma = i + s * d + min(d, 0);
  const int lo = min(i, j), hi = max(i, j);
  const uint32_t left = lo == gamma ? (kLeafRef | (uint32_t)gamma) : (uint32_t)gamma;
  const uint32_t right = hi == gamma + 1 ? (kLeafRef | (uint32_t)(gamma + 1)) : (uint32_t)(gamma + 1);
  child[2 * i] = left;
  child[2 * i + 1] = right;
  if (left & kLeafRef) parent_leaf[gamma] = (uint32_t)i; else parent_internal[gamma] = (uint32_t)i;
  if (right & kLeafRef) parent_leaf[gamma + 1] = (uint32_t)i; else parent_internal[gamma + 1] = (uint32_t)i;
  if (i == 0) parent_internal[0] = kNone;
}

// Boxes bottom-up: node ids are internal 0..n-2, leaf k -> (n-1) + k.  The second thread to reach a node fits it.
__global__ void fit_kernel(const float *P, const uint32_t *idx, const uint32_t *vals, int n, const uint32_t *child,
                           const uint32_t *parent_internal, const uint32_t *parent_leaf, uint32_t *visits, unsigned long long *bx) {
  const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (k >= n) return;
  float lo[3], hi[3];
  tri_box(P, idx, vals[k], lo, hi);
  box_store(bx, (uint32_t)(n - 1 + k), lo, hi);
  uint32_t node = parent_leaf[k], from = (uint32_t)(n - 1 + k);
  while (node != kNone) {
    __threadfence();
    if (atomicAdd(&visits[node], 1u) == 0u) return;  // the sibling subtree is not done yet
    __threadfence();
    // this thread carries the box of the child it came from; the other child's was written by another thread
    const uint32_t c0 = child[2 * node], c1 = child[2 * node + 1];
    const uint32_t i0 = (c0 & kLeafRef) ? (uint32_t)(n - 1) + (c0 & ~kLeafRef) : c0, i1 = (c1 & kLeafRef) ? (uint32_t)(n - 1) + (c1 & ~kLeafRef) : c1;
    float slo[3], shi[3];
    box_load<true>(bx, i0 == from ? i1 : i0, slo, shi);
    for (int a = 0; a < 3; a++) {
      lo[a] = fminf(lo[a], slo[a]);
      hi[a] = fmaxf(hi[a], shi[a]);
    }
    box_store(bx, node, lo, hi);
    from = node;
    node = parent_internal[node];
  }
}

struct CollapseItem {
  uint32_t node, quad, path;  // binary internal node, its quad slot, stack entries held above it
};

// One level of the top-down collapse: a quad node takes its binary node's two children and keeps opening the
// interior child with the largest surface area while the result fits four slots (the rule of capi.cpp
// make_quad_nodes; leaves hold one triangle here, so only interior children open).  Child boxes are quantised to
// the node's own 8-bit grid exactly as the host builder does it, enclosure checked in double arithmetic.
// counters: [0] quads allocated, [1] stack need (max); level_count[0] = items of this level, level_count[1] of the next
__global__ void collapse_kernel(const CollapseItem *items, uint32_t *level_count, int n, const uint32_t *child, const unsigned long long *bx,
                                uint4 *quads, CollapseItem *next, uint32_t *counters) {
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= level_count[0]) return;
  const CollapseItem it = items[w];
  struct Kid {
    float lo[3], hi[3];
    uint32_t c;  // child word of the radix tree
  } kids[4];
  int nk = 0;
  auto add = [&](uint32_t c) {
    const uint32_t id = (c & kLeafRef) ? (uint32_t)(n - 1) + (c & ~kLeafRef) : c;
    box_load<false>(bx, id, kids[nk].lo, kids[nk].hi);
    kids[nk].c = c;
    nk++;
  };
  add(child[2 * it.node]);
  add(child[2 * it.node + 1]);
  for (;;) {
    int best = -1;
    float best_area = -1.f;
    for (int k = 0; k < nk; k++) {
      if (kids[k].c & kLeafRef) continue;
      const float dx = kids[k].hi[0] - kids[k].lo[0], dy = kids[k].hi[1] - kids[k].lo[1], dz = kids[k].hi[2] - kids[k].lo[2];
      const float area = (dx * dy + dx * dz) + dy * dz;
      if (nk + 1 <= 4 && area > best_area) { best = k; best_area = area; }
    }
    if (best < 0) break;
    const uint32_t c = kids[best].c;
    kids[best] = kids[--nk];
    add(child[2 * c]);
    add(child[2 * c + 1]);
  }
  const uint32_t path = it.path + (uint32_t)(nk - 1);
  atomicMax(&counters[1], path);
  float me_lo[3], me_hi[3];
  box_load<false>(bx, it.node, me_lo, me_hi);
  uint32_t ebyte[3], qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};
  for (int a = 0; a < 3; a++) {
    const float origin = me_lo[a], extent = me_hi[a] - me_lo[a];
    // smallest power-of-two cell with 255 cells covering the extent (bumped while rounding pushes a plane past 255)
    int e = -126;
    if (extent > 0.f) {
      (void)frexpf(extent / 255.0f, &e);  // extent/255 = m * 2^e, m in [0.5, 1)  =>  2^e >= extent/255
      if (e < -126) e = -126;
    }
    for (; e <= 127; e++) {
      const float cell = ldexpf(1.0f, e);
      bool ok = true;
      uint32_t lo_bytes = 0, hi_bytes = 0;
      for (int k = 0; k < 4 && ok; k++) {
        if (k >= nk) { lo_bytes |= 255u << (8 * k); continue; }
        int ql = (int)floorf((kids[k].lo[a] - origin) / cell), qh = (int)ceilf((kids[k].hi[a] - origin) / cell);
        if (ql < 0) ql = 0;
        if (qh < 0) qh = 0;
        // enclosure checked in exact arithmetic: origin + q * cell fits a double without rounding
        const double o64 = origin, c64 = cell;
        while (ql > 0 && o64 + ql * c64 > (double)kids[k].lo[a]) ql--;
        while (qh <= 255 && o64 + qh * c64 < (double)kids[k].hi[a]) qh++;
        if (ql > 255 || qh > 255 || o64 + ql * c64 > (double)kids[k].lo[a]) { ok = false; break; }
        lo_bytes |= (uint32_t)ql << (8 * k);
        hi_bytes |= (uint32_t)qh << (8 * k);
      }
      if (ok) { qlo[a] = lo_bytes; qhi[a] = hi_bytes; break; }
    }
    ebyte[a] = (uint32_t)((e > 127 ? 127 : e) + 127);
  }
  uint32_t ref[4];
  for (int k = 0; k < 4; k++) {
    if (k >= nk) { ref[k] = kNone; continue; }
    const uint32_t c = kids[k].c;
    if (c & kLeafRef) {
      ref[k] = kLeafRef | (1u << 24) | (c & ~kLeafRef);  // one triangle, leaf slot = sorted position
    } else {
      ref[k] = atomicAdd(&counters[0], 1u);
      next[atomicAdd(&level_count[1], 1u)] = CollapseItem{c, ref[k], path};
    }
  }
  uint4 *q = quads + 4 * (size_t)it.quad;
  q[0] = make_uint4(__float_as_uint(me_lo[0]), __float_as_uint(me_lo[1]), __float_as_uint(me_lo[2]), ebyte[0] | (ebyte[1] << 8) | (ebyte[2] << 16));
  q[1] = make_uint4(qlo[0], qlo[1], qlo[2], qhi[0]);
  q[2] = make_uint4(qhi[1], qhi[2], (ebyte[0] << 7) | (ebyte[1] << 23), ebyte[2] << 7);
  q[3] = make_uint4(ref[0], ref[1], ref[2], ref[3]);
}

struct Tmp {
  void *p = nullptr;
  ~Tmp() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
  template <class T> T *as() { return (T *)p; }
};

}  // namespace

#define GB_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

hipError_t gpu_build_quads(const float *d_P, const uint32_t *d_idx, uint32_t n_tris, uint32_t *d_order, uint4 *d_quads,
                           uint32_t quad_capacity, GpuBuildInfo *info, hipStream_t stream) {
  const int n = (int)n_tris;
  if (n < 2) return hipErrorInvalidValue;  // (the caller builds trees of fewer than two triangles on the host)
  if (quad_capacity + 1u < n_tris) return hipErrorInvalidValue;  // a quad per interior node of the binary tree at most
  const dim3 block(256), grid_t((n_tris + 255u) / 256u);
  Tmp bounds, keys, keys_out, vals, sort_tmp, child, par_i, par_l, visits, bx, q0, q1, counters, level_counts;
  GB_TRY(bounds.alloc(6 * 4));
  GB_TRY(keys.alloc(4 * (size_t)n));
  GB_TRY(keys_out.alloc(4 * (size_t)n));
  GB_TRY(vals.alloc(4 * (size_t)n));
  GB_TRY(child.alloc(8 * (size_t)n));
  GB_TRY(par_i.alloc(4 * (size_t)n));
  GB_TRY(par_l.alloc(4 * (size_t)n));
  GB_TRY(visits.alloc(4 * (size_t)n));
  GB_TRY(bx.alloc(24 * (size_t)(2 * n)));
  GB_TRY(q0.alloc(sizeof(CollapseItem) * (size_t)n));
  GB_TRY(q1.alloc(sizeof(CollapseItem) * (size_t)n));
  GB_TRY(counters.alloc(2 * 4));
  GB_TRY(level_counts.alloc(64 * 4));
  size_t sort_bytes = 0;
  GB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, keys.as<uint32_t>(), keys_out.as<uint32_t>(), vals.as<uint32_t>(), d_order, n, 0, 30, stream));
  GB_TRY(sort_tmp.alloc(sort_bytes));

  hipEvent_t e0, e1;
  GB_TRY(hipEventCreate(&e0));
  GB_TRY(hipEventCreate(&e1));
  GB_TRY(hipEventRecord(e0, stream));
  const uint32_t init_bounds[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
  GB_TRY(hipMemcpyAsync(bounds.p, init_bounds, sizeof(init_bounds), hipMemcpyHostToDevice, stream));
  hipLaunchKernelGGL(centroid_bounds_kernel, dim3((n_tris + 2047u) / 2048u), block, 0, stream, d_P, d_idx, n_tris, bounds.as<uint32_t>());
  hipLaunchKernelGGL(morton_kernel, grid_t, block, 0, stream, d_P, d_idx, n_tris, bounds.as<uint32_t>(), keys.as<uint32_t>(), vals.as<uint32_t>());
  GB_TRY(hipGetLastError());
  GB_TRY(hipcub::DeviceRadixSort::SortPairs(sort_tmp.p, sort_bytes, keys.as<uint32_t>(), keys_out.as<uint32_t>(), vals.as<uint32_t>(), d_order, n, 0, 30, stream));
  GB_TRY(hipMemsetAsync(visits.p, 0, 4 * (size_t)n, stream));
  hipLaunchKernelGGL(radix_tree_kernel, grid_t, block, 0, stream, keys_out.as<uint32_t>(), n, child.as<uint32_t>(), par_i.as<uint32_t>(), par_l.as<uint32_t>());
  hipLaunchKernelGGL(fit_kernel, grid_t, block, 0, stream, d_P, d_idx, d_order, n, child.as<uint32_t>(), par_i.as<uint32_t>(), par_l.as<uint32_t>(),
                     visits.as<uint32_t>(), bx.as<unsigned long long>());
  GB_TRY(hipGetLastError());

  // top-down collapse, one launch per level of the quad tree.  The host does not know how many items a level holds
  // (the kernel reads the count the level above left on the device) nor how deep the tree is: it launches kBatch
  // levels blind -- level L holds at most min(4^L, n) items -- and looks at the device once per batch.
  constexpr uint32_t kBatch = 48;
  const CollapseItem root{0u, 0u, 0u};
  GB_TRY(hipMemcpyAsync(q0.p, &root, sizeof(root), hipMemcpyHostToDevice, stream));
  uint32_t h_counters[2] = {1u, 0u};  // quad 0 is the root's
  GB_TRY(hipMemcpyAsync(counters.p, h_counters, sizeof(h_counters), hipMemcpyHostToDevice, stream));
  uint32_t levels = 0;
  CollapseItem *cur = q0.as<CollapseItem>(), *nxt = q1.as<CollapseItem>();
  for (uint32_t first = 1;;) {  // `first`: items of the batch's first level
    std::vector<uint32_t> zeros(kBatch + 1, 0u);
    zeros[0] = first;
    GB_TRY(hipMemcpyAsync(level_counts.p, zeros.data(), 4 * zeros.size(), hipMemcpyHostToDevice, stream));
    GB_TRY(hipStreamSynchronize(stream));  // (zeros is a local)
    uint64_t bound = first;
    for (uint32_t l = 0; l < kBatch; l++) {
      const uint32_t cap = (uint32_t)std::min<uint64_t>(bound, (uint64_t)n);
      hipLaunchKernelGGL(collapse_kernel, dim3((cap + 127u) / 128u), dim3(128), 0, stream, cur, level_counts.as<uint32_t>() + l, n,
                         child.as<uint32_t>(), bx.as<unsigned long long>(), d_quads, nxt, counters.as<uint32_t>());
      CollapseItem *t = cur; cur = nxt; nxt = t;
      bound = std::min<uint64_t>(bound * 4u, (uint64_t)n);
    }
    GB_TRY(hipGetLastError());
    std::vector<uint32_t> got(kBatch + 1);
    GB_TRY(hipMemcpyAsync(got.data(), level_counts.p, 4 * got.size(), hipMemcpyDeviceToHost, stream));
    GB_TRY(hipStreamSynchronize(stream));
    for (uint32_t l = 0; l < kBatch && got[l]; l++) levels++;
    first = got[kBatch];
    if (!first) break;
  }
  GB_TRY(hipMemcpyAsync(h_counters, counters.p, sizeof(h_counters), hipMemcpyDeviceToHost, stream));
  float root_box[6];  // {lo.x lo.y lo.z hi.x hi.y hi.z} of node 0
  GB_TRY(hipMemcpyAsync(root_box, bx.p, 24, hipMemcpyDeviceToHost, stream));
  GB_TRY(hipEventRecord(e1, stream));
  GB_TRY(hipStreamSynchronize(stream));
  float ms = 0.f;
  GB_TRY(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  info->n_quads = h_counters[0];
  info->stack_need = h_counters[1];
  info->levels = levels;
  for (int a = 0; a < 3; a++) { info->root_lo[a] = root_box[a]; info->root_hi[a] = root_box[3 + a]; }
  info->build_ms = ms;
  return hipSuccess;
}

}  // namespace pbrt_hip
