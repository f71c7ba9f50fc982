// BVH construction on the device: Morton codes -> radix sort -> binary radix tree
// (Karras 2012) -> bottom-up box fit -> greedy collapse into the quantised 4-wide nodes the production walk
// reads.  The reference has no accelerator at all (core/api.rs:237 is a name), so there is
// nothing to conform to but the RESULT: by the tie rule of DESIGN.md 3.4 a ray's hit does not depend on the shape
// of the tree, so a scene built here renders the same film, bit for bit, as one built by the host's SAH builder
// (tests/test_gpu_parity.py::test_gpu_built_scene_*).  What differs is speed: an LBVH is built in milliseconds
// and walks slower than the SAH tree.  The canonical counters (oracle order) exist only for the host-built tree.
#include <hip/hip_runtime.h>
#include <stdint.h>

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

// bounds[0..2] = min, bounds[3..5] = max of the triangle-box centres, as order-preserving integers
__global__ void centroid_bounds_kernel(const float *P, const uint32_t *idx, uint32_t n, uint32_t *bounds) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  float c[3] = {0.f, 0.f, 0.f};
  const bool live = t < n;
  if (live) {
    float lo[3], hi[3];
    tri_box(P, idx, t, lo, hi);
    for (int a = 0; a < 3; a++) c[a] = 0.5f * lo[a] + 0.5f * hi[a];
  }
  for (int a = 0; a < 3; a++) {
    uint32_t mn = live ? f2ord(c[a]) : 0xffffffffu, mx = live ? f2ord(c[a]) : 0u;
    for (int off = 32; off > 0; off >>= 1) {
      mn = min(mn, (uint32_t)__shfl_down(mn, off, 64));
      mx = max(mx, (uint32_t)__shfl_down(mx, off, 64));
    }
    if ((threadIdx.x & 63u) == 0u) {
      atomicMin(&bounds[a], mn);
      atomicMax(&bounds[3 + a], mx);
    }
  }
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
                           const uint32_t *parent_internal, const uint32_t *parent_leaf, uint32_t *visits, float4 *blo, float4 *bhi) {
  const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (k >= n) return;
  float lo[3], hi[3];
  tri_box(P, idx, vals[k], lo, hi);
  blo[n - 1 + k] = make_float4(lo[0], lo[1], lo[2], 0.f);
  bhi[n - 1 + k] = make_float4(hi[0], hi[1], hi[2], 0.f);
  uint32_t node = parent_leaf[k];
  while (node != kNone) {
    __threadfence();
    if (atomicAdd(&visits[node], 1u) == 0u) return;  // the sibling subtree is not done yet
    __threadfence();
    const uint32_t c0 = child[2 * node], c1 = child[2 * node + 1];
    const uint32_t i0 = (c0 & kLeafRef) ? (uint32_t)(n - 1) + (c0 & ~kLeafRef) : c0, i1 = (c1 & kLeafRef) ? (uint32_t)(n - 1) + (c1 & ~kLeafRef) : c1;
    const volatile float4 *vlo = blo, *vhi = bhi;
    const float4 a0 = make_float4(vlo[i0].x, vlo[i0].y, vlo[i0].z, 0.f), a1 = make_float4(vlo[i1].x, vlo[i1].y, vlo[i1].z, 0.f);
    const float4 b0 = make_float4(vhi[i0].x, vhi[i0].y, vhi[i0].z, 0.f), b1 = make_float4(vhi[i1].x, vhi[i1].y, vhi[i1].z, 0.f);
    blo[node] = make_float4(fminf(a0.x, a1.x), fminf(a0.y, a1.y), fminf(a0.z, a1.z), 0.f);
    bhi[node] = make_float4(fmaxf(b0.x, b1.x), fmaxf(b0.y, b1.y), fmaxf(b0.z, b1.z), 0.f);
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
// counters: [0] quads allocated, [1] items of the next level, [2] stack need (max)
__global__ void collapse_kernel(const CollapseItem *items, uint32_t n_items, int n, const uint32_t *child, const float4 *blo,
                                const float4 *bhi, uint4 *quads, CollapseItem *next, uint32_t *counters) {
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_items) return;
  const CollapseItem it = items[w];
  struct Kid {
    float lo[3], hi[3];
    uint32_t c;  // child word of the radix tree
  } kids[4];
  int nk = 0;
  auto add = [&](uint32_t c) {
    const uint32_t id = (c & kLeafRef) ? (uint32_t)(n - 1) + (c & ~kLeafRef) : c;
    const float4 l = blo[id], h = bhi[id];
    kids[nk].lo[0] = l.x; kids[nk].lo[1] = l.y; kids[nk].lo[2] = l.z;
    kids[nk].hi[0] = h.x; kids[nk].hi[1] = h.y; kids[nk].hi[2] = h.z;
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
  atomicMax(&counters[2], path);
  const float4 mlo = blo[it.node], mhi = bhi[it.node];
  const float me_lo[3] = {mlo.x, mlo.y, mlo.z}, me_hi[3] = {mhi.x, mhi.y, mhi.z};
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
      next[atomicAdd(&counters[1], 1u)] = CollapseItem{c, ref[k], path};
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
  Tmp bounds, keys, keys_out, vals, sort_tmp, child, par_i, par_l, visits, blo, bhi, q0, q1, counters;
  GB_TRY(bounds.alloc(6 * 4));
  GB_TRY(keys.alloc(4 * (size_t)n));
  GB_TRY(keys_out.alloc(4 * (size_t)n));
  GB_TRY(vals.alloc(4 * (size_t)n));
  GB_TRY(child.alloc(8 * (size_t)n));
  GB_TRY(par_i.alloc(4 * (size_t)n));
  GB_TRY(par_l.alloc(4 * (size_t)n));
  GB_TRY(visits.alloc(4 * (size_t)n));
  GB_TRY(blo.alloc(16 * (size_t)(2 * n)));
  GB_TRY(bhi.alloc(16 * (size_t)(2 * n)));
  GB_TRY(q0.alloc(sizeof(CollapseItem) * (size_t)n));
  GB_TRY(q1.alloc(sizeof(CollapseItem) * (size_t)n));
  GB_TRY(counters.alloc(3 * 4));
  size_t sort_bytes = 0;
  GB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, keys.as<uint32_t>(), keys_out.as<uint32_t>(), vals.as<uint32_t>(), d_order, n, 0, 30, stream));
  GB_TRY(sort_tmp.alloc(sort_bytes));

  hipEvent_t e0, e1;
  GB_TRY(hipEventCreate(&e0));
  GB_TRY(hipEventCreate(&e1));
  GB_TRY(hipEventRecord(e0, stream));
  const uint32_t init_bounds[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
  GB_TRY(hipMemcpyAsync(bounds.p, init_bounds, sizeof(init_bounds), hipMemcpyHostToDevice, stream));
  hipLaunchKernelGGL(centroid_bounds_kernel, grid_t, block, 0, stream, d_P, d_idx, n_tris, bounds.as<uint32_t>());
  hipLaunchKernelGGL(morton_kernel, grid_t, block, 0, stream, d_P, d_idx, n_tris, bounds.as<uint32_t>(), keys.as<uint32_t>(), vals.as<uint32_t>());
  GB_TRY(hipGetLastError());
  GB_TRY(hipcub::DeviceRadixSort::SortPairs(sort_tmp.p, sort_bytes, keys.as<uint32_t>(), keys_out.as<uint32_t>(), vals.as<uint32_t>(), d_order, n, 0, 30, stream));
  GB_TRY(hipMemsetAsync(visits.p, 0, 4 * (size_t)n, stream));
  hipLaunchKernelGGL(radix_tree_kernel, grid_t, block, 0, stream, keys_out.as<uint32_t>(), n, child.as<uint32_t>(), par_i.as<uint32_t>(), par_l.as<uint32_t>());
  hipLaunchKernelGGL(fit_kernel, grid_t, block, 0, stream, d_P, d_idx, d_order, n, child.as<uint32_t>(), par_i.as<uint32_t>(), par_l.as<uint32_t>(),
                     visits.as<uint32_t>(), blo.as<float4>(), bhi.as<float4>());
  GB_TRY(hipGetLastError());

  // top-down collapse, one launch per level of the quad tree
  const CollapseItem root{0u, 0u, 0u};
  GB_TRY(hipMemcpyAsync(q0.p, &root, sizeof(root), hipMemcpyHostToDevice, stream));
  uint32_t h_counters[3] = {1u, 0u, 0u};  // quad 0 is the root's
  GB_TRY(hipMemcpyAsync(counters.p, h_counters, sizeof(h_counters), hipMemcpyHostToDevice, stream));
  uint32_t n_items = 1, levels = 0;
  CollapseItem *cur = q0.as<CollapseItem>(), *nxt = q1.as<CollapseItem>();
  while (n_items) {
    hipLaunchKernelGGL(collapse_kernel, dim3((n_items + 127u) / 128u), dim3(128), 0, stream, cur, n_items, n, child.as<uint32_t>(), blo.as<float4>(),
                       bhi.as<float4>(), d_quads, nxt, counters.as<uint32_t>());
    GB_TRY(hipGetLastError());
    GB_TRY(hipMemcpyAsync(h_counters, counters.p, sizeof(h_counters), hipMemcpyDeviceToHost, stream));
    GB_TRY(hipStreamSynchronize(stream));
    n_items = h_counters[1];
    const uint32_t zero = 0;
    GB_TRY(hipMemcpyAsync(counters.as<uint32_t>() + 1, &zero, 4, hipMemcpyHostToDevice, stream));
    CollapseItem *t = cur; cur = nxt; nxt = t;
    levels++;
  }
  float4 rl, rh;
  GB_TRY(hipMemcpyAsync(&rl, blo.p, 16, hipMemcpyDeviceToHost, stream));
  GB_TRY(hipMemcpyAsync(&rh, bhi.p, 16, hipMemcpyDeviceToHost, stream));
  GB_TRY(hipEventRecord(e1, stream));
  GB_TRY(hipStreamSynchronize(stream));
  float ms = 0.f;
  GB_TRY(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  info->n_quads = h_counters[0];
  info->stack_need = h_counters[2];
  info->levels = levels;
  info->root_lo[0] = rl.x; info->root_lo[1] = rl.y; info->root_lo[2] = rl.z;
  info->root_hi[0] = rh.x; info->root_hi[1] = rh.y; info->root_hi[2] = rh.z;
  info->build_ms = ms;
  return hipSuccess;
}

}  // namespace pbrt_hip
