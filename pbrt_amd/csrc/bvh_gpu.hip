// BVH construction on the device, three stages:
//   1. the host builder's algorithm -- top-down binned SAH (centre bounds -> widest axis -> 16 buckets -> plane of
//      least n_l A_l + n_r A_r) -- run LEVEL-SYNCHRONOUSLY over the Morton-sorted triangles (every segment of a level is split by
//      the same launches), down to leaves of one triangle;
//   2. the built tree OPTIMISED by parallel re-insertion (reinsert_core.hpp: every node looks for the position that lowers the
//      summed surface area most, conflicting moves are dropped, boxes refitted; up to 12 passes): 4.5 % fewer node fetches per ray
//      on BASELINE C3, 6.5 % on C2;
//   3. the host's dynamic-programming collapse into the quantised 4-wide nodes the production walk reads.
// 0.13 s for 1M triangles (34 ms without stage 2: PBRT_HIP_SCENE_PLAIN_TREE) against the host's second.  Round 1's LBVH and PLOC
// builders are records now (tools/experiments/r01_lbvh_ploc_builders.hip.txt).
// The reference has no accelerator at all (core/api.rs:237 is a name), so there is nothing to conform to but the RESULT:
// by the tie rule of DESIGN.md 3.4 a ray's hit does not depend on the shape of the tree, so a scene built here renders
// the same film, bit for bit, as one built by the host's SAH builder (tests/test_gpu_parity.py::test_gpu_built_scene_*).
// The canonical counters (the oracle's walk of ITS tree) of such a scene come from the oracle's tree, built on the host
// the first time they are asked for (capi.cpp ensure_canonical).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>  // (rocPRIM directly: hipCUB is its CUDA-compatibility face)
#include <rocprim/device/device_scan.hpp>

#include "device_types.h"
#include "reinsert_core.hpp"

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

__device__ __forceinline__ uint32_t node_of(uint32_t ref, int n) { return (ref & kLeafRef) ? (uint32_t)(n - 1) + (ref & ~kLeafRef) : ref; }

// ---- top-down binned SAH, level-synchronous (the host builder's split rule, DESIGN.md 3.3, on the device) ----
// The triangles (in Morton order to start with) form SEGMENTS of the position array; every level every segment of two
// or more triangles is split: bounds of the triangle-box centres -> widest axis -> 16 buckets with counts and boxes ->
// the plane of least n_l * A_l + n_r * A_r (planes with an empty side excluded) -> a stable partition (one global prefix
// sum of the "goes left" flags).  No plane (all centres in one bucket), or a level past kSahMedianLevel: the segment is
// halved as it stands.  Leaves hold one triangle.  Node ids: the root is 0, the segments of the next level get
// consecutive ids in segment order, so the numbering is deterministic.  Per-segment sums go through atomics, aggregated in
// LDS when a whole block lies in one segment (the top levels).
#ifndef PBRT_SAH_BUCKETS
#define PBRT_SAH_BUCKETS 16
#endif
constexpr int kSahBuckets = PBRT_SAH_BUCKETS;
constexpr uint32_t kSahMedianLevel = 40;  // beyond this level only halving: bounds the depth at ~ 40 + log2(n)
constexpr uint32_t kBinWords = kSahBuckets * 7;  // per bucket: count, box lo xyz, hi xyz (order-preserving integers)

struct SahSegs {  // one level's segment table (structure of arrays; capacity n / 2 + 1)
  uint32_t *start, *end, *node;
};
struct SahSplit {
  uint32_t axis, best, nl, mode;  // mode 0: bucket <= best goes left; 1: position < start + nl goes left
  float c0, scale;                // bucket = min(15, (int)((c - c0) * scale))
};

__global__ void sah_tri_boxes_kernel(const float *P, const uint32_t *idx, uint32_t n, float *tlo, float *thi) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  float lo[3], hi[3];
  tri_box(P, idx, t, lo, hi);
  for (int a = 0; a < 3; a++) { tlo[3 * (size_t)t + a] = lo[a]; thi[3 * (size_t)t + a] = hi[a]; }
}

// centre bounds of every active segment: cb[s][0..2] = min, [3..5] = max (order-preserving integers, preset to ~0 / 0)
__global__ void __launch_bounds__(256) sah_bounds_kernel(const uint32_t *ids, const uint32_t *seg_of, int n, const float *tlo, const float *thi, uint32_t *cb) {
  __shared__ uint32_t red[6];
  __shared__ uint32_t s_first, s_last;
  const int i = (int)(blockIdx.x * 256 + threadIdx.x);
  if (threadIdx.x == 0) { s_first = seg_of[blockIdx.x * 256]; s_last = seg_of[min((int)(blockIdx.x * 256 + 255), n - 1)]; }
  if (threadIdx.x < 6) red[threadIdx.x] = threadIdx.x < 3 ? 0xffffffffu : 0u;
  __syncthreads();
  const bool uniform = s_first == s_last && s_first != kNone;  // segments are contiguous: the whole block is in one
  const uint32_t s = i < n ? seg_of[i] : kNone;
  if (s != kNone) {
    const uint32_t t = ids[i];
    for (int a = 0; a < 3; a++) {
      const uint32_t c = f2ord(0.5f * tlo[3 * (size_t)t + a] + 0.5f * thi[3 * (size_t)t + a]);
      if (uniform) { atomicMin(&red[a], c); atomicMax(&red[3 + a], c); }
      else { atomicMin(&cb[6 * (size_t)s + a], c); atomicMax(&cb[6 * (size_t)s + 3 + a], c); }
    }
  }
  if (uniform) {
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&cb[6 * (size_t)s_first + threadIdx.x], red[threadIdx.x]);
    else if (threadIdx.x < 6) atomicMax(&cb[6 * (size_t)s_first + threadIdx.x], red[threadIdx.x]);
  }
}

__device__ __forceinline__ uint32_t sah_bucket(float c, float c0, float scale) {
  const int b = (int)((c - c0) * scale);
  return (uint32_t)(b < 0 ? 0 : (b >= kSahBuckets ? kSahBuckets - 1 : b));
}

// widest axis of the centre bounds and the bucket scale of every segment
__global__ void sah_prepare_kernel(int n_seg, const uint32_t *cb, SahSplit *sp) {
  const int s = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (s >= n_seg) return;
  float ext[3], mn[3];
  for (int a = 0; a < 3; a++) { mn[a] = ord2f(cb[6 * (size_t)s + a]); ext[a] = ord2f(cb[6 * (size_t)s + 3 + a]) - mn[a]; }
  const int axis = (ext[0] > ext[1] && ext[0] > ext[2]) ? 0 : (ext[1] > ext[2] ? 1 : 2);
  SahSplit o;
  o.axis = (uint32_t)axis;
  o.c0 = mn[axis];
  o.scale = ext[axis] > 0.f ? (float)kSahBuckets / ext[axis] : 0.f;
  o.best = 0; o.nl = 0; o.mode = 1;
  sp[s] = o;
}

// bucket counts and boxes of every segment (bins preset: count 0, lo ~0, hi 0)
__global__ void __launch_bounds__(256) sah_bins_kernel(const uint32_t *ids, const uint32_t *seg_of, int n, const float *tlo, const float *thi, const SahSplit *sp,
                                                       uint32_t *bins) {
  __shared__ uint32_t lb[kBinWords];
  __shared__ uint32_t s_first, s_last;
  const int i = (int)(blockIdx.x * 256 + threadIdx.x);
  if (threadIdx.x == 0) { s_first = seg_of[blockIdx.x * 256]; s_last = seg_of[min((int)(blockIdx.x * 256 + 255), n - 1)]; }
  if (threadIdx.x < kBinWords) lb[threadIdx.x] = (threadIdx.x % 7u == 0u || threadIdx.x % 7u > 3u) ? 0u : 0xffffffffu;
  __syncthreads();
  const bool uniform = s_first == s_last && s_first != kNone;
  const uint32_t s = i < n ? seg_of[i] : kNone;
  if (s != kNone) {
    const uint32_t t = ids[i];
    const SahSplit p = sp[s];
    const float c = 0.5f * tlo[3 * (size_t)t + p.axis] + 0.5f * thi[3 * (size_t)t + p.axis];
    const uint32_t b = sah_bucket(c, p.c0, p.scale);
    uint32_t *dst = uniform ? &lb[7u * b] : &bins[(size_t)s * kBinWords + 7u * b];
    atomicAdd(&dst[0], 1u);
    for (int a = 0; a < 3; a++) {
      atomicMin(&dst[1 + a], f2ord(tlo[3 * (size_t)t + a]));
      atomicMax(&dst[4 + a], f2ord(thi[3 * (size_t)t + a]));
    }
  }
  if (uniform) {
    __syncthreads();
    if (threadIdx.x < kBinWords) {
      uint32_t *dst = &bins[(size_t)s_first * kBinWords + threadIdx.x];
      const uint32_t k = threadIdx.x % 7u, v = lb[threadIdx.x];
      if (k == 0u) { if (v) atomicAdd(dst, v); }
      else if (k <= 3u) atomicMin(dst, v);
      else atomicMax(dst, v);
    }
  }
}

// the split of every segment: its own box (union of the buckets) goes to bx[node]; active[s] = children of >= 2 triangles
__global__ void sah_split_kernel(int n_seg, SahSegs segs, const uint32_t *bins, SahSplit *sp, uint32_t level, unsigned long long *bx, uint32_t *active) {
  const int s = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (s >= n_seg) return;
  const uint32_t *b = bins + (size_t)s * kBinWords;
  const uint32_t n = segs.end[s] - segs.start[s];
  uint32_t cnt[kSahBuckets];
  float lo[kSahBuckets][3], hi[kSahBuckets][3];
  float all_lo[3] = {__builtin_huge_valf(), __builtin_huge_valf(), __builtin_huge_valf()}, all_hi[3] = {-__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf()};
  for (int k = 0; k < kSahBuckets; k++) {
    cnt[k] = b[7 * k];
    for (int a = 0; a < 3; a++) {
      lo[k][a] = ord2f(b[7 * k + 1 + a]);
      hi[k][a] = ord2f(b[7 * k + 4 + a]);
      if (cnt[k]) { all_lo[a] = fminf(all_lo[a], lo[k][a]); all_hi[a] = fmaxf(all_hi[a], hi[k][a]); }
    }
  }
  box_store(bx, segs.node[s], all_lo, all_hi);
  // suffix sweep, then prefix sweep over the 15 planes
  float area_r[kSahBuckets];
  uint32_t cnt_r[kSahBuckets];
  {
    float rl[3] = {__builtin_huge_valf(), __builtin_huge_valf(), __builtin_huge_valf()}, rh[3] = {-__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf()};
    uint32_t c = 0;
    for (int k = kSahBuckets - 1; k >= 1; k--) {
      if (cnt[k]) { for (int a = 0; a < 3; a++) { rl[a] = fminf(rl[a], lo[k][a]); rh[a] = fmaxf(rh[a], hi[k][a]); } c += cnt[k]; }
      const float dx = rh[0] - rl[0], dy = rh[1] - rl[1], dz = rh[2] - rl[2];
      cnt_r[k - 1] = c;
      area_r[k - 1] = c ? (dx * dy + dx * dz) + dy * dz : 0.f;
    }
  }
  SahSplit o = sp[s];
  float best_cost = __builtin_huge_valf();
  int best = -1;
  {
    float ll[3] = {__builtin_huge_valf(), __builtin_huge_valf(), __builtin_huge_valf()}, lh[3] = {-__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf()};
    uint32_t c = 0;
    for (int k = 0; k < kSahBuckets - 1; k++) {
      if (cnt[k]) { for (int a = 0; a < 3; a++) { ll[a] = fminf(ll[a], lo[k][a]); lh[a] = fmaxf(lh[a], hi[k][a]); } c += cnt[k]; }
      if (c == 0 || cnt_r[k] == 0) continue;  // a plane with an empty side splits nothing
      const float dx = lh[0] - ll[0], dy = lh[1] - ll[1], dz = lh[2] - ll[2];
      const float cost = (float)c * ((dx * dy + dx * dz) + dy * dz) + (float)cnt_r[k] * area_r[k];
      if (cost < best_cost) { best_cost = cost; best = k; o.nl = c; }
    }
  }
  if (best < 0 || level >= kSahMedianLevel) { o.mode = 1; o.nl = n / 2; o.best = 0; }
  else { o.mode = 0; o.best = (uint32_t)best; }
  sp[s] = o;
  active[s] = (o.nl >= 2u ? 1u : 0u) + (n - o.nl >= 2u ? 1u : 0u);
}

__global__ void sah_flags_kernel(const uint32_t *ids, const uint32_t *seg_of, int n, const float *tlo, const float *thi, SahSegs segs, const SahSplit *sp, uint32_t *flags) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n) return;
  const uint32_t s = seg_of[i];
  uint32_t f = 0;
  if (s != kNone) {
    const SahSplit p = sp[s];
    if (p.mode) f = (uint32_t)i < segs.start[s] + p.nl ? 1u : 0u;
    else {
      const uint32_t t = ids[i];
      f = sah_bucket(0.5f * tlo[3 * (size_t)t + p.axis] + 0.5f * thi[3 * (size_t)t + p.axis], p.c0, p.scale) <= p.best ? 1u : 0u;
    }
  }
  flags[i] = f;
}

// stable partition of every segment (scan = exclusive prefix sum of flags over the whole array) and the segment each
// position belongs to on the next level (kNone: a finished leaf)
__global__ void sah_scatter_kernel(const uint32_t *ids, const uint32_t *seg_of, int n, SahSegs segs, const SahSplit *sp, const uint32_t *flags, const uint32_t *scan,
                                   const uint32_t *child_base, uint32_t *ids_out, uint32_t *seg_out) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n) return;
  const uint32_t s = seg_of[i];
  if (s == kNone) { ids_out[i] = ids[i]; seg_out[i] = kNone; return; }
  const uint32_t st = segs.start[s], nl = sp[s].nl, nr = segs.end[s] - st - nl;
  const uint32_t left_before = scan[i] - scan[st];
  const bool left = flags[i] != 0u;
  const uint32_t pos = left ? st + left_before : st + nl + ((uint32_t)i - st - left_before);
  ids_out[pos] = ids[i];
  const uint32_t first = child_base[s];  // next level's index of this segment's first active child
  seg_out[pos] = left ? (nl >= 2u ? first : kNone) : (nr >= 2u ? first + (nl >= 2u ? 1u : 0u) : kNone);
}

// the node of every segment gets its two children; the active children become the next level's segments
__global__ void sah_children_kernel(int n_seg, SahSegs segs, const SahSplit *sp, const uint32_t *child_base, uint32_t next_node_base, uint32_t *child, SahSegs next,
                                    uint32_t *parent_internal, uint32_t *parent_leaf) {
  const int s = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (s >= n_seg) return;
  const uint32_t st = segs.start[s], en = segs.end[s], nl = sp[s].nl, nr = en - st - nl, node = segs.node[s];
  uint32_t j = child_base[s];
  uint32_t left, right;
  if (nl >= 2u) { left = next_node_base + j; next.start[j] = st; next.end[j] = st + nl; next.node[j] = left; j++; }
  else left = kLeafRef | st;
  if (nr >= 2u) { right = next_node_base + j; next.start[j] = st + nl; next.end[j] = en; next.node[j] = right; }
  else right = kLeafRef | (st + nl);
  child[2 * (size_t)node] = left;
  child[2 * (size_t)node + 1] = right;
  // (parents: the collapse's dynamic programme walks the tree bottom-up)
  if (left & kLeafRef) parent_leaf[left & ~kLeafRef] = node; else parent_internal[left] = node;
  if (right & kLeafRef) parent_leaf[right & ~kLeafRef] = node; else parent_internal[right] = node;
  if (node == 0u) parent_internal[0] = kNone;
}

__global__ void sah_leaf_boxes_kernel(const uint32_t *ids, int n, const float *tlo, const float *thi, unsigned long long *bx, uint32_t *order) {
  const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (k >= n) return;
  const uint32_t t = ids[k];
  float lo[3], hi[3];
  for (int a = 0; a < 3; a++) { lo[a] = tlo[3 * (size_t)t + a]; hi[a] = thi[3 * (size_t)t + a]; }
  box_store(bx, (uint32_t)(n - 1 + k), lo, hi);
  order[k] = t;
}

// ---- parallel re-insertion over the built binary tree (reinsert_core.hpp has the algorithm and what each phase may touch) ----
// Unified node ids: interior nodes as the builder numbered them, [0, n - 1); the leaf of slot k = (n - 1) + k: the numbering
// of the box array `bx`.
__global__ void ri_links_kernel(int n, const uint32_t *child, uint32_t *par, uint32_t *kid) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n - 1) return;
  for (int k = 0; k < 2; k++) {
    const uint32_t c = child[2 * (size_t)i + k], id = node_of(c, n);
    kid[2 * (size_t)i + k] = id;
    par[id] = (uint32_t)i;
  }
  if (i == 0) par[0] = kNone;
}

// stats: [0] nodes visited by the searches, [1] searches that found a move, [2] moves applied (summed over the passes)
__global__ void __launch_bounds__(256) ri_search_kernel(reins::Tree t, reins::Search sp, uint32_t pass, uint32_t mu, uint32_t *mv_y, uint32_t *mv_lca,
                                                        float *mv_gain, unsigned long long *stats) {
  __shared__ unsigned int s_visits, s_found;
  if (threadIdx.x == 0) { s_visits = 0u; s_found = 0u; }
  __syncthreads();
  const uint32_t x = blockIdx.x * 256u + threadIdx.x, n_nodes = 2u * t.n_int + 1u;
  if (x < n_nodes) {
    reins::Move m;
    m.y = kNone; m.lca = kNone; m.gain = 0.f; m.visits = 0u;
    if ((x + pass) % mu == 0u) m = reins::find_move(t, x, sp);
    mv_y[x] = m.y;
    mv_lca[x] = m.lca;
    mv_gain[x] = m.gain;
    atomicAdd(&s_visits, m.visits);
    if (m.y != kNone) atomicAdd(&s_found, 1u);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_visits) atomicAdd(&stats[0], (unsigned long long)s_visits);
    if (s_found) atomicAdd(&stats[1], (unsigned long long)s_found);
  }
}

__global__ void ri_lock_kernel(reins::Tree t, const uint32_t *mv_y, const float *mv_gain, unsigned long long *lock) {
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= 2u * t.n_int + 1u || mv_y[x] == kNone) return;
  const unsigned long long key = reins::move_key(x, mv_gain[x]);
  reins::for_move_nodes(t, x, mv_y[x], [&](uint32_t q) { atomicMax(&lock[q], key); });
}

__global__ void ri_hold_kernel(reins::Tree t, const uint32_t *mv_y, const float *mv_gain, const unsigned long long *lock, uint32_t *holds) {
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= 2u * t.n_int + 1u) return;
  bool ok = mv_y[x] != kNone;
  if (ok) {
    const unsigned long long key = reins::move_key(x, mv_gain[x]);
    reins::for_move_nodes(t, x, mv_y[x], [&](uint32_t q) { if (lock[q] != key) ok = false; });
  }
  holds[x] = ok ? 1u : 0u;
}

// (a holder on the target's path blocks only with a LARGER key: reinsert_core.hpp target_path_is_free)
__global__ void ri_free_kernel(reins::Tree t, uint32_t *mv_y, const uint32_t *mv_lca, const float *mv_gain, const uint32_t *holds) {
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= 2u * t.n_int + 1u || mv_y[x] == kNone) return;
  const unsigned long long key = reins::move_key(x, mv_gain[x]);
  if (!(holds[x] && reins::target_path_is_free(t, x, mv_y[x], mv_lca[x], [&](uint32_t q) { return holds[q] != 0u && reins::move_key(q, mv_gain[q]) > key; })))
    mv_y[x] = kNone;
}

// (mv_from[x] = the grandparent x leaves: with its new parent, the two places whose ancestors' boxes the move changes)
__global__ void __launch_bounds__(256) ri_apply_kernel(reins::Tree t, const uint32_t *mv_y, uint32_t *mv_from, unsigned long long *stats) {
  __shared__ unsigned int s_applied;
  if (threadIdx.x == 0) s_applied = 0u;
  __syncthreads();
  const uint32_t x = blockIdx.x * 256u + threadIdx.x;
  if (x < 2u * t.n_int + 1u && mv_y[x] != kNone) {
    mv_from[x] = t.par[t.par[x]];  // (x's parent and grandparent are among the six nodes whose links this move alone may write)
    reins::apply_move(t, x, mv_y[x]);
    atomicAdd(&s_applied, 1u);
  }
  __syncthreads();
  if (threadIdx.x == 0 && s_applied) atomicAdd(&stats[2], (unsigned long long)s_applied);
}

// all boxes bottom-up over the links (one thread per leaf climbs; the second thread to reach a node fits it); COUNT: also the
// number of leaves below every interior node
template <bool COUNT, bool BOXES = true>
__global__ void ri_refit_kernel(reins::Tree t, uint32_t *visits, uint32_t *cnt) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > t.n_int) return;
  uint32_t node = t.n_int + k, c = 1u;
  float lo[3], hi[3];
  if (BOXES) box_load<false>(t.bx, node, lo, hi);  // (leaf boxes were written by an earlier launch)
  uint32_t p = t.par[node];
  while (p != kNone) {
    __threadfence();
    if (atomicAdd(&visits[p], 1u) == 0u) return;  // the sibling subtree is not done yet
    __threadfence();
    const uint32_t sib = reins::sibling(t, p, node);
    if (BOXES) {
      float slo[3], shi[3];
      box_load<true>(t.bx, sib, slo, shi);
      for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], slo[a]); hi[a] = fmaxf(hi[a], shi[a]); }
      box_store(t.bx, p, lo, hi);
    }
    if (COUNT) {
      c += sib >= t.n_int ? 1u : __hip_atomic_load(&cnt[sib], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      cnt[p] = c;
    }
    node = p;
    p = t.par[p];
  }
}

// ---- refit of what a pass moved (a pass moves 1.4 % of the nodes, the full refit above was 3/4 of its time) ----
// mark[i], interior node i: bit 31 = some box below i changed (i must be refitted), low bits = children that have arrived.
constexpr uint32_t kDirty = 0x80000000u;
// every applied move dirties the chain above x's new parent (the re-used node p, now y's parent too) and above the grandparent x
// left; a chain that meets a dirty node stops -- whoever dirtied it climbs on
__global__ void ri_mark_kernel(reins::Tree t, const uint32_t *mv_y, const uint32_t *mv_from, uint32_t *mark) {
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= 2u * t.n_int + 1u || mv_y[x] == kNone) return;
  for (int k = 0; k < 2; k++)
    for (uint32_t q = k ? mv_from[x] : t.par[x]; q != kNone; q = t.par[q])
      if (atomicOr(&mark[q], kDirty) & kDirty) break;
}
// One thread per interior node; the dirty nodes without a dirty child start (their children's boxes are final), fit their box and
// climb; at a parent whose other child is dirty too the first to arrive stops and the second fits (as in ri_refit_kernel).  The
// boxes that result are those of a full refit: a node that is not dirty has nothing changed below it, and min / max are exact.
__global__ void ri_refit_dirty_kernel(reins::Tree t, uint32_t *mark) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= t.n_int || !(mark[i] & kDirty)) return;
  auto dirty = [&](uint32_t c) { return c < t.n_int && (__hip_atomic_load(&mark[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & kDirty) != 0u; };
  uint32_t node = i;
  const uint32_t k0 = t.kid[2 * (size_t)i], k1 = t.kid[2 * (size_t)i + 1];
  if (dirty(k0) || dirty(k1)) return;  // (a thread from below arrives here later)
  float lo[3], hi[3], slo[3], shi[3];
  box_load<false>(t.bx, k0, lo, hi);  // (neither child's box changed in this pass)
  box_load<false>(t.bx, k1, slo, shi);
  for (;;) {
    for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], slo[a]); hi[a] = fmaxf(hi[a], shi[a]); }
    box_store(t.bx, node, lo, hi);  // (lo, hi) stays in registers: the parent is fitted from it and the sibling's box
    const uint32_t p = t.par[node];
    if (p == kNone) return;
    const uint32_t sib = reins::sibling(t, p, node);
    if (dirty(sib)) {
      // the sibling's subtree is being refitted too: the first to arrive has published its box and stops, the second reads it
      __threadfence();
      if ((atomicAdd(&mark[p], 1u) & ~kDirty) == 0u) return;
      __threadfence();
      box_load<true>(t.bx, sib, slo, shi);
    } else {
      box_load<false>(t.bx, sib, slo, shi);  // (untouched by this pass: no ordering to establish, a chain of one thread)
    }
    node = p;
  }
}
// The summed (half) surface area of the interior nodes in fixed point -- integers add up to the same sum in any order --, units of
// 2^-se with the root's area below 2^40 units (find_move's scale): out[0] += the sum, out[1] = se + 1024.
__global__ void __launch_bounds__(256) ri_cost_kernel(reins::Tree t, unsigned long long *out) {
  __shared__ unsigned long long s_sum[4];
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  int se = 0;
  (void)frexpf(reins::area(reins::load_box(t, 0u)), &se);
  se = 40 - se;
  se = se > 100 ? 100 : (se < -100 ? -100 : se);
  unsigned long long v = i < t.n_int ? (unsigned long long)(reins::area(reins::load_box(t, i)) * ldexpf(1.0f, se)) : 0ull;
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if ((threadIdx.x & 63u) == 0u) s_sum[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0u) {
    const unsigned long long b = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
    if (b) atomicAdd(&out[0], b);
  }
  if (i == 0u) out[1] = (unsigned long long)(se + 1024);
}

__global__ void ri_order_kernel(reins::Tree t) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < t.n_int) reins::order_children(t, i);
}

// a leaf's position in the depth-first order of the optimised tree = the leaves to its left: the triangle records follow the
// tree again (a moved triangle would otherwise keep its record where the first tree had it)
__global__ void ri_leafpos_kernel(reins::Tree t, const uint32_t *cnt, uint32_t *newslot) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > t.n_int) return;
  uint32_t node = t.n_int + k, pos = 0u;
  for (uint32_t p = t.par[node]; p != kNone; node = p, p = t.par[p]) {
    const uint32_t first = t.kid[2 * (size_t)p];
    if (first != node) pos += first >= t.n_int ? 1u : cnt[first];
  }
  newslot[k] = pos;
}

// back to the builder's own form (child words, parents of interior nodes and of leaf slots) with the new leaf slots
__global__ void ri_emit_kernel(reins::Tree t, const uint32_t *newslot, uint32_t *child, uint32_t *parent_internal, uint32_t *parent_leaf) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= t.n_int) return;
  for (int k = 0; k < 2; k++) {
    const uint32_t c = t.kid[2 * (size_t)i + k];
    if (c >= t.n_int) {
      const uint32_t s = newslot[c - t.n_int];
      child[2 * (size_t)i + k] = kLeafRef | s;
      parent_leaf[s] = i;
    } else {
      child[2 * (size_t)i + k] = c;
      parent_internal[c] = i;
    }
  }
  if (i == 0u) parent_internal[0] = kNone;
}
__global__ void ri_leaves_kernel(reins::Tree t, const uint32_t *newslot, const uint32_t *order_in, uint32_t *order_out, unsigned long long *leaf_bx_out) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k > t.n_int) return;
  const uint32_t s = newslot[k];
  order_out[s] = order_in[k];
  for (int w = 0; w < 3; w++) leaf_bx_out[3 * (size_t)s + w] = t.bx[3 * (size_t)(t.n_int + k) + w];
}

// ---- the collapse's dynamic programme (capi.cpp make_quad_nodes_as, after Ylitie, Karras, Laine 2017, section 3.2) ----
// F(n, k, d) = least expected work inside subtree n when n may occupy up to k child slots of the quad node made of its
// ancestor at binary distance d; G(n) = the work below n as a quad node of its own.  A child is reached with the probability
// of its box AS THE ANCESTOR'S 8-BIT GRID HOLDS IT (about one cell wider per axis).  Leaves hold one triangle here.
constexpr float kDpTriCost = 2.0f;
struct DpTables {
  float *F;   // [(4 * node + (k - 1)) * 3 + (d - 1)], internal nodes
  float *Fl;  // [3 * slot + (d - 1)], leaves
  float *G;   // internal nodes
};
__device__ __forceinline__ float dp_load(const float *p) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ float dp_f(const DpTables &T, uint32_t c, uint32_t k, uint32_t d) {
  return (c & kLeafRef) ? dp_load(&T.Fl[3 * (size_t)(c & ~kLeafRef) + (d - 1u)]) : dp_load(&T.F[(4 * (size_t)c + (k - 1u)) * 3 + (d - 1u)]);
}
// surface area of box (lo, hi) as the grid of a quad node with box (qlo, qhi) holds it
__device__ __forceinline__ float area_on_grid(const float lo[3], const float hi[3], const float qlo[3], const float qhi[3]) {
  float dd[3];
  for (int a = 0; a < 3; a++) {
    const float ext = qhi[a] - qlo[a];
    int e = -126;
    if (ext > 0.f) { (void)frexpf(ext / 255.0f, &e); if (e < -126) e = -126; }
    dd[a] = (hi[a] - lo[a]) + ldexpf(1.0f, e);
  }
  return (dd[0] * dd[1] + dd[0] * dd[2]) + dd[1] * dd[2];
}
__device__ __forceinline__ float dp_dist(const DpTables &T, uint32_t l, uint32_t r, uint32_t k, uint32_t d) {
  float best = INFINITY;
  for (uint32_t k1 = 1; k1 < k; k1++) best = fminf(best, dp_f(T, l, k1, d) + dp_f(T, r, k - k1, d));
  return best;
}
// bottom-up: one thread per leaf climbs; the second thread to reach a node computes it (its children are complete)
__global__ void collapse_dp_kernel(int n, const uint32_t *child, const uint32_t *parent_internal, const uint32_t *parent_leaf, uint32_t *visits,
                                   const unsigned long long *bx, DpTables T) {
  const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (k >= n) return;
  float lo[3], hi[3], qlo[3], qhi[3];
  box_load<false>(bx, (uint32_t)(n - 1 + k), lo, hi);  // (leaf boxes were written by an earlier launch)
  uint32_t anc = parent_leaf[k];
  for (uint32_t d = 1; d <= 3; d++) {
    box_load<false>(bx, anc, qlo, qhi);
    T.Fl[3 * (size_t)k + (d - 1u)] = kDpTriCost * area_on_grid(lo, hi, qlo, qhi);
    if (parent_internal[anc] != kNone) anc = parent_internal[anc];
  }
  uint32_t node = parent_leaf[k];
  while (node != kNone) {
    __threadfence();
    if (atomicAdd(&visits[node], 1u) == 0u) return;  // the sibling subtree is not done yet
    __threadfence();
    const uint32_t l = child[2 * (size_t)node], r = child[2 * (size_t)node + 1];
    const float g = dp_dist(T, l, r, 4u, 1u);
    T.G[node] = g;
    box_load<false>(bx, node, lo, hi);
    anc = node;
    for (uint32_t d = 1; d <= 3; d++) {
      if (parent_internal[anc] != kNone) anc = parent_internal[anc];
      box_load<false>(bx, anc, qlo, qhi);
      const float one = area_on_grid(lo, hi, qlo, qhi) + g;  // one step at the node when reached, plus what lies below
      T.F[(4 * (size_t)node + 0u) * 3 + (d - 1u)] = one;
      for (uint32_t kk = 2; kk <= 4; kk++) T.F[(4 * (size_t)node + (kk - 1u)) * 3 + (d - 1u)] = d < 3u ? fminf(one, dp_dist(T, l, r, kk, d + 1u)) : one;
    }
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
template <bool DP>
__global__ void collapse_kernel(const CollapseItem *items, uint32_t *level_count, int n, const uint32_t *child, const unsigned long long *bx,
                                uint4 *quads, CollapseItem *next, uint32_t *counters, DpTables T) {
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
  if (DP) {
    // follow the programme's minimising choices: node c with k slots at distance d is opened or presented as one child
    struct Open { uint32_t c, k, d; } st[8];
    int ns = 0;
    auto split = [&](uint32_t c, uint32_t k, uint32_t d) {  // the children of c share k slots at distance d; ties: the most even split
      const uint32_t l = child[2 * (size_t)c], r = child[2 * (size_t)c + 1];
      uint32_t bk = 1;
      float best = INFINITY;
      for (uint32_t k1 = 1; k1 < k; k1++) {
        const float v = dp_f(T, l, k1, d) + dp_f(T, r, k - k1, d);
        const int ev = abs((int)(2 * k1) - (int)k), eb = abs((int)(2 * bk) - (int)k);
        if (v < best || (v == best && ev < eb)) { best = v; bk = k1; }
      }
      st[ns++] = Open{r, k - bk, d};
      st[ns++] = Open{l, bk, d};
    };
    split(it.node, 4u, 1u);
    while (ns > 0) {
      const Open o = st[--ns];
      if (!(o.c & kLeafRef) && o.k >= 2u && o.d < 3u && dp_f(T, o.c, o.k, o.d) < dp_f(T, o.c, 1u, o.d)) split(o.c, o.k, o.d + 1u);
      else add(o.c);
    }
  } else {
  add(child[2 * it.node]);
  add(child[2 * it.node + 1]);
  }
  for (; !DP;) {
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
  // the interior children of a node are allocated side by side, as in the host's breadth-first array (one atomic for the
  // node: siblings share 128-byte lines, and a walk that has taken one child usually comes back for the next)
  uint32_t n_int = 0;
  for (int k = 0; k < nk; k++) n_int += (kids[k].c & kLeafRef) ? 0u : 1u;
  uint32_t quad = n_int ? atomicAdd(&counters[0], n_int) : 0u, slot = n_int ? atomicAdd(&level_count[1], n_int) : 0u;
  uint32_t ref[4];
  for (int k = 0; k < 4; k++) {
    if (k >= nk) { ref[k] = kEmptyLeafRef; continue; }
    const uint32_t c = kids[k].c;
    if (c & kLeafRef) {
      ref[k] = kLeafRef | (1u << 24) | (c & ~kLeafRef);  // one triangle, leaf slot = sorted position
    } else {
      ref[k] = quad * 64u;  // byte offset in the node array (capi.cpp make_quad_nodes)
      next[slot++] = CollapseItem{c, quad++, path};
    }
  }
  uint4 *q = quads + 4 * (size_t)it.quad;
  q[0] = make_uint4(__float_as_uint(me_lo[0]), __float_as_uint(me_lo[1]), __float_as_uint(me_lo[2]), ebyte[0] << 23);
  q[1] = make_uint4(qlo[0], qlo[1], qlo[2], qhi[0]);
  q[2] = make_uint4(qhi[1], qhi[2], ebyte[1] << 23, ebyte[2] << 23);
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
                           uint32_t quad_capacity, uint32_t flags, GpuBuildInfo *info, hipStream_t stream) {
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
  GB_TRY(rocprim::radix_sort_pairs(nullptr, sort_bytes, keys.as<uint32_t>(), keys_out.as<uint32_t>(), vals.as<uint32_t>(), d_order, (size_t)n, 0u, 30u, stream));
  GB_TRY(sort_tmp.alloc(sort_bytes));

  struct Events {  // (destroyed on every exit, early returns of GB_TRY included)
    hipEvent_t a = nullptr, b = nullptr;
    ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
  } ev;
  GB_TRY(hipEventCreate(&ev.a));
  GB_TRY(hipEventCreate(&ev.b));
  const hipEvent_t e0 = ev.a, e1 = ev.b;
  GB_TRY(hipEventRecord(e0, stream));
  const uint32_t init_bounds[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
  GB_TRY(hipMemcpyAsync(bounds.p, init_bounds, sizeof(init_bounds), hipMemcpyHostToDevice, stream));
  hipLaunchKernelGGL(centroid_bounds_kernel, dim3((n_tris + 2047u) / 2048u), block, 0, stream, d_P, d_idx, n_tris, bounds.as<uint32_t>());
  hipLaunchKernelGGL(morton_kernel, grid_t, block, 0, stream, d_P, d_idx, n_tris, bounds.as<uint32_t>(), keys.as<uint32_t>(), vals.as<uint32_t>());
  GB_TRY(hipGetLastError());
  GB_TRY(rocprim::radix_sort_pairs(sort_tmp.p, sort_bytes, keys.as<uint32_t>(), keys_out.as<uint32_t>(), vals.as<uint32_t>(), d_order, (size_t)n, 0u, 30u, stream));
  const uint32_t root_node = 0u;  // internal node the collapse starts from
  const bool sah_tree = true;
  {
    // ---- top-down binned SAH ----
    const size_t cap = (size_t)n / 2 + 2;  // segments of >= 2 triangles on one level
    Tmp tlo, thi, ids_b, seg_a, seg_b, cb, bins, split, active, cbase, flags, scan, scan_tmp, segtab;
    GB_TRY(tlo.alloc(12 * (size_t)n)); GB_TRY(thi.alloc(12 * (size_t)n));
    GB_TRY(ids_b.alloc(4 * (size_t)n)); GB_TRY(seg_a.alloc(4 * (size_t)n)); GB_TRY(seg_b.alloc(4 * (size_t)n));
    GB_TRY(cb.alloc(24 * cap)); GB_TRY(bins.alloc(4 * (size_t)kBinWords * cap)); GB_TRY(split.alloc(sizeof(SahSplit) * cap));
    GB_TRY(active.alloc(4 * cap)); GB_TRY(cbase.alloc(4 * cap)); GB_TRY(flags.alloc(4 * (size_t)n)); GB_TRY(scan.alloc(4 * (size_t)n));
    GB_TRY(segtab.alloc(4 * 6 * cap));
    size_t scan_bytes = 0;
    GB_TRY(rocprim::exclusive_scan(nullptr, scan_bytes, flags.as<uint32_t>(), scan.as<uint32_t>(), 0u, (size_t)n, rocprim::plus<uint32_t>(), stream));
    GB_TRY(scan_tmp.alloc(scan_bytes));
    SahSegs cur_s{segtab.as<uint32_t>(), segtab.as<uint32_t>() + cap, segtab.as<uint32_t>() + 2 * cap};
    SahSegs nxt_s{segtab.as<uint32_t>() + 3 * cap, segtab.as<uint32_t>() + 4 * cap, segtab.as<uint32_t>() + 5 * cap};
    hipLaunchKernelGGL(sah_tri_boxes_kernel, grid_t, block, 0, stream, d_P, d_idx, n_tris, tlo.as<float>(), thi.as<float>());
    // level 0: one segment [0, n) = node 0, triangles in Morton order (d_order holds the sorted ids)
    uint32_t *ids_in = d_order, *ids_out = ids_b.as<uint32_t>(), *seg_in = seg_a.as<uint32_t>(), *seg_out = seg_b.as<uint32_t>();
    GB_TRY(hipMemsetAsync(seg_in, 0, 4 * (size_t)n, stream));
    const uint32_t seg0[3] = {0u, (uint32_t)n, 0u};
    GB_TRY(hipMemcpyAsync(cur_s.start, &seg0[0], 4, hipMemcpyHostToDevice, stream));
    GB_TRY(hipMemcpyAsync(cur_s.end, &seg0[1], 4, hipMemcpyHostToDevice, stream));
    GB_TRY(hipMemcpyAsync(cur_s.node, &seg0[2], 4, hipMemcpyHostToDevice, stream));
    GB_TRY(hipStreamSynchronize(stream));  // (seg0 is a local)
    uint32_t n_seg = 1u, next_node = 1u;
    for (uint32_t level = 0; n_seg > 0u; level++) {
      if (level > 4096u) return hipErrorUnknown;  // (halving past kSahMedianLevel ends every segment long before)
      const dim3 gs((n_seg + 127u) / 128u), b128(128);
      // centre bounds: min words ~0, max words 0 (a strided memset over the second half of every 24-byte record)
      GB_TRY(hipMemsetAsync(cb.p, 0xff, 24 * (size_t)n_seg, stream));
      GB_TRY(hipMemset2DAsync((char *)cb.p + 12, 24, 0, 12, n_seg, stream));
      GB_TRY(hipMemsetAsync(bins.p, 0, 4 * (size_t)kBinWords * n_seg, stream));
      hipLaunchKernelGGL(sah_bounds_kernel, grid_t, block, 0, stream, ids_in, seg_in, n, tlo.as<float>(), thi.as<float>(), cb.as<uint32_t>());
      hipLaunchKernelGGL(sah_prepare_kernel, gs, b128, 0, stream, (int)n_seg, cb.as<uint32_t>(), split.as<SahSplit>());
      // bins: counts 0, box lo words ~0 (hi words 0 already)
      GB_TRY(hipMemset2DAsync((char *)bins.p + 4, 28, 0xff, 12, (size_t)n_seg * kSahBuckets, stream));
      hipLaunchKernelGGL(sah_bins_kernel, grid_t, block, 0, stream, ids_in, seg_in, n, tlo.as<float>(), thi.as<float>(), split.as<SahSplit>(), bins.as<uint32_t>());
      hipLaunchKernelGGL(sah_split_kernel, gs, b128, 0, stream, (int)n_seg, cur_s, bins.as<uint32_t>(), split.as<SahSplit>(), level, bx.as<unsigned long long>(),
                         active.as<uint32_t>());
      GB_TRY(rocprim::exclusive_scan(scan_tmp.p, scan_bytes, active.as<uint32_t>(), cbase.as<uint32_t>(), 0u, (size_t)n_seg, rocprim::plus<uint32_t>(), stream));
      hipLaunchKernelGGL(sah_flags_kernel, grid_t, block, 0, stream, ids_in, seg_in, n, tlo.as<float>(), thi.as<float>(), cur_s, split.as<SahSplit>(), flags.as<uint32_t>());
      GB_TRY(rocprim::exclusive_scan(scan_tmp.p, scan_bytes, flags.as<uint32_t>(), scan.as<uint32_t>(), 0u, (size_t)n, rocprim::plus<uint32_t>(), stream));
      hipLaunchKernelGGL(sah_scatter_kernel, grid_t, block, 0, stream, ids_in, seg_in, n, cur_s, split.as<SahSplit>(), flags.as<uint32_t>(), scan.as<uint32_t>(),
                         cbase.as<uint32_t>(), ids_out, seg_out);
      hipLaunchKernelGGL(sah_children_kernel, gs, b128, 0, stream, (int)n_seg, cur_s, split.as<SahSplit>(), cbase.as<uint32_t>(), next_node, child.as<uint32_t>(), nxt_s,
                         par_i.as<uint32_t>(), par_l.as<uint32_t>());
      GB_TRY(hipGetLastError());
      uint32_t last[2];  // exclusive sum and count of the last segment: the next level's segment count
      GB_TRY(hipMemcpyAsync(&last[0], cbase.as<uint32_t>() + (n_seg - 1u), 4, hipMemcpyDeviceToHost, stream));
      GB_TRY(hipMemcpyAsync(&last[1], active.as<uint32_t>() + (n_seg - 1u), 4, hipMemcpyDeviceToHost, stream));
      GB_TRY(hipStreamSynchronize(stream));
      n_seg = last[0] + last[1];
      next_node += n_seg;
      { uint32_t *t = ids_in; ids_in = ids_out; ids_out = t; }
      { uint32_t *t = seg_in; seg_in = seg_out; seg_out = t; }
      { SahSegs t = cur_s; cur_s = nxt_s; nxt_s = t; }
    }
    // leaf boxes and the final leaf order (slot k holds triangle ids[k])
    hipLaunchKernelGGL(sah_leaf_boxes_kernel, grid_t, block, 0, stream, ids_in, n, tlo.as<float>(), thi.as<float>(), bx.as<unsigned long long>(), ids_out);
    if (ids_out != d_order) GB_TRY(hipMemcpyAsync(d_order, ids_out, 4 * (size_t)n, hipMemcpyDeviceToDevice, stream));
    GB_TRY(hipGetLastError());
    GB_TRY(hipStreamSynchronize(stream));  // (the scratch above is freed when this block ends)
  }

  // ---- parallel re-insertion: the built tree optimised before it is collapsed (reinsert_core.hpp) ----
  // Passes of search / lock / check / apply / refit until a pass moves fewer than one node in 1024, `passes` at most, or the
  // searches have looked at more than kVisitBudget nodes per node of the tree in total (a mesh of coincident boxes, where
  // nothing prunes a search, costs a bounded amount).  The host reads two numbers per pass.
  info->reinsert_passes = 0;
  info->reinsert_moves = 0;
  info->reinsert_ms = 0.f;
  info->reinsert_cost_before = info->reinsert_cost_after = 0.;
  info->reinsert_undone = 0;
  {
    reins::StopRule stop;  // (shared with the host run of the pass: reinsert_core.hpp)
    int passes = stop.max_passes;
    uint32_t mu = 1;
    reins::Search sp;
    if (const char *v = debug_knob("PBRT_HIP_GPU_REINSERT")) passes = std::atoi(v);
    if (const char *v = debug_knob("PBRT_HIP_REINSERT_MU")) mu = (uint32_t)std::max(1, std::atoi(v));
    if (const char *v = debug_knob("PBRT_HIP_REINSERT_VISITS")) sp.max_visits = (uint32_t)std::max(1, std::atoi(v));
    if (const char *v = debug_knob("PBRT_HIP_REINSERT_MIN_REL")) sp.min_rel = (float)std::atof(v);
    if (const char *v = debug_knob("PBRT_HIP_REINSERT_QK")) sp.qk = (float)std::atof(v);
    if (const char *v = debug_knob("PBRT_HIP_REINSERT_QW")) sp.qw = (float)std::atof(v);
    const bool full_refit = debug_knob("PBRT_HIP_REINSERT_FULL_REFIT") != nullptr;  // (A-B: every box every pass, as until round 4)
    if (!(flags & kGpuBuildReinsert)) passes = 0;
    // Small trees are left as built: below StopRule::min_tris triangles (where the collapse is the greedy one, too) a walk is a handful
    // of steps in L1 and the surface-area objective decides nothing measurable -- BASELINE C4's 36 triangles rendered 2 % SLOWER with
    // the four moves the pass found (15 quad nodes instead of 19, profiles/r04p5_c4_ab.txt).  The test suite lowers the threshold
    // (tests/conftest.py) so that trees of 8 .. 300 triangles keep exercising the pass.
    uint32_t min_tris = stop.min_tris;
    if (const char *v = debug_knob("PBRT_HIP_REINSERT_MIN_TRIS")) min_tris = (uint32_t)std::max(8, std::atoi(v));
    if (sah_tree && n_tris >= min_tris && n >= 8 && passes > 0) {
      const uint32_t n_int = (uint32_t)n - 1u, n_nodes = 2u * n_int + 1u;
      const dim3 grid_n((n_nodes + 255u) / 256u), grid_i((n_int + 255u) / 256u);
      Tmp par, kid, mv_y, mv_lca, mv_gain, lock, holds, stats, cnt, newslot, order2, leaf_bx, par_b, kid_b, bx_b;
      GB_TRY(par.alloc(4 * (size_t)n_nodes)); GB_TRY(kid.alloc(8 * (size_t)n_int));
      GB_TRY(mv_y.alloc(4 * (size_t)n_nodes)); GB_TRY(mv_lca.alloc(4 * (size_t)n_nodes)); GB_TRY(mv_gain.alloc(4 * (size_t)n_nodes));
      GB_TRY(lock.alloc(8 * (size_t)n_nodes)); GB_TRY(holds.alloc(4 * (size_t)n_nodes)); GB_TRY(stats.alloc(5 * 8));
      GB_TRY(cnt.alloc(4 * (size_t)n_int)); GB_TRY(newslot.alloc(4 * (size_t)n)); GB_TRY(order2.alloc(4 * (size_t)n)); GB_TRY(leaf_bx.alloc(24 * (size_t)n));
      // the tree as it stood before the pass in flight (links + interior boxes): a pass that raises the summed area is undone
      GB_TRY(par_b.alloc(4 * (size_t)n_nodes)); GB_TRY(kid_b.alloc(8 * (size_t)n_int)); GB_TRY(bx_b.alloc(24 * (size_t)n_int));
      Events rev;
      GB_TRY(hipEventCreate(&rev.a));
      GB_TRY(hipEventCreate(&rev.b));
      GB_TRY(hipEventRecord(rev.a, stream));
      const reins::Tree t{n_int, par.as<uint32_t>(), kid.as<uint32_t>(), bx.as<unsigned long long>()};
      hipLaunchKernelGGL(ri_links_kernel, grid_i, block, 0, stream, n, child.as<uint32_t>(), par.as<uint32_t>(), kid.as<uint32_t>());
      // stats: [0] visits [1] found [2] applied (summed over the passes); [3] the tree's cost in fixed point, [4] its scale (ri_cost_kernel)
      GB_TRY(hipMemsetAsync(stats.p, 0, 5 * 8, stream));
      hipLaunchKernelGGL(ri_cost_kernel, grid_i, block, 0, stream, t, stats.as<unsigned long long>() + 3);
      unsigned long long h_stats[5] = {0, 0, 0, 0, 0}, applied_before = 0, cost = 0;
      GB_TRY(hipMemcpyAsync(h_stats, stats.p, sizeof(h_stats), hipMemcpyDeviceToHost, stream));
      GB_TRY(hipStreamSynchronize(stream));
      cost = h_stats[3];
      const double cost_unit = std::ldexp(1.0, -((int)h_stats[4] - 1024));
      info->reinsert_cost_before = info->reinsert_cost_after = (double)cost * cost_unit;
      int done = 0;
      for (int pass = 0; pass < passes; pass++) {
        GB_TRY(hipMemcpyAsync(par_b.p, par.p, 4 * (size_t)n_nodes, hipMemcpyDeviceToDevice, stream));
        GB_TRY(hipMemcpyAsync(kid_b.p, kid.p, 8 * (size_t)n_int, hipMemcpyDeviceToDevice, stream));
        GB_TRY(hipMemcpyAsync(bx_b.p, bx.p, 24 * (size_t)n_int, hipMemcpyDeviceToDevice, stream));
        hipLaunchKernelGGL(ri_search_kernel, grid_n, block, 0, stream, t, sp, (uint32_t)pass, mu, mv_y.as<uint32_t>(), mv_lca.as<uint32_t>(), mv_gain.as<float>(),
                           stats.as<unsigned long long>());
        GB_TRY(hipMemsetAsync(lock.p, 0, 8 * (size_t)n_nodes, stream));
        hipLaunchKernelGGL(ri_lock_kernel, grid_n, block, 0, stream, t, mv_y.as<uint32_t>(), mv_gain.as<float>(), lock.as<unsigned long long>());
        hipLaunchKernelGGL(ri_hold_kernel, grid_n, block, 0, stream, t, mv_y.as<uint32_t>(), mv_gain.as<float>(), lock.as<unsigned long long>(), holds.as<uint32_t>());
        hipLaunchKernelGGL(ri_free_kernel, grid_n, block, 0, stream, t, mv_y.as<uint32_t>(), mv_lca.as<uint32_t>(), mv_gain.as<float>(), holds.as<uint32_t>());
        // (mv_lca has served: the apply kernel leaves there the grandparent every moved node came from)
        hipLaunchKernelGGL(ri_apply_kernel, grid_n, block, 0, stream, t, mv_y.as<uint32_t>(), mv_lca.as<uint32_t>(), stats.as<unsigned long long>());
        GB_TRY(hipMemsetAsync(visits.p, 0, 4 * (size_t)n, stream));
        if (full_refit) {
          hipLaunchKernelGGL(ri_refit_kernel<false>, grid_t, block, 0, stream, t, visits.as<uint32_t>(), cnt.as<uint32_t>());
        } else {
          hipLaunchKernelGGL(ri_mark_kernel, grid_n, block, 0, stream, t, mv_y.as<uint32_t>(), mv_lca.as<uint32_t>(), visits.as<uint32_t>());
          hipLaunchKernelGGL(ri_refit_dirty_kernel, grid_i, block, 0, stream, t, visits.as<uint32_t>());
        }
        GB_TRY(hipMemsetAsync(stats.as<unsigned long long>() + 3, 0, 8, stream));
        hipLaunchKernelGGL(ri_cost_kernel, grid_i, block, 0, stream, t, stats.as<unsigned long long>() + 3);
        GB_TRY(hipGetLastError());
        GB_TRY(hipMemcpyAsync(h_stats, stats.p, sizeof(h_stats), hipMemcpyDeviceToHost, stream));
        GB_TRY(hipStreamSynchronize(stream));
        const unsigned long long applied = h_stats[2] - applied_before;
        if (h_stats[3] > cost) {
          // The moves of a pass hold six link nodes each, not their paths, so their gains need not add up: one pass in a
          // thousand or so comes out with MORE summed area than it went in with.  Such a pass is undone and ends the passes.
          GB_TRY(hipMemcpyAsync(par.p, par_b.p, 4 * (size_t)n_nodes, hipMemcpyDeviceToDevice, stream));
          GB_TRY(hipMemcpyAsync(kid.p, kid_b.p, 8 * (size_t)n_int, hipMemcpyDeviceToDevice, stream));
          GB_TRY(hipMemcpyAsync(bx.p, bx_b.p, 24 * (size_t)n_int, hipMemcpyDeviceToDevice, stream));
          h_stats[2] = applied_before;
          info->reinsert_undone = 1;
          break;
        }
        done = pass + 1;
        cost = h_stats[3];
        applied_before = h_stats[2];
        if (reins::stop_after_pass(stop, applied, h_stats[0], n_nodes)) break;
      }
      info->reinsert_cost_after = (double)cost * cost_unit;
      // child order, leaf counts, leaves renumbered in depth-first order, and back to the builder's arrays
      hipLaunchKernelGGL(ri_order_kernel, grid_i, block, 0, stream, t);
      GB_TRY(hipMemsetAsync(visits.p, 0, 4 * (size_t)n, stream));
      // (the leaves below every node; the boxes are right already -- the last pass refitted what it moved)
      hipLaunchKernelGGL((ri_refit_kernel<true, false>), grid_t, block, 0, stream, t, visits.as<uint32_t>(), cnt.as<uint32_t>());
      hipLaunchKernelGGL(ri_leafpos_kernel, grid_t, block, 0, stream, t, cnt.as<uint32_t>(), newslot.as<uint32_t>());
      hipLaunchKernelGGL(ri_emit_kernel, grid_i, block, 0, stream, t, newslot.as<uint32_t>(), child.as<uint32_t>(), par_i.as<uint32_t>(), par_l.as<uint32_t>());
      hipLaunchKernelGGL(ri_leaves_kernel, grid_t, block, 0, stream, t, newslot.as<uint32_t>(), d_order, order2.as<uint32_t>(), leaf_bx.as<unsigned long long>());
      GB_TRY(hipGetLastError());
      GB_TRY(hipMemcpyAsync(d_order, order2.p, 4 * (size_t)n, hipMemcpyDeviceToDevice, stream));
      GB_TRY(hipMemcpyAsync(bx.as<unsigned long long>() + 3 * (size_t)n_int, leaf_bx.p, 24 * (size_t)n, hipMemcpyDeviceToDevice, stream));
      GB_TRY(hipEventRecord(rev.b, stream));
      GB_TRY(hipStreamSynchronize(stream));  // (the scratch above is freed when this block ends)
      float rms = 0.f;
      GB_TRY(hipEventElapsedTime(&rms, rev.a, rev.b));
      info->reinsert_passes = (uint32_t)done;
      info->reinsert_moves = (uint32_t)std::min<unsigned long long>(h_stats[2], 0xffffffffull);
      info->reinsert_ms = rms;
    }
  }

  // top-down collapse, one launch per level of the quad tree.  The host does not know how many items a level holds
  // (the kernel reads the count the level above left on the device) nor how deep the tree is: it launches kBatch
  // levels blind -- level L holds at most min(4^L, n) items -- and looks at the device once per batch.
  // Which descendants become a quad node's children: for trees of kDpMinTris triangles and more the dynamic programme of
  // capi.cpp (the host's default there too), else -- and for the LBVH / PLOC trees -- the greedy largest-area rule.
  constexpr uint32_t kDpMinTris = 1024;
  const char *cm = debug_knob("PBRT_HIP_COLLAPSE");
  const bool use_dp = sah_tree && n >= 2 && ((n_tris >= kDpMinTris && !(cm && std::string(cm) == "greedy")) || (cm && std::string(cm) == "dp"));
  Tmp dpF, dpFl, dpG;
  DpTables dp{nullptr, nullptr, nullptr};
  if (use_dp) {
    GB_TRY(dpF.alloc(48 * (size_t)n));
    GB_TRY(dpFl.alloc(12 * (size_t)n));
    GB_TRY(dpG.alloc(4 * (size_t)n));
    dp = DpTables{dpF.as<float>(), dpFl.as<float>(), dpG.as<float>()};
    GB_TRY(hipMemsetAsync(visits.p, 0, 4 * (size_t)n, stream));
    hipLaunchKernelGGL(collapse_dp_kernel, grid_t, block, 0, stream, n, child.as<uint32_t>(), par_i.as<uint32_t>(), par_l.as<uint32_t>(), visits.as<uint32_t>(),
                       bx.as<unsigned long long>(), dp);
    GB_TRY(hipGetLastError());
  }
  constexpr uint32_t kBatch = 48;
  const CollapseItem root{root_node, 0u, 0u};
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
      if (use_dp)
        hipLaunchKernelGGL(collapse_kernel<true>, dim3((cap + 127u) / 128u), dim3(128), 0, stream, cur, level_counts.as<uint32_t>() + l, n,
                           child.as<uint32_t>(), bx.as<unsigned long long>(), d_quads, nxt, counters.as<uint32_t>(), dp);
      else
        hipLaunchKernelGGL(collapse_kernel<false>, dim3((cap + 127u) / 128u), dim3(128), 0, stream, cur, level_counts.as<uint32_t>() + l, n,
                           child.as<uint32_t>(), bx.as<unsigned long long>(), d_quads, nxt, counters.as<uint32_t>(), dp);
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
  float root_box[6];  // {lo.x lo.y lo.z hi.x hi.y hi.z} of the root
  GB_TRY(hipMemcpyAsync(root_box, bx.as<unsigned long long>() + 3 * (size_t)root_node, 24, hipMemcpyDeviceToHost, stream));
  GB_TRY(hipEventRecord(e1, stream));
  GB_TRY(hipStreamSynchronize(stream));
  float ms = 0.f;
  GB_TRY(hipEventElapsedTime(&ms, e0, e1));
  info->n_quads = h_counters[0];
  info->stack_need = h_counters[1];
  info->levels = levels;
  for (int a = 0; a < 3; a++) { info->root_lo[a] = root_box[a]; info->root_hi[a] = root_box[3 + a]; }
  info->build_ms = ms;
  return hipSuccess;
}

}  // namespace pbrt_hip
