// bvh_build.cpp -- see bvh_build.hpp.  Works on structure-of-arrays triangle bounds and a
// permutation of triangle ids; nodes are appended in pre-order so no pointer tree is ever built.
#include "bvh_build.hpp"

#include <algorithm>
#include <limits>

namespace pbrt_hip {
namespace {

constexpr int kBuckets = 16;
constexpr uint32_t kMaxLeaf = 4;
constexpr uint32_t kDepthGuard = 32;  // beyond this level: median splits only

struct Box {
  float lo[3], hi[3];
  void reset() {
    for (int a = 0; a < 3; a++) { lo[a] = std::numeric_limits<float>::infinity(); hi[a] = -lo[a]; }
  }
  void grow(const float *l, const float *h) {
    for (int a = 0; a < 3; a++) {
      if (l[a] < lo[a]) lo[a] = l[a];
      if (h[a] > hi[a]) hi[a] = h[a];
    }
  }
  void grow(const Box &b) { grow(b.lo, b.hi); }
  float area() const {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return 2.0f * ((dx * dy + dx * dz) + dy * dz);
  }
};

struct Builder {
  // per triangle id
  std::vector<float> lo, hi, cen;  // 3 floats each
  std::vector<uint32_t> ids, scratch;
  Bvh *out;

  uint32_t emit(uint32_t start, uint32_t end, uint32_t level) {
    const uint32_t me = (uint32_t)out->nodes.size();
    out->nodes.emplace_back();
    if (level + 1 > out->depth) out->depth = level + 1;
    const uint32_t n = end - start;

    Box all;
    all.reset();
    for (uint32_t i = start; i < end; i++) all.grow(&lo[3 * ids[i]], &hi[3 * ids[i]]);

    auto leaf = [&]() {
      BvhNode nd;
      for (int a = 0; a < 3; a++) { nd.lo[a] = all.lo[a]; nd.hi[a] = all.hi[a]; }
      nd.offset = (uint32_t)out->order.size();
      nd.count_axis = n;
      for (uint32_t i = start; i < end; i++) out->order.push_back(ids[i]);
      out->nodes[me] = nd;
      return me;
    };
    if (n == 1) return leaf();

    float cmin[3], cmax[3];
    for (int a = 0; a < 3; a++) { cmin[a] = std::numeric_limits<float>::infinity(); cmax[a] = -cmin[a]; }
    for (uint32_t i = start; i < end; i++) {
      const float *c = &cen[3 * ids[i]];
      for (int a = 0; a < 3; a++) {
        if (c[a] < cmin[a]) cmin[a] = c[a];
        if (c[a] > cmax[a]) cmax[a] = c[a];
      }
    }
    const float ex = cmax[0] - cmin[0], ey = cmax[1] - cmin[1], ez = cmax[2] - cmin[2];
    const int axis = (ex > ey && ex > ez) ? 0 : (ey > ez ? 1 : 2);
    const float c0 = cmin[axis], c1 = cmax[axis];
    uint32_t mid = (start + end) / 2;

    if (c1 == c0) {
      if (n <= 64) return leaf();
      // all centroids coincide on the widest axis: halve the range as it stands
    } else if (level >= kDepthGuard) {
      std::stable_sort(ids.begin() + start, ids.begin() + end,
                       [&](uint32_t a, uint32_t b) { return cen[3 * a + axis] < cen[3 * b + axis]; });
    } else if (n == 2) {
      if (cen[3 * ids[start + 1] + axis] < cen[3 * ids[start] + axis]) std::swap(ids[start], ids[start + 1]);
    } else {
      const float span = c1 - c0;
      auto bucket = [&](uint32_t id) {
        int b = (int)((float)kBuckets * ((cen[3 * id + axis] - c0) / span));
        return b == kBuckets ? kBuckets - 1 : b;
      };
      int count[kBuckets] = {0};
      Box bb[kBuckets];
      for (auto &b : bb) b.reset();
      for (uint32_t i = start; i < end; i++) {
        int b = bucket(ids[i]);
        count[b]++;
        bb[b].grow(&lo[3 * ids[i]], &hi[3 * ids[i]]);
      }
      // prefix / suffix sweeps: bounds and counts left of and right of each of the 15 planes
      float area_l[kBuckets - 1], area_r[kBuckets - 1];
      int cnt_l[kBuckets - 1], cnt_r[kBuckets - 1];
      Box acc;
      acc.reset();
      int c = 0;
      for (int i = 0; i < kBuckets - 1; i++) {
        if (count[i]) { acc.grow(bb[i]); c += count[i]; }
        cnt_l[i] = c;
        area_l[i] = c ? acc.area() : 0.f;
      }
      acc.reset();
      c = 0;
      for (int i = kBuckets - 1; i >= 1; i--) {
        if (count[i]) { acc.grow(bb[i]); c += count[i]; }
        cnt_r[i - 1] = c;
        area_r[i - 1] = c ? acc.area() : 0.f;
      }
      const float total_area = all.area();
      int best = 0;
      float best_cost = 0.f;
      for (int i = 0; i < kBuckets - 1; i++) {
        float cost = 1.0f + ((float)cnt_l[i] * area_l[i] + (float)cnt_r[i] * area_r[i]) / total_area;
        if (i == 0 || cost < best_cost) { best_cost = cost; best = i; }
      }
      if (n > kMaxLeaf || best_cost < (float)n) {
        // stable partition through the scratch array: left keeps its order, then right
        uint32_t nl = 0, nr = 0;
        for (uint32_t i = start; i < end; i++) {
          if (bucket(ids[i]) <= best) ids[start + nl++] = ids[i];
          else scratch[nr++] = ids[i];
        }
        for (uint32_t i = 0; i < nr; i++) ids[start + nl + i] = scratch[i];
        mid = start + nl;
      } else {
        return leaf();
      }
    }

    emit(start, mid, level + 1);
    const uint32_t second = emit(mid, end, level + 1);
    BvhNode nd;
    for (int a = 0; a < 3; a++) { nd.lo[a] = all.lo[a]; nd.hi[a] = all.hi[a]; }
    nd.offset = second;
    nd.count_axis = (uint32_t)axis << 16;
    out->nodes[me] = nd;
    return me;
  }
};

}  // namespace

void build_bvh(const float *P, const uint32_t *idx, uint32_t n_tris, Bvh *out) {
  out->nodes.clear();
  out->order.clear();
  out->depth = 0;
  if (n_tris == 0) return;
  Builder b;
  b.out = out;
  b.lo.resize(3 * (size_t)n_tris);
  b.hi.resize(3 * (size_t)n_tris);
  b.cen.resize(3 * (size_t)n_tris);
  b.ids.resize(n_tris);
  b.scratch.resize(n_tris);
  for (uint32_t t = 0; t < n_tris; t++) {
    b.ids[t] = t;
    for (int a = 0; a < 3; a++) {
      float v0 = P[3 * idx[3 * t] + a], v1 = P[3 * idx[3 * t + 1] + a], v2 = P[3 * idx[3 * t + 2] + a];
      float mn = v0 < v1 ? v0 : v1;
      mn = mn < v2 ? mn : v2;
      float mx = v0 > v1 ? v0 : v1;
      mx = mx > v2 ? mx : v2;
      b.lo[3 * t + a] = mn;
      b.hi[3 * t + a] = mx;
      b.cen[3 * t + a] = mn * 0.5f + mx * 0.5f;
    }
  }
  out->nodes.reserve(2 * (size_t)n_tris);
  out->order.reserve(n_tris);
  b.emit(0, n_tris, 0);
}

}  // namespace pbrt_hip
