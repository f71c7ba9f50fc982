// quad_walk.cpp -- CPU restatement of the PRODUCTION walk of the render kernel (pbrt_amd/csrc/kernels.hip trav_run,
// production instantiation) over the quantised 4-wide nodes of DESIGN.md section 4, one ray at a time.
//
// TEST INFRASTRUCTURE ONLY (see pbrt_oracle.h): tests/ and tools/ use it to check, without a GPU, that a tree a product
// builder emitted (pbrt_hip_quad_build_host*) finds the hits of the oracle's own BVH / of brute force -- the tie rule of
// DESIGN.md 3.4 makes a hit independent of the tree, so any difference is a builder bug (a clipped or quantised box that
// does not enclose what lies below it) -- and to count node steps / triangle tests per ray of a tree before it is ever
// uploaded.  The arithmetic of one step follows the kernel instruction for instruction: node-relative planes
// t = fma(q, cell * inv, -(g +- 3 eps |g|)), the (1 + 2^-19) pad, NaN-ignoring min / max, nearest hit child first,
// the other hit children stacked in slot order.  A lane of the kernel parks at a leaf until its triangles are tested, so
// the per-ray sequence of steps is the sequential one written here.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "pbrt_oracle.h"

namespace {

constexpr float kInf = __builtin_huge_valf();
constexpr float kRayTMin = 1e-4f;
constexpr float kBoxPad = 0x1.000004p+0f;  // 1 + 2^-19
constexpr float kOwnPad = 0x1.000001p+0f;  // 1 + 2^-21: the own-box rule of DESIGN.md 3.5
constexpr uint32_t kLeafRef = 0x80000000u, kDone = 0xffffffffu, kNoPrim = 0xffffffffu;

inline float fminn(float a, float b) { return std::fmin(a, b); }  // IEEE minNum / maxNum: a NaN operand is ignored
inline float fmaxn(float a, float b) { return std::fmax(a, b); }
inline float as_f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

struct V3 { float x, y, z; };
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {(a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x)}; }

struct Tree {
  const uint32_t *quads;
  uint32_t n_quads;
  uint32_t root_ref;
  const float *root_box;  // lo xyz, hi xyz
  const float *P;
  const uint32_t *idx, *order;
  const float *exact;  // experiment: unquantised child boxes (24 floats per node), or null
};

struct Out {
  float t = kInf, b1 = 0.f, b2 = 0.f;
  uint32_t prim = kNoPrim;
  bool occluded = false;
  uint64_t steps = 0, tris = 0;
  uint32_t max_stack = 0;
};

bool box_test(const float *b, V3 o, V3 inv, float tfar) {
  const bool nx = inv.x < 0.f, ny = inv.y < 0.f, nz = inv.z < 0.f;
  const float tnx = ((nx ? b[3] : b[0]) - o.x) * inv.x, tfx = ((nx ? b[0] : b[3]) - o.x) * inv.x;
  const float tny = ((ny ? b[4] : b[1]) - o.y) * inv.y, tfy = ((ny ? b[1] : b[4]) - o.y) * inv.y;
  const float tnz = ((nz ? b[5] : b[2]) - o.z) * inv.z, tfz = ((nz ? b[2] : b[5]) - o.z) * inv.z;
  const float tn = fmaxn(fmaxn(tnx, tny), fmaxn(tnz, kRayTMin));
  const float tf = fminn(fminn(tfx, tfy), fminn(tfz, tfar));
  return tn <= tf * kBoxPad;
}

// measurement aid (tools/treetop_sim.py): when set, every node step adds one to its node's counter
uint64_t *g_visits = nullptr;

void walk(const Tree &T, V3 o, V3 d, float tmax, bool any, Out *out) {
  Out r;
  uint32_t cur = kDone;
  const V3 inv1 = {1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
  const bool negx = inv1.x < 0.f, negy = inv1.y < 0.f, negz = inv1.z < 0.f;
  if (T.root_ref != kDone) {
    const float *b = T.root_box;
    const bool inside = o.x >= b[0] && o.x <= b[3] && o.y >= b[1] && o.y <= b[4] && o.z >= b[2] && o.z <= b[5];
    if (inside || box_test(b, o, inv1, tmax)) cur = (T.root_ref & kLeafRef) ? T.root_ref : 0u;
  }
  // the kernel's stand-in for 1 / 0 (kernels.hip trav_run; pbrt_amd/csrc/host_math.hpp inv_parallel_for_extent, restated)
  float big;
  {
    const float *b = T.root_box;
    const float extent = std::fmax(b[3] - b[0], std::fmax(b[4] - b[1], b[5] - b[2]));
    int x = 0;
    if (extent > 0.f && std::isfinite(extent)) (void)std::frexp(2.0f * extent, &x);
    int e = 123 - x;
    if (e > 120) e = 120;
    if (e < -100) e = -100;
    big = std::ldexp(1.0f, e);
  }
  const V3 inv = {d.x == 0.f ? std::copysign(big, inv1.x) : inv1.x, d.y == 0.f ? std::copysign(big, inv1.y) : inv1.y,
                  d.z == 0.f ? std::copysign(big, inv1.z) : inv1.z};
  std::vector<uint32_t> stack;
  stack.reserve(64);
  // experiment (ORC_WALK_CULL=n): every stacked entry carries its entry distance rounded DOWN to its n top bits (32: exact);
  // a popped entry farther than the closest hit so far is dropped without a fetch
  static const int cull_bits = std::getenv("ORC_WALK_CULL") ? std::atoi(std::getenv("ORC_WALK_CULL")) : 0;
  std::vector<float> skey;
  auto pop = [&]() -> uint32_t {
    while (!stack.empty()) {
      const uint32_t ref = stack.back();
      stack.pop_back();
      if (cull_bits) {
        const float k = skey.back();
        skey.pop_back();
        if (k > fminn(r.t, tmax)) continue;
      }
      return ref;
    }
    return kDone;
  };
  auto push = [&](uint32_t ref, float k) {
    stack.push_back(ref);
    if (cull_bits) {
      uint32_t u; std::memcpy(&u, &k, 4);
      if (cull_bits < 32) u &= ~((1u << (32 - cull_bits)) - 1u);  // k >= kRayTMin > 0: clearing low bits rounds down
      skey.push_back(as_f(u));
    }
  };
  while (cur != kDone) {
    if (!(cur & kLeafRef)) {
      const uint32_t *W = T.quads + (size_t)(cur / 64u) * 16u;
      r.steps++;
      if (g_visits) __atomic_fetch_add(&g_visits[cur / 64u], 1ull, __ATOMIC_RELAXED);
      const float tfar = fminn(r.t, tmax);
      const float gx = (o.x - as_f(W[0])) * inv.x, gy = (o.y - as_f(W[1])) * inv.y, gz = (o.z - as_f(W[2])) * inv.z;
      constexpr float kMargin = 0x1.8p-22f;
      const float gxn = std::fmaf(std::fabs(gx), kMargin, gx), gxf = std::fmaf(-std::fabs(gx), kMargin, gx);
      const float gyn = std::fmaf(std::fabs(gy), kMargin, gy), gyf = std::fmaf(-std::fabs(gy), kMargin, gy);
      const float gzn = std::fmaf(std::fabs(gz), kMargin, gz), gzf = std::fmaf(-std::fabs(gz), kMargin, gz);
      const float cix = as_f(W[3]) * inv.x, ciy = as_f(W[10]) * inv.y, ciz = as_f(W[11]) * inv.z;
      const uint32_t bnx = negx ? W[7] : W[4], bfx = negx ? W[4] : W[7];
      const uint32_t bny = negy ? W[8] : W[5], bfy = negy ? W[5] : W[8];
      const uint32_t bnz = negz ? W[9] : W[6], bfz = negz ? W[6] : W[9];
      float key[4];
      bool hit[4];
      for (int k = 0; k < 4; k++) {
        const float txn = std::fmaf((float)((bnx >> (8 * k)) & 0xffu), cix, -gxn), txf = std::fmaf((float)((bfx >> (8 * k)) & 0xffu), cix, -gxf);
        const float tyn = std::fmaf((float)((bny >> (8 * k)) & 0xffu), ciy, -gyn), tyf = std::fmaf((float)((bfy >> (8 * k)) & 0xffu), ciy, -gyf);
        const float tzn = std::fmaf((float)((bnz >> (8 * k)) & 0xffu), ciz, -gzn), tzf = std::fmaf((float)((bfz >> (8 * k)) & 0xffu), ciz, -gzf);
        const float tn = fmaxn(fmaxn(txn, tyn), fmaxn(tzn, kRayTMin));
        const float tf = fminn(fminn(txf, tyf), fminn(tzf, tfar));
        hit[k] = tn <= tf * kBoxPad;
        key[k] = tn;
      }
      if (T.exact) {  // experiment: what the walk would cost with full-precision child boxes
        const float *E = T.exact + (size_t)(cur / 64u) * 24u;
        for (int k = 0; k < 4; k++) {
          const float *b = E + 6 * k;
          const float tnx = ((negx ? b[3] : b[0]) - o.x) * inv.x, tfx = ((negx ? b[0] : b[3]) - o.x) * inv.x;
          const float tny = ((negy ? b[4] : b[1]) - o.y) * inv.y, tfy = ((negy ? b[1] : b[4]) - o.y) * inv.y;
          const float tnz = ((negz ? b[5] : b[2]) - o.z) * inv.z, tfz = ((negz ? b[2] : b[5]) - o.z) * inv.z;
          const float tn = fmaxn(fmaxn(tnx, tny), fmaxn(tnz, kRayTMin));
          const float tf = fminn(fminn(tfx, tfy), fminn(tfz, tfar));
          hit[k] = b[0] <= b[3] && tn <= tf * kBoxPad;
          key[k] = tn;
        }
      }
      float kmin = kInf;
      bool any_hit = false;
      for (int k = 0; k < 4; k++)
        if (hit[k]) { kmin = any_hit ? fminn(kmin, key[k]) : key[k]; any_hit = true; }
      int nearest = -1;
      for (int k = 0; k < 4 && nearest < 0; k++)
        if (hit[k] && key[k] == kmin) nearest = k;
      if (any_hit && nearest < 0) nearest = 3;  // (the kernel's n3 = none of the first three)
      static const int order_mode = std::getenv("ORC_WALK_ORDER") ? std::atoi(std::getenv("ORC_WALK_ORDER")) : 0;
      if (order_mode == 1) {  // experiment: the other hit children stacked by entry distance, the farthest deepest
        int ks[4], m = 0;
        for (int k = 0; k < 4; k++)
          if (hit[k] && k != nearest) ks[m++] = k;
        for (int i = 0; i < m; i++)
          for (int j = i + 1; j < m; j++)
            if (key[ks[j]] > key[ks[i]]) { const int t2 = ks[i]; ks[i] = ks[j]; ks[j] = t2; }
        for (int i = 0; i < m; i++) push(W[12 + ks[i]], key[ks[i]]);
      } else if (order_mode == 2) {  // experiment: slot order, reversed when the ray runs against the axis along which the
        // children's centres are spread most (what a builder that sorts children along that axis + one sign test would do)
        float c[4][3];
        int used = 0;
        for (int k = 0; k < 4; k++) {
          if (W[12 + k] == 0x80000000u) continue;
          used++;
          c[k][0] = (float)((W[4] >> (8 * k)) & 0xff) + (float)((W[7] >> (8 * k)) & 0xff);
          c[k][1] = (float)((W[5] >> (8 * k)) & 0xff) + (float)((W[8] >> (8 * k)) & 0xff);
          c[k][2] = (float)((W[6] >> (8 * k)) & 0xff) + (float)((W[9] >> (8 * k)) & 0xff);
        }
        const float cell[3] = {as_f(W[3]), as_f(W[10]), as_f(W[11])};
        int ax = 0;
        float best = -1.f;
        for (int a = 0; a < 3; a++) {
          float lo = kInf, hi = -kInf;
          for (int k = 0; k < 4; k++)
            if (W[12 + k] != 0x80000000u) { lo = fminn(lo, c[k][a]); hi = fmaxn(hi, c[k][a]); }
          const float spread = (hi - lo) * cell[a];
          if (spread > best) { best = spread; ax = a; }
        }
        const bool neg = ax == 0 ? negx : (ax == 1 ? negy : negz);
        int ks[4], m = 0;
        for (int k = 0; k < 4; k++)
          if (hit[k] && k != nearest) ks[m++] = k;
        // far side deepest: ascending centre for a negative ray (it meets high centres first), descending for a positive one
        for (int i = 0; i < m; i++)
          for (int j = i + 1; j < m; j++) {
            const bool swap = neg ? c[ks[j]][ax] < c[ks[i]][ax] : c[ks[j]][ax] > c[ks[i]][ax];
            if (swap) { const int t2 = ks[i]; ks[i] = ks[j]; ks[j] = t2; }
          }
        for (int i = 0; i < m; i++) push(W[12 + ks[i]], key[ks[i]]);
        (void)used;
      } else
      for (int k = 3; k >= 0; k--)
        if (hit[k] && k != nearest) push(W[12 + k], key[k]);
      if (stack.size() > r.max_stack) r.max_stack = (uint32_t)stack.size();
      cur = any_hit ? W[12 + nearest] : pop();
      continue;
    }
    // a leaf: its triangles in slot order (Moeller-Trumbore in the operation order of DESIGN.md 3.5)
    const uint32_t cnt = (cur >> 24) & 0x7fu, first = cur & 0xffffffu;
    bool stop = false;
    for (uint32_t i = 0; i < cnt && !stop; i++) {
      const uint32_t slot = first + i, id = T.order[slot];
      const float *a = T.P + 3 * (size_t)T.idx[3 * (size_t)id], *b = T.P + 3 * (size_t)T.idx[3 * (size_t)id + 1], *c = T.P + 3 * (size_t)T.idx[3 * (size_t)id + 2];
      r.tris++;
      const V3 p0 = {a[0], a[1], a[2]};
      const V3 e1 = sub({b[0], b[1], b[2]}, p0), e2 = sub({c[0], c[1], c[2]}, p0);
      const V3 pv = cross(d, e2);
      const float det = dot(e1, pv);
      const float idet = 1.0f / det;
      const V3 tv = sub(o, p0);
      const float u = dot(tv, pv) * idet;
      const V3 qv = cross(tv, e1);
      const float v = dot(d, qv) * idet;
      const float th = dot(e2, qv) * idet;
      // the own-box rule, with the TRUE 1 / d as in the kernel: slab distances of the three vertices, (p0 - o) = -tv
      const V3 w1 = sub({b[0], b[1], b[2]}, o), w2 = sub({c[0], c[1], c[2]}, o);
      const float x0 = (-tv.x) * inv1.x, x1 = w1.x * inv1.x, x2 = w2.x * inv1.x;
      const float y0 = (-tv.y) * inv1.y, y1 = w1.y * inv1.y, y2 = w2.y * inv1.y;
      const float z0 = (-tv.z) * inv1.z, z1 = w1.z * inv1.z, z2 = w2.z * inv1.z;
      const float otn = fmaxn(fmaxn(fminn(fminn(x0, x1), x2), fminn(fminn(y0, y1), y2)), fmaxn(fminn(fminn(z0, z1), z2), kRayTMin));
      const float otf = fminn(fminn(fmaxn(fmaxn(x0, x1), x2), fmaxn(fmaxn(y0, y1), y2)), fmaxn(fmaxn(z0, z1), z2));
      const bool in_own_box = otn <= otf * kOwnPad;   // the ray meets the triangle's own box ...
      const float ht = fmaxn(th, otn);                  // ... and the hit is not before the box's entry
      const bool valid = in_own_box && !(std::fabs(det) < 1e-8f) && (u >= 0.f) && (v >= 0.f) && (u + v <= 1.0f) && (th > kRayTMin) && (ht < tmax);
      if (valid && any) { r.occluded = true; stop = true; }
      if (valid && !any && (ht < r.t || (ht == r.t && id < r.prim))) { r.t = ht; r.prim = id; r.b1 = u; r.b2 = v; }
    }
    cur = stop ? kDone : pop();
  }
  *out = r;
}

}  // namespace

// The production node step's test of child k of quad node W against [kRayTMin, tfar] -- the arithmetic of walk() above, for one child
static bool child_passes(const uint32_t *W, int k, V3 o, V3 inv, bool negx, bool negy, bool negz, float tfar) {
  const float gx = (o.x - as_f(W[0])) * inv.x, gy = (o.y - as_f(W[1])) * inv.y, gz = (o.z - as_f(W[2])) * inv.z;
  constexpr float kMargin = 0x1.8p-22f;
  const float gxn = std::fmaf(std::fabs(gx), kMargin, gx), gxf = std::fmaf(-std::fabs(gx), kMargin, gx);
  const float gyn = std::fmaf(std::fabs(gy), kMargin, gy), gyf = std::fmaf(-std::fabs(gy), kMargin, gy);
  const float gzn = std::fmaf(std::fabs(gz), kMargin, gz), gzf = std::fmaf(-std::fabs(gz), kMargin, gz);
  const float cix = as_f(W[3]) * inv.x, ciy = as_f(W[10]) * inv.y, ciz = as_f(W[11]) * inv.z;
  const uint32_t bnx = negx ? W[7] : W[4], bfx = negx ? W[4] : W[7];
  const uint32_t bny = negy ? W[8] : W[5], bfy = negy ? W[5] : W[8];
  const uint32_t bnz = negz ? W[9] : W[6], bfz = negz ? W[6] : W[9];
  const float txn = std::fmaf((float)((bnx >> (8 * k)) & 0xffu), cix, -gxn), txf = std::fmaf((float)((bfx >> (8 * k)) & 0xffu), cix, -gxf);
  const float tyn = std::fmaf((float)((bny >> (8 * k)) & 0xffu), ciy, -gyn), tyf = std::fmaf((float)((bfy >> (8 * k)) & 0xffu), ciy, -gyf);
  const float tzn = std::fmaf((float)((bnz >> (8 * k)) & 0xffu), ciz, -gzn), tzf = std::fmaf((float)((bfz >> (8 * k)) & 0xffu), ciz, -gzf);
  const float tn = fmaxn(fmaxn(txn, tyn), fmaxn(tzn, kRayTMin));
  const float tf = fminn(fminn(txf, tyf), fminn(tzf, tfar));
  return tn <= tf * kBoxPad;
}

// What the own-box rule of DESIGN.md 3.5 PROMISES, checked node by node on a product builder's quantised tree: for ray i and a triangle
// tri[i] whose candidate at t = th[i] the rule accepted, every node test on the way from the root to the triangle's leaf slot passes with
// tfar = th[i] (so that no walk whose best hit is still >= th[i] can miss the triangle).  fails[i] = the nodes on that path whose test fails
// (0 = the promise holds); a triangle that is not in the tree counts 1000.
extern "C" void orc_quad_path_check(const uint32_t *quads, uint32_t n_quads, const float root_box[6], const uint32_t *order, uint32_t n_tris, int64_t n,
                                    const float *o, const float *d, const uint32_t *tri, const float *th, uint32_t *fails) {
  struct Up { uint32_t node; int k; };
  std::vector<Up> node_up(n_quads, Up{kNoPrim, 0}), slot_up(n_tris, Up{kNoPrim, 0});
  for (uint32_t q = 0; q < n_quads; q++)
    for (int k = 0; k < 4; k++) {
      const uint32_t ref = quads[(size_t)q * 16 + 12 + k];
      if (ref & kLeafRef) {
        const uint32_t cnt = (ref >> 24) & 0x7fu, first = ref & 0xffffffu;
        for (uint32_t j = 0; j < cnt && first + j < n_tris; j++) slot_up[first + j] = Up{q, k};
      } else if (ref / 64u < n_quads && ref != 0u) {
        node_up[ref / 64u] = Up{q, k};
      }
    }
  std::vector<uint32_t> slot_of(n_tris, kNoPrim);
  for (uint32_t sl = 0; sl < n_tris; sl++)
    if (order[sl] < n_tris) slot_of[order[sl]] = sl;
  for (int64_t i = 0; i < n; i++) {
    fails[i] = 0;
    if (tri[i] >= n_tris || slot_of[tri[i]] == kNoPrim || n_quads == 0) { fails[i] = 1000; continue; }
    const V3 oo = {o[3 * i], o[3 * i + 1], o[3 * i + 2]}, dd = {d[3 * i], d[3 * i + 1], d[3 * i + 2]};
    const V3 inv1 = {1.0f / dd.x, 1.0f / dd.y, 1.0f / dd.z};
    const bool negx = inv1.x < 0.f, negy = inv1.y < 0.f, negz = inv1.z < 0.f;
    float big;
    {
      const float extent = std::fmax(root_box[3] - root_box[0], std::fmax(root_box[4] - root_box[1], root_box[5] - root_box[2]));
      int x = 0;
      if (extent > 0.f && std::isfinite(extent)) (void)std::frexp(2.0f * extent, &x);
      int e = 123 - x;
      e = e > 120 ? 120 : (e < -100 ? -100 : e);
      big = std::ldexp(1.0f, e);
    }
    const V3 inv = {dd.x == 0.f ? std::copysign(big, inv1.x) : inv1.x, dd.y == 0.f ? std::copysign(big, inv1.y) : inv1.y, dd.z == 0.f ? std::copysign(big, inv1.z) : inv1.z};
    const bool inside = oo.x >= root_box[0] && oo.x <= root_box[3] && oo.y >= root_box[1] && oo.y <= root_box[4] && oo.z >= root_box[2] && oo.z <= root_box[5];
    if (!inside && !box_test(root_box, oo, inv1, th[i])) fails[i]++;
    Up u = slot_up[slot_of[tri[i]]];
    for (int guard = 0; u.node != kNoPrim && guard < 4096; guard++) {
      if (!child_passes(quads + (size_t)u.node * 16, u.k, oo, inv, negx, negy, negz, th[i])) fails[i]++;
      if (u.node == 0u) break;
      u = node_up[u.node];
    }
  }
}

// per-node visit counters of the walks that follow (n_quads uint64 words, or null to stop counting)
extern "C" void orc_quad_walk_count_visits(uint64_t *per_node) { g_visits = per_node; }

extern "C" void orc_quad_walk(const uint32_t *quads, uint32_t n_quads, uint32_t root_ref, const float root_box[6], const float *P,
                              const uint32_t *idx, const uint32_t *order, int64_t n, const float *o, const float *d,
                              const float *tmax, int any_hit, float *t, uint32_t *prim, float *b1, float *b2, uint8_t *occluded,
                              uint32_t *steps, uint32_t *tris, uint32_t *max_stack, int n_threads, const float *exact_boxes) {
  const Tree T = {quads, n_quads, root_ref, root_box, P, idx, order, exact_boxes};
  if (n_threads < 1) n_threads = 1;
  uint32_t worst = 0;
  std::vector<uint32_t> worst_of((size_t)n_threads, 0u);
  auto work = [&](int w) {
    for (int64_t i = w; i < n; i += n_threads) {
      Out r;
      walk(T, {o[3 * i], o[3 * i + 1], o[3 * i + 2]}, {d[3 * i], d[3 * i + 1], d[3 * i + 2]}, tmax[i], any_hit != 0, &r);
      if (t) t[i] = r.t;
      if (prim) prim[i] = r.prim;
      if (b1) b1[i] = r.b1;
      if (b2) b2[i] = r.b2;
      if (occluded) occluded[i] = r.occluded ? 1 : 0;
      if (steps) steps[i] = (uint32_t)r.steps;
      if (tris) tris[i] = (uint32_t)r.tris | (std::getenv("ORC_WALK_STACK_IN_TRIS") ? r.max_stack << 16 : 0u);  // (diagnostics: the ray's deepest stack in the high half)
      if (r.max_stack > worst_of[w]) worst_of[w] = r.max_stack;
    }
  };
  std::vector<std::thread> th;
  for (int w = 1; w < n_threads; w++) th.emplace_back(work, w);
  work(0);
  for (auto &x : th) x.join();
  for (uint32_t v : worst_of) worst = v > worst ? v : worst;
  if (max_stack) *max_stack = worst;
}
