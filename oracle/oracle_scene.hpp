// oracle_scene.hpp -- scene, BVH build + traversal, shapes of the CPU ORACLE
// (test infrastructure only, see pbrt_oracle.h).  The reference has none of this
// (accelerator is a name only, api.rs:237; Shape is rejected by the
// parser, parser.rs:300); class and method names follow pbrt-v3 (Primitive::WorldBound /
// Intersect / IntersectP, BVHAccel) as SURVEY.md section 8(b) asks, the arithmetic follows
// DESIGN.md section 3.
#pragma once
#include <algorithm>
#include <cstring>
#include <vector>

#include "oracle_math.hpp"
#include "pbrt_oracle.h"

namespace orc {

struct Bounds3 {
  Vec3 pmin{kInf, kInf, kInf}, pmax{-kInf, -kInf, -kInf};
};
static inline float fmin2(float a, float b) { return a < b ? a : b; }
static inline float fmax2(float a, float b) { return a > b ? a : b; }
static inline Bounds3 Union(const Bounds3 &a, const Bounds3 &b) {
  Bounds3 r;
  r.pmin = {fmin2(a.pmin.x, b.pmin.x), fmin2(a.pmin.y, b.pmin.y), fmin2(a.pmin.z, b.pmin.z)};
  r.pmax = {fmax2(a.pmax.x, b.pmax.x), fmax2(a.pmax.y, b.pmax.y), fmax2(a.pmax.z, b.pmax.z)};
  return r;
}
static inline Bounds3 Union(const Bounds3 &a, Vec3 p) {
  Bounds3 r;
  r.pmin = {fmin2(a.pmin.x, p.x), fmin2(a.pmin.y, p.y), fmin2(a.pmin.z, p.z)};
  r.pmax = {fmax2(a.pmax.x, p.x), fmax2(a.pmax.y, p.y), fmax2(a.pmax.z, p.z)};
  return r;
}
static inline float SurfaceArea(const Bounds3 &b) {
  Vec3 d = b.pmax - b.pmin;
  return 2.0f * ((d.x * d.y + d.x * d.z) + d.y * d.z);
}

struct Ray {
  Vec3 o, d;
  float tmax;
};

struct Hit {
  float t = kInf;
  uint32_t prim = 0xffffffffu;
  float b1 = 0.f, b2 = 0.f;
};

struct Counters {
  uint64_t nodes = 0, tris = 0;
};

// 32-byte flattened node: bounds, offset (first ordered prim of a leaf, or the
// second child of an interior node; the first child is the next node), prim count, split axis.
struct LinearBVHNode {
  float bmin[3];
  float bmax[3];
  uint32_t offset;
  uint16_t n_prims;
  uint8_t axis;
  uint8_t pad;
};
static_assert(sizeof(LinearBVHNode) == 32, "node must be 32 bytes");

struct LightRec {
  uint32_t type;  // 0 point, 1 distant, 2 infinite, 3 triangle
  Vec3 p0, p1, p2;  // point: p0 = position; distant: p0 = direction to light; triangle: vertices
  Vec3 c;           // I / L / Le
  Vec3 n;           // triangle: geometric normal
  float area;       // triangle
};

class Scene {
 public:
  std::vector<Vec3> P;
  std::vector<uint32_t> idx;
  std::vector<uint16_t> mat_id;
  std::vector<orc_material> mats;
  std::vector<orc_texture> textures;
  std::vector<float> tri_uv;          // 6 per triangle (corner u, v), or empty
  std::vector<orc_sphere> spheres;
  std::vector<LightRec> lights;
  Vec3 le_infinite{0, 0, 0};
  bool has_infinite = false;
  // camera
  Mat4 c2w;
  float cam_ax, cam_bx, cam_ay, cam_by;
  int xres, yres;
  float crop[4];
  int32_t cropped[4];  // x0 y0 x1 y1
  // accelerator
  std::vector<LinearBVHNode> nodes;
  std::vector<uint32_t> order;  // ordered prim slot -> triangle id
  uint32_t depth = 0;

  uint32_t n_tris() const { return (uint32_t)mat_id.size(); }

  void tri_verts(uint32_t t, Vec3 *p0, Vec3 *p1, Vec3 *p2) const {
    *p0 = P[idx[3 * t]];
    *p1 = P[idx[3 * t + 1]];
    *p2 = P[idx[3 * t + 2]];
  }

  // ---- BVHAccel build: binned SAH, 16 buckets, <= 4 prims per leaf ----
  struct PrimInfo {
    uint32_t id;
    Bounds3 b;
    Vec3 c;
  };
  struct BuildNode {
    Bounds3 b;
    int child[2] = {-1, -1};
    uint32_t first = 0, n = 0;
    int axis = 0;
  };
  std::vector<BuildNode> bnodes;

  int recursive_build(std::vector<PrimInfo> &pi, size_t start, size_t end, uint32_t d) {
    int me = (int)bnodes.size();
    bnodes.emplace_back();
    if (d + 1 > depth) depth = d + 1;
    Bounds3 bounds;
    for (size_t i = start; i < end; i++) bounds = Union(bounds, pi[i].b);
    bnodes[me].b = bounds;
    size_t n = end - start;
    auto make_leaf = [&]() {
      bnodes[me].first = (uint32_t)order.size();
      bnodes[me].n = (uint32_t)n;
      for (size_t i = start; i < end; i++) order.push_back(pi[i].id);
      return me;
    };
    if (n == 1) return make_leaf();
    Bounds3 cb;
    for (size_t i = start; i < end; i++) cb = Union(cb, pi[i].c);
    Vec3 ext = cb.pmax - cb.pmin;
    int dim = (ext.x > ext.y && ext.x > ext.z) ? 0 : (ext.y > ext.z ? 1 : 2);  // pbrt MaximumExtent
    float cmin = cb.pmin[dim], cmax = cb.pmax[dim];
    size_t mid = (start + end) / 2;
    if (cmax == cmin) {
      if (n <= 64) return make_leaf();
      // degenerate: split by position in the current order
    } else if (d >= 32) {
      // depth guard: median split, deterministic through a stable sort
      std::stable_sort(pi.begin() + start, pi.begin() + end,
                       [dim](const PrimInfo &a, const PrimInfo &b) { return a.c[dim] < b.c[dim]; });
    } else if (n == 2) {
      if (pi[start + 1].c[dim] < pi[start].c[dim]) std::swap(pi[start], pi[start + 1]);
    } else {
      const int nB = 16;
      int cnt[nB];
      Bounds3 bb[nB];
      for (int i = 0; i < nB; i++) cnt[i] = 0;
      auto bucket_of = [&](const PrimInfo &p) {
        int b = (int)((float)nB * ((p.c[dim] - cmin) / (cmax - cmin)));
        if (b == nB) b = nB - 1;
        return b;
      };
      for (size_t i = start; i < end; i++) {
        int b = bucket_of(pi[i]);
        cnt[b]++;
        bb[b] = Union(bb[b], pi[i].b);
      }
      float cost[nB - 1];
      float sa = SurfaceArea(bounds);
      for (int i = 0; i < nB - 1; i++) {
        Bounds3 b0, b1;
        int c0 = 0, c1 = 0;
        for (int j = 0; j <= i; j++)
          if (cnt[j]) { b0 = Union(b0, bb[j]); c0 += cnt[j]; }
        for (int j = i + 1; j < nB; j++)
          if (cnt[j]) { b1 = Union(b1, bb[j]); c1 += cnt[j]; }
        float s0 = c0 ? SurfaceArea(b0) : 0.f, s1 = c1 ? SurfaceArea(b1) : 0.f;
        cost[i] = 1.0f + ((float)c0 * s0 + (float)c1 * s1) / sa;
      }
      float min_cost = cost[0];
      int min_b = 0;
      for (int i = 1; i < nB - 1; i++)
        if (cost[i] < min_cost) { min_cost = cost[i]; min_b = i; }
      float leaf_cost = (float)n;
      if (n > 4 || min_cost < leaf_cost) {
        auto it = std::stable_partition(pi.begin() + start, pi.begin() + end,
                                        [&](const PrimInfo &p) { return bucket_of(p) <= min_b; });
        mid = (size_t)(it - pi.begin());
      } else {
        return make_leaf();
      }
    }
    bnodes[me].axis = dim;
    int c0 = recursive_build(pi, start, mid, d + 1);
    bnodes[me].child[0] = c0;
    int c1 = recursive_build(pi, mid, end, d + 1);
    bnodes[me].child[1] = c1;
    return me;
  }

  uint32_t flatten(int bn) {
    uint32_t me = (uint32_t)nodes.size();
    nodes.emplace_back();
    const BuildNode &b = bnodes[bn];
    LinearBVHNode ln;
    ln.bmin[0] = b.b.pmin.x; ln.bmin[1] = b.b.pmin.y; ln.bmin[2] = b.b.pmin.z;
    ln.bmax[0] = b.b.pmax.x; ln.bmax[1] = b.b.pmax.y; ln.bmax[2] = b.b.pmax.z;
    ln.pad = 0;
    if (b.n > 0) {
      ln.offset = b.first;
      ln.n_prims = (uint16_t)b.n;
      ln.axis = 0;
      nodes[me] = ln;
    } else {
      ln.n_prims = 0;
      ln.axis = (uint8_t)b.axis;
      flatten(b.child[0]);
      ln.offset = flatten(b.child[1]);
      nodes[me] = ln;
    }
    return me;
  }

  // A sphere's own box: [c - r, c + r] per component in fp32 -- what the builder bounds it by and what the own-box rule tests
  static void sphere_box(const orc_sphere &sp, Vec3 *lo, Vec3 *hi) {
    *lo = {sp.c[0] - sp.r, sp.c[1] - sp.r, sp.c[2] - sp.r};
    *hi = {sp.c[0] + sp.r, sp.c[1] + sp.r, sp.c[2] + sp.r};
  }

  // The primitives of the tree: the triangles, then the spheres (primitive n_tris + s: round 6 -- until round 5 every ray tested every
  // sphere after the walk), each bounded by its box, centroid = the box's centre.
  void build_bvh() {
    const uint32_t nt = n_tris(), np = nt + (uint32_t)spheres.size();
    if (np == 0) return;
    std::vector<PrimInfo> pi(np);
    for (uint32_t t = 0; t < nt; t++) {
      Vec3 p0, p1, p2;
      tri_verts(t, &p0, &p1, &p2);
      Bounds3 b;
      b = Union(b, p0); b = Union(b, p1); b = Union(b, p2);
      pi[t].id = t;
      pi[t].b = b;
      pi[t].c = b.pmin * 0.5f + b.pmax * 0.5f;
    }
    for (uint32_t s = 0; s < spheres.size(); s++) {
      Vec3 lo, hi;
      sphere_box(spheres[s], &lo, &hi);
      Bounds3 b;
      b = Union(b, lo); b = Union(b, hi);
      pi[nt + s].id = nt + s;
      pi[nt + s].b = b;
      pi[nt + s].c = b.pmin * 0.5f + b.pmax * 0.5f;
    }
    order.reserve(np);
    bnodes.reserve(2 * (size_t)np);
    recursive_build(pi, 0, np, 0);
    nodes.reserve(bnodes.size());
    flatten(0);
    bnodes.clear();
    bnodes.shrink_to_fit();
  }

  // (tests only: the rule of in_own_box switched off, to show that the adversarial-ray tests can see what it closes)
  static bool &own_box_rule_on() { static bool on = true; return on; }

  // ---- Triangle::Intersect: Moeller-Trumbore in the operation order of SURVEY A5 ----
  bool tri_intersect(uint32_t t, const Ray &r, float *tt, float *uu, float *vv) const {
    Vec3 p0, p1, p2;
    tri_verts(t, &p0, &p1, &p2);
    Vec3 e1 = p1 - p0, e2 = p2 - p0;
    Vec3 pv = cross(r.d, e2);
    float det = dot(e1, pv);
    if (std::fabs(det) < 1e-8f) return false;
    float inv = 1.0f / det;
    Vec3 tv = r.o - p0;
    float u = dot(tv, pv) * inv;
    Vec3 qv = cross(tv, e1);
    float v = dot(r.d, qv) * inv;
    float th = dot(e2, qv) * inv;
    if (!(u >= 0.f) || !(v >= 0.f) || !(u + v <= 1.0f)) return false;
    if (!(th > kRayTMin) || !(th < r.tmax)) return false;
    if (!own_box_rule(p0, p1, p2, r, &th)) return false;
    *tt = th; *uu = u; *vv = v;
    return true;
  }

  // The own-box rule (round 6).  A candidate of fp32 Moeller-Trumbore at distance th counts as a hit only if the RAY MEETS THE
  // PRIMITIVE'S OWN BOUNDING BOX -- the node test of 3.4 on that box: slab distances of the three vertices with the node test's arithmetic
  // ((p - o) * (1 / d), two roundings), their minimum / maximum per axis (NaN-ignoring), tn = max(near, 1e-4), tf = min(far), tn <= tf * kOwnPad
  // with kOwnPad < kBoxPad --, and its distance is AT LEAST THE BOX'S ENTRY: t = max(th, tn), still < tmax.  Subtraction and multiplication
  // are monotone, so every box that encloses the primitive has tn' <= tn <= t and tf' >= tf: EVERY walk over EVERY tree whose best hit is
  // still >= t reaches the primitive, and whether and where a ray hits a primitive is a function of the ray and the primitive alone.
  // (Without the rule an ill-conditioned test "hits" outside the triangle's box, where a walk over tight boxes culls the triangle and a walk
  // through wider ones does not.  Raising t to the box's entry instead of REJECTING a candidate that lies before it matters for triangles
  // that lie flat in an axis plane -- every wall of a Cornell box: there entry = exit = the plane's own slab distance, Moeller-Trumbore's t
  // differs from it by rounding (by far more for a sliver), and a rule that rejected would punch holes into walls.)
  static bool own_box_rule(Vec3 p0, Vec3 p1, Vec3 p2, const Ray &r, float *th) {
    if (!own_box_rule_on()) return true;  // (tests only: orc_debug_own_box_rule(0) shows what the rule is for)
    const Vec3 inv = {1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z};
    const float x0 = (p0.x - r.o.x) * inv.x, x1 = (p1.x - r.o.x) * inv.x, x2 = (p2.x - r.o.x) * inv.x;
    const float y0 = (p0.y - r.o.y) * inv.y, y1 = (p1.y - r.o.y) * inv.y, y2 = (p2.y - r.o.y) * inv.y;
    const float z0 = (p0.z - r.o.z) * inv.z, z1 = (p1.z - r.o.z) * inv.z, z2 = (p2.z - r.o.z) * inv.z;
    const float nx = std::fmin(std::fmin(x0, x1), x2), fx = std::fmax(std::fmax(x0, x1), x2);
    const float ny = std::fmin(std::fmin(y0, y1), y2), fy = std::fmax(std::fmax(y0, y1), y2);
    const float nz = std::fmin(std::fmin(z0, z1), z2), fz = std::fmax(std::fmax(z0, z1), z2);
    const float tn = std::fmax(std::fmax(nx, ny), std::fmax(nz, kRayTMin));
    const float tf = std::fmin(std::fmin(fx, fy), fz);
    if (!(tn <= tf * kOwnPad)) return false;
    const float t = std::fmax(*th, tn);
    if (!(t < r.tmax)) return false;
    *th = t;
    return true;
  }

  // ---- Sphere::Intersect, quadratic per lib.rs:181-203 ----
  bool sphere_intersect(uint32_t s, const Ray &r, float *tt) const {
    const orc_sphere &sp = spheres[s];
    Vec3 oc = r.o - v3(sp.c[0], sp.c[1], sp.c[2]);
    float a = dot(r.d, r.d);
    float b = 2.0f * dot(r.d, oc);
    float c = dot(oc, oc) - sp.r * sp.r;
    float t0, t1;
    if (!quadratic(a, b, c, &t0, &t1)) return false;
    float th = t0;
    if (!(th > kRayTMin && th < r.tmax)) {
      th = t1;
      if (!(th > kRayTMin && th < r.tmax)) return false;
    }
    // the own-box rule on the sphere's box [c - r, c + r]: the "vertices" lo, hi, lo
    Vec3 lo, hi;
    sphere_box(sp, &lo, &hi);
    if (!own_box_rule(lo, hi, lo, r, &th)) return false;
    *tt = th;
    return true;
  }

  // one primitive of a leaf: triangle t < n_tris, else sphere t - n_tris (u = v = 0)
  bool prim_intersect(uint32_t t, const Ray &r, float *tt, float *uu, float *vv) const {
    if (t < n_tris()) return tri_intersect(t, r, tt, uu, vv);
    *uu = 0.f; *vv = 0.f;
    return sphere_intersect(t - n_tris(), r, tt);
  }

  // Bounds3::IntersectP: slab test against [kRayTMin, tfar].  Near / far plane per axis chosen by
  // the sign of the inverse direction (pbrt-v3 dirIsNeg); a 0 * inf = NaN (ray parallel to a slab
  // and starting exactly on its plane) is ignored by fmin / fmax, which keeps the test
  // conservative; far side padded by kBoxPad (1 + 2^-19 since round 6; pbrt-v3: 1 + 2*gamma(3)).
  static bool box_hit(const LinearBVHNode &n, const Ray &r, Vec3 inv, const int neg[3], float tfar) {
    float nx = ((neg[0] ? n.bmax[0] : n.bmin[0]) - r.o.x) * inv.x, fx = ((neg[0] ? n.bmin[0] : n.bmax[0]) - r.o.x) * inv.x;
    float ny = ((neg[1] ? n.bmax[1] : n.bmin[1]) - r.o.y) * inv.y, fy = ((neg[1] ? n.bmin[1] : n.bmax[1]) - r.o.y) * inv.y;
    float nz = ((neg[2] ? n.bmax[2] : n.bmin[2]) - r.o.z) * inv.z, fz = ((neg[2] ? n.bmin[2] : n.bmax[2]) - r.o.z) * inv.z;
    float tn = std::fmax(std::fmax(nx, ny), std::fmax(nz, kRayTMin));
    float tf = std::fmin(std::fmin(fx, fy), std::fmin(fz, tfar));
    return tn <= tf * kBoxPad;
  }

  static inline void consider(Hit *h, float t, uint32_t prim, float u, float v) {
    if (t < h->t || (t == h->t && prim < h->prim)) {
      h->t = t; h->prim = prim; h->b1 = u; h->b2 = v;
    }
  }

  // BVHAccel::Intersect: closest hit; ties on t resolved towards the lower primitive id so the
  // answer does not depend on the tree.
  Hit Intersect(const Ray &r, Counters *ctr) const {
    Hit h;
    if (!nodes.empty()) {
      Vec3 inv = {1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z};
      int neg[3] = {inv.x < 0.f, inv.y < 0.f, inv.z < 0.f};
      uint32_t stack[64];
      int sp = 0;
      uint32_t cur = 0;
      for (;;) {
        const LinearBVHNode &n = nodes[cur];
        if (ctr) ctr->nodes++;
        float tfar = h.t < r.tmax ? h.t : r.tmax;
        if (box_hit(n, r, inv, neg, tfar)) {
          if (n.n_prims > 0) {
            for (uint32_t i = 0; i < n.n_prims; i++) {
              uint32_t t = order[n.offset + i];
              if (ctr) ctr->tris++;
              float tt, u, v;
              if (prim_intersect(t, r, &tt, &u, &v)) consider(&h, tt, t, u, v);
            }
            if (sp == 0) break;
            cur = stack[--sp];
          } else if (neg[n.axis]) {
            stack[sp++] = cur + 1;
            cur = n.offset;
          } else {
            stack[sp++] = n.offset;
            cur = cur + 1;
          }
        } else {
          if (sp == 0) break;
          cur = stack[--sp];
        }
      }
    }
    return h;
  }

  // BVHAccel::IntersectP: any hit in (kRayTMin, tmax)
  bool IntersectP(const Ray &r, Counters *ctr) const {
    if (!nodes.empty()) {
      Vec3 inv = {1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z};
      int neg[3] = {inv.x < 0.f, inv.y < 0.f, inv.z < 0.f};
      uint32_t stack[64];
      int sp = 0;
      uint32_t cur = 0;
      for (;;) {
        const LinearBVHNode &n = nodes[cur];
        if (ctr) ctr->nodes++;
        if (box_hit(n, r, inv, neg, r.tmax)) {
          if (n.n_prims > 0) {
            for (uint32_t i = 0; i < n.n_prims; i++) {
              if (ctr) ctr->tris++;
              float tt, u, v;
              if (prim_intersect(order[n.offset + i], r, &tt, &u, &v)) return true;
            }
            if (sp == 0) break;
            cur = stack[--sp];
          } else if (neg[n.axis]) {
            stack[sp++] = cur + 1;
            cur = n.offset;
          } else {
            stack[sp++] = n.offset;
            cur = cur + 1;
          }
        } else {
          if (sp == 0) break;
          cur = stack[--sp];
        }
      }
    }
    return false;
  }

  // No accelerator at all: the independent cross-check of the BVH.
  Hit IntersectBrute(const Ray &r) const {
    Hit h;
    for (uint32_t t = 0; t < n_tris(); t++) {
      float tt, u, v;
      if (tri_intersect(t, r, &tt, &u, &v)) consider(&h, tt, t, u, v);
    }
    for (uint32_t s = 0; s < spheres.size(); s++) {
      float tt;
      if (sphere_intersect(s, r, &tt)) consider(&h, tt, n_tris() + s, 0.f, 0.f);
    }
    return h;
  }
  bool IntersectPBrute(const Ray &r) const {
    float tt, u, v;
    for (uint32_t t = 0; t < n_tris(); t++)
      if (tri_intersect(t, r, &tt, &u, &v)) return true;
    for (uint32_t s = 0; s < spheres.size(); s++)
      if (sphere_intersect(s, r, &tt)) return true;
    return false;
  }

  // PerspectiveCamera::GenerateRay
  Ray camera_ray(float fx, float fy) const {
    Vec3 pc = {fx * cam_ax + cam_bx, fy * cam_ay + cam_by, 1.0f};
    Vec3 dc = normalize(pc);
    Ray r;
    r.d = {(c2w.m[0][0] * dc.x + c2w.m[0][1] * dc.y) + c2w.m[0][2] * dc.z,
           (c2w.m[1][0] * dc.x + c2w.m[1][1] * dc.y) + c2w.m[1][2] * dc.z,
           (c2w.m[2][0] * dc.x + c2w.m[2][1] * dc.y) + c2w.m[2][2] * dc.z};
    r.o = {c2w.m[0][3], c2w.m[1][3], c2w.m[2][3]};
    r.tmax = kInf;
    return r;
  }
};

// Film::new cropped pixel bounds, /root/reference/src/core/film.rs:92-101
static inline void film_cropped_bounds(int xres, int yres, const float crop[4], int32_t out[4]) {
  out[0] = (int32_t)std::ceil((float)xres * crop[0]);
  out[1] = (int32_t)std::ceil((float)yres * crop[2]);
  out[2] = (int32_t)std::ceil((float)xres * crop[1]);
  out[3] = (int32_t)std::ceil((float)yres * crop[3]);
}

static inline void setup_camera(Scene &s, const float c2w[16], float fov, int xres, int yres, const float crop[4]) {
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) s.c2w.m[i][j] = c2w[4 * i + j];
  s.xres = xres;
  s.yres = yres;
  for (int i = 0; i < 4; i++) s.crop[i] = crop[i];
  film_cropped_bounds(xres, yres, crop, s.cropped);
  float aspect = (float)xres / (float)yres;
  float sxmin, sxmax, symin, symax;
  if (aspect > 1.f) { sxmin = -aspect; sxmax = aspect; symin = -1.f; symax = 1.f; }
  else { sxmin = -1.f; sxmax = 1.f; symin = -1.f / aspect; symax = 1.f / aspect; }
  float tan_half = (float)std::tan((double)fov * (3.14159265358979323846 / 180.0) * 0.5);
  s.cam_ax = ((sxmax - sxmin) / (float)xres) * tan_half;
  s.cam_bx = sxmin * tan_half;
  s.cam_ay = -((symax - symin) / (float)yres) * tan_half;
  s.cam_by = symax * tan_half;
}

}  // namespace orc
