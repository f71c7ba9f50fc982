// kernels.hip -- the CDNA4 (gfx950) kernels of the render path.
//
//   render_kernel    one 64-lane wavefront per 8x8 pixel tile (4 waves = one 16x16 FilmTile per
//                    workgroup); every lane owns one pixel and walks its samples in order:
//                    stratified camera sample -> BVH closest hit -> emission -> one-light direct
//                    estimate (any-hit shadow ray) -> BSDF sample -> Russian roulette.  A lane
//                    whose path ends regenerates its next camera sample at once, so the wave
//                    stays full until the tile runs out of samples; `__ballot` decides the
//                    wave-uniform exits.  Per-lane traversal stack in LDS, laid out
//                    stack[level][lane] (conflict-free: lane l -> bank l).
//   intersect_kernel the traversal loop alone over a ray batch (parity + roofline of the loop).
//   pack_tris_kernel builds the leaf-ordered 48-byte triangle records from the uploaded
//                    vertex / index buffers.
//   assemble_kernel  scatters a rank's tile-major slab into the row-major film.
//
// The reference has no renderer (core/api.rs:446-453 is a comment); the arithmetic below is
// DESIGN.md section 3, and is written so that every fp32 operation happens in the same order as
// in the CPU oracle: build with -ffp-contract=off, never -ffast-math.  No MFMA: this is branchy
// gather work (BASELINE.json north_star).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_types.h"

namespace pbrt_hip {
namespace {

constexpr float kInf = __builtin_huge_valf();
constexpr float kRayTMin = 1e-4f;
constexpr float kSpawnEps = 1e-4f;
constexpr float kShadowShrink = 0.9999f;
constexpr float kBoxPad = 0x1.000006p+0f;  // 1 + 2*gamma(3)
constexpr float kInvPi = 0.31830988618379067154f;
constexpr float kPiOver4 = 0.78539816339744830961f;
constexpr float kOneMinusEps = 0x1.fffffep-1f;  // 1 - f32::EPSILON, core/rng.rs:19
constexpr uint32_t kNoPrim = 0xffffffffu;

struct V3 {
  float x, y, z;
};
__device__ __forceinline__ V3 mk(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return {(a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x)};
}
__device__ __forceinline__ V3 unit(V3 a) { return a / sqrtf(dot(a, a)); }
__device__ __forceinline__ V3 xyz(float4 v) { return {v.x, v.y, v.z}; }

// ---- PCG32, core/rng.rs:46-93 ----
struct Pcg {
  uint64_t state, inc;
};
__device__ __forceinline__ uint32_t pcg_u32(Pcg &r) {
  uint64_t old = r.state;
  r.state = old * 0x5851f42d4c957f2dULL + r.inc;
  uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
  uint32_t rot = (uint32_t)(old >> 59u);
  return (xs >> rot) | (xs << ((0u - rot) & 31u));
}
__device__ __forceinline__ void pcg_seq(Pcg &r, uint64_t seq) {
  r.state = 0;
  r.inc = (seq << 1) | 1u;
  pcg_u32(r);
  r.state += 0x853c49e6748fea9bULL;
  pcg_u32(r);
}
__device__ __forceinline__ float pcg_float(Pcg &r) {
  return fminf(kOneMinusEps, (float)pcg_u32(r) * 2.3283064365386963e-10f);
}

// ---- fixed sin / cos polynomials on [-pi/4, pi/4] ----
__device__ __forceinline__ float poly_sin(float x) {
  float z = x * x;
  float p = -1.9515295891e-4f * z + 8.3321608736e-3f;
  p = p * z - 1.6666654611e-1f;
  return (p * z) * x + x;
}
__device__ __forceinline__ float poly_cos(float x) {
  float z = x * x;
  float p = 2.443315711809948e-5f * z - 1.388731625493765e-3f;
  p = p * z + 4.166664568298827e-2f;
  return ((p * z) * z - 0.5f * z) + 1.0f;
}

// cosine-weighted direction about n; returns local z (0 => pdf 0)
__device__ __forceinline__ float cosine_about(V3 n, float u1, float u2, V3 &wi) {
  float ox = 2.0f * u1 - 1.0f, oy = 2.0f * u2 - 1.0f;
  float dx, dy;
  if (ox == 0.f && oy == 0.f) {
    dx = 0.f;
    dy = 0.f;
  } else if (fabsf(ox) > fabsf(oy)) {
    float phi = kPiOver4 * (oy / ox);
    dx = ox * poly_cos(phi);
    dy = ox * poly_sin(phi);
  } else {
    float phi = kPiOver4 * (ox / oy);
    dx = oy * poly_sin(phi);
    dy = oy * poly_cos(phi);
  }
  float zz = (1.0f - dx * dx) - dy * dy;
  float z = sqrtf(zz > 0.f ? zz : 0.f);
  V3 v2;
  if (fabsf(n.x) > fabsf(n.y)) {
    float l = sqrtf(n.x * n.x + n.z * n.z);
    v2 = {-n.z / l, 0.f, n.x / l};
  } else {
    float l = sqrtf(n.y * n.y + n.z * n.z);
    v2 = {0.f, n.z / l, -n.y / l};
  }
  V3 v3 = cross(n, v2);
  wi = (v2 * dx + v3 * dy) + n * z;
  return z;
}

struct HitRec {
  float t;
  uint32_t prim;  // triangle id (or n_tris + sphere index); kNoPrim on a miss
  uint32_t slot;  // leaf slot of a triangle hit
  float b1, b2;
};

// lib.rs:181-203 quadratic with its f64 discriminant
__device__ __forceinline__ bool quadratic(float af, float bf, float cf, float &t0, float &t1) {
  double a = af, b = bf, c = cf;
  double disc = b * b - 4. * a * c;
  if (disc < 0.) return false;
  double rd = sqrt(disc);
  double q = (b < 0.) ? -0.5 * (b - rd) : -0.5 * (b + rd);
  float r0 = (float)(q / a), r1 = (float)(c / q);
  if (r0 > r1) { t0 = r1; t1 = r0; } else { t0 = r0; t1 = r1; }
  return true;
}

__device__ __forceinline__ bool sphere_hit(const DevScene &S, uint32_t s, V3 o, V3 d, float tmax, float &th) {
  float4 cr = S.spheres[2 * s];
  V3 oc = o - xyz(cr);
  float a = dot(d, d);
  float b = 2.0f * dot(d, oc);
  float c = dot(oc, oc) - cr.w * cr.w;
  float t0, t1;
  if (!quadratic(a, b, c, t0, t1)) return false;
  th = t0;
  if (!(th > kRayTMin && th < tmax)) {
    th = t1;
    if (!(th > kRayTMin && th < tmax)) return false;
  }
  return true;
}

// Closest hit (any == false) or any hit (any == true; returns true when occluded) of one ray per
// lane.  `stk` points at this lane's column of the LDS stack; consecutive levels are 64 dwords
// apart.  Tie rule for equal t: the lower primitive id wins, so the answer does not depend on the
// shape of the tree.
template <bool SPH, bool COUNT>
__device__ __forceinline__ bool traverse(const DevScene &S, V3 o, V3 d, float tmax, bool any, uint32_t *stk,
                                         HitRec &h, unsigned long long &cn, unsigned long long &ct) {
  h.t = kInf;
  h.prim = kNoPrim;
  h.slot = kNoPrim;
  h.b1 = 0.f;
  h.b2 = 0.f;
  if (S.n_nodes) {
    const V3 inv = {1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    const uint32_t neg = (inv.x < 0.f ? 1u : 0u) | (inv.y < 0.f ? 2u : 0u) | (inv.z < 0.f ? 4u : 0u);
    uint32_t cur = 0;
    int sp = 0;
    for (;;) {
      const uint4 n0 = S.nodes[2 * cur];
      const uint4 n1 = S.nodes[2 * cur + 1];
      if (COUNT) cn++;
      const float tfar = fminf(h.t, tmax);
      // near / far plane per axis by the sign of the inverse direction; fmin / fmax ignore a
      // 0 * inf = NaN, which keeps the test conservative
      const float lx = __uint_as_float(n0.x), ly = __uint_as_float(n0.y), lz = __uint_as_float(n0.z);
      const float hx = __uint_as_float(n0.w), hy = __uint_as_float(n1.x), hz = __uint_as_float(n1.y);
      const float nx = ((neg & 1u ? hx : lx) - o.x) * inv.x, fx = ((neg & 1u ? lx : hx) - o.x) * inv.x;
      const float ny = ((neg & 2u ? hy : ly) - o.y) * inv.y, fy = ((neg & 2u ? ly : hy) - o.y) * inv.y;
      const float nz = ((neg & 4u ? hz : lz) - o.z) * inv.z, fz = ((neg & 4u ? lz : hz) - o.z) * inv.z;
      const float tn = fmaxf(fmaxf(nx, ny), fmaxf(nz, kRayTMin));
      const float tf = fminf(fminf(fx, fy), fminf(fz, tfar));
      bool pop = true;
      if (tn <= tf * kBoxPad) {
        const uint32_t cnt = n1.w & 0xffffu;
        if (cnt) {
          for (uint32_t i = 0; i < cnt; i++) {
            const uint32_t slot = n1.z + i;
            const float4 a = S.tris[3 * slot], b = S.tris[3 * slot + 1], c = S.tris[3 * slot + 2];
            if (COUNT) ct++;
            const V3 p0 = xyz(a);
            const V3 e1 = xyz(b) - p0, e2 = xyz(c) - p0;
            const V3 pv = cross(d, e2);
            const float det = dot(e1, pv);
            if (fabsf(det) < 1e-8f) continue;
            const float idet = 1.0f / det;
            const V3 tv = o - p0;
            const float u = dot(tv, pv) * idet;
            const V3 qv = cross(tv, e1);
            const float v = dot(d, qv) * idet;
            const float th = dot(e2, qv) * idet;
            if (!(u >= 0.f) || !(v >= 0.f) || !(u + v <= 1.0f)) continue;
            if (!(th > kRayTMin) || !(th < tmax)) continue;
            if (any) return true;
            const uint32_t id = __float_as_uint(a.w);
            if (th < h.t || (th == h.t && id < h.prim)) {
              h.t = th; h.prim = id; h.slot = slot; h.b1 = u; h.b2 = v;
            }
          }
        } else {
          const uint32_t axis = n1.w >> 16;
          pop = false;
          if ((neg >> axis) & 1u) {
            stk[sp * 64] = cur + 1;
            cur = n1.z;
          } else {
            stk[sp * 64] = n1.z;
            cur = cur + 1;
          }
          sp++;
        }
      }
      if (pop) {
        if (sp == 0) break;
        sp--;
        cur = stk[sp * 64];
      }
    }
  }
  if (SPH) {
    for (uint32_t s = 0; s < S.n_spheres; s++) {
      float th;
      if (sphere_hit(S, s, o, d, tmax, th)) {
        if (any) return true;
        const uint32_t id = S.n_tris + s;
        if (th < h.t || (th == h.t && id < h.prim)) {
          h.t = th; h.prim = id; h.slot = kNoPrim; h.b1 = 0.f; h.b2 = 0.f;
        }
      }
    }
  }
  return false;
}

// One light of UniformSampleOneLight.  false: geometry rules the light out.
__device__ __forceinline__ bool sample_light(const DevScene &S, uint32_t li, V3 po, V3 nf, V3 kd, float u1, float u2,
                                             float nLf, V3 &Ld, V3 &wi, float &tmax) {
  const float4 l0 = S.lights[5 * li];
  const float4 l3 = S.lights[5 * li + 3];
  const uint32_t type = __float_as_uint(l0.x);
  const V3 p0 = {l0.y, l0.z, l0.w};
  const V3 lc = xyz(l3);
  const V3 f = kd * kInvPi;
  if (type == 0u) {
    V3 dv = p0 - po;
    float dist2 = dot(dv, dv);
    if (!(dist2 > 0.f)) return false;
    float dist = sqrtf(dist2);
    wi = dv / dist;
    float cs = dot(wi, nf);
    if (!(cs > 0.f)) return false;
    float scale = (cs / dist2) * nLf;
    Ld = (f * lc) * scale;
    tmax = dist * kShadowShrink;
    return true;
  } else if (type == 1u) {
    wi = p0;
    float cs = dot(wi, nf);
    if (!(cs > 0.f)) return false;
    float scale = cs * nLf;
    Ld = (f * lc) * scale;
    tmax = kInf;
    return true;
  } else if (type == 2u) {
    float z = cosine_about(nf, u1, u2, wi);
    if (z == 0.f) return false;
    Ld = (kd * lc) * nLf;
    tmax = kInf;
    return true;
  } else {
    const float4 l1 = S.lights[5 * li + 1];
    const float4 l2 = S.lights[5 * li + 2];
    const float4 l4 = S.lights[5 * li + 4];
    float su0 = sqrtf(u1);
    float b0 = 1.0f - su0;
    float b1 = u2 * su0;
    float b2 = (1.0f - b0) - b1;
    V3 pl = (p0 * b0 + xyz(l1) * b1) + xyz(l2) * b2;
    V3 dv = pl - po;
    float dist2 = dot(dv, dv);
    if (!(dist2 > 0.f)) return false;
    float dist = sqrtf(dist2);
    wi = dv / dist;
    float cs = dot(wi, nf);
    if (!(cs > 0.f)) return false;
    float cl = -dot(wi, xyz(l4));
    if (!(cl > 0.f)) return false;
    float scale = (((cs * cl) * l1.w) / dist2) * nLf;
    Ld = (f * lc) * scale;
    tmax = dist * kShadowShrink;
    return true;
  }
}

enum : uint32_t { ST_NEW = 0, ST_CLOSEST = 1, ST_SHADOW = 2, ST_DONE = 3 };

template <bool SPH, bool COUNT, int STACK>
__global__ void __launch_bounds__(256) render_kernel(const DevScene S, const RenderParams R) {
  __shared__ uint32_t lds_stack[4][STACK][64];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t *stk = &lds_stack[wave][0][lane];

  // block -> (local super-tile, 16x16 tile inside it); wave -> 8x8 quadrant; lane -> pixel
  const int32_t W = S.cx1 - S.cx0, H = S.cy1 - S.cy0;
  const uint32_t stx = (uint32_t)(W + 63) >> 6;
  const uint32_t jsup = blockIdx.x >> 4, sub = blockIdx.x & 15u;
  const uint32_t tsup = R.rank + jsup * R.world;
  const uint32_t pxs = (sub & 3u) * 16u + (wave & 1u) * 8u + (lane & 7u);
  const uint32_t pys = (sub >> 2) * 16u + (wave >> 1) * 8u + (lane >> 3);
  const int32_t xr = (int32_t)((tsup % stx) * 64u + pxs), yr = (int32_t)((tsup / stx) * 64u + pys);
  const bool valid = xr < W && yr < H;
  const int32_t px = S.cx0 + xr, py = S.cy0 + yr;

  const uint32_t spp = R.spp_x * R.spp_y;
  const uint32_t nL = S.n_lights;
  const float nLf = (float)nL;
  const bool direct_only = R.integrator == 1u;

  Pcg rng;
  pcg_seq(rng, R.seed * (uint64_t)S.xres * (uint64_t)S.yres + (uint64_t)py * (uint64_t)S.xres + (uint64_t)px);

  V3 sum = {0.f, 0.f, 0.f};
  V3 L = {0.f, 0.f, 0.f}, beta = {1.f, 1.f, 1.f};
  V3 ro = {0.f, 0.f, 0.f}, rd = {0.f, 0.f, 1.f};
  float rtmax = kInf;
  V3 wi_next = {0.f, 0.f, 0.f}, Lpend = {0.f, 0.f, 0.f};
  uint32_t s = 0, bounces = 0, state = valid ? ST_NEW : ST_DONE;
  bool specular = false, cont = false;
  unsigned long long c_cam = 0, c_bounce = 0, c_shadow = 0, c_nodes = 0, c_tris = 0;

  for (;;) {
    if (state == ST_NEW) {
      if (s == spp) {
        state = ST_DONE;
      } else {
        // stratified camera sample and PerspectiveCamera ray (3.2)
        const uint32_t sx = s % R.spp_x, sy = s / R.spp_x;
        const float u1 = pcg_float(rng), u2 = pcg_float(rng);
        const float jx = fminf(((float)sx + u1) * R.inv_nx, kOneMinusEps);
        const float jy = fminf(((float)sy + u2) * R.inv_ny, kOneMinusEps);
        const float fx = (float)px + jx, fy = (float)py + jy;
        const V3 dc = unit(mk(fx * S.cam_ax + S.cam_bx, fy * S.cam_ay + S.cam_by, 1.0f));
        rd = {(S.c2w[0] * dc.x + S.c2w[1] * dc.y) + S.c2w[2] * dc.z,
              (S.c2w[4] * dc.x + S.c2w[5] * dc.y) + S.c2w[6] * dc.z,
              (S.c2w[8] * dc.x + S.c2w[9] * dc.y) + S.c2w[10] * dc.z};
        ro = {S.c2w[3], S.c2w[7], S.c2w[11]};
        rtmax = kInf;
        L = {0.f, 0.f, 0.f};
        beta = {1.f, 1.f, 1.f};
        specular = false;
        bounces = 0;
        state = ST_CLOSEST;
        if (COUNT) c_cam++;
      }
    }
    if (__ballot(state != ST_DONE) == 0ull) break;
    if (state != ST_DONE) {
    HitRec h;
    const bool any = state == ST_SHADOW;
    const bool occluded = traverse<SPH, COUNT>(S, ro, rd, rtmax, any, stk, h, c_nodes, c_tris);

    bool advance = false;  // take the prepared bounce (or end the sample)
    if (any) {
      if (!occluded) L = L + Lpend;
      advance = true;
    } else {
      const bool hit = h.prim != kNoPrim;
      V3 p = {0.f, 0.f, 0.f}, ng = {0.f, 0.f, 1.f};
      float4 m0 = {0.f, 0.f, 0.f, 0.f}, m1 = {0.f, 0.f, 0.f, 0.f};
      const V3 wo = -rd;
      if (hit) {
        uint32_t mid;
        if (!SPH || h.prim < S.n_tris) {
          const float4 a = S.tris[3 * h.slot], b = S.tris[3 * h.slot + 1], c = S.tris[3 * h.slot + 2];
          const V3 p0 = xyz(a), p1 = xyz(b), p2 = xyz(c);
          ng = unit(cross(p1 - p0, p2 - p0));
          const float w = (1.0f - h.b1) - h.b2;
          p = (p0 * w + p1 * h.b1) + p2 * h.b2;
          mid = __float_as_uint(b.w);
        } else {
          const uint32_t si = h.prim - S.n_tris;
          const float4 cr = S.spheres[2 * si];
          const V3 c = xyz(cr);
          const V3 ph = (ro - c) + rd * h.t;
          ng = ph / cr.w;
          p = c + ph;
          mid = __float_as_uint(S.spheres[2 * si + 1].x);
        }
        m0 = S.mats[2 * mid];
        m1 = S.mats[2 * mid + 1];
      }
      if (bounces == 0 || specular) {
        if (hit) {
          const V3 le = xyz(m1);
          if ((le.x > 0.f || le.y > 0.f || le.z > 0.f) && dot(ng, wo) > 0.f) L = L + beta * le;
        } else if (S.has_inf) {
          L = L + beta * mk(S.le_inf[0], S.le_inf[1], S.le_inf[2]);
        }
      }
      cont = false;
      bool need_shadow = false;
      if (hit && bounces < R.max_depth) {
        const V3 nf = dot(ng, wo) < 0.f ? -ng : ng;
        const V3 po = p + nf * kSpawnEps;
        const V3 k = {m0.y, m0.z, m0.w};
        V3 sh_d = {0.f, 0.f, 1.f};
        float sh_tmax = kInf;
        bool alive = true;
        if (__float_as_uint(m0.x) == 0u) {  // matte
          if (nL > 0u) {
            const float xi = pcg_float(rng), u1 = pcg_float(rng), u2 = pcg_float(rng);
            uint32_t li = (uint32_t)(xi * nLf);
            if (li > nL - 1u) li = nL - 1u;
            V3 Ld;
            if (sample_light(S, li, po, nf, k, u1, u2, nLf, Ld, sh_d, sh_tmax)) {
              need_shadow = true;
              Lpend = beta * Ld;
            }
          }
          if (direct_only) {
            alive = false;
          } else {
            const float u1 = pcg_float(rng), u2 = pcg_float(rng);
            const float z = cosine_about(nf, u1, u2, wi_next);
            if (z == 0.f) alive = false;
            else { beta = beta * k; specular = false; }
          }
        } else {  // mirror
          const float c = dot(wo, nf);
          wi_next = -wo + nf * (2.0f * c);
          beta = beta * k;
          specular = true;
        }
        if (alive && beta.x == 0.f && beta.y == 0.f && beta.z == 0.f) alive = false;
        if (alive && bounces > 3u) {
          const float mx = fmaxf(beta.x, fmaxf(beta.y, beta.z));
          const float q = fmaxf(0.05f, 1.0f - mx);
          if (pcg_float(rng) < q) alive = false;
          else beta = beta / (1.0f - q);
        }
        cont = alive;
        ro = po;  // shadow ray and bounce ray both leave from the offset point
        if (need_shadow) {
          rd = sh_d;
          rtmax = sh_tmax;
          state = ST_SHADOW;
          if (COUNT) c_shadow++;
        }
      }
      if (!need_shadow) advance = true;
    }

    if (advance) {
      bool go = cont;
      if (go) {
        bounces++;
        // a ray at the depth limit can only collect emission, and only after a specular bounce
        if (bounces >= R.max_depth && !specular) go = false;
      }
      if (go) {
        rd = wi_next;
        rtmax = kInf;
        state = ST_CLOSEST;
        if (COUNT) c_bounce++;
      } else {
        // radiance sanitising of SamplerIntegrator::Render, then FilmTile::AddSample (box filter)
        const float y = (0.212671f * L.x + 0.715160f * L.y) + 0.072169f * L.z;
        if (isnan(L.x) || isnan(L.y) || isnan(L.z) || y < -1e-5f || isinf(y)) L = {0.f, 0.f, 0.f};
        sum = sum + L;
        s++;
        state = ST_NEW;
      }
      cont = false;
    }
    }  // state != ST_DONE
  }

  if (valid) {
    // Film::merge_film_tile (core/film.rs:313-326): xyz = rgb_to_xyz(contrib_sum), weight = spp
    float4 o;
    o.x = 0.412453f * sum.x + 0.357580f * sum.y + 0.180423f * sum.z;
    o.y = 0.212671f * sum.x + 0.715160f * sum.y + 0.072169f * sum.z;
    o.z = 0.019334f * sum.x + 0.119193f * sum.y + 0.950227f * sum.z;
    o.w = (float)spp;
    R.slab[(size_t)jsup * 4096u + pys * 64u + pxs] = o;
  }
  if (COUNT) {
    unsigned long long v[5] = {c_cam, c_bounce, c_shadow, c_nodes, c_tris};
    for (int i = 0; i < 5; i++) {
      unsigned long long x = v[i];
      for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
      if (lane == 0) atomicAdd(&R.counters[i], x);
    }
  }
}

template <bool SPH, bool COUNT, int STACK>
__global__ void __launch_bounds__(256) intersect_kernel(const DevScene S, const RayBatch B, const int any_hit) {
  __shared__ uint32_t lds_stack[4][STACK][64];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t *stk = &lds_stack[wave][0][lane];
  unsigned long long cn = 0, ct = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < B.n; i += (int64_t)gridDim.x * 256) {
    const V3 o = {B.o[3 * i], B.o[3 * i + 1], B.o[3 * i + 2]};
    const V3 d = {B.d[3 * i], B.d[3 * i + 1], B.d[3 * i + 2]};
    HitRec h;
    const bool occ = traverse<SPH, COUNT>(S, o, d, B.tmax[i], any_hit != 0, stk, h, cn, ct);
    if (any_hit) {
      B.occluded[i] = occ ? 1 : 0;
    } else {
      B.t[i] = h.t;
      B.prim[i] = h.prim;
      B.b1[i] = h.b1;
      B.b2[i] = h.b2;
    }
  }
  if (COUNT) {
    for (int off = 32; off > 0; off >>= 1) {
      cn += __shfl_down(cn, off, 64);
      ct += __shfl_down(ct, off, 64);
    }
    if (lane == 0) {
      atomicAdd(&B.counters[0], cn);
      atomicAdd(&B.counters[1], ct);
    }
  }
}

__global__ void pack_tris_kernel(const float *P, const uint32_t *idx, const uint16_t *mat_id, const uint32_t *order,
                                 uint32_t n_tris, float4 *tris) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= n_tris) return;
  const uint32_t t = order[slot];
  const uint32_t i0 = idx[3 * t], i1 = idx[3 * t + 1], i2 = idx[3 * t + 2];
  tris[3 * slot] = make_float4(P[3 * i0], P[3 * i0 + 1], P[3 * i0 + 2], __uint_as_float(t));
  tris[3 * slot + 1] = make_float4(P[3 * i1], P[3 * i1 + 1], P[3 * i1 + 2], __uint_as_float((uint32_t)mat_id[t]));
  tris[3 * slot + 2] = make_float4(P[3 * i2], P[3 * i2 + 1], P[3 * i2 + 2], 0.f);
}

__global__ void assemble_kernel(const float4 *slab, float4 *film, int32_t w, int32_t h, uint32_t rank, uint32_t world,
                                uint32_t n_local_super) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_local_super * 4096u) return;
  const uint32_t j = i >> 12, pys = (i >> 6) & 63u, pxs = i & 63u;
  const uint32_t stx = (uint32_t)(w + 63) >> 6;
  const uint32_t t = rank + j * world;
  const int32_t x = (int32_t)((t % stx) * 64u + pxs), y = (int32_t)((t / stx) * 64u + pys);
  if (x < w && y < h) film[(size_t)y * w + x] = slab[i];
}

}  // namespace

template <bool SPH, bool COUNT>
static hipError_t launch_render_t(const DevScene &S, const RenderParams &R, uint32_t n_local_super, uint32_t depth,
                                  hipStream_t st) {
  const dim3 grid(n_local_super * 16u), block(256);
  if (depth <= 32) hipLaunchKernelGGL((render_kernel<SPH, COUNT, 32>), grid, block, 0, st, S, R);
  else hipLaunchKernelGGL((render_kernel<SPH, COUNT, 64>), grid, block, 0, st, S, R);
  return hipGetLastError();
}

hipError_t launch_render(const DevScene &S, const RenderParams &R, uint32_t n_local_super, uint32_t bvh_depth,
                         bool counters, hipStream_t stream) {
  if (n_local_super == 0) return hipSuccess;
  const bool sph = S.n_spheres > 0;
  if (sph) return counters ? launch_render_t<true, true>(S, R, n_local_super, bvh_depth, stream)
                           : launch_render_t<true, false>(S, R, n_local_super, bvh_depth, stream);
  return counters ? launch_render_t<false, true>(S, R, n_local_super, bvh_depth, stream)
                  : launch_render_t<false, false>(S, R, n_local_super, bvh_depth, stream);
}

template <bool SPH, bool COUNT>
static hipError_t launch_intersect_t(const DevScene &S, const RayBatch &B, bool any_hit, uint32_t depth, hipStream_t st) {
  int64_t blocks = (B.n + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  const dim3 grid((uint32_t)blocks), block(256);
  if (depth <= 32) hipLaunchKernelGGL((intersect_kernel<SPH, COUNT, 32>), grid, block, 0, st, S, B, any_hit ? 1 : 0);
  else hipLaunchKernelGGL((intersect_kernel<SPH, COUNT, 64>), grid, block, 0, st, S, B, any_hit ? 1 : 0);
  return hipGetLastError();
}

hipError_t launch_intersect(const DevScene &S, const RayBatch &B, bool any_hit, uint32_t bvh_depth, hipStream_t stream) {
  if (B.n == 0) return hipSuccess;
  const bool sph = S.n_spheres > 0, cnt = B.counters != nullptr;
  if (sph) return cnt ? launch_intersect_t<true, true>(S, B, any_hit, bvh_depth, stream)
                      : launch_intersect_t<true, false>(S, B, any_hit, bvh_depth, stream);
  return cnt ? launch_intersect_t<false, true>(S, B, any_hit, bvh_depth, stream)
             : launch_intersect_t<false, false>(S, B, any_hit, bvh_depth, stream);
}

hipError_t launch_pack_tris(const float *P, const uint32_t *idx, const uint16_t *mat_id, const uint32_t *order,
                            uint32_t n_tris, float4 *tris, hipStream_t stream) {
  if (n_tris == 0) return hipSuccess;
  hipLaunchKernelGGL(pack_tris_kernel, dim3((n_tris + 255) / 256), dim3(256), 0, stream, P, idx, mat_id, order, n_tris,
                     tris);
  return hipGetLastError();
}

hipError_t launch_assemble(const float4 *slab, float4 *film, int32_t w, int32_t h, uint32_t rank, uint32_t world,
                           uint32_t n_local_super, hipStream_t stream) {
  if (n_local_super == 0) return hipSuccess;
  const uint32_t n = n_local_super * 4096u;
  hipLaunchKernelGGL(assemble_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, slab, film, w, h, rank, world,
                     n_local_super);
  return hipGetLastError();
}

}  // namespace pbrt_hip
