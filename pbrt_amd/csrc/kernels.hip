// kernels.hip -- the CDNA4 (gfx950) kernels of the render path.
//
//   render_kernel    persistent one-wave workgroups (as many as the device holds at once); every lane draws a
//                    pixel from the rank's pixel list (XCD-aware hand-out, one atomic per wave and round), keeps
//                    it for all its samples, in order: stratified camera sample -> BVH closest hit -> emission
//                    -> one-light direct estimate (any-hit shadow ray) -> BSDF sample -> Russian roulette; then
//                    writes the film pixel and draws the next one.  The wave walks a quantised 4-wide BVH in a
//                    "while-while" loop with parked leaves; `__ballot` + popcount take every scheduling decision
//                    wave-uniformly.  Per-lane traversal stack in LDS, laid out stack[level][lane]
//                    (conflict-free: lane l -> bank l); path state parked in coalesced HBM records.
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

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "device_types.h"

// node steps of the production walk between two scheduling checks (trav_run)
#ifndef PBRT_STEPS_PER_CHECK
#define PBRT_STEPS_PER_CHECK 3
#endif

namespace pbrt_hip {
namespace {

constexpr float kInf = __builtin_huge_valf();
constexpr float kRayTMin = 1e-4f;
constexpr float kSpawnEps = 1e-4f;
constexpr float kShadowShrink = 0.9999f;
constexpr float kBoxPad = 0x1.000004p+0f;  // 1 + 2^-19: the node test's far-side pad (pbrt-v3 pads by 1 + 2 gamma(3) = 1 + 6 * 2^-24)
constexpr float kOwnPad = 0x1.000001p+0f;  // 1 + 2^-21: the own-box rule's pad (3.5), strictly inside kBoxPad
constexpr float kInvPi = 0.31830988618379067154f;
constexpr float kPiOver4 = 0.78539816339744830961f;
constexpr float kOneMinusEps = 0x1.fffffcp-1f;  // 1 - f32::EPSILON = 1 - 2^-23, core/rng.rs:19 (NOT pbrt-v3's 1 - 2^-24)
constexpr uint32_t kNoPrim = 0xffffffffu;

typedef float f32x2 __attribute__((ext_vector_type(2)));

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

// ---- a sphere's (u, v) for 2-D textures (pbrt-v3 Sphere::Intersect: u = phi / 2 pi with phi = atan2(y, x) in [0, 2 pi),
// v = (theta - pi) / (0 - pi) with theta = acos(z)) on the unit normal n = (p - c) / r, the sphere's own frame being the world's axes.  atan
// and asin are the Cephes single-precision polynomials written out (|error| < 2e-7), the same operations on CPU and GPU. ----
__device__ __forceinline__ float poly_atan_pos(float x) {  // x >= 0 (+inf included): atan(x) in [0, pi / 2]
  float y0 = 0.f;
  if (x > 2.414213562373095f) {
    y0 = 1.5707963267948966f;
    x = -(1.0f / x);
  } else if (x > 0.4142135623730950f) {
    y0 = 0.7853981633974483f;
    x = (x - 1.0f) / (x + 1.0f);
  }
  const float z = x * x;
  float p = 8.05374449538e-2f * z - 1.38776856032e-1f;
  p = p * z + 1.99777106478e-1f;
  p = p * z - 3.33329491539e-1f;
  return y0 + ((p * z) * x + x);
}
__device__ __forceinline__ float poly_asin_small(float a) {  // |a| <= 0.5
  const float z = a * a;
  float p = 4.2163199048e-2f * z + 2.4181311049e-2f;
  p = p * z + 4.5470025998e-2f;
  p = p * z + 7.4953002686e-2f;
  p = p * z + 1.6666752422e-1f;
  return (p * z) * a + a;
}
__device__ __forceinline__ float poly_acos(float x) {  // x in [-1, 1]
  if (x < -0.5f) return 3.14159265358979323846f - 2.0f * poly_asin_small(sqrtf(0.5f * (1.0f + x)));
  if (x > 0.5f) return 2.0f * poly_asin_small(sqrtf(0.5f * (1.0f - x)));
  return 1.5707963267948966f - poly_asin_small(x);
}
__device__ __forceinline__ void sphere_uv(float nx, float ny, float nz, float *u, float *v) {
  const float ax = fabsf(nx), ay = fabsf(ny);
  float phi = (ax == 0.f && ay == 0.f) ? 0.f : poly_atan_pos(ay / ax);  // first quadrant (ax == 0: atan(+inf) = pi / 2)
  if (nx < 0.f) phi = 3.14159265358979323846f - phi;
  if (ny < 0.f) phi = 6.28318530717958647692f - phi;
  const float zc = nz < -1.0f ? -1.0f : (nz > 1.0f ? 1.0f : nz);
  const float theta = poly_acos(zc);
  *u = phi * 0.15915494309189533577f;  // 1 / (2 pi)
  *v = (theta - 3.14159265358979323846f) / (0.f - 3.14159265358979323846f);
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

__device__ __forceinline__ bool sphere_hit(const float4 cr, V3 o, V3 d, float tmax, float &th) {  // cr = {centre, radius}
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

constexpr uint32_t kDone = 0xffffffffu;
constexpr uint32_t kLeafRef = 0x80000000u;  // ref = kLeafRef | n_prims << 24 | first slot (n_prims <= 64)

// The production walk addresses its LDS stack by 32-bit LDS byte addresses kept in a register (one v_add per push, no
// shift / or to form an address: tools/ubench/valu_issue.hip shows v_lshl_or_b32 and friends issue at half rate).
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ uint32_t lds_addr(const uint32_t *p) { return (uint32_t)(uintptr_t)(const lds_u32 *)p; }
__device__ __forceinline__ void lds_store(uint32_t a, uint32_t v) { *(lds_u32 *)(uintptr_t)a = v; }
__device__ __forceinline__ uint32_t lds_load(uint32_t a) { return *(const lds_u32 *)(uintptr_t)a; }
constexpr uint32_t kRowBytes = 256u;  // one stack row = 64 lanes x 4 bytes: consecutive entries of a lane are one row apart
// row of the entry at LDS address `a` of the stack whose lane column starts at `stk`: measured from the ARRAY's base, a
// link-time constant, so that no per-lane limit has to be kept in a register (the lane's 4-byte column offset is below a row)
// the lane number where it is needed only on a rare path: opaque to the optimiser, so that nothing derived from it is
// hoisted out of the kernel's loop into a register that lives for the whole kernel
__device__ __forceinline__ uint32_t lane_here() { uint32_t l = threadIdx.x & 63u; asm volatile("" : "+v"(l)); return l; }
__device__ __forceinline__ uint32_t lds_row(uint32_t a, const uint32_t *stk) { return (a - lds_addr(stk - (threadIdx.x & 63u))) / kRowBytes; }

// Per-lane traversal state.  It lives in registers across iterations of the kernels' outer loops,
// so a lane can be suspended in the middle of a walk while other lanes of the wave are served.
struct Trav {
  V3 o, d;
  float tmax;
  uint32_t cur;       // ref to process next: interior = step, leaf = the lane is PARKED there; kDone = walk over
  uint32_t sp;        // exact walk: stack entries in use; production walk: LDS byte address of the lane's first free entry
  uint32_t any;       // bit 0: any-hit (shadow) ray; bit 1: its result, occluded
  HitRec h;           // closest-hit result
};

struct TravTuning {
  uint32_t min_walkers;  // leave the loop when fewer lanes are walking and some lane waits for service
  uint32_t min_parked;   // test triangles once this many lanes are parked at a leaf
};

// s_setprio (A-B: -DPBRT_NO_PRIO compiles the priorities out; the levels can be overridden)
#ifndef PBRT_PRIO_FETCH
#define PBRT_PRIO_FETCH 3
#endif
#ifndef PBRT_PRIO_ARITH
#define PBRT_PRIO_ARITH 0
#endif
#ifndef PBRT_PRIO_SERVICE
#define PBRT_PRIO_SERVICE 1
#endif
__device__ __forceinline__ void wave_prio(int p) {
#ifndef PBRT_NO_PRIO
  switch (p) {  // (the builtin wants a constant)
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
  }
#endif
}

// The slab test of DESIGN.md 3.4 against [kRayTMin, tfar]: near / far plane per axis by the sign of
// the inverse direction; fmin / fmax ignore a 0 * inf = NaN (conservative); far side padded.
__device__ __forceinline__ bool box_test(float lx, float ly, float lz, float hx, float hy, float hz, V3 o, V3 inv,
                                         bool negx, bool negy, bool negz, float tfar, float &tn) {
  const float nx = ((negx ? hx : lx) - o.x) * inv.x, fx = ((negx ? lx : hx) - o.x) * inv.x;
  const float ny = ((negy ? hy : ly) - o.y) * inv.y, fy = ((negy ? ly : hy) - o.y) * inv.y;
  const float nz = ((negz ? hz : lz) - o.z) * inv.z, fz = ((negz ? lz : hz) - o.z) * inv.z;
  tn = fmaxf(fmaxf(nx, ny), fmaxf(nz, kRayTMin));
  const float tf = fminf(fminf(fx, fy), fminf(fz, tfar));
  return tn <= tf * kBoxPad;
}

// Next node from the stack.  EXACT (the counting instantiation): every entry carries the entry
// distance tn of its box (NaN if the box already failed when it was pushed); the pop counts the node
// as visited and re-tests `tn <= tfar * pad`, which is equivalent to the oracle's slab test of the
// popped node with the current tfar (the far-plane part of that test can only have loosened).
// Otherwise entries are bare refs and a popped node is simply processed (a superset walk).
template <bool EXACT, uint32_t OVFR>  // OVFR: rows of the LDS part when deeper entries go to HBM (overflow variant), else 0
__device__ __forceinline__ uint32_t trav_pop(Trav &T, uint32_t *stk, float *stkt, uint32_t *ovf, unsigned long long &cn) {
  if (EXACT) {
    while (T.sp != 0u) {
      T.sp--;
      const uint32_t ref = stk[T.sp * 64u];
      const float tn = stkt[T.sp * 64u];
      cn++;  // EXACT implies counting
      if (tn <= fminf(T.h.t, T.tmax) * kBoxPad) return ref;
    }
    return kDone;
  }
  // production walk: entry 0 is the sentinel kDone (trav_begin), so a pop needs no emptiness test
  T.sp -= kRowBytes;
  if (OVFR == 0u) return lds_load(T.sp);
  const uint32_t e = lds_row(T.sp, stk);
  return e < OVFR - 1u ? lds_load(T.sp) : ovf[(e - (OVFR - 1u)) * 64u + lane_here()];
}
// production walk: the LDS part of a lane's stack has OVFR rows = OVFR - 1 entries (the
// sentinel first) + one scratch row, which the branch-free pushes below write when they do not push.  Only for
// trees whose worst-case bound exceeds that (OVF) do the deeper entries go to a per-lane HBM area.
template <uint32_t OVFR>
__device__ __forceinline__ void trav_push(Trav &T, uint32_t *stk, uint32_t *ovf, uint32_t ref) {
  if (OVFR == 0u) {
    lds_store(T.sp, ref);
  } else {
    const uint32_t e = lds_row(T.sp, stk);
    if (e < OVFR - 1u) lds_store(T.sp, ref);
    else ovf[(e - (OVFR - 1u)) * 64u + lane_here()] = ref;
  }
  T.sp += kRowBytes;
}

__device__ __forceinline__ void trav_enter(Trav &T, uint32_t ref) { T.cur = ref; }
__device__ __forceinline__ bool trav_parked(const Trav &T) { return T.cur != kDone && (T.cur & kLeafRef) != 0u; }
__device__ __forceinline__ uint32_t trav_leaf_cnt(const Trav &T) { return trav_parked(T) ? (T.cur >> 24) & 0x7fu : 0u; }

// Measurement aids (phase probe, ray log, per-pixel trace, A-B sensitivity loads / instructions, measured-negative
// variants kept for the record) live in experiments.inc and exist only in builds made with one of its switches
// (PBRT_PHASE_PROBE, PBRT_RAY_LOG, PBRT_DEBUG_PIXEL_X/Y, PBRT_EXTRA_VALU, PBRT_EXTRA_LOADS, PBRT_PREFETCH_POP: tools/README.md);
// in the product build every hook below expands to nothing.
#include "experiments.inc"
template <bool EXACT>
__device__ __forceinline__ void trav_begin(const DevScene &S, Trav &T, uint32_t *stk, V3 o, V3 d, float tmax, bool any,
                                           unsigned long long &cn) {
  T.o = o;
  T.d = d;
  T.tmax = tmax;
  T.sp = 0;
  if (!EXACT) {  // production walk: entry 0 is a sentinel, so that popping needs no emptiness test
    lds_store(lds_addr(stk), kDone);
    T.sp = lds_addr(stk) + kRowBytes;
  }
  T.any = any ? 1u : 0u;
  T.h.t = kInf;
  T.h.prim = kNoPrim;
  T.h.slot = kNoPrim;
  T.h.b1 = 0.f;
  T.h.b2 = 0.f;
  T.cur = kDone;
  if (S.n_nodes) {  // the root is the one node whose box is not held by a parent
    if (EXACT) cn++;
    // production walk: a ray that starts inside the root box needs no test (entering a node the ray might miss is
    // always allowed in a superset walk), and bounce / shadow rays always do: the three divisions are skipped
    if (!EXACT && o.x >= S.root_lo[0] && o.x <= S.root_hi[0] && o.y >= S.root_lo[1] && o.y <= S.root_hi[1] && o.z >= S.root_lo[2] &&
        o.z <= S.root_hi[2]) {
      trav_enter(T, !(S.root_ref & kLeafRef) ? 0u : S.root_ref);
      return;
    }
    const V3 inv = {1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    float tn;
    if (box_test(S.root_lo[0], S.root_lo[1], S.root_lo[2], S.root_hi[0], S.root_hi[1], S.root_hi[2], o, inv,
                 inv.x < 0.f, inv.y < 0.f, inv.z < 0.f, tmax, tn))
      trav_enter(T, (!EXACT && !(S.root_ref & kLeafRef)) ? 0u : S.root_ref);
  }
}

// The traversal loop of a whole wave ("while-while" with parked leaves) over the child-pair nodes.
// Every lane walks its own ray through the binary tree of DESIGN.md 3.3 in the order of 3.4 (near
// child by the sign of the split axis first, far child pushed); one step fetches ONE 64-byte record
// and tests BOTH children of an interior node, leaves are never fetched (their ref holds slot and
// count).  With EXACT the sequence of nodes visited and triangles tested is the oracle's, counter
// for counter; without it the far child is dropped at once if its box fails and is not re-tested
// when popped -- a superset walk whose RESULT is identical because of the tie rule (lower primitive
// id wins at equal t).  What is scheduling, and never changes a lane's arithmetic:
//   * a lane that reaches a leaf PARKS there; the wave tests triangles (one per parked lane per
//     pass) only when `min_parked` lanes are parked or nobody can step, so the long
//     Moeller-Trumbore body runs with many lanes instead of one or two;
//   * the loop EXITS when no lane walks, or when fewer than `min_walkers` do and some lane whose
//     walk is over is waiting to be served (shade / regenerate / fetch the next ray); walking
//     lanes keep their state and resume on the next call.
// `__ballot` + popcount make both decisions wave-uniform.  `alive`: this lane has work for the
// caller once its walk is over.
// SPH: the scene has spheres -- leaf records flagged as such (pack_tris_kernel: word 3 of the third float4) take the sphere test
template <bool EXACT, bool COUNT, uint32_t OVFR, int STEPS = PBRT_STEPS_PER_CHECK, bool SPH = false>
__device__ __forceinline__ void trav_run(const DevScene &S, Trav &T, uint32_t *stk, float *stkt, uint32_t *ovf,
                                         const bool alive, const TravTuning tune, unsigned long long &cn,
                                         unsigned long long &ct) {
  const V3 o = T.o, d = T.d;
  const V3 inv1 = {1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
  const bool negx = inv1.x < 0.f, negy = inv1.y < 0.f, negz = inv1.z < 0.f;
  const uint32_t negbits = (negx ? 1u : 0u) | (negy ? 2u : 0u) | (negz ? 4u : 0u);
  // Production walk: a ray PARALLEL to a slab (d exactly 0 on that axis; 1 / d = +-inf) multiplies by the scene's huge finite power of
  // two instead (device_types.h inv_parallel, host_math.hpp): with inf every quantised plane's t = q * inf - inf is NaN, the axis drops
  // out of the slab test and a ray along an axis -- every shadow ray towards a sun straight overhead -- walks the whole tree.  The
  // canonical walk (EXACT) keeps 1 / 0 = inf: its planes are real floats ((lo - o) * inf is +-inf with the right sign) and its visit
  // counters are the oracle's.
#ifdef PBRT_INV_INF  // A-B switch: the walk as it was before the stand-in (tools/experiments/README.md, round 5)
  const V3 inv = inv1;
#else
  const V3 inv = EXACT ? inv1
                       : V3{d.x == 0.f ? copysignf(S.inv_parallel, inv1.x) : inv1.x, d.y == 0.f ? copysignf(S.inv_parallel, inv1.y) : inv1.y,
                            d.z == 0.f ? copysignf(S.inv_parallel, inv1.z) : inv1.z};
#endif
  const char *nodes = reinterpret_cast<const char *>(S.nodes);
  const char *quads = reinterpret_cast<const char *>(S.quads);
  const char *tris = reinterpret_cast<const char *>(S.tris);
  for (;;) {
    const bool walking = T.cur != kDone;
    const unsigned long long mwalk = __ballot(walking);
    if (mwalk == 0ull) break;
    // (min_walkers is meant for a full wave: it scales with the lanes that still have work at all, so that a wave whose
    // pixel list has run dry does not visit the service stage for every single ray)
    if ((uint32_t)__popcll(mwalk) * 64u < tune.min_walkers * (uint32_t)__popcll(__ballot(alive)) && __ballot(!walking && alive) != 0ull) break;

    if (EXACT && walking && !trav_parked(T)) {
      // ---- one step: both children of interior node T.cur ----
      const uint32_t off = T.cur * 64u;
      const uint4 q0 = *reinterpret_cast<const uint4 *>(nodes + off);
      const uint4 q1 = *reinterpret_cast<const uint4 *>(nodes + off + 16u);
      const uint4 q2 = *reinterpret_cast<const uint4 *>(nodes + off + 32u);
      const uint4 q3 = *reinterpret_cast<const uint4 *>(nodes + off + 48u);
      const float tfar = fminf(T.h.t, T.tmax);
      // The slab test of DESIGN.md 3.4 for child 0 and child 1 side by side: element 0 / 1 of each
      // float2 belongs to child 0 / 1, so the six subtractions and six multiplications of the two
      // boxes are six packed instructions (v_pk_add_f32 / v_pk_mul_f32, IEEE per element: the
      // same bits as the scalar form).
      const f32x2 lx = {__uint_as_float(q0.x), __uint_as_float(q1.z)}, hx = {__uint_as_float(q0.w), __uint_as_float(q2.y)};
      const f32x2 ly = {__uint_as_float(q0.y), __uint_as_float(q1.w)}, hy = {__uint_as_float(q1.x), __uint_as_float(q2.z)};
      const f32x2 lz = {__uint_as_float(q0.z), __uint_as_float(q2.x)}, hz = {__uint_as_float(q1.y), __uint_as_float(q2.w)};
      const f32x2 nx = ((negx ? hx : lx) - o.x) * inv.x, fx = ((negx ? lx : hx) - o.x) * inv.x;
      const f32x2 ny = ((negy ? hy : ly) - o.y) * inv.y, fy = ((negy ? ly : hy) - o.y) * inv.y;
      const f32x2 nz = ((negz ? hz : lz) - o.z) * inv.z, fz = ((negz ? lz : hz) - o.z) * inv.z;
      const float tn0 = fmaxf(fmaxf(nx.x, ny.x), fmaxf(nz.x, kRayTMin));
      const float tn1 = fmaxf(fmaxf(nx.y, ny.y), fmaxf(nz.y, kRayTMin));
      const float tf0 = fminf(fminf(fx.x, fy.x), fminf(fz.x, tfar));
      const float tf1 = fminf(fminf(fx.y, fy.y), fminf(fz.y, tfar));
      const bool hit0 = tn0 <= tf0 * kBoxPad, hit1 = tn1 <= tf1 * kBoxPad;
      const bool far_first = ((negbits >> q3.z) & 1u) != 0u;  // child 1 is the near one
      const uint32_t ref_near = far_first ? q3.y : q3.x, ref_far = far_first ? q3.x : q3.y;
      const bool hit_near = (far_first && hit1) || (!far_first && hit0);
      const bool hit_far = (far_first && hit0) || (!far_first && hit1);
      if (EXACT) {
        cn++;  // the near child is visited now; the far one when it is popped
        stk[T.sp * 64u] = ref_far;
        stkt[T.sp * 64u] = hit_far ? (far_first ? tn0 : tn1) : __builtin_nanf("");
        T.sp++;
      }
      trav_enter(T, hit_near ? ref_near : trav_pop<EXACT, OVFR>(T, stk, stkt, ovf, cn));
    }

    // production walk: STEPS node steps between two scheduling checks (a lane that parks or
    // finishes in the first one idles through the rest; the checks cost about a fifth of a step)
#pragma unroll
    for (int rep = 0; !EXACT && rep < STEPS; rep++) {
    EXP_PROBE_LANES(0, T.cur != kDone && !trav_parked(T));
    if (T.cur != kDone && !trav_parked(T)) {
      // ---- one step of the production walk: the four children of quantised quad node T.cur (64 bytes) ----
      const uint32_t off = T.cur;  // the ref of an interior quad node IS its byte offset (node number x 64)
      // Wave priority (s_setprio; the SIMD's arbiter picks the ready wave of highest priority, the oldest among equals): 3
      // while a step issues its node fetch, 0 for the arithmetic on the node -- a wave that is about to wait ~700 cycles for
      // its next node gets its loads out before the other waves' decode and slab tests.  With the same around the leaf
      // pass's triangle fetch and 1 for the service stage: C3 +3.5 %, C2 +0.7 % (tools/experiments/README.md).
      wave_prio(PBRT_PRIO_FETCH);
      const uint4 W0 = EXP_NODE_LOAD(reinterpret_cast<const uint4 *>(quads + off));
      const uint4 W1 = EXP_NODE_LOAD(reinterpret_cast<const uint4 *>(quads + off + 16u));
      const uint4 W2 = EXP_NODE_LOAD(reinterpret_cast<const uint4 *>(quads + off + 32u));
      const uint4 W3 = EXP_NODE_LOAD(reinterpret_cast<const uint4 *>(quads + off + 48u));
      EXP_STEP_EXTRA_LOADS(quads, off, T);
      wave_prio(PBRT_PRIO_ARITH);
      if (COUNT) cn++;  // one 64-byte fetch
      const float tfar = fminf(T.h.t, T.tmax);
      EXP_STEP_EXTRA_VALU(W0, tfar, T);
      // Node-relative slab test.  A decoded plane is the REAL number origin + q * cell (the builder
      // checks in exact arithmetic that these planes enclose the true box), so
      //     t = (origin + q*cell - o) * inv = q * (cell*inv) - (o - origin)*inv = fma(q, ci, -gi):
      // one cvt + one fma per plane.  ci = cell * inv is exact (cell is a power of two); the two
      // roundings inside gi = fl(fl(o - origin) * inv) are absolute errors <= 2 eps |gi| in t, covered by
      // the margin m = 3 eps |gi|: near planes subtract gi + m, far planes gi - m, so every computed
      // t_near / t_far lies outside the true one and the walk stays a superset of the exact walk; the
      // fma's own relative rounding is what kBoxPad of DESIGN.md 3.4 is for (1 + 2^-19 since round 6).  A ray
      // parallel to the slab (d = 0) has the scene's finite stand-in for 1 / 0 in inv (above): t is then
      // negative huge or positive huge by the side of the plane the origin is on; an inv that is infinite
      // because d is a denormal yields NaN or +-inf, which fmin / fmax ignore or keep conservative.
      const float gx = (o.x - __uint_as_float(W0.x)) * inv.x, gy = (o.y - __uint_as_float(W0.y)) * inv.y;
      const float gz = (o.z - __uint_as_float(W0.z)) * inv.z;
      // g +- 3 eps |g| as one fma each (|x| and -x are operand modifiers): rounded once instead of twice, at least
      // g +- 5/2 eps |g|, still beyond the 2 eps |g| the margin has to cover
      constexpr float kMargin = 0x1.8p-22f;
      const float gxn = __builtin_fmaf(fabsf(gx), kMargin, gx), gxf = __builtin_fmaf(-fabsf(gx), kMargin, gx);  // near, far: subtracted below
      const float gyn = __builtin_fmaf(fabsf(gy), kMargin, gy), gyf = __builtin_fmaf(-fabsf(gy), kMargin, gy);
      const float gzn = __builtin_fmaf(fabsf(gz), kMargin, gz), gzf = __builtin_fmaf(-fabsf(gz), kMargin, gz);
      const float cix = __uint_as_float(W0.w) * inv.x, ciy = __uint_as_float(W2.z) * inv.y, ciz = __uint_as_float(W2.w) * inv.z;
      // near / far planes by the sign of the inverse direction: one select per axis serves all four
      // children (a dword holds the four children's bytes of one plane)
      const uint32_t bnx = negx ? W1.w : W1.x, bfx = negx ? W1.x : W1.w;
      const uint32_t bny = negy ? W2.x : W1.y, bfy = negy ? W1.y : W2.x;
      const uint32_t bnz = negz ? W2.y : W1.z, bfz = negz ? W1.z : W2.y;
      // How the 24 plane fmas are issued.  A gfx950 SIMD issues, per quad-cycle, one instruction of any kind plus one of
      // the "simple" class (fma, mul, add, logic, right shift, mov) from another wave; a packed instruction goes alone
      // (tools/ubench/valu_pairing.hip, profiles/r03w_valu_pairing_ubench.txt).  This step has three converts / min /
      // max / compares / selects for every simple instruction, so 24 scalar v_fma_f32 ride along with them where 12
      // v_pk_fma_f32 take 12 quad-cycles of their own -- but they are 12 more instructions for a wave that issues one
      // every ~5 cycles at best.  Measured (profiles/r03w_scalar_fma_ab.txt): trees that sit in L2 (no wait to hide: C4
      // +4.7 %, C2 +1.6 %) gain from the scalar form; the deep trees of the overflow variant, whose waves spend their time
      // waiting for nodes, lose (C3 -0.8 %, 12 M triangles -6.5 %) and keep the packed one ({near, far} of a child side by
      // side; IEEE per element: the same bits either way).
      constexpr bool kScalarFma = OVFR == 0u;
      const f32x2 gxx = {gxn, gxf}, gyy = {gyn, gyf}, gzz = {gzn, gzf}, cxx = {cix, cix}, cyy = {ciy, ciy}, czz = {ciz, ciz};
      float key[4];
      bool hit[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const float qxn = (float)((bnx >> (8 * k)) & 0xffu), qxf = (float)((bfx >> (8 * k)) & 0xffu);
        const float qyn = (float)((bny >> (8 * k)) & 0xffu), qyf = (float)((bfy >> (8 * k)) & 0xffu);
        const float qzn = (float)((bnz >> (8 * k)) & 0xffu), qzf = (float)((bfz >> (8 * k)) & 0xffu);
        f32x2 tx, ty, tz;  // {near, far}
        if (kScalarFma) {
          tx = f32x2{__builtin_fmaf(qxn, cix, -gxn), __builtin_fmaf(qxf, cix, -gxf)};
          ty = f32x2{__builtin_fmaf(qyn, ciy, -gyn), __builtin_fmaf(qyf, ciy, -gyf)};
          tz = f32x2{__builtin_fmaf(qzn, ciz, -gzn), __builtin_fmaf(qzf, ciz, -gzf)};
        } else {
          tx = __builtin_elementwise_fma(f32x2{qxn, qxf}, cxx, -gxx);
          ty = __builtin_elementwise_fma(f32x2{qyn, qyf}, cyy, -gyy);
          tz = __builtin_elementwise_fma(f32x2{qzn, qzf}, czz, -gzz);
        }
        const float tn = fmaxf(fmaxf(tx.x, ty.x), fmaxf(tz.x, kRayTMin));
        const float tf = fminf(fminf(tx.y, ty.y), fminf(tz.y, tfar));
        hit[k] = tn <= tf * kBoxPad;
        key[k] = tn;
      }
      // (an unused child slot holds kEmptyLeafRef behind an inverted box: if a degenerate ray gets through that box the
      // lane parks at a leaf without triangles and pops -- no test for it here)
#ifdef PBRT_PRIO_SELECT  // (A-B: raised priority from the child selection on)
      wave_prio(PBRT_PRIO_SELECT);
#endif
      // The nearest child hit is entered, the other hit ones are stacked in slot order.  Order affects only
      // speed (tie rule of 3.4) -- but a lot: visiting the hit children in slot order alone costs C3 49 node steps per
      // ray instead of 41 (measured, r02), and sorting the stacked ones cost more than it saved (r01).
#pragma unroll
      // (a missed child's key is a NaN with all bits set -- an inline constant of the select, where +inf would need a
      // register; fminf ignores it, and when every child is missed nothing below uses kmin)
      for (int k = 0; k < 4; k++) key[k] = hit[k] ? key[k] : __uint_as_float(0xffffffffu);
      const float kmin = fminf(fminf(key[0], key[1]), fminf(key[2], key[3]));
      const bool n0 = key[0] == kmin, n1 = !n0 && key[1] == kmin, n2 = !n0 && !n1 && key[2] == kmin;
      const bool n3 = !n0 && !n1 && !n2;
      const bool any_hit = hit[0] || hit[1] || hit[2] || hit[3];
      const uint32_t nearest = n0 ? W3.x : (n1 ? W3.y : (n2 ? W3.z : W3.w));
      // Overflow variant (trees whose worst-case stack bound exceeds the LDS part): one wave-uniform test per step -- is
      // any lane within four rows of the end of its LDS part? -- picks the slow form with predicated pushes that go
      // to HBM beyond it; stacks rarely get that deep, so nearly every step takes the branch-free form below.  (One
      // compare against a constant: the stack array's base is a link-time constant, the lane's column offset is
      // smaller than a row.  A test per batch of steps instead, with a threshold three times as far from the end,
      // measured 1.5 % slower.  __builtin_expect moves the slow form out of line: the fast form falls through, +1.2 %.)
      if (OVFR != 0u && __builtin_expect(__ballot(T.sp >= lds_addr(stk - (threadIdx.x & 63u)) + (OVFR - 4u) * kRowBytes) != 0ull, 0)) {
        if (hit[3] && !n3) trav_push<OVFR>(T, stk, ovf, W3.w);
        if (hit[2] && !n2) trav_push<OVFR>(T, stk, ovf, W3.z);
        if (hit[1] && !n1) trav_push<OVFR>(T, stk, ovf, W3.y);
        if (hit[0] && !n0) trav_push<OVFR>(T, stk, ovf, W3.x);
        trav_enter(T, any_hit ? nearest : trav_pop<false, OVFR>(T, stk, stkt, ovf, cn));
      } else {
        // branch-free: each ref is written above the stack top in any case (one LDS row beyond the entries is
        // scratch) and the top advances by the hit mask; entry 0 is the sentinel kDone, so the entry below the top
        // can be read in any case.  T.sp is the LDS ADDRESS of the top: a push is one ds_write + one v_add, no
        // address arithmetic (v_lshl_or_b32 and the other three-operand integer forms issue at half rate on gfx950).
        // the entry below the top is read BEFORE the pushes (a lane that pops has pushed nothing in this step): the read
        // does not wait behind four writes, and the next node's address is known that much earlier
        const uint32_t below = T.sp - kRowBytes, top = lds_load(below);
        const uint32_t next = any_hit ? nearest : top;
        lds_store(T.sp, W3.w); T.sp += (hit[3] && !n3) ? kRowBytes : 0u;
        lds_store(T.sp, W3.z); T.sp += (hit[2] && !n2) ? kRowBytes : 0u;
        lds_store(T.sp, W3.y); T.sp += (hit[1] && !n1) ? kRowBytes : 0u;
        lds_store(T.sp, W3.x); T.sp += (hit[0] && !n0) ? kRowBytes : 0u;
        T.sp = any_hit ? T.sp : below;
        trav_enter(T, next);
      }
    }
    }

    // ---- leaf flush (wave-uniform decision) ----
    const bool parked = trav_parked(T);
    const unsigned long long mleaf = __ballot(parked);
    if (mleaf != 0ull &&
        ((uint32_t)__popcll(mleaf) >= tune.min_parked ||
         // ... or when the parked lanes are at least half as many as the lanes that can still step (with few
         // steppers left, waiting for min_parked only idles the parked ones; this also covers "nobody can step")
         (uint32_t)__popcll(mleaf) * 2u >= (uint32_t)__popcll(__ballot(T.cur != kDone && !parked)))) {
      const uint32_t cnt = parked ? (T.cur >> 24) & 0x7fu : 0u, first = T.cur & 0xffffffu;
      bool stop = false;  // any-hit ray found its hit
      EXP_LEAF_PREFETCH_BEGIN(parked, quads, T);
      EXP_PROBE_FLUSH(cnt);
      for (uint32_t i = 0;; i++) {
        if (__ballot(cnt > i && !stop) == 0ull) break;
        EXP_PROBE_LANES(2, cnt > i && !stop);
        if (cnt > i && !stop) {
          const uint32_t slot = first + i;
          wave_prio(PBRT_PRIO_FETCH);  // (as for the node fetch)
          const float4 a = EXP_TRI_LOAD(reinterpret_cast<const float4 *>(tris + slot * (16u * kTriStride)));
          const float4 b = EXP_TRI_LOAD(reinterpret_cast<const float4 *>(tris + slot * (16u * kTriStride) + 16u));
          const float4 c = EXP_TRI_LOAD(reinterpret_cast<const float4 *>(tris + slot * (16u * kTriStride) + 32u));
          wave_prio(PBRT_PRIO_ARITH);
          if (COUNT) ct++;
              // Moeller-Trumbore, operation order of DESIGN.md 3.5
          const V3 p0 = xyz(a);
          const V3 e1 = xyz(b) - p0, e2 = xyz(c) - p0;
          const V3 pv = cross(d, e2);
          const float det = dot(e1, pv);
          // Branch-free from here: every lane of the pass computes u, v and t (a degenerate triangle's 1 / det is inf or
          // NaN and fails the tests below like any miss) and the hit record is updated by selects.  The nested early-outs
          // this replaces skipped work only when ALL lanes of the pass failed the same test, and the compiler paid for
          // them with copies of the six hit-record registers at every level (about 50 v_mov per pass).
          const float idet = 1.0f / det;
          const V3 tv = o - p0;
          const float u = dot(tv, pv) * idet;
          const V3 qv = cross(tv, e1);
          const float v = dot(d, qv) * idet;
          const float th = dot(e2, qv) * idet;
          // The own-box rule (round 6): the ray must MEET the triangle's own box -- the node test of 3.4 on it: slab distances
          // of the three vertices with the TRUE 1 / d (two roundings each, as the canonical node test; p0 - o = -tv exactly), their min / max
          // per axis, pad kOwnPad < kBoxPad -- and the hit's distance is at least the box's entry: t = max(th, entry).  Monotone arithmetic:
          // every enclosing box of every tree then passes its own test while the walk's best hit is still >= t, so an accepted hit is reached
          // by every walk and a hit is a function of (ray, triangle) alone.  (Raising t instead of rejecting: a triangle flat in an axis plane
          // has entry = exit = the plane's slab distance, which Moeller-Trumbore's t misses by rounding.)
#ifdef PBRT_NO_OWN_BOX_RULE  // A-B switch: the leaf pass as it was until round 5 (what the rule costs; films differ where it rejects)
          const bool in_own_box = true;
          const float tsnap = th;
#else
          const V3 w1 = xyz(b) - o, w2 = xyz(c) - o;
          const float x0 = (-tv.x) * inv1.x, x1 = w1.x * inv1.x, x2 = w2.x * inv1.x;
          const float y0 = (-tv.y) * inv1.y, y1 = w1.y * inv1.y, y2 = w2.y * inv1.y;
          const float z0 = (-tv.z) * inv1.z, z1 = w1.z * inv1.z, z2 = w2.z * inv1.z;
          const float otn = fmaxf(fmaxf(fminf(fminf(x0, x1), x2), fminf(fminf(y0, y1), y2)), fmaxf(fminf(fminf(z0, z1), z2), kRayTMin));
          const float otf = fminf(fminf(fmaxf(fmaxf(x0, x1), x2), fmaxf(fmaxf(y0, y1), y2)), fmaxf(fmaxf(z0, z1), z2));
          const bool in_own_box = otn <= otf * kOwnPad;
          const float tsnap = fmaxf(th, otn);
#endif
          bool valid = in_own_box && !(fabsf(det) < 1e-8f) && (u >= 0.f) && (v >= 0.f) && (u + v <= 1.0f) && (th > kRayTMin) && (tsnap < T.tmax);
          float ht = tsnap, hu = u, hv = v;
          if (SPH && __float_as_uint(c.w) != 0u) {
            // a SPHERE's record (round 6: spheres are primitives of the tree): {centre, primitive id}{radius, -, -, material}
            // {-, -, -, 1}.  Sphere::Intersect with the f64 quadratic of lib.rs:181-203, then the own-box rule on [c - r, c + r] (the
            // "vertices" lo, hi, lo) exactly as for a triangle.
            const float r = b.x;
            float ts = 0.f;
            valid = sphere_hit(make_float4(a.x, a.y, a.z, r), o, d, T.tmax, ts);
            const V3 lo = {a.x - r, a.y - r, a.z - r}, hi = {a.x + r, a.y + r, a.z + r};
            const float sx0 = (lo.x - o.x) * inv1.x, sx1 = (hi.x - o.x) * inv1.x, sy0 = (lo.y - o.y) * inv1.y, sy1 = (hi.y - o.y) * inv1.y;
            const float sz0 = (lo.z - o.z) * inv1.z, sz1 = (hi.z - o.z) * inv1.z;
            const float stn = fmaxf(fmaxf(fminf(fminf(sx0, sx1), sx0), fminf(fminf(sy0, sy1), sy0)), fmaxf(fminf(fminf(sz0, sz1), sz0), kRayTMin));
            const float stf = fminf(fminf(fmaxf(fmaxf(sx0, sx1), sx0), fmaxf(fmaxf(sy0, sy1), sy0)), fmaxf(fmaxf(sz0, sz1), sz0));
            ht = fmaxf(ts, stn);
            valid = valid && stn <= stf * kOwnPad && ht < T.tmax;
            hu = 0.f; hv = 0.f;
          }
          const uint32_t id = __float_as_uint(a.w);
          const bool occl = valid && T.any != 0u;  // any-hit ray: the walk ends at the first valid hit
          const bool closer = valid && T.any == 0u && (ht < T.h.t || (ht == T.h.t && id < T.h.prim));
          T.any = occl ? 3u : T.any;
          stop = stop || occl;
          T.h.t = closer ? ht : T.h.t;
          T.h.prim = closer ? id : T.h.prim;
          T.h.slot = closer ? slot : T.h.slot;
          T.h.b1 = closer ? hu : T.h.b1;
          T.h.b2 = closer ? hv : T.h.b2;
        }
      }
      EXP_LEAF_PREFETCH_END();
      // (OVF: is any entry about to be popped one of the rare ones beyond the LDS part?  wave-uniform, as for the pushes)
      const bool far_pop = OVFR != 0u && !EXACT &&
                           __ballot(parked && !stop && T.sp >= lds_addr(stk - (threadIdx.x & 63u)) + OVFR * kRowBytes) != 0ull;
      if (parked) {  // leave the leaf: the walk is over (any-hit found) or the next node comes off the stack
        if (stop) {
          T.cur = kDone;
          T.sp = 0u;
        } else if (OVFR != 0u && __builtin_expect(far_pop, 0)) {
          trav_enter(T, trav_pop<EXACT, OVFR>(T, stk, stkt, ovf, cn));
        } else {
          trav_enter(T, trav_pop<EXACT, 0u>(T, stk, stkt, ovf, cn));
        }
      }
    }
  }
}

// One light of UniformSampleOneLight.  false: geometry rules the light out.
// mis: the estimate weighted with the power heuristic pl^2 / (pl^2 + pb^2), pl = this strategy's density for the
// direction (light picked with 1 / nL), pb = cos / pi the BSDF's; delta lights keep weight 1
__device__ __forceinline__ bool sample_light(const DevScene &S, uint32_t li, V3 po, V3 nf, V3 kd, float u1, float u2,
                                             float nLf, V3 &Ld, V3 &wi, float &tmax, const bool mis = false) {
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
    if (mis) Ld = Ld * (1.0f / (1.0f + nLf * nLf));  // pl = pb / nL
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
    if (mis) {
      const float pl = (dist2 / (cl * l1.w)) / nLf, pbl = cs * kInvPi;
      scale = scale * ((pl * pl) / (pl * pl + pbl * pbl));
    }
    Ld = (f * lc) * scale;
    tmax = dist * kShadowShrink;
    return true;
  }
}

enum : uint32_t { ST_NEW = 0, ST_CLOSEST = 1, ST_SHADOW = 2, ST_DONE = 3, ST_FETCH = 4 };

// waves per SIMD the register allocator must leave room for (launch_bounds' 2nd argument)
#ifndef PBRT_RENDER_WAVES_PER_SIMD
#define PBRT_RENDER_WAVES_PER_SIMD 5
#endif

// Path state of one work item (a CHUNK of a pixel's samples) while its lane is busy walking the BVH:
// five 16-byte records per lane in HBM, laid out [record][lane] per wave so that a wave's access is one coalesced
// 1 KB transaction.  It is loaded and stored only in the service stage (once per ray, against ~41 gather steps),
// which keeps these 20 dwords out of the registers that are live across the traversal loop.
struct PathState {
  V3 L, beta;  // radiance and throughput of the sample in flight
  V3 wi_next;  // prepared bounce direction (taken after the shadow ray returns)
  Pcg rng;     // stratified sampler: rng.inc is recomputed from the item, only the state is stored.  Sobol sampler:
               // rng.state = the pixel's scramble key | the request counter of the sample in flight << 32
  uint32_t s, bounces;
  bool specular, cont;
};
// Records 0..2 hold what every visit of the service stage needs; record 3 (beta * Ld of the light sample, added if the
// shadow ray comes back unoccluded) and record 4 (the chunk's partial film sum so far) are read and written only where
// they are used -- by the lanes whose ray was a shadow ray, and once per finished sample -- and never sit in registers
// beside the shading arithmetic (r01 loaded all five on every visit: 6 more live VGPRs, 40 % more record traffic).
// The lane's five records lie 1 KB apart around a wave-uniform base that points at record 2: -2048 ... +2048 bytes, all
// within the immediate offset of a global load / store.  The address is formed at each access from the uniform base (an
// SGPR pair) and the lane's 32-bit byte offset, which is made opaque so that base + offset is not hoisted out of the
// kernel's loop as a 64-bit per-lane pointer: one long-lived VGPR instead of the four the compiler kept (a pointer pair for
// records 0..3 and a second one for record 4, which was out of immediate range from record 0).
struct LaneRecords {
  char *base;    // wave-uniform: record 2 of lane 0
  uint32_t off;  // lane * 16
};
constexpr int32_t kRecL = -2048, kRecBeta = -1024, kRecWi = 0, kRecLpend = 1024, kRecSum = 2048;  // byte offsets
__device__ __forceinline__ float4 rec_load(const LaneRecords &r, int32_t k) {
  uint32_t o = r.off;
  asm volatile("" : "+v"(o));
  return *reinterpret_cast<const float4 *>(r.base + o + k);
}
__device__ __forceinline__ void rec_store(const LaneRecords &r, int32_t k, float4 v) {
  uint32_t o = r.off;
  asm volatile("" : "+v"(o));
  *reinterpret_cast<float4 *>(r.base + o + k) = v;
}
__device__ __forceinline__ void path_store(const LaneRecords &rec, const PathState &P) {
  rec_store(rec, kRecL, make_float4(P.L.x, P.L.y, P.L.z,
                                    __uint_as_float(P.s | (P.bounces << 20) | (P.specular ? 1u << 30 : 0u) | (P.cont ? 1u << 31 : 0u))));
  rec_store(rec, kRecBeta, make_float4(P.beta.x, P.beta.y, P.beta.z, __uint_as_float((uint32_t)P.rng.state)));
  rec_store(rec, kRecWi, make_float4(P.wi_next.x, P.wi_next.y, P.wi_next.z, __uint_as_float((uint32_t)(P.rng.state >> 32))));
}
__device__ __forceinline__ void path_load(const LaneRecords &rec, PathState &P) {
  const float4 a = rec_load(rec, kRecL), b = rec_load(rec, kRecBeta), c = rec_load(rec, kRecWi);
  P.L = {a.x, a.y, a.z};
  P.beta = {b.x, b.y, b.z};
  P.wi_next = {c.x, c.y, c.z};
  const uint32_t w = __float_as_uint(a.w);
  P.s = w & 0xfffffu;
  P.bounces = (w >> 20) & 0x3ffu;
  P.specular = (w >> 30) & 1u;
  P.cont = (w >> 31) & 1u;
  P.rng.state = (uint64_t)__float_as_uint(b.w) | ((uint64_t)__float_as_uint(c.w) << 32);
}

// ---- samplers (DESIGN.md 3.1 stratified, 3.10 padded (0,2)-sequence) ----
// (K = 2^kb chunks per pixel, kb = RenderParams::chunk_shift: device_types.h sample_chunk_shift)
__device__ __forceinline__ uint32_t chunk_begin(uint32_t c, uint32_t spp, uint32_t kb) { return (c * spp) >> kb; }  // spp <= 2^20, c <= 16
__device__ __forceinline__ uint32_t mix32(uint32_t v) {  // lowbias32
  v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16;
  return v;
}
// One 2-D request of the sample in flight.  Sobol: point (s ^ mask_j) of the first two Sobol' dimensions -- the
// van der Corput sequence (bit reversal) and the dimension whose generator matrix has the columns v, v ^ v >> 1, ...
// (Joe-Kuo s = 1, a = 0, m = 1) -- XOR-scrambled with keys hashed from the pixel and the request number j.
// SND (sampler 2): requests 0 .. 15 of a sample take their own Sobol' dimensions (2j, 2j + 1) from the
// generator matrices in `mat` at point index s, XOR-scrambled per dimension; later requests are the padded ones below.
// Sampler 3: dimension d of the Halton sampler at point index i under the pixel's key: the radical inverse of i in
// base b = the d-th prime, the D digits the frame's largest sample index can have (b^D > spp_mask) each scrambled by a random linear
// bijection of Z_b, all higher digits -- zeros for every sample of the frame -- as one random tail.  tab = {b, K, ceil(2^32 / b), bits of
// 1 / b^K} (host_math.hpp halton_table; b and the reciprocal are used): n / b by the reciprocal, the estimate is the quotient or one more.
__device__ __forceinline__ float halton_dim(const uint32_t *tab, uint32_t d, uint32_t i, uint32_t key, uint32_t spp_mask) {
  const uint4 t = reinterpret_cast<const uint4 *>(tab)[d];
  const uint32_t b = t.x;
  uint32_t h = mix32(key + (d + 1u) * 0x9e3779b9u);
  if (b == 2u) return fminf(kOneMinusEps, (float)(__builtin_bitreverse32(i) ^ h) * 2.3283064365386963e-10f);
  uint32_t v = 0u, n = i, pw = 1u;
  do {
    pw *= b;
    uint32_t q = __umulhi(n, t.z);
    if (q * b > n) q--;
    const uint32_t a = n - q * b;
    n = q;
    h = h * 0x9e3779b1u + 0x7f4a7c15u;
    const uint32_t w = a * (1u + (((h >> 16) * (b - 1u)) >> 16)) + (((h & 0xffffu) * b) >> 16);  // a m + c < b^2
    uint32_t wq = __umulhi(w, t.z);
    if (wq * b > w) wq--;
    v = v * b + (w - wq * b);
  } while (pw <= spp_mask);
  h = h * 0x9e3779b1u + 0x7f4a7c15u;
  return fminf(kOneMinusEps, ((float)v + (float)h * 2.3283064365386963e-10f) * (1.0f / (float)pw));
}
// HAL (with SND): the table sampler in use is the Halton one (sampler 3), `mat` its table
template <bool SND = false>
__device__ __forceinline__ void sample_2d(PathState &P, const bool sobol, const uint32_t spp_mask, float &u1, float &u2, const uint32_t *mat = nullptr,
                                          const bool halton = false) {
  if (!sobol) {
    u1 = pcg_float(P.rng);
    u2 = pcg_float(P.rng);
    return;
  }
  if (SND && halton && (uint32_t)(P.rng.state >> 32) < kSobolNdRequests) {
    const uint32_t key = (uint32_t)P.rng.state, d0 = 2u * (uint32_t)(P.rng.state >> 32);
    P.rng.state += 1ull << 32;  // next request
    u1 = halton_dim(mat, d0, P.s, key, spp_mask);
    u2 = halton_dim(mat, d0 + 1u, P.s, key, spp_mask);
    return;
  }
  if (SND && (uint32_t)(P.rng.state >> 32) < kSobolNdRequests) {
    const uint32_t key = (uint32_t)P.rng.state, d0 = 2u * (uint32_t)(P.rng.state >> 32);
    P.rng.state += 1ull << 32;  // next request
    // x = XOR of the columns of dimension d0's matrix at the set bits of the sample index, y likewise for d0 + 1.  Branch-free
    // over the bits a sample index of this frame can have (wave-uniform count), four columns per 16-byte load: the loads of a
    // request are in flight together, where a loop over the set bits waited for two dependent loads per bit (r03: sampler 2
    // at 64 spp 369 -> 4xx Msamples/s, profiles/r03z_variant_throughput.txt)
    const uint4 *m0 = reinterpret_cast<const uint4 *>(mat + d0 * 32u);
    uint32_t x = 0u, y = 0u;
    const uint32_t nq = ((uint32_t)__popc(spp_mask) + 3u) >> 2;  // groups of four bits below 2^ceil(log2 spp)
    for (uint32_t q = 0u, k = P.s; q < nq; q++, k >>= 4) {
      const uint4 cx = m0[q], cy = m0[8u + q];
      x ^= (cx.x & (0u - (k & 1u))) ^ (cx.y & (0u - ((k >> 1) & 1u))) ^ (cx.z & (0u - ((k >> 2) & 1u))) ^ (cx.w & (0u - ((k >> 3) & 1u)));
      y ^= (cy.x & (0u - (k & 1u))) ^ (cy.y & (0u - ((k >> 1) & 1u))) ^ (cy.z & (0u - ((k >> 2) & 1u))) ^ (cy.w & (0u - ((k >> 3) & 1u)));
    }
    x ^= mix32(key + (d0 + 1u) * 0x9e3779b9u);
    y ^= mix32(key + (d0 + 2u) * 0x9e3779b9u);
    u1 = fminf(kOneMinusEps, (float)x * 2.3283064365386963e-10f);
    u2 = fminf(kOneMinusEps, (float)y * 2.3283064365386963e-10f);
    return;
  }
  const uint32_t a = mix32((uint32_t)P.rng.state + (uint32_t)(P.rng.state >> 32) * 0x9e3779b9u);
  P.rng.state += 1ull << 32;  // next request
  const uint32_t i = P.s ^ (a & spp_mask);
  uint32_t x = __builtin_bitreverse32(i), y = 0u;
  for (uint32_t k = i, v = 0x80000000u; k != 0u; k >>= 1, v ^= v >> 1)
    if (k & 1u) y ^= v;
  x ^= mix32(a ^ 0x68e31da4u);
  y ^= mix32(a ^ 0xb5297a4du);
  u1 = fminf(kOneMinusEps, (float)x * 2.3283064365386963e-10f);
  u2 = fminf(kOneMinusEps, (float)y * 2.3283064365386963e-10f);
}
template <bool SND = false>
__device__ __forceinline__ float sample_1d(PathState &P, const bool sobol, const uint32_t spp_mask, const uint32_t *mat = nullptr, const bool halton = false) {
  if (!sobol) return pcg_float(P.rng);
  if (SND && halton && (uint32_t)(P.rng.state >> 32) < kSobolNdRequests) {  // (a 1-D request takes the first coordinate of its pair)
    const float u = halton_dim(mat, 2u * (uint32_t)(P.rng.state >> 32), P.s, (uint32_t)P.rng.state, spp_mask);
    P.rng.state += 1ull << 32;
    return u;
  }
  float u1, u2;
  sample_2d<SND>(P, true, spp_mask, u1, u2, mat, halton);
  return u1;
}

// COUNT: accumulate ray / visit counters.  EXACT (needs COUNT): walk the tree in exactly the oracle's
// order so that the counters are the oracle's; COUNT without EXACT counts the production walk itself
// (64-byte fetches and triangle tests).
// One workgroup = one wavefront = one 8x8 pixel tile; 64 workgroups per 64x64 super-tile.
// STACK: LDS entries of the exact walk; for the production walk the LDS rows of the overflow variant (deeper entries
// in the HBM overflow area), or 0 = the whole stack in LDS.
// WIDE: a box filter radius other than 0.5: a sample is added to every pixel within the radius, into
// fixed-point accumulators with atomics, instead of to its chunk's partial sum.
// SND: sampler 2, the Sobol' sampler with its own dimensions per request (3.12): generator-matrix lookups in the service stage.
// (Both variants keep the default path's register budget: 5 waves per SIMD, no spill -- tests/test_host.py.)
// Wide box filter: 16 footprint slots per lane, two float4 records each ({r, g} and {b, samples, footprint}), a slot's records
// of the 64 lanes side by side: 16 x 2 x 64 float4 per one-wave workgroup (RenderParams::wide_slots)
constexpr uint32_t kWideSlotFloat4 = 16u * 2u * 64u;
__device__ __forceinline__ void wide_slots_clear(float4 *slots, uint32_t lane) {
  uint32_t lo = lane * 16u;
  asm volatile("" : "+v"(lo));
  char *lb = reinterpret_cast<char *>(slots + (size_t)blockIdx.x * kWideSlotFloat4) + lo;
#pragma nounroll
  for (uint32_t sl = 0u; sl < 16u; sl++) *reinterpret_cast<float4 *>(lb + sl * 2048u + 1024u) = make_float4(0.f, 0.f, 0.f, 0.f);
}
// `n` samples of summed fixed-point radiance (r, g, b) to every pixel [x0, x1) x [y0, y1) of the cropped window
__device__ __forceinline__ void film_add(unsigned long long *acc, const DevScene &S, int32_t x0, int32_t x1, int32_t y0, int32_t y1,
                                         unsigned long long r, unsigned long long g, unsigned long long b, uint32_t n) {
  for (int32_t py = y0; py < y1; py++)
    for (int32_t px = x0; px < x1; px++) {
      unsigned long long *a = acc + 4u * ((size_t)(py - S.cy0) * (size_t)(S.cx1 - S.cx0) + (size_t)(px - S.cx0));
      atomicAdd(a, r); atomicAdd(a + 1, g); atomicAdd(a + 2, b); atomicAdd(a + 3, (unsigned long long)n);
    }
}

template <bool SPH, bool COUNT, bool EXACT, int STACK, int STEPS = PBRT_STEPS_PER_CHECK, bool WIDE = false, bool SND = false>
// (scenes with spheres -- C0 / C1: a handful of primitives, nothing to gain from occupancy -- get the register budget
// of 3 waves per SIMD: the f64 quadratic of lib.rs:181-203 does not fit 128 VGPRs beside the path state)
__global__ void __launch_bounds__(64, (COUNT ? 1 : (SPH ? 3 : PBRT_RENDER_WAVES_PER_SIMD))) render_kernel(const DevScene S, const RenderParams R) {
  constexpr bool MIS = false, TEX = false;  // (the variants: render_kernel_x below)
  (void)MIS;
#include "render_body.inc"
}
// The variants of the path that BASELINE's configs do not use, in a kernel of their own name so that the instantiations above keep
// theirs (and their machine code): MIS = multiple importance sampling of the direct-light estimate, TEX = materials
// whose Kd is a checkerboard texture (3.15) -- and every combination of the two with the table samplers (SND: 3.12, 3.13) and a box
// filter radius other than 0.5 (WIDE: 3.11), which render_kernel instantiates one at a time.  No counters.
template <bool SPH, int STACK, bool MIS, bool TEX, bool SND, bool WIDE>
__global__ void __launch_bounds__(64, (SPH ? 3 : PBRT_RENDER_WAVES_PER_SIMD)) render_kernel_x(const DevScene S, const RenderParams R) {
  constexpr bool COUNT = false, EXACT = false;
  constexpr int STEPS = PBRT_STEPS_PER_CHECK;
#include "render_body.inc"
}

#ifndef PBRT_INTERSECT_WAVES_PER_SIMD
#define PBRT_INTERSECT_WAVES_PER_SIMD 8
#endif
// The traversal loop alone over a ray batch, as persistent waves with dynamic fetch: a lane whose
// walk is over writes its result and pulls its next ray while the other lanes keep walking.
template <bool SPH, bool COUNT, int STACK>
// (scenes with spheres: the f64 quadratic of the leaf pass does not fit the 64 VGPRs of eight waves per SIMD -- four, as few rays as such scenes trace)
__global__ void __launch_bounds__(256, (COUNT ? 1 : (SPH ? 4 : PBRT_INTERSECT_WAVES_PER_SIMD))) intersect_kernel(const DevScene S, const RayBatch B, const int any_hit) {
  __shared__ uint32_t lds_stack[4][COUNT ? STACK : kIntersectLdsStack][64];
  __shared__ float lds_tn[COUNT ? 4 : 1][COUNT ? STACK : 1][64];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t *stk = &lds_stack[wave][0][lane];
  float *stkt = &lds_tn[COUNT ? wave : 0][0][lane];
  uint32_t *ovf = B.stack_overflow + ((size_t)blockIdx.x * 4u + (uint32_t)__builtin_amdgcn_readfirstlane(wave)) * B.stack_overflow_entries * 64u;  // wave-uniform
  const TravTuning tune = {B.min_walkers, B.min_parked};
  unsigned long long cn = 0, ct = 0;
  EXP_PROBE_INIT_BLOCK();
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t next = (int64_t)blockIdx.x * 256 + threadIdx.x, idx = 0;
  bool have = false;
  Trav T;
  T.o = mk(0.f, 0.f, 0.f);
  T.d = mk(0.f, 0.f, 1.f);
  T.tmax = 0.f;
  T.cur = kDone;
  T.sp = 0;
  T.any = 0;
  T.h = HitRec{kInf, kNoPrim, kNoPrim, 0.f, 0.f};
  for (;;) {
    if (T.cur == kDone) {
      if (have) {
        if (any_hit) {
          B.occluded[idx] = T.any == 3u ? 1 : 0;
        } else {
          B.t[idx] = T.h.t;
          B.prim[idx] = T.h.prim;
          B.b1[idx] = T.h.b1;
          B.b2[idx] = T.h.b2;
        }
        have = false;
      }
      if (next < B.n) {
        idx = next;
        next += stride;
        const V3 ro = mk(B.o[3 * idx], B.o[3 * idx + 1], B.o[3 * idx + 2]), rd = mk(B.d[3 * idx], B.d[3 * idx + 1], B.d[3 * idx + 2]);
        const float rt = B.tmax[idx];
        // a probe ray with a non-finite origin / direction or a NaN tmax hits nothing and is not walked (with NaN slabs nothing prunes: such
        // a ray would visit every node of the tree; oracle.cpp probe_ray_is_sane is the same rule) -- the renderer's own rays are finite
        const bool sane = fabsf(ro.x) < kInf && fabsf(ro.y) < kInf && fabsf(ro.z) < kInf && fabsf(rd.x) < kInf && fabsf(rd.y) < kInf && fabsf(rd.z) < kInf && rt == rt;
        if (sane) {
          trav_begin<COUNT>(S, T, stk, ro, rd, rt, any_hit != 0, cn);
        } else {  // the miss is written when the loop comes round
          T.o = mk(0.f, 0.f, 0.f);
          T.d = mk(0.f, 0.f, 1.f);
          T.tmax = 0.f;
          T.any = 0;
          T.h = HitRec{kInf, kNoPrim, kNoPrim, 0.f, 0.f};
          T.cur = kDone;
        }
        have = true;
      }
    }
    if (__ballot(have) == 0ull) break;
    trav_run<COUNT, COUNT, (COUNT ? 0u : kIntersectLdsStack), PBRT_STEPS_PER_CHECK, SPH>(S, T, stk, stkt, ovf, have, tune, cn, ct);
  }
  EXP_PROBE_FINI_BLOCK();
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

// n_prims = n_tris + the spheres: primitive t >= n_tris is sphere t - n_tris (its place in the vertex / index buffers is taken by a
// degenerate proxy triangle spanning its box, so that every builder bounds it: capi.cpp) and gets a sphere's record
__global__ void pack_tris_kernel(const float *P, const uint32_t *idx, const uint16_t *mat_id, const uint32_t *order,
                                 uint32_t n_prims, uint32_t n_tris, const float4 *spheres, float4 *tris) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= n_prims) return;
  const uint32_t t = order[slot];
  if (t >= n_tris) {
    const float4 cr = spheres[2 * (t - n_tris)], m = spheres[2 * (t - n_tris) + 1];
    tris[kTriStride * slot] = make_float4(cr.x, cr.y, cr.z, __uint_as_float(t));
    tris[kTriStride * slot + 1] = make_float4(cr.w, 0.f, 0.f, m.x);
    tris[kTriStride * slot + 2] = make_float4(0.f, 0.f, 0.f, __uint_as_float(1u));
    return;
  }
  const uint32_t i0 = idx[3 * t], i1 = idx[3 * t + 1], i2 = idx[3 * t + 2];
  tris[kTriStride * slot] = make_float4(P[3 * i0], P[3 * i0 + 1], P[3 * i0 + 2], __uint_as_float(t));
  tris[kTriStride * slot + 1] = make_float4(P[3 * i1], P[3 * i1 + 1], P[3 * i1 + 2], __uint_as_float((uint32_t)mat_id[t]));
  tris[kTriStride * slot + 2] = make_float4(P[3 * i2], P[3 * i2 + 1], P[3 * i2 + 2], 0.f);
}

// corner (u, v) of the triangle in leaf slot `slot` (textured scenes: DESIGN.md 3.15)
__global__ void pack_uv_kernel(const float *tri_uv, const uint32_t *order, uint32_t n_prims, uint32_t n_tris, float2 *out) {
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= n_prims) return;
  const uint32_t t = order[slot];
  for (int v = 0; v < 3; v++) out[3u * slot + v] = t < n_tris ? make_float2(tri_uv[6 * (size_t)t + 2 * v], tri_uv[6 * (size_t)t + 2 * v + 1]) : make_float2(0.f, 0.f);  // (a sphere has its own (u, v))
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

// Film::merge_film_tile (core/film.rs:313-326) for one pixel of a rank's slab: contrib_sum = the K = 2^kb partial sums of
// its chunks added in chunk order, xyz = rgb_to_xyz(contrib_sum) (spectrum.rs:139-145), weight = spp.
// (rank, world, n_local_super) are those of the LAUNCH that wrote `partials`: a frame rendered in P passes (capi.cpp partials_passes) hands the
// rank's super-tiles j = pass + P * j' to pass `pass`, which is rank + world * pass of world * P; its tile j' lands at slab tile j0 + jstride * j'.
__global__ void merge_kernel(const float4 *partials, float4 *slab, int32_t w, int32_t h, uint32_t rank, uint32_t world,
                             uint32_t n_local_super, float weight, uint32_t kb, uint32_t j0, uint32_t jstride) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_local_super * 4096u) return;
  const uint32_t j = i >> 12, pys = (i >> 6) & 63u, pxs = i & 63u;
  const uint32_t stx = (uint32_t)(w + 63) >> 6;
  const uint32_t t = rank + j * world;
  const int32_t x = (int32_t)((t % stx) * 64u + pxs), y = (int32_t)((t / stx) * 64u + pys);
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  if (x < w && y < h) {  // (pixels of a ragged super-tile outside the image were never rendered)
    V3 sum = {0.f, 0.f, 0.f};
    for (uint32_t c = 0; c < (1u << kb); c++) {
      const float4 p = partials[((size_t)i << kb) + c];
      sum = sum + mk(p.x, p.y, p.z);
    }
    o.x = 0.412453f * sum.x + 0.357580f * sum.y + 0.180423f * sum.z;
    o.y = 0.212671f * sum.x + 0.715160f * sum.y + 0.072169f * sum.z;
    o.z = 0.019334f * sum.x + 0.119193f * sum.y + 0.950227f * sum.z;
    o.w = weight;
  }
  slab[((size_t)(j0 + jstride * j) << 12) + (i & 4095u)] = o;
}

// fixed-point accumulators {r, g, b, samples} -> Film pixel {XYZ of the radiance sum, weight} (film.rs:313-326)
__global__ void film_from_acc_kernel(const unsigned long long *acc, float4 *film, size_t n_px) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_px) return;
  const float inv = 1.0f / kFixedOne;
  const V3 sum = {(float)(long long)acc[4 * i] * inv, (float)(long long)acc[4 * i + 1] * inv, (float)(long long)acc[4 * i + 2] * inv};
  float4 o;
  o.x = 0.412453f * sum.x + 0.357580f * sum.y + 0.180423f * sum.z;
  o.y = 0.212671f * sum.x + 0.715160f * sum.y + 0.072169f * sum.z;
  o.z = 0.019334f * sum.x + 0.119193f * sum.y + 0.950227f * sum.z;
  o.w = (float)(long long)acc[4 * i + 3];
  film[i] = o;
}

}  // namespace

// the fixed-point accumulators of two ranks on one device added (multi_gpu.cpp's loopback exchange; with one rank per device ncclReduce adds them)
__global__ void acc_add_kernel(unsigned long long *dst, const unsigned long long *src, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}
hipError_t launch_acc_add(unsigned long long *dst, const unsigned long long *src, size_t n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(acc_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dst, src, n);
  return hipGetLastError();
}

hipError_t launch_film_from_acc(const unsigned long long *acc, float4 *film, size_t n_px, hipStream_t stream) {
  if (n_px == 0) return hipSuccess;
  hipLaunchKernelGGL(film_from_acc_kernel, dim3((unsigned)((n_px + 255) / 256)), dim3(256), 0, stream, acc, film, n_px);
  return hipGetLastError();
}

// the variants of the production walk outside the default path (no counting instantiations): WIDE = a box filter radius
// other than 0.5 (3.11), SND = the Sobol' sampler with its own dimensions per request (3.12)
template <bool SPH, bool WIDE, bool SND>
static hipError_t launch_render_variant(const DevScene &S, const RenderParams &R, hipStream_t st) {
  const dim3 grid(R.n_workgroups), block(64);
  const RenderStackPlan plan = render_stack_plan(S.quad_stack_need, render_force_overflow(), render_prefer_lds());
  const uint32_t lds = plan.rows * 256u;
  if (plan.overflow && plan.rows == kQuadLdsStackOvfDeep && kQuadLdsStackOvfDeep != kQuadLdsStackOvf)
    hipLaunchKernelGGL((render_kernel<SPH, false, false, (int)kQuadLdsStackOvfDeep, PBRT_STEPS_PER_CHECK, WIDE, SND>), grid, block, lds, st, S, R);
  else if (plan.overflow) hipLaunchKernelGGL((render_kernel<SPH, false, false, (int)kQuadLdsStackOvf, PBRT_STEPS_PER_CHECK, WIDE, SND>), grid, block, lds, st, S, R);
  else hipLaunchKernelGGL((render_kernel<SPH, false, false, 0, PBRT_STEPS_PER_CHECK, WIDE, SND>), grid, block, lds, st, S, R);
  return hipGetLastError();
}

template <bool SPH, bool COUNT, bool EXACT>
static hipError_t launch_render_t(const DevScene &S, const RenderParams &R, uint32_t n_local_super, uint32_t depth,
                                  hipStream_t st) {
  const dim3 grid(R.n_workgroups), block(64);
  if constexpr (!EXACT) {  // production walk: LDS rows per scene; the overflow variant for deep quad trees
    // node steps per scheduling check: 3 for deep trees (C3 +1 %, C2 +2 % over 2), 2 for shallow ones whose walks
    // are a few steps long (C4: 3 would cost 5 %)
    const RenderStackPlan plan = render_stack_plan(S.quad_stack_need, render_force_overflow(), render_prefer_lds());
    const uint32_t lds = plan.rows * 256u;
    // (production walk: STACK = the LDS rows of the overflow variant, 0 = whole stack in LDS)
    if (plan.overflow && plan.rows == kQuadLdsStackOvfDeep && kQuadLdsStackOvfDeep != kQuadLdsStackOvf)
      hipLaunchKernelGGL((render_kernel<SPH, COUNT, false, (int)kQuadLdsStackOvfDeep>), grid, block, lds, st, S, R);
    else if (plan.overflow) hipLaunchKernelGGL((render_kernel<SPH, COUNT, false, (int)kQuadLdsStackOvf>), grid, block, lds, st, S, R);
    else if (!COUNT && S.quad_stack_need <= 16u) hipLaunchKernelGGL((render_kernel<SPH, COUNT, false, 0, 2>), grid, block, lds, st, S, R);
    else hipLaunchKernelGGL((render_kernel<SPH, COUNT, false, 0>), grid, block, lds, st, S, R);
    return hipGetLastError();
  } else {
    // exact walk: the LDS stack is sized to the tree: the walk holds at most depth - 1 entries (refs + entry distances)
    const uint32_t need = depth > 0 ? depth - 1 : 0;
    if (need > 40) hipLaunchKernelGGL((render_kernel<SPH, COUNT, EXACT, 64>), grid, block, 64 * 512, st, S, R);
    else if (need > 32) hipLaunchKernelGGL((render_kernel<SPH, COUNT, EXACT, 40>), grid, block, 40 * 512, st, S, R);
    else if (need > 26) hipLaunchKernelGGL((render_kernel<SPH, COUNT, EXACT, 32>), grid, block, 32 * 512, st, S, R);
    else if (need > 20) hipLaunchKernelGGL((render_kernel<SPH, COUNT, EXACT, 26>), grid, block, 26 * 512, st, S, R);
    else hipLaunchKernelGGL((render_kernel<SPH, COUNT, EXACT, 20>), grid, block, 20 * 512, st, S, R);
    return hipGetLastError();
  }
}

template <bool SPH, bool MIS, bool TEX, bool SND, bool WIDE>
static hipError_t launch_render_x(const DevScene &S, const RenderParams &R, hipStream_t st) {
  const dim3 grid(R.n_workgroups), block(64);
  const RenderStackPlan plan = render_stack_plan(S.quad_stack_need, render_force_overflow(), render_prefer_lds());
  const uint32_t lds = plan.rows * 256u;
  if (plan.overflow) hipLaunchKernelGGL((render_kernel_x<SPH, (int)kQuadLdsStackOvf, MIS, TEX, SND, WIDE>), grid, block, lds, st, S, R);
  else hipLaunchKernelGGL((render_kernel_x<SPH, 0, MIS, TEX, SND, WIDE>), grid, block, lds, st, S, R);
  return hipGetLastError();
}
// (mis, tex, snd, wide) -> the instantiation: every combination render_kernel does not cover itself
template <bool SPH, bool MIS, bool TEX>
static hipError_t launch_render_x_sw(const DevScene &S, const RenderParams &R, bool snd, bool wide, hipStream_t st) {
  if (snd) return wide ? launch_render_x<SPH, MIS, TEX, true, true>(S, R, st) : launch_render_x<SPH, MIS, TEX, true, false>(S, R, st);
  return wide ? launch_render_x<SPH, MIS, TEX, false, true>(S, R, st) : launch_render_x<SPH, MIS, TEX, false, false>(S, R, st);
}
template <bool SPH>
static hipError_t launch_render_x_pick(const DevScene &S, const RenderParams &R, bool mis, bool tex, bool snd, bool wide, hipStream_t st) {
  if (mis && tex) return launch_render_x_sw<SPH, true, true>(S, R, snd, wide, st);
  if (mis) return launch_render_x_sw<SPH, true, false>(S, R, snd, wide, st);
  if (tex) return launch_render_x_sw<SPH, false, true>(S, R, snd, wide, st);
  return launch_render_x<SPH, false, false, true, true>(S, R, st);  // (a table sampler under a wide filter: the one pair left)
}

hipError_t launch_render(const DevScene &S, const RenderParams &R, uint32_t n_local_super, uint32_t bvh_depth,
                         int counters, bool wide_filter, bool sobol_nd, hipStream_t stream, bool mis, bool textured) {
  if (n_local_super == 0) return hipSuccess;
  const bool sph = S.n_spheres > 0;
  if (mis || textured || (wide_filter && sobol_nd)) {  // the variants (render_kernel_x): no counters -- refused by check_render_desc
    if (counters != 0) return hipErrorInvalidValue;
    return sph ? launch_render_x_pick<true>(S, R, mis, textured, sobol_nd, wide_filter, stream)
               : launch_render_x_pick<false>(S, R, mis, textured, sobol_nd, wide_filter, stream);
  }
  if (wide_filter) return sph ? launch_render_variant<true, true, false>(S, R, stream) : launch_render_variant<false, true, false>(S, R, stream);
  if (sobol_nd) return sph ? launch_render_variant<true, false, true>(S, R, stream) : launch_render_variant<false, false, true>(S, R, stream);
  if (counters == 0 && kExperimentLaunch) {  // (ray log / phase probe builds: experiments.inc)
    hipError_t e = hipSuccess;
    if (experiment_launch_begin(&e)) return e;
    e = sph ? launch_render_t<true, false, false>(S, R, n_local_super, bvh_depth, stream)
            : launch_render_t<false, false, false>(S, R, n_local_super, bvh_depth, stream);
    experiment_launch_end(stream);
    return e;
  }
  if (counters == 1)
    return sph ? launch_render_t<true, true, true>(S, R, n_local_super, bvh_depth, stream)
               : launch_render_t<false, true, true>(S, R, n_local_super, bvh_depth, stream);
  if (counters == 2)
    return sph ? launch_render_t<true, true, false>(S, R, n_local_super, bvh_depth, stream)
               : launch_render_t<false, true, false>(S, R, n_local_super, bvh_depth, stream);
  return sph ? launch_render_t<true, false, false>(S, R, n_local_super, bvh_depth, stream)
             : launch_render_t<false, false, false>(S, R, n_local_super, bvh_depth, stream);
}

template <bool SPH, bool COUNT>
static hipError_t launch_intersect_t(const DevScene &S, const RayBatch &B, bool any_hit, uint32_t depth, hipStream_t st) {
  int64_t blocks = (B.n + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  const dim3 grid((uint32_t)blocks), block(256);
  if (depth <= 32) hipLaunchKernelGGL((intersect_kernel<SPH, COUNT, 32>), grid, block, 0, st, S, B, any_hit ? 1 : 0);
  else hipLaunchKernelGGL((intersect_kernel<SPH, COUNT, 64>), grid, block, 0, st, S, B, any_hit ? 1 : 0);
  if (kExperimentLaunch) experiment_launch_end(st);
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
                            uint32_t n_prims, uint32_t n_tris, const float4 *spheres, float4 *tris, hipStream_t stream) {
  if (n_prims == 0) return hipSuccess;
  hipLaunchKernelGGL(pack_tris_kernel, dim3((n_prims + 255) / 256), dim3(256), 0, stream, P, idx, mat_id, order, n_prims, n_tris, spheres,
                     tris);
  return hipGetLastError();
}

hipError_t launch_pack_uv(const float *tri_uv, const uint32_t *order, uint32_t n_prims, uint32_t n_tris, float2 *out, hipStream_t stream) {
  if (n_prims == 0) return hipSuccess;
  hipLaunchKernelGGL(pack_uv_kernel, dim3((n_prims + 255) / 256), dim3(256), 0, stream, tri_uv, order, n_prims, n_tris, out);
  return hipGetLastError();
}

hipError_t launch_merge(const float4 *partials, float4 *slab, int32_t w, int32_t h, uint32_t rank, uint32_t world,
                        uint32_t n_local_super, uint32_t spp, hipStream_t stream, uint32_t j0, uint32_t jstride) {
  if (n_local_super == 0) return hipSuccess;
  const uint32_t n = n_local_super * 4096u;
  hipLaunchKernelGGL(merge_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, partials, slab, w, h, rank, world, n_local_super,
                     (float)spp, sample_chunk_shift(spp), j0, jstride);
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
