// oracle_math.hpp -- scalar fp32 building blocks of the CPU ORACLE (test infrastructure only,
// see pbrt_oracle.h).  Everything here is +,-,*,/,sqrt and comparisons in a FIXED operation
// order; the library is compiled with -ffp-contract=off so that no
// fused multiply-add is formed and the HIP kernels can reproduce every bit.
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>

namespace orc {

// ---------------------------------------------------------------------------------------------
// Vectors.  Follows /root/reference/src/core/geometry/vector.rs: cross :314-324,
// length_squared :182-184 (x*x + y*y + z*z, left to right), normalize = self / length :165-167
// with a per-component division :210-220.
// ---------------------------------------------------------------------------------------------
struct Vec3 {
  float x, y, z;
  float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
static inline Vec3 v3(float x, float y, float z) { return Vec3{x, y, z}; }
static inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline Vec3 operator-(Vec3 a) { return {-a.x, -a.y, -a.z}; }
static inline Vec3 operator*(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static inline Vec3 operator*(Vec3 a, Vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
static inline Vec3 operator/(Vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
static inline float dot(Vec3 a, Vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline Vec3 cross(Vec3 a, Vec3 b) {
  return {(a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x)};
}
static inline float length(Vec3 a) { return std::sqrt(dot(a, a)); }
static inline Vec3 normalize(Vec3 a) { return a / length(a); }

static const float kPi = 3.14159265358979323846f;
static const float kInvPi = 0.31830988618379067154f;
static const float kPiOver4 = 0.78539816339744830961f;
static const float kInf = std::numeric_limits<float>::infinity();
static const float kRayTMin = 1e-4f;      // absolute ray t_min
static const float kSpawnEps = 1e-4f;     // spawned-ray offset along the facing normal
static const float kShadowShrink = 0.9999f;  // shadow ray tmax = dist * (1 - 1e-4)
static const float kBoxPad = 0x1.000004p+0f;  // 1 + 2^-19: the far-side pad of the node test (pbrt-v3 Bounds3::IntersectP pads by 1 + 2 gamma(3) = 1 + 6 * 2^-24;
                                              // 32 * 2^-24 here since round 6: room for the own-box rule's pad INSIDE it)
static const float kOwnPad = 0x1.000001p+0f;  // 1 + 2^-21: the pad of the own-box rule of Triangle::Intersect -- kOwnPad * (1 + 2^-21) < kBoxPad

// ---------------------------------------------------------------------------------------------
// PCG32.  Bit-exact restatement of /root/reference/src/core/rng.rs:19-93.
// ---------------------------------------------------------------------------------------------
struct Rng {
  uint64_t state = 0x853c49e6748fea9bULL;  // rng.rs:21
  uint64_t inc = 0xda3e39cb94b95bdbULL;    // rng.rs:22
  Rng() {}
  explicit Rng(uint64_t sequence_index) { set_sequence(sequence_index); }  // rng.rs:46-50
  void set_sequence(uint64_t sequence_index) {                             // rng.rs:53-59
    state = 0;
    inc = (sequence_index << 1) | 1;
    uniform_u32();
    state += 0x853c49e6748fea9bULL;
    uniform_u32();
  }
  uint32_t uniform_u32() {  // rng.rs:62-76
    uint64_t oldstate = state;
    state = oldstate * 0x5851f42d4c957f2dULL + inc;
    uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
    uint32_t rot = (uint32_t)(oldstate >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
  }
  uint32_t uniform_u32_threshold(uint32_t b) {  // rng.rs:79-87
    uint32_t threshold = (~b + 1u) % b;
    for (;;) {
      uint32_t r = uniform_u32();
      if (r >= threshold) return r % b;
    }
  }
  float uniform_float() {  // rng.rs:91-93: min(1 - eps, u32 * 2^-32)
    const float one_minus_eps = 1.0f - std::numeric_limits<float>::epsilon();
    float f = (float)uniform_u32() * 2.3283064365386963e-10f;
    return one_minus_eps < f ? one_minus_eps : f;
  }
};

// ---------------------------------------------------------------------------------------------
// Colour.  /root/reference/src/core/spectrum.rs:129-145 (constants and left-to-right sums).
// ---------------------------------------------------------------------------------------------
static inline void xyz_to_rgb(const float xyz[3], float rgb[3]) {
  rgb[0] = 3.240479f * xyz[0] - 1.537150f * xyz[1] - 0.498535f * xyz[2];
  rgb[1] = -0.969256f * xyz[0] + 1.875991f * xyz[1] + 0.041556f * xyz[2];
  rgb[2] = 0.055648f * xyz[0] - 0.204043f * xyz[1] + 1.057311f * xyz[2];
}
static inline void rgb_to_xyz(const float rgb[3], float xyz[3]) {
  xyz[0] = 0.412453f * rgb[0] + 0.357580f * rgb[1] + 0.180423f * rgb[2];
  xyz[1] = 0.212671f * rgb[0] + 0.715160f * rgb[1] + 0.072169f * rgb[2];
  xyz[2] = 0.019334f * rgb[0] + 0.119193f * rgb[1] + 0.950227f * rgb[2];
}

// /root/reference/src/lib.rs:93-99
static inline float gamma_correct(float value) {
  if (value <= 0.0031308f) return 12.92f * value;
  return 1.055f * std::pow(value, 1.0f / 2.4f) - 0.055f;
}
// /root/reference/src/core/imageio.rs:66-68 with lib.rs:115-126 clamp; `as u8` truncates
static inline uint8_t to_byte(float v) {
  float x = 255.0f * gamma_correct(v) + 0.5f;
  if (x < 0.0f) x = 0.0f; else if (x > 255.0f) x = 255.0f;
  return (uint8_t)x;
}

// /root/reference/src/lib.rs:181-203: f64 discriminant, ordered roots returned as f32
static inline bool quadratic(float af, float bf, float cf, float *t0, float *t1) {
  double a = af, b = bf, c = cf;
  double discrim = b * b - 4. * a * c;
  if (discrim < 0.) return false;
  double root_discrim = std::sqrt(discrim);
  double q = (b < 0.) ? -0.5 * (b - root_discrim) : -0.5 * (b + root_discrim);
  float r0 = (float)(q / a);
  float r1 = (float)(c / q);
  if (r0 > r1) { *t0 = r1; *t1 = r0; } else { *t0 = r0; *t1 = r1; }
  return true;
}

// ---------------------------------------------------------------------------------------------
// 4x4 matrices, row-major.  /root/reference/src/core/transform.rs:75-77, inverse :162-234,
// mul :270-282, look_at :485-520.
// ---------------------------------------------------------------------------------------------
struct Mat4 { float m[4][4]; };
static inline Mat4 identity() {
  Mat4 r{};
  for (int i = 0; i < 4; i++) r.m[i][i] = 1.f;
  return r;
}
static inline Mat4 transpose(const Mat4 &a) {  // transform.rs:129-139
  Mat4 r;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) r.m[i][j] = a.m[j][i];
  return r;
}
static inline Mat4 mul(const Mat4 &a, const Mat4 &b) {
  Mat4 r;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
  return r;
}
static inline Mat4 inverse(const Mat4 &in) {
  int indxc[4] = {0, 0, 0, 0}, indxr[4] = {0, 0, 0, 0}, ipiv[4] = {0, 0, 0, 0};
  Mat4 minv = in;
  for (int i = 0; i < 4; i++) {
    int irow = 0, icol = 0;
    float big = 0.f;
    for (int j = 0; j < 4; j++) {
      if (ipiv[j] != 1) {
        for (int k = 0; k < 4; k++) {
          if (ipiv[k] == 0) {
            if (std::fabs(minv.m[j][k]) >= big) {
              big = std::fabs(minv.m[j][k]);
              irow = j;
              icol = k;
            }
          }
        }
      }
    }
    ipiv[icol] += 1;
    if (irow != icol)
      for (int k = 0; k < 4; k++) { float t = minv.m[irow][k]; minv.m[irow][k] = minv.m[icol][k]; minv.m[icol][k] = t; }
    indxr[i] = irow;
    indxc[i] = icol;
    float pivinv = 1.0f / minv.m[icol][icol];  // f32::recip
    minv.m[icol][icol] = 1.f;
    for (int j = 0; j < 4; j++) minv.m[icol][j] *= pivinv;
    for (int j = 0; j < 4; j++) {
      if (j != icol) {
        float save = minv.m[j][icol];
        minv.m[j][icol] = 0.f;
        for (int k = 0; k < 4; k++) minv.m[j][k] -= minv.m[icol][k] * save;
      }
    }
  }
  for (int j = 3; j >= 0; j--) {
    if (indxr[j] != indxc[j])
      for (int k = 0; k < 4; k++) { float t = minv.m[k][indxr[j]]; minv.m[k][indxr[j]] = minv.m[k][indxc[j]]; minv.m[k][indxc[j]] = t; }
  }
  return minv;
}
// returns camera_to_world in c2w, and its inverse (the Transform's `m`) in w2c
static inline void look_at(Vec3 pos, Vec3 look, Vec3 up, Mat4 *w2c, Mat4 *c2w) {
  Mat4 m = identity();
  m.m[0][3] = pos.x; m.m[1][3] = pos.y; m.m[2][3] = pos.z; m.m[3][3] = 1.f;
  Vec3 dir = normalize(look - pos);
  Vec3 right = normalize(cross(normalize(up), dir));
  Vec3 new_up = cross(dir, right);
  m.m[0][0] = right.x; m.m[1][0] = right.y; m.m[2][0] = right.z; m.m[3][0] = 0.f;
  m.m[0][1] = new_up.x; m.m[1][1] = new_up.y; m.m[2][1] = new_up.z; m.m[3][1] = 0.f;
  m.m[0][2] = dir.x; m.m[1][2] = dir.y; m.m[2][2] = dir.z; m.m[3][2] = 0.f;
  *c2w = m;
  *w2c = inverse(m);
}

// lib.rs:115-127 clamp, :139-141 lerp; transform.rs:59-71 solve_linear_system_2x2 (not called on this path -- pbrt-v3 uses it for a
// triangle's dp/du, which no kernel here needs -- but SURVEY 8(c) lists their doc-tests among the vectors to pin the restatement to)
template <class T>
static inline T clamp_ref(T val, T low, T high) { return val < low ? low : (val > high ? high : val); }
static inline float lerp_ref(float t, float v1, float v2) { return (1.f - t) * v1 + t * v2; }
static inline bool solve_linear_system_2x2(const float a[2][2], const float b[2], float x[2]) {
  const float det = a[0][0] * a[1][1] - a[0][1] * a[1][0];
  if (std::fabs(det) < 1e-10f) return false;
  const float x0 = (a[1][1] * b[0] - a[0][1] * b[1]) / det, x1 = (a[0][0] * b[1] - a[1][0] * b[0]) / det;
  if (std::isnan(x0) || std::isnan(x1)) return false;
  x[0] = x0;
  x[1] = x1;
  return true;
}

// ---------------------------------------------------------------------------------------------
// sin / cos on [-pi/4, pi/4] as fixed polynomials (Cephes single-precision kernels) so that CPU
// and GPU agree bit for bit -- no libm / ocml call on the path.
// ---------------------------------------------------------------------------------------------
static inline float poly_sin(float x) {
  float z = x * x;
  float p = -1.9515295891e-4f * z + 8.3321608736e-3f;
  p = p * z - 1.6666654611e-1f;
  return (p * z) * x + x;
}
static inline float poly_cos(float x) {
  float z = x * x;
  float p = 2.443315711809948e-5f * z - 1.388731625493765e-3f;
  p = p * z + 4.166664568298827e-2f;
  return ((p * z) * z - 0.5f * z) + 1.0f;
}

// ---- a sphere's (u, v) for 2-D textures (pbrt-v3 Sphere::Intersect: u = phi / 2 pi with phi = atan2(y, x) in [0, 2 pi),
// v = (theta - pi) / (0 - pi) with theta = acos(z)) on the unit normal n = (p - c) / r, the sphere's own frame being the world's axes.  atan
// and asin are the Cephes single-precision polynomials written out (|error| < 2e-7), the same operations on CPU and GPU. ----
static inline float poly_atan_pos(float x) {  // x >= 0 (+inf included): atan(x) in [0, pi / 2]
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
static inline float poly_asin_small(float a) {  // |a| <= 0.5
  const float z = a * a;
  float p = 4.2163199048e-2f * z + 2.4181311049e-2f;
  p = p * z + 4.5470025998e-2f;
  p = p * z + 7.4953002686e-2f;
  p = p * z + 1.6666752422e-1f;
  return (p * z) * a + a;
}
static inline float poly_acos(float x) {  // x in [-1, 1]
  if (x < -0.5f) return 3.14159265358979323846f - 2.0f * poly_asin_small(std::sqrt(0.5f * (1.0f + x)));
  if (x > 0.5f) return 2.0f * poly_asin_small(std::sqrt(0.5f * (1.0f - x)));
  return 1.5707963267948966f - poly_asin_small(x);
}
static inline void sphere_uv(float nx, float ny, float nz, float *u, float *v) {
  const float ax = std::fabs(nx), ay = std::fabs(ny);
  float phi = (ax == 0.f && ay == 0.f) ? 0.f : poly_atan_pos(ay / ax);  // first quadrant (ax == 0: atan(+inf) = pi / 2)
  if (nx < 0.f) phi = 3.14159265358979323846f - phi;
  if (ny < 0.f) phi = 6.28318530717958647692f - phi;
  const float zc = nz < -1.0f ? -1.0f : (nz > 1.0f ? 1.0f : nz);
  const float theta = poly_acos(zc);
  *u = phi * 0.15915494309189533577f;  // 1 / (2 pi)
  *v = (theta - 3.14159265358979323846f) / (0.f - 3.14159265358979323846f);
}

}  // namespace orc
