// host_math.hpp -- host-side pieces of the render path that the reference crate implements and
// this library must agree with: Matrix4x4 / look_at, Film geometry, RGB<->XYZ, sRGB quantisation.
// Plain fp32, fixed operation order, compiled with -ffp-contract=off.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace pbrt_hip {

struct F3 {
  float x, y, z;
};
inline F3 sub(F3 a, F3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float dot3(F3 a, F3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// core/geometry/vector.rs:314-324
inline F3 cross3(F3 a, F3 b) { return {(a.y * b.z) - (a.z * b.y), (a.z * b.x) - (a.x * b.z), (a.x * b.y) - (a.y * b.x)}; }
// core/geometry/vector.rs:165-167, 204-206: self / self.length(), per-component division
inline F3 unit3(F3 a) {
  float len = std::sqrt(dot3(a, a));
  return {a.x / len, a.y / len, a.z / len};
}

// Matrix4x4: 16 floats, row-major (core/transform.rs:75-77)
inline void mat_identity(float m[16]) {
  for (int i = 0; i < 16; i++) m[i] = (i % 5 == 0) ? 1.f : 0.f;
}

// Gauss-Jordan with full pivoting, core/transform.rs:162-234
inline void mat_inverse(const float in[16], float out[16]) {
  float a[4][4];
  std::memcpy(a, in, 64);
  int row_of[4], col_of[4], used[4] = {0, 0, 0, 0};
  for (int step = 0; step < 4; step++) {
    int pr = 0, pc = 0;
    float best = 0.f;
    for (int r = 0; r < 4; r++) {
      if (used[r] == 1) continue;
      for (int c = 0; c < 4; c++) {
        if (used[c] != 0) continue;
        float v = std::fabs(a[r][c]);
        if (v >= best) { best = v; pr = r; pc = c; }
      }
    }
    used[pc]++;
    if (pr != pc)
      for (int k = 0; k < 4; k++) { float t = a[pr][k]; a[pr][k] = a[pc][k]; a[pc][k] = t; }
    row_of[step] = pr;
    col_of[step] = pc;
    float piv = 1.0f / a[pc][pc];
    a[pc][pc] = 1.f;
    for (int k = 0; k < 4; k++) a[pc][k] *= piv;
    for (int r = 0; r < 4; r++) {
      if (r == pc) continue;
      float f = a[r][pc];
      a[r][pc] = 0.f;
      for (int k = 0; k < 4; k++) a[r][k] -= a[pc][k] * f;
    }
  }
  for (int step = 3; step >= 0; step--) {
    if (row_of[step] == col_of[step]) continue;
    for (int r = 0; r < 4; r++) { float t = a[r][row_of[step]]; a[r][row_of[step]] = a[r][col_of[step]]; a[r][col_of[step]] = t; }
  }
  std::memcpy(out, a, 64);
}

// Transform::look_at, core/transform.rs:485-520.  Left-handed; m = inverse(camera_to_world).
inline void look_at(const float pos[3], const float look[3], const float up[3], float m[16], float m_inv[16]) {
  F3 p = {pos[0], pos[1], pos[2]}, l = {look[0], look[1], look[2]}, u = {up[0], up[1], up[2]};
  F3 dir = unit3(sub(l, p));
  F3 right = unit3(cross3(unit3(u), dir));
  F3 new_up = cross3(dir, right);
  float c2w[16] = {right.x, new_up.x, dir.x, p.x,  //
                   right.y, new_up.y, dir.y, p.y,  //
                   right.z, new_up.z, dir.z, p.z,  //
                   0.f,     0.f,      0.f,   1.f};
  std::memcpy(m_inv, c2w, 64);
  mat_inverse(c2w, m);
}

// Film::new, core/film.rs:92-101: cropped_pixel_bounds = ceil(resolution * crop)
inline int32_t ceil_to_i32(float v) {  // ceil, saturated; NaN -> 0 (the cast of a float outside int32 is undefined behaviour)
  if (!(v == v)) return 0;
  v = std::ceil(v);
  if (v >= 2147483648.f) return 2147483647;
  if (v <= -2147483648.f) return -2147483647 - 1;
  return (int32_t)v;
}
inline void film_cropped_bounds(int xres, int yres, const float crop[4], int32_t b[4]) {
  b[0] = ceil_to_i32((float)xres * crop[0]);
  b[1] = ceil_to_i32((float)yres * crop[2]);
  b[2] = ceil_to_i32((float)xres * crop[1]);
  b[3] = ceil_to_i32((float)yres * crop[3]);
}

// core/spectrum.rs:129-145
inline void xyz_to_rgb(const float xyz[3], float rgb[3]) {
  rgb[0] = 3.240479f * xyz[0] - 1.537150f * xyz[1] - 0.498535f * xyz[2];
  rgb[1] = -0.969256f * xyz[0] + 1.875991f * xyz[1] + 0.041556f * xyz[2];
  rgb[2] = 0.055648f * xyz[0] - 0.204043f * xyz[1] + 1.057311f * xyz[2];
}

// lib.rs:93-99 and core/imageio.rs:66-68
inline float gamma_correct(float v) { return v <= 0.0031308f ? 12.92f * v : 1.055f * std::pow(v, 1.f / 2.4f) - 0.055f; }
inline uint8_t to_byte(float v) {
  float q = 255.f * gamma_correct(v) + 0.5f;
  q = q < 0.f ? 0.f : (q > 255.f ? 255.f : q);
  return (uint8_t)q;
}

// Generator matrices of the first 128 Sobol' dimensions, 32 columns each: dimension 0 is the van der
// Corput sequence (column b = bit 31 - b); dimension d >= 1 from the Joe-Kuo direction numbers (joe_kuo.inc: degree s, coefficient
// bits a, initial numbers m_1 .. m_s; m_i = XOR_{k=1..s-1} a_k 2^k m_{i-k} ^ 2^s m_{i-s} ^ m_{i-s}; column i = m_i << (32 - i)).
// These are rows 0 .. 127 of the reference's SOBOL_MATRICES32 (sobolmatrices.rs:81, 52 columns per dimension, of which a
// 2^20-sample pixel needs 20); tests compare them with the oracle's own construction and, where the reference is mounted, with
// its table.  Sampler 2 gives request j < kSobolNdDims / 2 the dimensions (2j, 2j + 1): 64 requests, every one a path of maxdepth
// 16 can make.
constexpr int kSobolNdDims = 128;
inline void sobol_nd_matrices(uint32_t out[kSobolNdDims * 32]) {
  static const struct { uint32_t s, a, m[10]; } jk[kSobolNdDims - 1] = {
#include "joe_kuo.inc"
  };
  for (int b = 0; b < 32; b++) out[b] = 1u << (31 - b);
  for (int d = 1; d < kSobolNdDims; d++) {
    const uint32_t s = jk[d - 1].s, a = jk[d - 1].a;
    uint64_t m[33];
    for (uint32_t i = 1; i <= s; i++) m[i] = jk[d - 1].m[i - 1];
    for (uint32_t i = s + 1; i <= 32; i++) {
      uint64_t v = m[i - s] ^ (m[i - s] << s);
      for (uint32_t k = 1; k < s; k++)
        if ((a >> (s - 1 - k)) & 1u) v ^= m[i - k] << k;
      m[i] = v;
    }
    for (uint32_t i = 1; i <= 32; i++) out[d * 32 + (i - 1)] = (uint32_t)(m[i] << (32 - i));
  }
}

// The Halton sampler's table, four words per dimension d < 128: {base b = the d-th prime, K = the largest K with
// b^K < 2^32 (32 for b = 2), ceil(2^32 / b), the bits of the float 1 / (float)b^K}.  The kernel's digit loop divides by b with the
// reciprocal (kernels.hip halton_dim); the oracle divides (oracle.cpp HaltonSampler).
constexpr int kHaltonDims = 128;
// The production walk computes a quantised plane's t as fma(q, cell * inv, -(o - origin) * inv).  With inv = 1 / 0 = inf both terms are
// infinite and every t of that axis is NaN: the slab test ignores the axis, and a ray parallel to two axes (a shadow ray towards a sun
// straight overhead: pbrt-v3's default distant light points along z) walks every node its third coordinate allows -- measured 2 500 x
// slower on 1 M triangles.  For a direction component that is exactly 0 the walk therefore multiplies by this FINITE power of two
// instead (with the sign of 1 / d, so that the near / far planes keep their roles): the sign of t is then the sign of
// (plane - o) -- negative huge or positive huge, i.e. outside any ray interval on the right side -- and (o - origin) * it cannot
// overflow while o lies inside the root box (extent = the root box's largest side; outside it an overflow to +-inf still says "missed",
// which is true there).  Conservative like the rest of the walk: the margins of 3 eps |g| apply unchanged (a power of two scales exactly).
inline float inv_parallel_for_extent(float extent) {
  int x = 0;
  if (extent > 0.f && std::isfinite(extent)) (void)std::frexp(2.0f * extent, &x);  // 2 extent < 2^x (a quantised plane may lie a cell beyond the box)
  int e = 123 - x;  // 2^x * 2^e * (1 + margins) < 2^124
  if (e > 120) e = 120;
  if (e < -100) e = -100;
  return std::ldexp(1.0f, e);
}

inline void halton_table(uint32_t out[kHaltonDims * 4]) {
  int n = 0;
  for (uint32_t c = 2; n < kHaltonDims; c++) {
    bool prime = true;
    for (uint32_t q = 2; q * q <= c; q++) if (c % q == 0) { prime = false; break; }
    if (!prime) continue;
    uint32_t K = 0;
    uint64_t bk = 1;
    while (bk * c < (1ull << 32)) { bk *= c; K++; }
    const float inv = c == 2u ? 2.3283064365386963e-10f : 1.0f / (float)(uint32_t)bk;
    uint32_t bits;
    std::memcpy(&bits, &inv, 4);
    out[4 * n] = c;
    out[4 * n + 1] = c == 2u ? 32u : K;
    out[4 * n + 2] = (uint32_t)(((1ull << 32) + c - 1u) / c);
    out[4 * n + 3] = bits;
    n++;
  }
}

}  // namespace pbrt_hip
