// oracle.cpp -- integrator, film and C API of the CPU ORACLE (test infrastructure only, see
// pbrt_oracle.h).  The reference's render call is a comment (api.rs:446-453); the loop below is
// the pbrt-v3 SamplerIntegrator::Render / PathIntegrator::Li structure that comment sketches,
// with every arithmetic choice fixed in DESIGN.md section 3.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <thread>

#include "oracle_scene.hpp"

namespace orc {
static bool g_debug_li = false;  // ORC_DEBUG_LI=1: orc_pixel_samples traces every bounce to stderr (parity debugging)


// -------- Samplers --------
// A pixel's spp samples are cut into K CHUNKS, K = sample_chunks(spp) = the largest power of two <= 16 that leaves a chunk
// at least 32 samples (1 below 64 spp): chunk c holds the samples s with floor(c * spp / K) <= s < floor((c + 1) * spp / K)
// and is a unit of its own -- its own PCG32 stream Rng::new((seed * W * H + y * W + x) * K + c) (rng.rs:46-59 fixes only
// set_sequence) and its own partial film sum; the pixel's contrib_sum is the sum of the K partial sums in chunk order.
// (Round 1 ran all samples of a pixel on one stream; the chunks exist so that the GPU can hand out work in pieces of about
// 32 samples: the frame no longer waits for the sequential samples of its most expensive pixels.)
static inline uint32_t sample_chunks(uint32_t spp) {
  uint32_t k = 1;
  while (2 * k <= 16 && 2 * k * 32 <= spp) k *= 2;
  return k;
}
static inline uint32_t chunk_begin(uint32_t c, uint32_t spp) { return (uint32_t)(((uint64_t)c * spp) / sample_chunks(spp)); }

// lowbias32 (integer hash; every operation is defined on uint32): the scrambles of the Sobol sampler come from it
static inline uint32_t mix32(uint32_t v) {
  v ^= v >> 16; v *= 0x7feb352du; v ^= v >> 15; v *= 0x846ca68bu; v ^= v >> 16;
  return v;
}

// Interface of pbrt-v3's samplers as far as PathIntegrator::Li uses it.
class Sampler {
 public:
  virtual ~Sampler() {}
  virtual void StartChunk(int x, int y, uint32_t chunk) = 0;  // positions the sampler on the chunk's first sample
  virtual bool ChunkDone() const = 0;
  virtual void GetCameraSample(float *fx, float *fy) = 0;
  virtual float Get1D() = 0;
  virtual void Get2D(float *u1, float *u2) = 0;
  virtual void StartNextSample() = 0;
};

// stratified pixel position + independent later dimensions
class StratifiedSampler : public Sampler {
 public:
  // pad: pixels the sample bounds reach beyond the image on each side (0 for the default filter)
  StratifiedSampler(uint32_t nx, uint32_t ny, uint64_t seed, const Scene &s, int pad_x = 0, int pad_y = 0)
      : nx_(nx), ny_(ny), seed_(seed), w_((uint64_t)(s.xres + 2 * pad_x)), h_((uint64_t)(s.yres + 2 * pad_y)), pad_x_(pad_x), pad_y_(pad_y) {
    inv_nx_ = 1.0f / (float)nx;
    inv_ny_ = 1.0f / (float)ny;
  }
  void StartChunk(int x, int y, uint32_t chunk) override {
    rng_.set_sequence((seed_ * w_ * h_ + (uint64_t)(y + pad_y_) * w_ + (uint64_t)(x + pad_x_)) * sample_chunks(nx_ * ny_) + chunk);
    px_ = x; py_ = y;
    s_ = chunk_begin(chunk, nx_ * ny_);
    s_end_ = chunk_begin(chunk + 1, nx_ * ny_);
  }
  bool ChunkDone() const override { return s_ >= s_end_; }
  // film position of the current sample: stratum (s mod nx, s div nx), jittered
  void GetCameraSample(float *fx, float *fy) override {
    const float one_minus_eps = 1.0f - std::numeric_limits<float>::epsilon();
    uint32_t sx = s_ % nx_, sy = s_ / nx_;
    float u1 = rng_.uniform_float();
    float u2 = rng_.uniform_float();
    float jx = ((float)sx + u1) * inv_nx_;
    float jy = ((float)sy + u2) * inv_ny_;
    if (jx > one_minus_eps) jx = one_minus_eps;
    if (jy > one_minus_eps) jy = one_minus_eps;
    *fx = (float)px_ + jx;
    *fy = (float)py_ + jy;
  }
  float Get1D() override { return rng_.uniform_float(); }
  void Get2D(float *u1, float *u2) override { *u1 = rng_.uniform_float(); *u2 = rng_.uniform_float(); }
  void StartNextSample() override { ++s_; }

 private:
  uint32_t nx_, ny_;
  uint64_t seed_, w_, h_;
  int pad_x_, pad_y_;
  float inv_nx_, inv_ny_;
  Rng rng_;
  int px_ = 0, py_ = 0;
  uint32_t s_ = 0, s_end_ = 0;
};

// Generator matrices of the Sobol' sequence from the Joe-Kuo direction numbers (S. Joe, F. Y. Kuo, "Constructing
// Sobol sequences with better two-dimensional projections", SIAM J. Sci. Comput. 30, 2008; table new-joe-kuo-6):
// dimension 1 is the van der Corput sequence, dimension j >= 2 has a primitive polynomial of degree s with
// coefficient bits a and initial numbers m_1 .. m_s.  Column i (1-based) of the 32-bit matrix is m_i * 2^(32 - i)
// (m_i >> (i - 32) beyond column 32), with m_i = 2 a_1 m_{i-1} ^ 4 a_2 m_{i-2} ^ ... ^ 2^s m_{i-s} ^ m_{i-s}.
// The same layout as the reference's SOBOL_MATRICES32 (sobolmatrices.rs:81: 52 columns per dimension), which a
// test compares it with where the reference is mounted.
struct JoeKuo { uint32_t s, a; uint32_t m[10]; };
static const JoeKuo kJoeKuo[] = {  // dimensions 2 .. 128 (rows 1 .. 127 of the reference's table)
#include "joe_kuo.inc"
};
constexpr int kSobolDims = 1 + (int)(sizeof(kJoeKuo) / sizeof(kJoeKuo[0]));
constexpr int kSobolColumns = 52;
static void sobol_matrix(int dim, uint32_t out[kSobolColumns]) {
  if (dim == 0) {
    for (int i = 0; i < kSobolColumns; i++) out[i] = i < 32 ? 1u << (31 - i) : 0u;
    return;
  }
  const JoeKuo &d = kJoeKuo[dim - 1];
  uint64_t m[kSobolColumns + 1];
  for (uint32_t i = 1; i <= d.s; i++) m[i] = d.m[i - 1];
  for (uint32_t i = d.s + 1; i <= (uint32_t)kSobolColumns; i++) {
    uint64_t v = m[i - d.s] ^ (m[i - d.s] << d.s);
    for (uint32_t k = 1; k < d.s; k++)
      if ((d.a >> (d.s - 1 - k)) & 1u) v ^= m[i - k] << k;
    m[i] = v;
  }
  for (int i = 1; i <= kSobolColumns; i++) out[i - 1] = i <= 32 ? (uint32_t)(m[i] << (32 - i)) : (uint32_t)(m[i] >> (i - 32));
}

// Padded (0,2)-sequence sampler: every request j of a sample (0: the camera sample; then light pick,
// light point, BSDF direction, roulette ... in program order) takes point number (s ^ mask_j) of the first two
// Sobol' dimensions, XOR-scrambled: the 2^k points of a pixel are the same (0,2)-net in every request, visited in a
// different order and with different digit scrambles (keys from mix32 of the pixel and j).  No state besides (pixel, s,
// j): any chunk of a pixel's samples can be generated on its own.
class SobolSampler : public Sampler {
 public:
  SobolSampler(uint32_t nx, uint32_t ny, uint64_t seed, const Scene &s, int pad_x = 0, int pad_y = 0)
      : spp_(nx * ny), seed_(seed), w_((uint64_t)(s.xres + 2 * pad_x)), h_((uint64_t)(s.yres + 2 * pad_y)), pad_x_(pad_x), pad_y_(pad_y) {
    sobol_matrix(1, c1_);
    log2_ = 0;
    while ((1u << log2_) < spp_) log2_++;
  }
  void StartChunk(int x, int y, uint32_t chunk) override {
    const uint64_t q = seed_ * w_ * h_ + (uint64_t)(y + pad_y_) * w_ + (uint64_t)(x + pad_x_);
    key_ = mix32((uint32_t)q ^ mix32((uint32_t)(q >> 32) + 0x9e3779b9u));
    px_ = x; py_ = y;
    s_ = chunk_begin(chunk, spp_);
    s_end_ = chunk_begin(chunk + 1, spp_);
    j_ = 0;
  }
  bool ChunkDone() const override { return s_ >= s_end_; }
  void GetCameraSample(float *fx, float *fy) override {
    float u1, u2;
    Get2D(&u1, &u2);
    *fx = (float)px_ + u1;
    *fy = (float)py_ + u2;
  }
  float Get1D() override { float u1, u2; Get2D(&u1, &u2); return u1; }
  void Get2D(float *u1, float *u2) override {
    const float one_minus_eps = 1.0f - std::numeric_limits<float>::epsilon();
    const uint32_t a = mix32(key_ + j_ * 0x9e3779b9u);
    j_++;
    const uint32_t i = s_ ^ (a & ((1u << log2_) - 1u));
    uint32_t x = 0, y = 0;
    for (uint32_t b = 0; b < 32 && (i >> b) != 0u; b++)
      if ((i >> b) & 1u) { x ^= 1u << (31 - b); y ^= c1_[b]; }
    x ^= mix32(a ^ 0x68e31da4u);
    y ^= mix32(a ^ 0xb5297a4du);
    float f1 = (float)x * 2.3283064365386963e-10f, f2 = (float)y * 2.3283064365386963e-10f;
    *u1 = f1 > one_minus_eps ? one_minus_eps : f1;
    *u2 = f2 > one_minus_eps ? one_minus_eps : f2;
  }
  void StartNextSample() override { ++s_; j_ = 0; }

 protected:
  uint32_t spp_, log2_ = 0;
  uint64_t seed_, w_, h_;
  int pad_x_, pad_y_;
  uint32_t c1_[kSobolColumns];
  uint32_t key_ = 0, j_ = 0;
  int px_ = 0, py_ = 0;
  uint32_t s_ = 0, s_end_ = 0;
};

// Sobol' sampler proper (Sampler "sobol"): request j of a sample takes ITS OWN pair of Sobol' dimensions
// (2j, 2j + 1) for j = 0 .. 63 -- the first 128 dimensions, whose generator matrices are pinned to the reference's
// SOBOL_MATRICES32 (sobolmatrices.rs:81) -- at point index i = the sample number, each coordinate XOR-scrambled with mix32(key +
// (d + 1) * 0x9e3779b9), d the dimension.  64 requests cover every request a path of maxdepth 16 can make (the camera sample, three
// per vertex, the roulette from the fifth vertex on: 61); later requests are the padded (0,2)-sequence requests of 3.10 with the
// same request counter.  Integer arithmetic only.
class SobolNdSampler : public SobolSampler {
 public:
  SobolNdSampler(uint32_t nx, uint32_t ny, uint64_t seed, const Scene &s, int pad_x = 0, int pad_y = 0) : SobolSampler(nx, ny, seed, s, pad_x, pad_y) {
    for (int d = 0; d < kSobolDims; d++) sobol_matrix(d, m_[d]);
  }
  void Get2D(float *u1, float *u2) override {
    if (j_ >= (uint32_t)(kSobolDims / 2)) { SobolSampler::Get2D(u1, u2); return; }
    const float one_minus_eps = 1.0f - std::numeric_limits<float>::epsilon();
    const uint32_t d0 = 2u * j_;
    j_++;
    uint32_t x = 0, y = 0;
    for (uint32_t b = 0; b < 32 && (s_ >> b) != 0u; b++)
      if ((s_ >> b) & 1u) { x ^= m_[d0][b]; y ^= m_[d0 + 1][b]; }
    x ^= mix32(key_ + (d0 + 1u) * 0x9e3779b9u);
    y ^= mix32(key_ + (d0 + 2u) * 0x9e3779b9u);
    const float f1 = (float)x * 2.3283064365386963e-10f, f2 = (float)y * 2.3283064365386963e-10f;
    *u1 = f1 > one_minus_eps ? one_minus_eps : f1;
    *u2 = f2 > one_minus_eps ? one_minus_eps : f2;
  }

 private:
  uint32_t m_[kSobolDims][kSobolColumns];
};

// Halton sampler proper (Sampler "halton", the reference's default sampler NAME, api.rs:235 -- it has no sampler
// code): request j < 64 of a sample takes the dimensions (2j, 2j + 1), dimension d the radical inverse of the sample's number in
// the pixel in base p_d (the d-th prime, 2 .. 719), every digit scrambled per pixel and dimension, the digits no sample of the frame
// has drawn together as one random tail.  Integer arithmetic up to the last step:
//   base 2 (d = 0):  v = bit-reversal of the index, XOR mix32(key + 0x9e3779b9);  u = v * 2^-32
//   base b > 2:      D = the digits the frame's largest sample index can have: the smallest D with b^D > spp_mask (spp_mask =
//                    2^ceil(log2 spp) - 1);  h_0 = mix32(key + (d + 1) 0x9e3779b9), h_{k+1} = h_k 0x9e3779b1 + 0x7f4a7c15;
//                    digits a_0 .. a_{D-1} of the index (least significant first; leading zeros are digits like any other):
//                    m_k = 1 + (((h_{k+1} >> 16) (b - 1)) >> 16), c_k = ((h_{k+1} & 0xffff) b) >> 16, a'_k = (a_k m_k + c_k) mod b
//                    (a random linear bijection of Z_b per digit position); head = sum a'_k b^(D-1-k);
//                    u = ((float)head + (float)h_{D+1} * 2^-32) * (1 / (float)b^D)
//   u = min(u, 1 - eps).  Requests j >= 64 are the padded requests of 3.10 (as for sampler 2).
static const uint32_t *halton_primes() {
  static uint32_t p[128];
  if (!p[0]) {
    int n = 0;
    for (uint32_t c = 2; n < 128; c++) {
      bool prime = true;
      for (uint32_t q = 2; q * q <= c; q++) if (c % q == 0) { prime = false; break; }
      if (prime) p[n++] = c;
    }
  }
  return p;
}
class HaltonSampler : public SobolSampler {
 public:
  HaltonSampler(uint32_t nx, uint32_t ny, uint64_t seed, const Scene &s, int pad_x = 0, int pad_y = 0) : SobolSampler(nx, ny, seed, s, pad_x, pad_y) {}
  // -> the integer head (base 2: the scrambled 32-bit value); *pw_out = b^D (base 2: 0)
  static uint32_t scrambled_radical_inverse(uint32_t d, uint32_t index, uint32_t key, uint32_t spp_mask, float *u, uint32_t *pw_out = nullptr) {
    const float one_minus_eps = 1.0f - std::numeric_limits<float>::epsilon();
    const uint32_t b = halton_primes()[d];
    uint32_t v = 0, pw = 0;
    float f;
    if (b == 2u) {
      for (uint32_t k = 0; k < 32; k++) v |= ((index >> k) & 1u) << (31 - k);
      v ^= mix32(key + (d + 1u) * 0x9e3779b9u);
      f = (float)v * 2.3283064365386963e-10f;
    } else {
      uint32_t h = mix32(key + (d + 1u) * 0x9e3779b9u), n = index;
      pw = 1u;
      do {
        pw *= b;
        const uint32_t a = n % b;
        n /= b;
        h = h * 0x9e3779b1u + 0x7f4a7c15u;
        const uint32_t m = 1u + (((h >> 16) * (b - 1u)) >> 16), c = ((h & 0xffffu) * b) >> 16;
        v = v * b + (a * m + c) % b;
      } while (pw <= spp_mask);
      h = h * 0x9e3779b1u + 0x7f4a7c15u;
      f = ((float)v + (float)h * 2.3283064365386963e-10f) * (1.0f / (float)pw);
    }
    *u = f > one_minus_eps ? one_minus_eps : f;
    if (pw_out) *pw_out = pw;
    return v;
  }
  void Get2D(float *u1, float *u2) override {
    if (j_ >= 64u) { SobolSampler::Get2D(u1, u2); return; }
    const uint32_t d0 = 2u * j_, mask = (1u << log2_) - 1u;
    j_++;
    scrambled_radical_inverse(d0, s_, key_, mask, u1);
    scrambled_radical_inverse(d0 + 1u, s_, key_, mask, u2);
  }
};

// pbrt-v3 ConcentricSampleDisk with the fixed polynomials instead of libm sin/cos
static inline void concentric_sample_disk(float u1, float u2, float *dx, float *dy) {
  float ox = 2.0f * u1 - 1.0f;
  float oy = 2.0f * u2 - 1.0f;
  if (ox == 0.f && oy == 0.f) { *dx = 0.f; *dy = 0.f; return; }
  if (std::fabs(ox) > std::fabs(oy)) {
    float phi = kPiOver4 * (oy / ox);
    *dx = ox * poly_cos(phi);
    *dy = ox * poly_sin(phi);
  } else {
    float phi = kPiOver4 * (ox / oy);
    *dx = oy * poly_sin(phi);
    *dy = oy * poly_cos(phi);
  }
}

// cosine-weighted direction about n (pbrt-v3 CosineSampleHemisphere + CoordinateSystem);
// returns the local z (= cos theta); z == 0 means pdf == 0
static inline float cosine_sample_about(Vec3 n, float u1, float u2, Vec3 *wi) {
  float dx, dy;
  concentric_sample_disk(u1, u2, &dx, &dy);
  float zz = (1.0f - dx * dx) - dy * dy;
  float z = std::sqrt(zz > 0.f ? zz : 0.f);
  Vec3 v2;
  if (std::fabs(n.x) > std::fabs(n.y)) {
    float l = std::sqrt(n.x * n.x + n.z * n.z);
    v2 = {-n.z / l, 0.f, n.x / l};
  } else {
    float l = std::sqrt(n.y * n.y + n.z * n.z);
    v2 = {0.f, n.z / l, -n.y / l};
  }
  Vec3 v3_ = cross(n, v2);
  *wi = (v2 * dx + v3_ * dy) + n * z;
  return z;
}

struct RayStats {
  uint64_t camera = 0, bounce = 0, shadow = 0;
  Counters c;
};

// -------- Integrator --------
class PathIntegrator {
 public:
  PathIntegrator(const Scene &s, uint32_t max_depth, bool direct_only, bool mis = false)
      : scene(s), max_depth_(max_depth), direct_only_(direct_only), mis_(mis && !direct_only) {}

  // PathIntegrator::Li.  `direct_only_` turns it into the direct-lighting
  // integrator: first non-specular vertex gets its one-light estimate and the path ends.
  Vec3 Li(Ray ray, Sampler &sampler, RayStats &st) const {
    if (g_debug_li) std::fprintf(stderr, "ORC sample begins\n");
    Vec3 L = {0, 0, 0}, beta = {1, 1, 1};
    bool specular = false;
    float pb = 0.f;  // MIS: the solid-angle density with which the ray in flight was drawn from the BSDF (cos / pi)
    const uint32_t nL = (uint32_t)scene.lights.size();
    const float nLf = (float)nL;
    for (uint32_t bounces = 0;; bounces++) {
      // a ray at the depth limit can only collect emission, and only after a specular bounce -- or, with MIS, as the BSDF-sampled
      // half of the last vertex's direct-light estimate
      if (bounces > 0 && bounces >= max_depth_ && !specular && !mis_) break;
      if (bounces == 0) st.camera++; else st.bounce++;
      Hit h = scene.Intersect(ray, &st.c);
      bool hit = h.prim != 0xffffffffu;
      if (g_debug_li) {
        uint32_t tb, lb, bb; std::memcpy(&tb, &h.t, 4); std::memcpy(&lb, &L.x, 4); std::memcpy(&bb, &beta.x, 4);
        std::fprintf(stderr, "ORC bounce %u prim %u t %08x b1 %a b2 %a L %08x beta %08x\n", bounces, h.prim, tb, h.b1, h.b2, lb, bb);
        std::fprintf(stderr, "ORC   ray o %a %a %a d %a %a %a\n", ray.o.x, ray.o.y, ray.o.z, ray.d.x, ray.d.y, ray.d.z);
      }
      Vec3 p{}, ng{};
      const orc_material *m = nullptr;
      Vec3 wo = -ray.d;
      if (hit) {
        if (h.prim < scene.n_tris()) {
          Vec3 p0, p1, p2;
          scene.tri_verts(h.prim, &p0, &p1, &p2);
          ng = normalize(cross(p1 - p0, p2 - p0));
          float w = (1.0f - h.b1) - h.b2;
          p = (p0 * w + p1 * h.b1) + p2 * h.b2;
          m = &scene.mats[scene.mat_id[h.prim]];
        } else {
          const orc_sphere &sp = scene.spheres[h.prim - scene.n_tris()];
          Vec3 c = v3(sp.c[0], sp.c[1], sp.c[2]);
          Vec3 ph = (ray.o - c) + ray.d * h.t;
          ng = ph / sp.r;
          p = c + ph;
          m = &scene.mats[sp.mat];
        }
      }
      if (bounces == 0 || specular) {
        if (hit) {
          Vec3 le = v3(m->le[0], m->le[1], m->le[2]);
          if ((le.x > 0.f || le.y > 0.f || le.z > 0.f) && dot(ng, wo) > 0.f) L = L + beta * le;
        } else if (scene.has_infinite) {
          L = L + beta * scene.le_infinite;
        }
      } else if (mis_) {
        // the ray was drawn from the BSDF with density pb; the light-sampling strategy would have drawn this direction
        // with pl -- an emissive triangle (picked with 1 / nL, a uniform point of it): (t^2 / (cos_l A)) / nL; the constant
        // environment (picked with 1 / nL, cosine-sampled): pb / nL.  Power heuristic: w = pb^2 / (pb^2 + pl^2).
        if (hit && h.prim < scene.n_tris()) {
          Vec3 le = v3(m->le[0], m->le[1], m->le[2]);
          const float cl = dot(ng, wo);
          if ((le.x > 0.f || le.y > 0.f || le.z > 0.f) && cl > 0.f) {
            Vec3 p0, p1, p2;
            scene.tri_verts(h.prim, &p0, &p1, &p2);
            const float area = 0.5f * length(cross(p1 - p0, p2 - p0));
            const float pl = ((h.t * h.t) / (cl * area)) / nLf;
            const float w = (pb * pb) / (pb * pb + pl * pl);
            L = L + (beta * le) * w;
          }
        } else if (!hit && scene.has_infinite) {
          const float w = (nLf * nLf) / (nLf * nLf + 1.0f);
          L = L + (beta * scene.le_infinite) * w;
        }
      }
      if (!hit || bounces >= max_depth_) break;
      Vec3 nf = dot(ng, wo) < 0.f ? -ng : ng;
      Vec3 po = p + nf * kSpawnEps;
      Vec3 k = v3(m->k[0], m->k[1], m->k[2]);
      if (m->type == 0 && m->kd_tex != 0u && (h.prim >= scene.n_tris() || !scene.tri_uv.empty())) {
        // Kd from a texture (pbrt-v3 Checkerboard2DTexture over UVMapping2D, aamode none): the corner (u, v) of the
        // triangle interpolated with the hit's barycentrics in the order the hit point is -- a sphere: its own (u, v), sphere_uv on the
        // normal --, then (s, t) = (su u + du, sv v + dv); tex1 where floor(s) + floor(t) is even
        const orc_texture &tx = scene.textures[m->kd_tex - 1u];
        float u, v;
        if (h.prim < scene.n_tris()) {
          const float *uv = &scene.tri_uv[6 * (size_t)h.prim];
          const float w = (1.0f - h.b1) - h.b2;
          u = (uv[0] * w + uv[2] * h.b1) + uv[4] * h.b2;
          v = (uv[1] * w + uv[3] * h.b1) + uv[5] * h.b2;
        } else {
          sphere_uv(ng.x, ng.y, ng.z, &u, &v);
        }
        const float ss = tx.su * u + tx.du, tt = tx.sv * v + tx.dv;
        // (the parity in float on both sides: an (int) cast of a huge s is undefined here and saturates on the GPU)
        const float cf = std::floor(ss) + std::floor(tt);
        k = (cf - 2.0f * std::floor(0.5f * cf)) == 0.f ? v3(tx.tex1[0], tx.tex1[1], tx.tex1[2]) : v3(tx.tex2[0], tx.tex2[1], tx.tex2[2]);
      }
      Vec3 wi;
      if (m->type == 0) {  // matte
        if (nL > 0) {
          float xi = sampler.Get1D();
          float u1, u2;
          sampler.Get2D(&u1, &u2);
          uint32_t li = (uint32_t)(xi * nLf);
          if (li > nL - 1) li = nL - 1;
          Vec3 Ld;
          Ray sh;
          if (sample_light(scene.lights[li], po, nf, k, u1, u2, nLf, &Ld, &sh, mis_)) {
            st.shadow++;
            const bool occ = scene.IntersectP(sh, &st.c);
            if (g_debug_li) { uint32_t a; std::memcpy(&a, &Ld.x, 4); std::fprintf(stderr, "ORC   light %u Ld %08x occluded %d tmax %a\n", li, a, (int)occ, sh.tmax); }
            if (!occ) L = L + beta * Ld;
          }
        }
        if (direct_only_) break;
        float u1, u2;
        sampler.Get2D(&u1, &u2);
        float z = cosine_sample_about(nf, u1, u2, &wi);
        if (g_debug_li) {
          float dx, dy; concentric_sample_disk(u1, u2, &dx, &dy);
          std::fprintf(stderr, "ORC   cos u1 %a u2 %a dx %a dy %a z %a nf %a %a %a wi %a %a %a\n", u1, u2, dx, dy, z, nf.x, nf.y, nf.z, wi.x, wi.y, wi.z);
        }
        if (z == 0.f) break;
        beta = beta * k;
        specular = false;
        pb = z * kInvPi;
      } else {  // mirror
        float c = dot(wo, nf);
        wi = -wo + nf * (2.0f * c);
        beta = beta * k;
        specular = true;
      }
      if (beta.x == 0.f && beta.y == 0.f && beta.z == 0.f) break;
      ray.o = po;
      ray.d = wi;
      ray.tmax = kInf;
      if (bounces > 3) {
        float mx = fmax2(beta.x, fmax2(beta.y, beta.z));
        float q = fmax2(0.05f, 1.0f - mx);
        if (sampler.Get1D() < q) break;
        beta = beta / (1.0f - q);
      }
    }
    return L;
  }

  // UniformSampleOneLight's per-light part.  Returns false when geometry rules the
  // light out (no shadow ray is cast then).
  // mis: the estimate is weighted with the power heuristic pl^2 / (pl^2 + pb^2) -- pl the density of this
  // strategy for the direction (light picked with 1 / nL), pb = cos / pi the BSDF's; delta lights keep weight 1.
  static bool sample_light(const LightRec &l, Vec3 po, Vec3 nf, Vec3 kd, float u1, float u2, float nLf,
                           Vec3 *Ld, Ray *sh, bool mis = false) {
    Vec3 f = kd * kInvPi;
    sh->o = po;
    if (l.type == 0) {  // point
      Vec3 dv = l.p0 - po;
      float dist2 = dot(dv, dv);
      if (!(dist2 > 0.f)) return false;
      float dist = std::sqrt(dist2);
      Vec3 wi = dv / dist;
      float cs = dot(wi, nf);
      if (!(cs > 0.f)) return false;
      float scale = (cs / dist2) * nLf;
      *Ld = (f * l.c) * scale;
      sh->d = wi;
      sh->tmax = dist * kShadowShrink;
      return true;
    } else if (l.type == 1) {  // distant
      Vec3 wi = l.p0;
      float cs = dot(wi, nf);
      if (!(cs > 0.f)) return false;
      float scale = cs * nLf;
      *Ld = (f * l.c) * scale;
      sh->d = wi;
      sh->tmax = kInf;
      return true;
    } else if (l.type == 2) {  // constant infinite: cosine sampled, f*cos/pdf = Kd
      Vec3 wi;
      float z = cosine_sample_about(nf, u1, u2, &wi);
      if (z == 0.f) return false;
      *Ld = (kd * l.c) * nLf;
      if (mis) *Ld = *Ld * (1.0f / (1.0f + nLf * nLf));  // pl = pb / nL
      sh->d = wi;
      sh->tmax = kInf;
      return true;
    } else {  // emissive triangle, uniform area sampling (pbrt-v3 UniformSampleTriangle)
      float su0 = std::sqrt(u1);
      float b0 = 1.0f - su0;
      float b1 = u2 * su0;
      float b2 = (1.0f - b0) - b1;
      Vec3 pl = (l.p0 * b0 + l.p1 * b1) + l.p2 * b2;
      Vec3 dv = pl - po;
      float dist2 = dot(dv, dv);
      if (!(dist2 > 0.f)) return false;
      float dist = std::sqrt(dist2);
      Vec3 wi = dv / dist;
      float cs = dot(wi, nf);
      if (!(cs > 0.f)) return false;
      float cl = -dot(wi, l.n);
      if (!(cl > 0.f)) return false;
      float scale = (((cs * cl) * l.area) / dist2) * nLf;
      if (mis) {
        const float pl = (dist2 / (cl * l.area)) / nLf, pbl = cs * kInvPi;
        scale = scale * ((pl * pl) / (pl * pl + pbl * pbl));
      }
      *Ld = (f * l.c) * scale;
      sh->d = wi;
      sh->tmax = dist * kShadowShrink;
      return true;
    }
  }

  const Scene &scene;

 private:
  uint32_t max_depth_;
  bool direct_only_;
  bool mis_;  // integrator 2: the direct-light estimate with multiple importance sampling
};

// pbrt-v3 SamplerIntegrator::Render's radiance sanitising before FilmTile::AddSample
static inline Vec3 sanitize(Vec3 L) {
  float y = (0.212671f * L.x + 0.715160f * L.y) + 0.072169f * L.z;
  if (std::isnan(L.x) || std::isnan(L.y) || std::isnan(L.z) || y < -1e-5f || std::isinf(y)) return {0, 0, 0};
  return L;
}

// FilmTile::AddSample's clamp (pbrt-v3; the reference stores the bound, film.rs:75,279): a sample brighter than
// max_sample_luminance is scaled down to it.  0 = no bound.
static inline Vec3 clamp_luminance(Vec3 L, float max_lum) {
  const float y = (0.212671f * L.x + 0.715160f * L.y) + 0.072169f * L.z;
  if (max_lum > 0.f && y > max_lum) {
    const float sc = max_lum / y;
    return {L.x * sc, L.y * sc, L.z * sc};
  }
  return L;
}

// ---- box filter radii other than 0.5 ----
// Sample bounds (film.rs:166-175) reach pad = ceil(radius - 0.5) pixels beyond the cropped window; every sample at film
// point p adds its radiance, weight 1, to the pixels [ceil(p - 0.5 - radius), floor(p - 0.5 + radius)] inside the window
// (pbrt-v3 FilmTile::AddSample with the box filter's table of ones).  The sums are kept in 64-bit fixed point, 2^-24
// units, a component clamped to [0, 2^15] first: integer addition is associative, so the result does not depend on the
// order in which samples -- of one thread or of several ranks -- arrive, which is what lets the GPU add them with atomics.
struct WideFilter {
  float rx, ry;
  int pad_x, pad_y;
  int32_t sb[4];  // sample bounds x0 y0 x1 y1
};
static inline float filter_radius(float w) { return w == 0.f ? 0.5f : w; }
static inline bool is_default_filter(const orc_render_desc &r) { return filter_radius(r.filter_xwidth) == 0.5f && filter_radius(r.filter_ywidth) == 0.5f; }
static WideFilter wide_filter(const Scene &s, const orc_render_desc &r) {
  WideFilter f;
  f.rx = filter_radius(r.filter_xwidth);
  f.ry = filter_radius(r.filter_ywidth);
  f.sb[0] = (int32_t)std::floor(((float)s.cropped[0] + 0.5f) - f.rx);
  f.sb[1] = (int32_t)std::floor(((float)s.cropped[1] + 0.5f) - f.ry);
  f.sb[2] = (int32_t)std::ceil(((float)s.cropped[2] - 0.5f) + f.rx);
  f.sb[3] = (int32_t)std::ceil(((float)s.cropped[3] - 0.5f) + f.ry);
  f.pad_x = std::max(0, (int)std::ceil(f.rx - 0.5f));
  f.pad_y = std::max(0, (int)std::ceil(f.ry - 0.5f));
  return f;
}
constexpr float kFixedOne = 16777216.0f;        // 2^24
constexpr float kFixedMax = 32768.0f;           // 2^15: largest component a sample can add
static inline int64_t to_fixed(float v) {
  const float c = v > 0.f ? (v < kFixedMax ? v : kFixedMax) : 0.f;
  return (int64_t)(c * kFixedOne);
}
static void render_pixel_wide(const Scene &s, const PathIntegrator &integ, const orc_render_desc &r, const WideFilter &f, int x, int y,
                              int64_t *acc, RayStats &st) {
  StratifiedSampler strat(r.spp_x, r.spp_y, r.seed, s, f.pad_x, f.pad_y);
  SobolSampler sobol(r.spp_x, r.spp_y, r.seed, s, f.pad_x, f.pad_y);
  SobolNdSampler sobol_nd(r.spp_x, r.spp_y, r.seed, s, f.pad_x, f.pad_y);
  HaltonSampler halton(r.spp_x, r.spp_y, r.seed, s, f.pad_x, f.pad_y);
  Sampler &sampler = r.sampler == 3 ? (Sampler &)halton : (r.sampler == 2 ? (Sampler &)sobol_nd : (r.sampler == 1 ? (Sampler &)sobol : (Sampler &)strat));
  const int W = s.cropped[2] - s.cropped[0];
  for (uint32_t c = 0, n_chunks = sample_chunks(r.spp_x * r.spp_y); c < n_chunks; c++)
    for (sampler.StartChunk(x, y, c); !sampler.ChunkDone(); sampler.StartNextSample()) {
      float fx, fy;
      sampler.GetCameraSample(&fx, &fy);
      Ray ray = s.camera_ray(fx, fy);
      const Vec3 L = clamp_luminance(sanitize(integ.Li(ray, sampler, st)), r.max_sample_luminance);
      const int64_t q[4] = {to_fixed(L.x), to_fixed(L.y), to_fixed(L.z), 1};
      const float dx = fx - 0.5f, dy = fy - 0.5f;
      int x0 = (int)std::ceil(dx - f.rx), x1 = (int)std::floor(dx + f.rx) + 1;
      int y0 = (int)std::ceil(dy - f.ry), y1 = (int)std::floor(dy + f.ry) + 1;
      x0 = std::max(x0, s.cropped[0]); x1 = std::min(x1, s.cropped[2]);
      y0 = std::max(y0, s.cropped[1]); y1 = std::min(y1, s.cropped[3]);
      for (int py = y0; py < y1; py++)
        for (int px = x0; px < x1; px++) {
          int64_t *a = acc + 4 * ((size_t)(py - s.cropped[1]) * W + (size_t)(px - s.cropped[0]));
          for (int k = 0; k < 4; k++) __atomic_fetch_add(&a[k], q[k], __ATOMIC_RELAXED);
        }
    }
}
static inline void film_from_acc(const int64_t a[4], float out_xyzw[4]) {
  const float inv = 1.0f / kFixedOne;
  float rgb[3] = {(float)a[0] * inv, (float)a[1] * inv, (float)a[2] * inv}, xyz[3];
  rgb_to_xyz(rgb, xyz);
  out_xyzw[0] = xyz[0]; out_xyzw[1] = xyz[1]; out_xyzw[2] = xyz[2]; out_xyzw[3] = (float)a[3];
}

static void render_pixel(const Scene &s, const PathIntegrator &integ, const orc_render_desc &r, int x, int y,
                         float out_xyzw[4], float *per_sample, RayStats &st) {
  const WideFilter f = wide_filter(s, r);  // (pads 0 for the default filter; orc_pixel_samples under a wide filter numbers its streams as render_pixel_wide does)
  StratifiedSampler strat(r.spp_x, r.spp_y, r.seed, s, f.pad_x, f.pad_y);
  SobolSampler sobol(r.spp_x, r.spp_y, r.seed, s, f.pad_x, f.pad_y);
  SobolNdSampler sobol_nd(r.spp_x, r.spp_y, r.seed, s, f.pad_x, f.pad_y);
  HaltonSampler halton(r.spp_x, r.spp_y, r.seed, s, f.pad_x, f.pad_y);
  Sampler &sampler = r.sampler == 3 ? (Sampler &)halton : (r.sampler == 2 ? (Sampler &)sobol_nd : (r.sampler == 1 ? (Sampler &)sobol : (Sampler &)strat));
  Vec3 sum = {0, 0, 0};
  uint32_t i = 0;
  for (uint32_t c = 0, n_chunks = sample_chunks(r.spp_x * r.spp_y); c < n_chunks; c++) {
    Vec3 part = {0, 0, 0};
    for (sampler.StartChunk(x, y, c); !sampler.ChunkDone(); sampler.StartNextSample()) {
      float fx, fy;
      sampler.GetCameraSample(&fx, &fy);
      Ray ray = s.camera_ray(fx, fy);
      Vec3 L = clamp_luminance(sanitize(integ.Li(ray, sampler, st)), r.max_sample_luminance);
      if (per_sample) { per_sample[3 * i] = L.x; per_sample[3 * i + 1] = L.y; per_sample[3 * i + 2] = L.z; }
      part = part + L;  // FilmTile::AddSample with the box filter: weight 1, this pixel only
      i++;
    }
    sum = sum + part;  // the chunks' partial sums in chunk order
  }
  // Film::merge_film_tile (film.rs:313-326): xyz += to_xyz(contrib_sum); weight += filter_weight_sum (= spp)
  float rgb[3] = {sum.x, sum.y, sum.z}, xyz[3];
  rgb_to_xyz(rgb, xyz);
  out_xyzw[0] = xyz[0]; out_xyzw[1] = xyz[1]; out_xyzw[2] = xyz[2]; out_xyzw[3] = (float)(r.spp_x * r.spp_y);
}

}  // namespace orc

using namespace orc;

struct orc_scene {
  Scene s;
};

extern "C" {

int orc_sobol_dims(void) { return kSobolDims; }
void orc_debug_own_box_rule(int on) { Scene::own_box_rule_on() = on != 0; }
// Triangle::Intersect (Moeller-Trumbore + the own-box rule) of ray i against triangle tri[i] alone: ok[i], and t[i] where accepted
void orc_tri_accepts(const orc_scene *s, int64_t n, const float *o, const float *d, const float *tmax, const uint32_t *tri, uint8_t *ok, float *t) {
  for (int64_t i = 0; i < n; i++) {
    Ray r;
    r.o = v3(o[3 * i], o[3 * i + 1], o[3 * i + 2]);
    r.d = v3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    r.tmax = tmax[i];
    float tt = 0.f, u, v;
    ok[i] = tri[i] < s->s.n_tris() && s->s.tri_intersect(tri[i], r, &tt, &u, &v) ? 1 : 0;
    t[i] = tt;
  }
}
// Halton sampler: u and the integer head of dimension d for indices 0 .. n - 1 under `key` in a frame whose largest sample index is
// spp_mask (tests of the radical inverse); returns b^D, the head's modulus (0 for base 2)
uint32_t orc_halton_points(uint32_t d, uint32_t key, uint32_t spp_mask, uint32_t n, float *u, uint32_t *v) {
  uint32_t pw = 0;
  for (uint32_t i = 0; i < n; i++) v[i] = HaltonSampler::scrambled_radical_inverse(d, i, key, spp_mask, &u[i], &pw);
  return pw;
}
void orc_sobol_matrix(int dim, uint32_t *out52) { sobol_matrix(dim, out52); }
void orc_sobol_points(uint32_t key_seed, uint32_t n, float *out2n) {
  // the first n points of the unscrambled (0,2)-sequence (dimensions 1 and 2), for the net-property test
  uint32_t c1[kSobolColumns];
  sobol_matrix(1, c1);
  (void)key_seed;
  for (uint32_t i = 0; i < n; i++) {
    uint32_t x = 0, y = 0;
    for (uint32_t b = 0; b < 32 && (i >> b) != 0u; b++)
      if ((i >> b) & 1u) { x ^= 1u << (31 - b); y ^= c1[b]; }
    out2n[2 * i] = (float)x * 2.3283064365386963e-10f;
    out2n[2 * i + 1] = (float)y * 2.3283064365386963e-10f;
  }
}

void orc_rng_default_u32(uint32_t *out, int n) { Rng r; for (int i = 0; i < n; i++) out[i] = r.uniform_u32(); }
void orc_rng_default_float(float *out, int n) { Rng r; for (int i = 0; i < n; i++) out[i] = r.uniform_float(); }
void orc_rng_default_threshold(uint32_t b, uint32_t *out, int n) { Rng r; for (int i = 0; i < n; i++) out[i] = r.uniform_u32_threshold(b); }
void orc_rng_seq_u32(uint64_t seq, uint32_t *out, int n) { Rng r(seq); for (int i = 0; i < n; i++) out[i] = r.uniform_u32(); }
void orc_rng_seq_float(uint64_t seq, float *out, int n) { Rng r(seq); for (int i = 0; i < n; i++) out[i] = r.uniform_float(); }

void orc_film_cropped_bounds(int xres, int yres, const float crop[4], int32_t out[4]) { film_cropped_bounds(xres, yres, crop, out); }

// Film::get_sample_bounds, film.rs:166-175
void orc_film_sample_bounds(int xres, int yres, const float crop[4], float rx, float ry, int32_t out[4]) {
  int32_t c[4];
  film_cropped_bounds(xres, yres, crop, c);
  out[0] = (int32_t)std::floor(((float)c[0] + 0.5f) - rx);
  out[1] = (int32_t)std::floor(((float)c[1] + 0.5f) - ry);
  out[2] = (int32_t)std::ceil(((float)c[2] - 0.5f) + rx);
  out[3] = (int32_t)std::ceil(((float)c[3] - 0.5f) + ry);
}

// Film::get_film_tile, film.rs:264-281 (bounds are x0 y0 x1 y1)
void orc_film_tile_bounds(int xres, int yres, const float crop[4], float rx, float ry, const int32_t sb[4], int32_t out[4]) {
  int32_t c[4];
  film_cropped_bounds(xres, yres, crop, c);
  int32_t p0x = (int32_t)std::ceil(((float)sb[0] - 0.5f) - rx);
  int32_t p0y = (int32_t)std::ceil(((float)sb[1] - 0.5f) - ry);
  int32_t p1x = (int32_t)(std::floor(((float)sb[2] - 0.5f) + rx) + 1.f);
  int32_t p1y = (int32_t)(std::floor(((float)sb[3] - 0.5f) + ry) + 1.f);
  out[0] = std::max(p0x, c[0]); out[1] = std::max(p0y, c[1]);
  out[2] = std::min(p1x, c[2]); out[3] = std::min(p1y, c[3]);
}

// Film::get_physical_extent, film.rs:218-227
void orc_film_physical_extent(int xres, int yres, float diagonal_mm, float out[4]) {
  float diag = diagonal_mm * 0.001f;
  float aspect = (float)yres / (float)xres;
  float x = std::sqrt(diag * diag / (1.f + aspect * aspect));
  float y = aspect * x;
  out[0] = -x / 2.f; out[1] = -y / 2.f; out[2] = x / 2.f; out[3] = y / 2.f;
}

void orc_rgb_to_xyz(const float rgb[3], float xyz[3]) { rgb_to_xyz(rgb, xyz); }
void orc_xyz_to_rgb(const float xyz[3], float rgb[3]) { xyz_to_rgb(xyz, rgb); }

// Film::write_image's per-pixel arithmetic, film.rs:346-372 (no splats: splat_xyz is always 0)
void orc_film_write_rgb(const float *xyzw, int64_t n_px, float scale, float *rgb) {
  for (int64_t i = 0; i < n_px; i++) {
    float c[3];
    xyz_to_rgb(xyzw + 4 * i, c);
    float w = xyzw[4 * i + 3];
    if (w != 0.f) {
      float inv = 1.f / w;
      for (int k = 0; k < 3; k++) { float v = c[k] * inv; c[k] = v > 0.f ? v : 0.f; }
    }
    for (int k = 0; k < 3; k++) rgb[3 * i + k] = c[k] * scale;
  }
}

void orc_look_at(const float pos[3], const float look[3], const float up[3], float m[16], float m_inv[16]) {
  Mat4 w2c, c2w;
  look_at(v3(pos[0], pos[1], pos[2]), v3(look[0], look[1], look[2]), v3(up[0], up[1], up[2]), &w2c, &c2w);
  memcpy(m, w2c.m, 64);
  memcpy(m_inv, c2w.m, 64);
}
void orc_matrix_transpose(const float m[16], float out[16]) { Mat4 a; memcpy(a.m, m, 64); Mat4 r = transpose(a); memcpy(out, r.m, 64); }
void orc_matrix_inverse(const float m[16], float out[16]) { Mat4 a; memcpy(a.m, m, 64); Mat4 r = inverse(a); memcpy(out, r.m, 64); }
void orc_matrix_mul(const float a[16], const float b[16], float out[16]) { Mat4 x, y; memcpy(x.m, a, 64); memcpy(y.m, b, 64); Mat4 r = mul(x, y); memcpy(out, r.m, 64); }
int orc_quadratic(float a, float b, float c, float *t0, float *t1) { return quadratic(a, b, c, t0, t1) ? 1 : 0; }
float orc_clamp_f(float v, float lo, float hi) { return clamp_ref(v, lo, hi); }
long orc_clamp_i(long v, long lo, long hi) { return clamp_ref(v, lo, hi); }
float orc_lerp(float t, float v1, float v2) { return lerp_ref(t, v1, v2); }
int orc_solve_2x2(const float a[4], const float b[2], float x[2]) {
  const float m[2][2] = {{a[0], a[1]}, {a[2], a[3]}};
  return solve_linear_system_2x2(m, b, x) ? 1 : 0;
}
float orc_gamma_correct(float v) { return gamma_correct(v); }
uint8_t orc_to_byte(float v) { return to_byte(v); }

orc_scene *orc_scene_create(const orc_scene_desc *d) {
  orc_scene *h = new orc_scene();
  Scene &s = h->s;
  s.P.resize(d->n_verts);
  for (uint32_t i = 0; i < d->n_verts; i++) s.P[i] = v3(d->P[3 * i], d->P[3 * i + 1], d->P[3 * i + 2]);
  s.idx.assign(d->idx, d->idx + 3 * (size_t)d->n_tris);
  s.mat_id.assign(d->mat_id, d->mat_id + d->n_tris);
  s.mats.assign(d->mats, d->mats + d->n_mats);
  if (d->n_spheres) s.spheres.assign(d->spheres, d->spheres + d->n_spheres);
  if (d->n_textures) s.textures.assign(d->textures, d->textures + d->n_textures);
  if (d->tri_uv) s.tri_uv.assign(d->tri_uv, d->tri_uv + 6 * (size_t)d->n_tris);
  // light list: explicit lights in order, then every emissive triangle in index order
  for (uint32_t i = 0; i < d->n_lights; i++) {
    LightRec l{};
    l.type = d->lights[i].type;
    l.p0 = v3(d->lights[i].p[0], d->lights[i].p[1], d->lights[i].p[2]);
    l.c = v3(d->lights[i].c[0], d->lights[i].c[1], d->lights[i].c[2]);
    s.lights.push_back(l);
    if (l.type == 2) { s.le_infinite = s.le_infinite + l.c; s.has_infinite = true; }
  }
  for (uint32_t t = 0; t < d->n_tris; t++) {
    const orc_material &m = s.mats[s.mat_id[t]];
    if (m.le[0] > 0.f || m.le[1] > 0.f || m.le[2] > 0.f) {
      LightRec l{};
      l.type = 3;
      s.tri_verts(t, &l.p0, &l.p1, &l.p2);
      l.c = v3(m.le[0], m.le[1], m.le[2]);
      Vec3 cr = cross(l.p1 - l.p0, l.p2 - l.p0);
      float len = length(cr);
      l.n = cr / len;
      l.area = 0.5f * len;
      s.lights.push_back(l);
    }
  }
  setup_camera(s, d->cam_to_world, d->fov, d->xres, d->yres, d->crop);
  s.build_bvh();
  return h;
}
void orc_scene_destroy(orc_scene *s) { delete s; }
uint32_t orc_bvh_node_count(const orc_scene *s) { return (uint32_t)s->s.nodes.size(); }
uint32_t orc_bvh_depth(const orc_scene *s) { return s->s.depth; }
void orc_bvh_export(const orc_scene *s, uint32_t *nodes, uint32_t *order) {
  if (!s->s.nodes.empty()) memcpy(nodes, s->s.nodes.data(), s->s.nodes.size() * 32);
  if (!s->s.order.empty()) memcpy(order, s->s.order.data(), s->s.order.size() * 4);
}
uint32_t orc_light_count(const orc_scene *s) { return (uint32_t)s->s.lights.size(); }

// A probe ray (pbrt_hip_intersect / pbrt_hip_occluded) with a non-finite origin or direction, or a NaN tmax, hits nothing -- which is what
// the arithmetic makes of it anyway (every test against a NaN fails) -- and is not walked at all: with NaN slabs nothing prunes, and a
// batch of such rays would keep a GPU busy for seconds per ray on a large scene.  (The renderer's own rays are finite.)
static bool probe_ray_is_sane(const float *o, const float *d, float tmax) {
  return std::isfinite(o[0]) && std::isfinite(o[1]) && std::isfinite(o[2]) && std::isfinite(d[0]) && std::isfinite(d[1]) && std::isfinite(d[2]) &&
         !std::isnan(tmax);
}
void orc_intersect(const orc_scene *sc, int64_t n, const float *o, const float *d, const float *tmax, float *t,
                   uint32_t *prim, float *b1, float *b2, uint64_t *counters, int brute_force) {
  const Scene &s = sc->s;
  Counters c;
  for (int64_t i = 0; i < n; i++) {
    Ray r{v3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), v3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), tmax[i]};
    if (!probe_ray_is_sane(o + 3 * i, d + 3 * i, tmax[i])) {
      const Hit miss;
      t[i] = miss.t; prim[i] = miss.prim; b1[i] = miss.b1; b2[i] = miss.b2;
      continue;
    }
    Hit h = brute_force ? s.IntersectBrute(r) : s.Intersect(r, &c);
    t[i] = h.t; prim[i] = h.prim; b1[i] = h.b1; b2[i] = h.b2;
  }
  if (counters) { counters[0] = c.nodes; counters[1] = c.tris; }
}
void orc_occluded(const orc_scene *sc, int64_t n, const float *o, const float *d, const float *tmax, uint8_t *hit,
                  int brute_force) {
  const Scene &s = sc->s;
  for (int64_t i = 0; i < n; i++) {
    Ray r{v3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), v3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), tmax[i]};
    if (!probe_ray_is_sane(o + 3 * i, d + 3 * i, tmax[i])) { hit[i] = 0; continue; }
    hit[i] = (brute_force ? s.IntersectPBrute(r) : s.IntersectP(r, nullptr)) ? 1 : 0;
  }
}
void orc_camera_ray(const orc_scene *sc, float fx, float fy, float o[3], float d[3]) {
  Ray r = sc->s.camera_ray(fx, fy);
  o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z;
  d[0] = r.d.x; d[1] = r.d.y; d[2] = r.d.z;
}
void orc_pixel_samples(const orc_scene *sc, const orc_render_desc *r, int x, int y, float *out) {
  orc::g_debug_li = std::getenv("ORC_DEBUG_LI") != nullptr;
  PathIntegrator integ(sc->s, r->max_depth, r->integrator == 1, r->integrator == 2);
  RayStats st;
  float px[4];
  render_pixel(sc->s, integ, *r, x, y, px, out, st);
  orc::g_debug_li = false;
}

// the render loop of both film paths: acc == nullptr: the default box filter, pixels straight into `film`; else the
// fixed-point accumulators of DESIGN.md 3.11 over the sample bounds
static int render_any(const orc_scene *sc, const orc_render_desc *r, float *film, int64_t *acc, orc_stats *out, int n_threads) {
  const Scene &s = sc->s;
  if (r->spp_x == 0 || r->spp_y == 0 || r->world_size == 0 || r->rank >= r->world_size || r->sampler > 3 || r->integrator > 2) return -1;
  if (!(filter_radius(r->filter_xwidth) > 0.f) || !(filter_radius(r->filter_ywidth) > 0.f)) return -1;
  PathIntegrator integ(s, r->max_depth, r->integrator == 1, r->integrator == 2);
  const WideFilter wf = wide_filter(s, *r);
  // the 64x64 super-tiles cover the SAMPLE bounds (= the cropped window for the default filter)
  const int x0 = acc ? wf.sb[0] : s.cropped[0], y0 = acc ? wf.sb[1] : s.cropped[1];
  const int x1 = acc ? wf.sb[2] : s.cropped[2], y1 = acc ? wf.sb[3] : s.cropped[3];
  const int W = x1 - x0, H = y1 - y0;
  if (W <= 0 || H <= 0 || s.cropped[2] <= s.cropped[0] || s.cropped[3] <= s.cropped[1]) return 0;
  // 16x16 work tiles inside the 64x64 super-tiles this rank owns
  const int stx = (W + 63) / 64;
  const int ntx = (W + 15) / 16, nty = (H + 15) / 16;
  std::atomic<int> next(0);
  if (n_threads < 1) n_threads = 1;
  std::vector<RayStats> stats(n_threads);
  auto t_start = std::chrono::steady_clock::now();
  auto worker = [&](int tid) {
    RayStats &st = stats[tid];
    for (;;) {
      int tile = next.fetch_add(1);
      if (tile >= ntx * nty) break;
      int tx = tile % ntx, ty = tile / ntx;
      int super = (ty / 4) * stx + (tx / 4);
      if ((uint32_t)super % r->world_size != r->rank) continue;
      for (int yy = ty * 16; yy < std::min(ty * 16 + 16, H); yy++)
        for (int xx = tx * 16; xx < std::min(tx * 16 + 16, W); xx++)
          if (acc) render_pixel_wide(s, integ, *r, wf, x0 + xx, y0 + yy, acc, st);
          else render_pixel(s, integ, *r, x0 + xx, y0 + yy, film + 4 * ((size_t)yy * W + xx), nullptr, st);
    }
  };
  std::vector<std::thread> th;
  for (int i = 1; i < n_threads; i++) th.emplace_back(worker, i);
  worker(0);
  for (auto &t : th) t.join();
  auto t_end = std::chrono::steady_clock::now();
  if (out) {
    *out = orc_stats{};
    for (auto &st : stats) {
      out->camera_rays += st.camera; out->bounce_rays += st.bounce; out->shadow_rays += st.shadow;
      out->nodes_visited += st.c.nodes; out->tris_tested += st.c.tris;
    }
    out->seconds = std::chrono::duration<double>(t_end - t_start).count();
  }
  return 0;
}

int orc_render_acc(const orc_scene *sc, const orc_render_desc *r, int64_t *acc, orc_stats *out, int n_threads) {
  return render_any(sc, r, nullptr, acc, out, n_threads);
}
void orc_film_from_acc(const int64_t *acc, int64_t n_px, float *film) {
  for (int64_t i = 0; i < n_px; i++) film_from_acc(acc + 4 * i, film + 4 * i);
}
int orc_render(const orc_scene *sc, const orc_render_desc *r, float *film, orc_stats *out, int n_threads) {
  if (is_default_filter(*r)) return render_any(sc, r, film, nullptr, out, n_threads);
  // another radius: this rank's samples into accumulators, then converted (ranks combine by adding ACCUMULATORS)
  const Scene &s = sc->s;
  const int64_t n_px = (int64_t)std::max(0, s.cropped[2] - s.cropped[0]) * (int64_t)std::max(0, s.cropped[3] - s.cropped[1]);
  std::vector<int64_t> acc((size_t)(4 * n_px), 0);
  const int rc = render_any(sc, r, nullptr, acc.data(), out, n_threads);
  if (rc == 0) orc_film_from_acc(acc.data(), n_px, film);
  return rc;
}

}  // extern "C"
