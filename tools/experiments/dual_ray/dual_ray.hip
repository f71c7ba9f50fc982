// dual_ray.hip -- EXPERIMENT (round 3): TWO rays per lane in the traversal-only kernel.
//
// Not part of the product: this translation unit includes pbrt_amd/csrc/kernels.hip whole (its device functions are
// reused as they are) and adds one kernel and one C entry point; tools/experiments/dual_ray/probe.py builds it into its
// own shared library beside the product's and hands it the scene handle of the product library.
//
// Idea: half of a wave's lanes idle in every node-step pass (a lane whose ray is parked at a leaf, or whose walk is over,
// waits).  Here a lane owns two rays, each with its own walk state in registers and its own half of the lane's LDS stack:
// before every batch of node steps a lane whose ACTIVE ray cannot step swaps in its SPARE one (v_swap_b32 under EXEC: one
// instruction per register).  Results are the product kernel's bit for bit (a ray's arithmetic does not depend on
// scheduling); what is measured is rays per second against intersect_kernel on the same rays.
#include "../../../pbrt_amd/csrc/kernels.hip"
#include "../../../pbrt_amd/csrc/capi_internal.hpp"

namespace pbrt_hip {
namespace {

struct Ctx {            // one ray of a lane
  V3 o, d, inv;
  float tmax, ht, hb1, hb2;
  uint32_t hprim, cur, sp, any, base;  // base: LDS address of row 0 of this ray's stack
  int32_t idx;                         // index of the ray in the batch, -1: none
};

__device__ __forceinline__ void swap_reg(float &a, float &b) { asm volatile("v_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap_reg(uint32_t &a, uint32_t &b) { asm volatile("v_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap_reg(int32_t &a, int32_t &b) { asm volatile("v_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap_ctx(Ctx &a, Ctx &b) {
  swap_reg(a.o.x, b.o.x); swap_reg(a.o.y, b.o.y); swap_reg(a.o.z, b.o.z);
  swap_reg(a.d.x, b.d.x); swap_reg(a.d.y, b.d.y); swap_reg(a.d.z, b.d.z);
  swap_reg(a.inv.x, b.inv.x); swap_reg(a.inv.y, b.inv.y); swap_reg(a.inv.z, b.inv.z);
  swap_reg(a.tmax, b.tmax); swap_reg(a.ht, b.ht); swap_reg(a.hb1, b.hb1); swap_reg(a.hb2, b.hb2);
  swap_reg(a.hprim, b.hprim); swap_reg(a.cur, b.cur); swap_reg(a.sp, b.sp); swap_reg(a.any, b.any); swap_reg(a.base, b.base);
  swap_reg(a.idx, b.idx);
}
__device__ __forceinline__ bool can_step(const Ctx &c) { return c.cur != kDone && !(c.cur & kLeafRef); }
__device__ __forceinline__ bool is_parked(const Ctx &c) { return c.cur != kDone && (c.cur & kLeafRef) != 0u; }
__device__ __forceinline__ bool is_finished(const Ctx &c) { return c.cur == kDone && c.idx >= 0; }

// stack of R rows per ray: rows 0 .. R-2 in LDS (row 0 the sentinel), deeper entries in HBM (rare)
template <uint32_t R>
__device__ __forceinline__ void push_d(Ctx &c, uint32_t *ovf, uint32_t E, uint32_t stk0, uint32_t ref) {
  const uint32_t e = (c.sp - c.base) / kRowBytes;
  if (e < R - 1u) lds_store(c.sp, ref);
  else ovf[(((c.base - stk0) / (R * kRowBytes)) * E + (e - (R - 1u))) * 64u + lane_here()] = ref;
  c.sp += kRowBytes;
}
template <uint32_t R>
__device__ __forceinline__ uint32_t pop_d(Ctx &c, const uint32_t *ovf, uint32_t E, uint32_t stk0) {
  c.sp -= kRowBytes;
  const uint32_t e = (c.sp - c.base) / kRowBytes;
  return e < R - 1u ? lds_load(c.sp) : ovf[(((c.base - stk0) / (R * kRowBytes)) * E + (e - (R - 1u))) * 64u + lane_here()];
}

template <bool SPH, int STEPS, uint32_t R, bool DUAL, int WAVES = 4>
__global__ void __launch_bounds__(256, WAVES) intersect_dual_kernel(const DevScene S, const RayBatch B, const int any_hit, const uint32_t min_done,
                                                                unsigned long long *probe) {
  __shared__ uint32_t lds_stack[4][DUAL ? 2 * R : R][64];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t stk0 = lds_addr(&lds_stack[wave][0][lane]);
  const uint32_t E = B.stack_overflow_entries;
  uint32_t *ovf = B.stack_overflow + ((size_t)blockIdx.x * 4u + (uint32_t)__builtin_amdgcn_readfirstlane(wave)) * (2u * E) * 64u;
  const char *quads = reinterpret_cast<const char *>(S.quads);
  const char *tris = reinterpret_cast<const char *>(S.tris);
  const int32_t stride = (int32_t)(gridDim.x * 256u);
  int32_t next = (int32_t)(blockIdx.x * 256u + threadIdx.x);
  const int32_t n = (int32_t)B.n;
  Ctx A, Z;
  A.o = A.d = A.inv = Z.o = Z.d = Z.inv = mk(0.f, 0.f, 1.f);
  A.tmax = Z.tmax = 0.f;
  A.ht = Z.ht = kInf; A.hb1 = A.hb2 = Z.hb1 = Z.hb2 = 0.f;
  A.hprim = Z.hprim = kNoPrim;
  A.cur = Z.cur = kDone;
  A.any = Z.any = 0u;
  A.base = stk0; Z.base = stk0 + R * kRowBytes;
  A.sp = A.base; Z.sp = Z.base;
  A.idx = Z.idx = -1;
  unsigned long long p_steps = 0, p_lanes = 0, p_leaf = 0, p_leaf_lanes = 0, p_swaps = 0;
  const TravTuning tune = {B.min_walkers, B.min_parked};

  auto serve = [&](Ctx &c) {  // a finished ray's result out, the lane's next ray in
    if (c.cur != kDone) return;
    if (c.idx >= 0) {
      if (any_hit) {
        B.occluded[c.idx] = c.any == 3u ? 1 : 0;
      } else {
        B.t[c.idx] = c.ht; B.prim[c.idx] = c.hprim; B.b1[c.idx] = c.hb1; B.b2[c.idx] = c.hb2;
      }
      c.idx = -1;
    }
    if (next < n) {
      c.idx = next;
      next += stride;
      c.o = mk(B.o[3 * c.idx], B.o[3 * c.idx + 1], B.o[3 * c.idx + 2]);
      c.d = mk(B.d[3 * c.idx], B.d[3 * c.idx + 1], B.d[3 * c.idx + 2]);
      c.inv = mk(1.0f / c.d.x, 1.0f / c.d.y, 1.0f / c.d.z);
      c.tmax = B.tmax[c.idx];
      c.any = any_hit ? 1u : 0u;
      c.ht = kInf; c.hprim = kNoPrim; c.hb1 = 0.f; c.hb2 = 0.f;
      lds_store(c.base, kDone);
      c.sp = c.base + kRowBytes;
      c.cur = kDone;
      if (S.n_nodes) {
        const bool inside = c.o.x >= S.root_lo[0] && c.o.x <= S.root_hi[0] && c.o.y >= S.root_lo[1] && c.o.y <= S.root_hi[1] &&
                            c.o.z >= S.root_lo[2] && c.o.z <= S.root_hi[2];
        float tn;
        if (inside || box_test(S.root_lo[0], S.root_lo[1], S.root_lo[2], S.root_hi[0], S.root_hi[1], S.root_hi[2], c.o, c.inv,
                               c.inv.x < 0.f, c.inv.y < 0.f, c.inv.z < 0.f, c.tmax, tn))
          c.cur = !(S.root_ref & kLeafRef) ? 0u : S.root_ref;
      }
    }
  };

  for (;;) {
    serve(A);
    if (DUAL) serve(Z);
    if (__ballot(A.idx >= 0 || (DUAL && Z.idx >= 0)) == 0ull) break;
    // ---- the walk: until enough lanes hold a finished ray, or nothing can move ----
    for (;;) {
      if (DUAL) {
        // the active ray should be one that can step; failing that, one that is parked (so that the leaf pass sees it)
        const bool sw = !can_step(A) && (can_step(Z) || (!is_parked(A) && is_parked(Z)));
        if (__ballot(sw) != 0ull) {
          if (sw) swap_ctx(A, Z);
          p_swaps++;
        }
      }
      const unsigned long long mstep = __ballot(can_step(A)), mpark = __ballot(is_parked(A));
      if (mstep == 0ull && mpark == 0ull) break;
      const uint32_t n_fin = (uint32_t)__popcll(__ballot(is_finished(A) || (DUAL && is_finished(Z))));
      if (n_fin >= min_done && (uint32_t)__popcll(mstep) * 64u < tune.min_walkers * 64u) break;
#pragma unroll
      for (int rep = 0; rep < STEPS; rep++) {
        if (probe) { const unsigned long long m = __ballot(can_step(A)); if (m) { p_steps++; p_lanes += __popcll(m); } }
        if (can_step(A)) {
          const uint32_t off = A.cur;
          wave_prio(PBRT_PRIO_FETCH);
          const uint4 W0 = *reinterpret_cast<const uint4 *>(quads + off);
          const uint4 W1 = *reinterpret_cast<const uint4 *>(quads + off + 16u);
          const uint4 W2 = *reinterpret_cast<const uint4 *>(quads + off + 32u);
          const uint4 W3 = *reinterpret_cast<const uint4 *>(quads + off + 48u);
          wave_prio(PBRT_PRIO_ARITH);
          const V3 o = A.o, inv = A.inv;
          const bool negx = inv.x < 0.f, negy = inv.y < 0.f, negz = inv.z < 0.f;
          const float tfar = fminf(A.ht, A.tmax);
          const float gx = (o.x - __uint_as_float(W0.x)) * inv.x, gy = (o.y - __uint_as_float(W0.y)) * inv.y;
          const float gz = (o.z - __uint_as_float(W0.z)) * inv.z;
          constexpr float kMargin = 0x1.8p-22f;
          const f32x2 gxx = {__builtin_fmaf(fabsf(gx), kMargin, gx), __builtin_fmaf(-fabsf(gx), kMargin, gx)};
          const f32x2 gyy = {__builtin_fmaf(fabsf(gy), kMargin, gy), __builtin_fmaf(-fabsf(gy), kMargin, gy)};
          const f32x2 gzz = {__builtin_fmaf(fabsf(gz), kMargin, gz), __builtin_fmaf(-fabsf(gz), kMargin, gz)};
          const float cix = __uint_as_float(W0.w) * inv.x, ciy = __uint_as_float(W2.z) * inv.y, ciz = __uint_as_float(W2.w) * inv.z;
          const uint32_t bnx = negx ? W1.w : W1.x, bfx = negx ? W1.x : W1.w;
          const uint32_t bny = negy ? W2.x : W1.y, bfy = negy ? W1.y : W2.x;
          const uint32_t bnz = negz ? W2.y : W1.z, bfz = negz ? W1.z : W2.y;
          const f32x2 cxx = {cix, cix}, cyy = {ciy, ciy}, czz = {ciz, ciz};
          float key[4];
          bool hit[4];
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const f32x2 qx = {(float)((bnx >> (8 * k)) & 0xffu), (float)((bfx >> (8 * k)) & 0xffu)};
            const f32x2 qy = {(float)((bny >> (8 * k)) & 0xffu), (float)((bfy >> (8 * k)) & 0xffu)};
            const f32x2 qz = {(float)((bnz >> (8 * k)) & 0xffu), (float)((bfz >> (8 * k)) & 0xffu)};
            const f32x2 tx = __builtin_elementwise_fma(qx, cxx, -gxx), ty = __builtin_elementwise_fma(qy, cyy, -gyy);
            const f32x2 tz = __builtin_elementwise_fma(qz, czz, -gzz);
            const float tn = fmaxf(fmaxf(tx.x, ty.x), fmaxf(tz.x, kRayTMin));
            const float tf = fminf(fminf(tx.y, ty.y), fminf(tz.y, tfar));
            hit[k] = tn <= tf * kBoxPad;
            key[k] = tn;
          }
#pragma unroll
          for (int k = 0; k < 4; k++) key[k] = hit[k] ? key[k] : __uint_as_float(0xffffffffu);
          const float kmin = fminf(fminf(key[0], key[1]), fminf(key[2], key[3]));
          const bool n0 = key[0] == kmin, n1 = !n0 && key[1] == kmin, n2 = !n0 && !n1 && key[2] == kmin;
          const bool n3 = !n0 && !n1 && !n2;
          const bool any = hit[0] || hit[1] || hit[2] || hit[3];
          const uint32_t nearest = n0 ? W3.x : (n1 ? W3.y : (n2 ? W3.z : W3.w));
          if (__builtin_expect(__ballot(A.sp - A.base >= (R - 5u) * kRowBytes) != 0ull, 0)) {  // some lane near the end of its LDS part
            if (hit[3] && !n3) push_d<R>(A, ovf, E, stk0, W3.w);
            if (hit[2] && !n2) push_d<R>(A, ovf, E, stk0, W3.z);
            if (hit[1] && !n1) push_d<R>(A, ovf, E, stk0, W3.y);
            if (hit[0] && !n0) push_d<R>(A, ovf, E, stk0, W3.x);
            A.cur = any ? nearest : pop_d<R>(A, ovf, E, stk0);
          } else {
            const uint32_t below = A.sp - kRowBytes, top = lds_load(below);
            const uint32_t nxt = any ? nearest : top;
            lds_store(A.sp, W3.w); A.sp += (hit[3] && !n3) ? kRowBytes : 0u;
            lds_store(A.sp, W3.z); A.sp += (hit[2] && !n2) ? kRowBytes : 0u;
            lds_store(A.sp, W3.y); A.sp += (hit[1] && !n1) ? kRowBytes : 0u;
            lds_store(A.sp, W3.x); A.sp += (hit[0] && !n0) ? kRowBytes : 0u;
            A.sp = any ? A.sp : below;
            A.cur = nxt;
          }
        }
      }
      // ---- leaf pass: every lane that holds a parked ray takes part (a parked SPARE ray is swapped in for it) ----
      const bool pA = is_parked(A), pZ = DUAL && is_parked(Z);
      const unsigned long long mleaf = __ballot(pA || pZ);
      if (mleaf != 0ull && ((uint32_t)__popcll(mleaf) >= tune.min_parked ||
                            (uint32_t)__popcll(mleaf) * 2u >= (uint32_t)__popcll(__ballot(can_step(A) || (DUAL && can_step(Z)))))) {
        if (DUAL && __ballot(!pA && pZ) != 0ull) {
          if (!pA && pZ) swap_ctx(A, Z);
          p_swaps++;
        }
        const bool parked = is_parked(A);
        const uint32_t cnt = parked ? (A.cur >> 24) & 0x7fu : 0u, first = A.cur & 0xffffffu;
        bool stop = false;
        for (uint32_t i = 0;; i++) {
          const unsigned long long m = __ballot(cnt > i && !stop);
          if (m == 0ull) break;
          if (probe) { p_leaf++; p_leaf_lanes += __popcll(m); }
          if (cnt > i && !stop) {
            const uint32_t slot = first + i;
            wave_prio(PBRT_PRIO_FETCH);
            const float4 a = *reinterpret_cast<const float4 *>(tris + slot * (16u * kTriStride));
            const float4 b = *reinterpret_cast<const float4 *>(tris + slot * (16u * kTriStride) + 16u);
            const float4 c = *reinterpret_cast<const float4 *>(tris + slot * (16u * kTriStride) + 32u);
            wave_prio(PBRT_PRIO_ARITH);
            const V3 p0 = xyz(a);
            const V3 e1 = xyz(b) - p0, e2 = xyz(c) - p0;
            const V3 pv = cross(A.d, e2);
            const float det = dot(e1, pv);
            const float idet = 1.0f / det;
            const V3 tv = A.o - p0;
            const float u = dot(tv, pv) * idet;
            const V3 qv = cross(tv, e1);
            const float v = dot(A.d, qv) * idet;
            const float th = dot(e2, qv) * idet;
            const bool valid = !(fabsf(det) < 1e-8f) && (u >= 0.f) && (v >= 0.f) && (u + v <= 1.0f) && (th > kRayTMin) && (th < A.tmax);
            const uint32_t id = __float_as_uint(a.w);
            const bool occl = valid && A.any != 0u;
            const bool closer = valid && A.any == 0u && (th < A.ht || (th == A.ht && id < A.hprim));
            A.any = occl ? 3u : A.any;
            stop = stop || occl;
            A.ht = closer ? th : A.ht;
            A.hprim = closer ? id : A.hprim;
            A.hb1 = closer ? u : A.hb1;
            A.hb2 = closer ? v : A.hb2;
          }
        }
        if (parked) {
          if (stop) { A.cur = kDone; A.sp = A.base; }
          else A.cur = pop_d<R>(A, ovf, E, stk0);
        }
      }
    }
  }
  if (probe && lane == 0) {
    atomicAdd(&probe[0], p_steps); atomicAdd(&probe[1], p_lanes); atomicAdd(&probe[2], p_leaf); atomicAdd(&probe[3], p_leaf_lanes);
    atomicAdd(&probe[4], p_swaps);
  }
}

}  // namespace
}  // namespace pbrt_hip

// C entry of the experiment: rays already on the host; returns the kernel's time.  mode 0: the product's intersect_kernel;
// 1: this kernel with one ray per lane (the control for its own structure); 2: two rays per lane; 3 / 4 / 5: one ray per lane
// at 6 / 8 / 5 resident waves per SIMD (26 / 20 / 31 LDS stack rows; deeper entries in HBM).
extern "C" int exp_dual_intersect(pbrt_hip_scene *s, int64_t n, const float *o, const float *d, const float *tmax, float *t, uint32_t *prim,
                                  float *b1, float *b2, uint8_t *occ, int any_hit, int mode, int steps, uint32_t min_done, uint32_t min_walkers,
                                  uint32_t min_parked, float *ms_out, unsigned long long *probe_out) {
  using namespace pbrt_hip;
  if (!s || n <= 0 || n >= (1ll << 31)) return 1;
#define X_TRY(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); return 2; } } while (0)
  X_TRY(hipSetDevice(s->device));
  float *d_o, *d_d, *d_tmax, *d_t, *d_b1, *d_b2;
  uint32_t *d_prim, *d_ovf;
  uint8_t *d_occ;
  unsigned long long *d_probe;
  X_TRY(hipMalloc(&d_o, 12 * n)); X_TRY(hipMalloc(&d_d, 12 * n)); X_TRY(hipMalloc(&d_tmax, 4 * n));
  X_TRY(hipMalloc(&d_t, 4 * n)); X_TRY(hipMalloc(&d_b1, 4 * n)); X_TRY(hipMalloc(&d_b2, 4 * n)); X_TRY(hipMalloc(&d_prim, 4 * n));
  X_TRY(hipMalloc(&d_occ, n)); X_TRY(hipMalloc(&d_probe, 64));
  X_TRY(hipMemset(d_probe, 0, 64));
  X_TRY(hipMemcpy(d_o, o, 12 * n, hipMemcpyHostToDevice)); X_TRY(hipMemcpy(d_d, d, 12 * n, hipMemcpyHostToDevice));
  X_TRY(hipMemcpy(d_tmax, tmax, 4 * n, hipMemcpyHostToDevice));
  constexpr uint32_t R = kQuadLdsStack / 2;
  const uint32_t E = s->dev.quad_stack_need + 4u;  // per ray, generous
  const size_t blocks = 4096;
  X_TRY(hipMalloc(&d_ovf, blocks * 4 * 2 * E * 64 * 4));
  RayBatch B{};
  B.o = d_o; B.d = d_d; B.tmax = d_tmax; B.n = n; B.t = d_t; B.prim = d_prim; B.b1 = d_b1; B.b2 = d_b2; B.occluded = d_occ;
  B.min_walkers = min_walkers; B.min_parked = min_parked; B.stack_overflow = d_ovf; B.stack_overflow_entries = E;
  hipEvent_t e0, e1;
  X_TRY(hipEventCreate(&e0)); X_TRY(hipEventCreate(&e1));
  int64_t nb = (n + 255) / 256;
  if (nb > (int64_t)blocks) nb = blocks;
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    X_TRY(hipMemset(d_probe, 0, 64));
    X_TRY(hipEventRecord(e0, nullptr));
    if (mode == 0) {
      RayBatch B0 = B;
      B0.stack_overflow_entries = 2 * E;  // (its own layout: one stack per lane)
      X_TRY(launch_intersect(s->dev, B0, any_hit != 0, s->bvh.depth, nullptr));
    } else {
      const dim3 grid((uint32_t)nb), block(256);
#define LAUNCH_D(STEPS_, DUAL_) hipLaunchKernelGGL((intersect_dual_kernel<false, STEPS_, (DUAL_ ? R : 2 * R), DUAL_>), grid, block, 0, nullptr, s->dev, B, any_hit, min_done, probe_out ? d_probe : nullptr)
      if (mode == 2) { if (steps == 1) LAUNCH_D(1, true); else if (steps == 2) LAUNCH_D(2, true); else LAUNCH_D(3, true); }
      else if (mode == 3) hipLaunchKernelGGL((intersect_dual_kernel<false, 3, 26u, false, 6>), grid, block, 0, nullptr, s->dev, B, any_hit, min_done, probe_out ? d_probe : nullptr);  // 6 waves per SIMD: 26 rows
      else if (mode == 4) hipLaunchKernelGGL((intersect_dual_kernel<false, 3, 20u, false, 8>), grid, block, 0, nullptr, s->dev, B, any_hit, min_done, probe_out ? d_probe : nullptr);  // 8 waves per SIMD: 20 rows
      else if (mode == 5) hipLaunchKernelGGL((intersect_dual_kernel<false, 3, 31u, false, 5>), grid, block, 0, nullptr, s->dev, B, any_hit, min_done, probe_out ? d_probe : nullptr);  // 5 waves per SIMD: 31 rows
      else { if (steps == 1) LAUNCH_D(1, false); else if (steps == 2) LAUNCH_D(2, false); else LAUNCH_D(3, false); }
      X_TRY(hipGetLastError());
    }
    X_TRY(hipEventRecord(e1, nullptr));
    X_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    X_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  *ms_out = best;
  if (any_hit) X_TRY(hipMemcpy(occ, d_occ, n, hipMemcpyDeviceToHost));
  else {
    X_TRY(hipMemcpy(t, d_t, 4 * n, hipMemcpyDeviceToHost)); X_TRY(hipMemcpy(prim, d_prim, 4 * n, hipMemcpyDeviceToHost));
    X_TRY(hipMemcpy(b1, d_b1, 4 * n, hipMemcpyDeviceToHost)); X_TRY(hipMemcpy(b2, d_b2, 4 * n, hipMemcpyDeviceToHost));
  }
  if (probe_out) X_TRY(hipMemcpy(probe_out, d_probe, 40, hipMemcpyDeviceToHost));
  (void)hipFree(d_o); (void)hipFree(d_d); (void)hipFree(d_tmax); (void)hipFree(d_t); (void)hipFree(d_b1); (void)hipFree(d_b2); (void)hipFree(d_prim);
  (void)hipFree(d_occ); (void)hipFree(d_probe); (void)hipFree(d_ovf);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return 0;
}
