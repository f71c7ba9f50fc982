// ray_pool.hip -- EXPERIMENT (round 4): a wave that keeps a POOL of rays in LDS and deals them to its
// lanes per pass, in a traversal-only kernel.
//
// Not part of the product: this translation unit includes pbrt_amd/csrc/kernels.hip whole (constants and device helpers are
// reused) and adds one kernel and one C entry point; probe.py builds it into its own shared library beside the product's and
// hands it the scene handle of the product library (the harness of tools/experiments/dual_ray).
//
// Idea.  In the product's walk a lane owns its ray: a node-step pass runs with the 38 of 64 lanes whose ray can step, a leaf pass
// with the 20 whose ray is parked at a leaf.  Here the rays live in LDS -- P slots per one-wave workgroup: origin, direction,
// inverse direction, best hit, current ref, stack depth; the stack itself in R rows of LDS per slot -- and every pass DEALS up to
// 64 rays of ONE state to the lanes (ballots over the slots' states, ranks by popcount, a 64-entry table in LDS), runs up to
// STEPS node steps (or one triangle test) for them with the product's arithmetic, and writes back what changed.  Results are the
// product's bit for bit (a ray's arithmetic does not depend on scheduling; tie rule of DESIGN.md 3.4).  What is measured: rays per
// second against intersect_kernel on the same rays, and the lanes per pass.
#include "../../../pbrt_amd/csrc/kernels.hip"
#include "../../../pbrt_amd/csrc/capi_internal.hpp"

namespace pbrt_hip {
namespace {

constexpr uint32_t kEmpty = 0xfffffffeu;  // a slot without a ray (cur): nothing to write out

// P slots (a multiple of 64, <= 256), R LDS stack rows per slot (row 0: the sentinel kDone), STEPS node steps per deal
template <uint32_t P, uint32_t R, int STEPS, int WAVES>
__global__ void __launch_bounds__(64, WAVES) pool_kernel(const DevScene S, const RayBatch B, const int any_hit, uint32_t *next_ray,
                                                         uint32_t *ovf_all, const uint32_t E, unsigned long long *probe) {
  constexpr uint32_t G = P / 64u;  // slots per lane when the states are classified
  __shared__ float4 s_o[P];    // origin, tmax
  __shared__ float4 s_d[P];    // direction, best t
  __shared__ float4 s_inv[P];  // inverse direction, b1
  __shared__ float s_b2[P];
  __shared__ uint32_t s_prim[P], s_cur[P], s_sp[P], s_any[P];
  __shared__ int32_t s_idx[P];
  __shared__ uint32_t s_stk[R][P];
  __shared__ uint32_t s_deal[64];
  const uint32_t lane = threadIdx.x & 63u;
  const char *quads = reinterpret_cast<const char *>(S.quads);
  const char *tris = reinterpret_cast<const char *>(S.tris);
  uint32_t *ovf = ovf_all + (size_t)blockIdx.x * P * E;  // [slot][entry beyond the LDS rows]
  const uint32_t n = (uint32_t)B.n;
  unsigned long long p_steps = 0, p_lanes = 0, p_leaf = 0, p_leaf_lanes = 0, p_deals = 0;
  for (uint32_t g = 0; g < G; g++) { s_cur[lane + 64u * g] = kEmpty; s_idx[lane + 64u * g] = -1; s_sp[lane + 64u * g] = 1u; s_stk[0][lane + 64u * g] = kDone; }
  __syncthreads();
  bool more = true;  // rays left in the batch (wave-uniform)

  auto push = [&](uint32_t slot, uint32_t &sp, uint32_t ref) {
    if (sp < R) s_stk[sp][slot] = ref; else ovf[(size_t)slot * E + (sp - R)] = ref;
    sp++;
  };
  auto pop = [&](uint32_t slot, uint32_t &sp) -> uint32_t {
    sp--;
    return sp < R ? s_stk[sp][slot] : ovf[(size_t)slot * E + (sp - R)];
  };

  for (;;) {
    // ---- the slots' states: S can step, L is parked at a leaf, F is finished or empty ----
    unsigned long long mS[G], mL[G], mF[G];
    uint32_t nS = 0, nL = 0, nF = 0;
#pragma unroll
    for (uint32_t g = 0; g < G; g++) {
      const uint32_t c = s_cur[lane + 64u * g];
      const bool fin = c == kDone || c == kEmpty;
      mS[g] = __ballot(!fin && !(c & kLeafRef));
      mL[g] = __ballot(!fin && (c & kLeafRef));
      mF[g] = __ballot(fin);
      nS += (uint32_t)__popcll(mS[g]); nL += (uint32_t)__popcll(mL[g]); nF += (uint32_t)__popcll(mF[g]);
    }
    if (nS == 0u && nL == 0u && !more) break;
    // the fullest pass first (a refill counts with the slots it can fill)
    const uint32_t cS = nS < 64u ? nS : 64u, cL = nL < 64u ? nL : 64u, cF = more ? (nF < 64u ? nF : 64u) : 0u;
    int pick = 0;
    if (cL > cS) pick = 1;
    if (cF > (cS > cL ? cS : cL)) pick = 2;
    // ---- deal: the first 64 slots of the picked state, slot j to lane rank(j) ----
    uint32_t before = 0u;
    const unsigned long long lt = lane == 0u ? 0ull : (~0ull >> (64u - lane));
#pragma unroll
    for (uint32_t g = 0; g < G; g++) {
      const unsigned long long m = pick == 0 ? mS[g] : (pick == 1 ? mL[g] : mF[g]);
      if ((m >> lane) & 1ull) {
        const uint32_t r = before + (uint32_t)__popcll(m & lt);
        if (r < 64u) s_deal[r] = lane + 64u * g;
      }
      before += (uint32_t)__popcll(m);
    }
    const uint32_t count = pick == 0 ? cS : (pick == 1 ? cL : cF);
    __syncthreads();
    const bool on = lane < count;
    const uint32_t slot = on ? s_deal[lane] : 0u;
    if (probe) p_deals++;

    if (pick == 2) {
      // ---- refill: a finished ray's result out, the next ray of the batch in ----
      uint32_t base = 0u;
      if (lane == 0u) base = atomicAdd(next_ray, count);
      base = (uint32_t)__builtin_amdgcn_readfirstlane(base);
      if (base + count >= n) more = false;
      if (on) {
        const int32_t old = s_idx[slot];
        if (old >= 0) {
          if (any_hit) B.occluded[old] = s_any[slot] == 3u ? 1 : 0;
          else { B.t[old] = s_d[slot].w; B.prim[old] = s_prim[slot]; B.b1[old] = s_inv[slot].w; B.b2[old] = s_b2[slot]; }
        }
        const uint32_t i = base + lane;
        if (i < n) {
          const V3 o = mk(B.o[3 * (size_t)i], B.o[3 * (size_t)i + 1], B.o[3 * (size_t)i + 2]), d = mk(B.d[3 * (size_t)i], B.d[3 * (size_t)i + 1], B.d[3 * (size_t)i + 2]);
          const V3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
          const float tmax = B.tmax[i];
          uint32_t cur = kDone;
          if (S.n_nodes) {
            const bool inside = o.x >= S.root_lo[0] && o.x <= S.root_hi[0] && o.y >= S.root_lo[1] && o.y <= S.root_hi[1] && o.z >= S.root_lo[2] && o.z <= S.root_hi[2];
            float tn;
            if (inside || box_test(S.root_lo[0], S.root_lo[1], S.root_lo[2], S.root_hi[0], S.root_hi[1], S.root_hi[2], o, inv, inv.x < 0.f, inv.y < 0.f, inv.z < 0.f, tmax, tn))
              cur = !(S.root_ref & kLeafRef) ? 0u : S.root_ref;
          }
          s_o[slot] = make_float4(o.x, o.y, o.z, tmax);
          s_d[slot] = make_float4(d.x, d.y, d.z, kInf);
          s_inv[slot] = make_float4(inv.x, inv.y, inv.z, 0.f);
          s_b2[slot] = 0.f; s_prim[slot] = kNoPrim; s_any[slot] = any_hit ? 1u : 0u;
          s_idx[slot] = (int32_t)i; s_cur[slot] = cur; s_sp[slot] = 1u;
        } else {
          s_idx[slot] = -1; s_cur[slot] = kEmpty;
        }
      }
    } else if (pick == 0) {
      // ---- node steps: the product's step (kernels.hip trav_run, production instantiation), STEPS of them for the dealt rays ----
      float4 ro = make_float4(0, 0, 0, 0), rd = ro, ri = ro;
      uint32_t cur = kDone, sp = 1u;
      if (on) { ro = s_o[slot]; rd = s_d[slot]; ri = s_inv[slot]; cur = s_cur[slot]; sp = s_sp[slot]; }
      const bool negx = ri.x < 0.f, negy = ri.y < 0.f, negz = ri.z < 0.f;
#pragma unroll
      for (int rep = 0; rep < STEPS; rep++) {
        const bool act = on && cur != kDone && !(cur & kLeafRef);
        if (probe) { const unsigned long long m = __ballot(act); if (m) { p_steps++; p_lanes += __popcll(m); } }
        if (act) {
          const uint32_t off = cur;
          wave_prio(PBRT_PRIO_FETCH);
          const uint4 W0 = *reinterpret_cast<const uint4 *>(quads + off);
          const uint4 W1 = *reinterpret_cast<const uint4 *>(quads + off + 16u);
          const uint4 W2 = *reinterpret_cast<const uint4 *>(quads + off + 32u);
          const uint4 W3 = *reinterpret_cast<const uint4 *>(quads + off + 48u);
          wave_prio(PBRT_PRIO_ARITH);
          const float tfar = fminf(rd.w, ro.w);
          const float gx = (ro.x - __uint_as_float(W0.x)) * ri.x, gy = (ro.y - __uint_as_float(W0.y)) * ri.y;
          const float gz = (ro.z - __uint_as_float(W0.z)) * ri.z;
          constexpr float kMargin = 0x1.8p-22f;
          const f32x2 gxx = {__builtin_fmaf(fabsf(gx), kMargin, gx), __builtin_fmaf(-fabsf(gx), kMargin, gx)};
          const f32x2 gyy = {__builtin_fmaf(fabsf(gy), kMargin, gy), __builtin_fmaf(-fabsf(gy), kMargin, gy)};
          const f32x2 gzz = {__builtin_fmaf(fabsf(gz), kMargin, gz), __builtin_fmaf(-fabsf(gz), kMargin, gz)};
          const float cix = __uint_as_float(W0.w) * ri.x, ciy = __uint_as_float(W2.z) * ri.y, ciz = __uint_as_float(W2.w) * ri.z;
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
          if (hit[3] && !n3) push(slot, sp, W3.w);
          if (hit[2] && !n2) push(slot, sp, W3.z);
          if (hit[1] && !n1) push(slot, sp, W3.y);
          if (hit[0] && !n0) push(slot, sp, W3.x);
          cur = any ? nearest : pop(slot, sp);
        }
      }
      if (on) { s_cur[slot] = cur; s_sp[slot] = sp; }
    } else {
      // ---- one triangle test for every dealt ray (leaves hold single triangles in the device-built tree; a leaf of more is
      // worked off one triangle per pass: its ref's count and first slot are rewritten) ----
      if (probe) { p_leaf++; p_leaf_lanes += count; }
      if (on) {
        const float4 ro = s_o[slot], rd = s_d[slot];
        uint32_t cur = s_cur[slot], sp = s_sp[slot], any = s_any[slot];
        const uint32_t cnt = (cur >> 24) & 0x7fu, first = cur & 0xffffffu;
        bool stop = false;
        if (cnt) {
          wave_prio(PBRT_PRIO_FETCH);
          const float4 a = *reinterpret_cast<const float4 *>(tris + first * (16u * kTriStride));
          const float4 b = *reinterpret_cast<const float4 *>(tris + first * (16u * kTriStride) + 16u);
          const float4 c = *reinterpret_cast<const float4 *>(tris + first * (16u * kTriStride) + 32u);
          wave_prio(PBRT_PRIO_ARITH);
          const V3 o = mk(ro.x, ro.y, ro.z), d = mk(rd.x, rd.y, rd.z);
          const V3 p0 = xyz(a);
          const V3 e1 = xyz(b) - p0, e2 = xyz(c) - p0;
          const V3 pv = cross(d, e2);
          const float det = dot(e1, pv);
          const float idet = 1.0f / det;
          const V3 tv = o - p0;
          const float u = dot(tv, pv) * idet;
          const V3 qv = cross(tv, e1);
          const float v = dot(d, qv) * idet;
          const float th = dot(e2, qv) * idet;
          const bool valid = !(fabsf(det) < 1e-8f) && (u >= 0.f) && (v >= 0.f) && (u + v <= 1.0f) && (th > kRayTMin) && (th < ro.w);
          const uint32_t id = __float_as_uint(a.w);
          const bool occl = valid && any != 0u;
          const bool closer = valid && any == 0u && (th < rd.w || (th == rd.w && id < s_prim[slot]));
          if (occl) { s_any[slot] = 3u; stop = true; }
          if (closer) { s_d[slot].w = th; s_prim[slot] = id; s_inv[slot].w = u; s_b2[slot] = v; }
        }
        if (stop) { cur = kDone; sp = 1u; }
        else if (cnt > 1u) cur = kLeafRef | ((cnt - 1u) << 24) | (first + 1u);
        else cur = pop(slot, sp);
        s_cur[slot] = cur; s_sp[slot] = sp;
      }
    }
    __syncthreads();
  }
  // ---- the last finished rays ----
  for (uint32_t g = 0; g < G; g++) {
    const uint32_t slot = lane + 64u * g;
    const int32_t old = s_idx[slot];
    if (old >= 0 && s_cur[slot] == kDone) {
      if (any_hit) B.occluded[old] = s_any[slot] == 3u ? 1 : 0;
      else { B.t[old] = s_d[slot].w; B.prim[old] = s_prim[slot]; B.b1[old] = s_inv[slot].w; B.b2[old] = s_b2[slot]; }
    }
  }
  if (probe && lane == 0) {
    atomicAdd(&probe[0], p_steps); atomicAdd(&probe[1], p_lanes); atomicAdd(&probe[2], p_leaf); atomicAdd(&probe[3], p_leaf_lanes);
    atomicAdd(&probe[4], p_deals);
  }
}

}  // namespace
}  // namespace pbrt_hip

// C entry of the experiment: rays already on the host; returns the kernel's best time of three launches.
// mode 0: the product's intersect_kernel; 1: pool of 128 slots, 12 LDS stack rows, 3 steps per deal; 2: 128 / 12 / 2 steps;
// 3: 192 slots / 8 rows / 3 steps; 4: 128 / 12 / 4 steps; 5: 64 slots (one ray per lane, dealt: the control for the dealing's cost)
extern "C" int exp_pool_intersect(pbrt_hip_scene *s, int64_t n, const float *o, const float *d, const float *tmax, float *t, uint32_t *prim,
                                  float *b1, float *b2, uint8_t *occ, int any_hit, int mode, float *ms_out, unsigned long long *probe_out) {
  using namespace pbrt_hip;
  if (!s || n <= 0 || n >= (1ll << 31)) return 1;
#define X_TRY(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); return 2; } } while (0)
  X_TRY(hipSetDevice(s->device));
  float *d_o, *d_d, *d_tmax, *d_t, *d_b1, *d_b2;
  uint32_t *d_prim, *d_ovf, *d_next;
  uint8_t *d_occ;
  unsigned long long *d_probe;
  X_TRY(hipMalloc(&d_o, 12 * n)); X_TRY(hipMalloc(&d_d, 12 * n)); X_TRY(hipMalloc(&d_tmax, 4 * n));
  X_TRY(hipMalloc(&d_t, 4 * n)); X_TRY(hipMalloc(&d_b1, 4 * n)); X_TRY(hipMalloc(&d_b2, 4 * n)); X_TRY(hipMalloc(&d_prim, 4 * n));
  X_TRY(hipMalloc(&d_occ, n)); X_TRY(hipMalloc(&d_probe, 64)); X_TRY(hipMalloc(&d_next, 4));
  X_TRY(hipMemcpy(d_o, o, 12 * n, hipMemcpyHostToDevice)); X_TRY(hipMemcpy(d_d, d, 12 * n, hipMemcpyHostToDevice));
  X_TRY(hipMemcpy(d_tmax, tmax, 4 * n, hipMemcpyHostToDevice));
  const uint32_t E = s->dev.quad_stack_need + 4u;  // entries beyond the LDS rows, per slot: generous
  const uint32_t blocks = s->n_cu * 10u;           // about what the device holds at once (LDS: 9-10 workgroups of 16 KB per CU)
  X_TRY(hipMalloc(&d_ovf, (size_t)blocks * 256 * E * 4));
  RayBatch B{};
  B.o = d_o; B.d = d_d; B.tmax = d_tmax; B.n = n; B.t = d_t; B.prim = d_prim; B.b1 = d_b1; B.b2 = d_b2; B.occluded = d_occ;
  B.min_walkers = 36; B.min_parked = 16; B.stack_overflow = d_ovf; B.stack_overflow_entries = E;
  hipEvent_t e0, e1;
  X_TRY(hipEventCreate(&e0)); X_TRY(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    X_TRY(hipMemset(d_probe, 0, 64));
    X_TRY(hipMemset(d_next, 0, 4));
    X_TRY(hipEventRecord(e0, nullptr));
    if (mode == 0) {
      RayBatch B0 = B;
      X_TRY(hipFree(d_ovf));
      X_TRY(hipMalloc(&d_ovf, (size_t)4096 * 4 * 2 * E * 64 * 4));
      B0.stack_overflow = d_ovf;
      B0.stack_overflow_entries = 2 * E;
      X_TRY(launch_intersect(s->dev, B0, any_hit != 0, s->bvh.depth, nullptr));
    } else {
      const dim3 grid(blocks), block(64);
      unsigned long long *pr = probe_out ? d_probe : nullptr;
      if (mode == 1) hipLaunchKernelGGL((pool_kernel<128u, 12u, 3, 2>), grid, block, 0, nullptr, s->dev, B, any_hit, d_next, d_ovf, E, pr);
      else if (mode == 2) hipLaunchKernelGGL((pool_kernel<128u, 12u, 2, 2>), grid, block, 0, nullptr, s->dev, B, any_hit, d_next, d_ovf, E, pr);
      else if (mode == 3) hipLaunchKernelGGL((pool_kernel<192u, 8u, 3, 2>), grid, block, 0, nullptr, s->dev, B, any_hit, d_next, d_ovf, E, pr);
      else if (mode == 4) hipLaunchKernelGGL((pool_kernel<128u, 12u, 4, 2>), grid, block, 0, nullptr, s->dev, B, any_hit, d_next, d_ovf, E, pr);
      else hipLaunchKernelGGL((pool_kernel<64u, 24u, 3, 4>), grid, block, 0, nullptr, s->dev, B, any_hit, d_next, d_ovf, E, pr);
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
  (void)hipFree(d_occ); (void)hipFree(d_probe); (void)hipFree(d_ovf); (void)hipFree(d_next);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return 0;
}
