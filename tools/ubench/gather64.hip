// Microbenchmark: throughput of random 64-byte record gathers on MI355X, for the access shapes the
// BVH step can use.  Build: hipcc -O3 --offload-arch=gfx950 gather64.hip -o gather64
//   A  every active lane loads its own record with 4 x global_load_dwordx4
//   B  quad-cooperative: load k gives the 4 lanes of a quad the 4 pieces of quad-lane k's record (registers)
//   C  as B through LDS-DMA (global_load_lds_dwordx4) + ds_read_b128 by the owner
//   D  every active lane loads 2 x dwordx4 (a 32-byte record)
// `active` = lanes per wave that take part (the traversal has ~30 of 64 stepping).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ void __launch_bounds__(64) gather(const char *tab, uint32_t nrec, int iters, uint32_t active_mask_lo, uint32_t active_mask_hi, float *out) {
  __shared__ __attribute__((aligned(16))) uint32_t buf[4 * 260];
  const uint32_t lane = threadIdx.x;
  const unsigned long long amask = ((unsigned long long)active_mask_hi << 32) | active_mask_lo;
  const bool act = (amask >> lane) & 1ull;
  uint32_t seed = blockIdx.x * 64u + lane;
  float acc = 0.f;
  for (int it = 0; it < iters; it++) {
    seed = hash(seed + it);
    const uint32_t rec = act ? seed % nrec : 0xffffffffu;
    if (MODE == 0) {
      if (act) {
        const uint4 *p = reinterpret_cast<const uint4 *>(tab + (size_t)rec * 64u);
        uint4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += __uint_as_float(a.x ^ b.y ^ c.z ^ d.w);
      }
    } else if (MODE == 3) {
      if (act) {
        const uint4 *p = reinterpret_cast<const uint4 *>(tab + (size_t)rec * 64u);
        uint4 a = p[0], b = p[1];
        acc += __uint_as_float(a.x ^ b.y);
      }
    } else {
      const uint32_t piece = (lane & 3u) * 16u;
#define COOP(K)                                                                                                   \
  {                                                                                                               \
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rec, K * 0x55, 0xf, 0xf, false);             \
    if (o != 0xffffffffu) {                                                                                       \
      if (MODE == 1) {                                                                                            \
        uint4 v = *reinterpret_cast<const uint4 *>(tab + (size_t)o * 64u + piece);                                \
        acc += __uint_as_float(v.x ^ v.w);                                                                        \
      } else {                                                                                                    \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(tab + (size_t)o * 64u + piece), \
                                         (__attribute__((address_space(3))) void *)(buf + K * 260), 16, 0, 0);    \
      }                                                                                                           \
    }                                                                                                             \
  }
      COOP(0) COOP(1) COOP(2) COOP(3)
      if (MODE == 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (act) {
          const uint4 *r = reinterpret_cast<const uint4 *>(buf + (lane & 3u) * 260u + (lane >> 2) * 16u);
          uint4 a = r[0], b = r[1], c = r[2], d = r[3];
          acc += __uint_as_float(a.x ^ b.y ^ c.z ^ d.w);
        }
      }
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <int MODE>
double run(const char *tab, uint32_t nrec, int iters, unsigned long long amask, float *out, int blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(gather<MODE>, dim3(blocks), dim3(64), 0, 0, tab, nrec, 8, (uint32_t)amask, (uint32_t)(amask >> 32), out);
  hipEventRecord(e0);
  hipLaunchKernelGGL(gather<MODE>, dim3(blocks), dim3(64), 0, 0, tab, nrec, iters, (uint32_t)amask, (uint32_t)(amask >> 32), out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main(int argc, char **argv) {
  const size_t mb = argc > 1 ? atoi(argv[1]) : 64;
  const uint32_t nrec = (uint32_t)(mb * 1024 * 1024 / 64);
  char *tab; float *out;
  hipMalloc(&tab, (size_t)nrec * 64); hipMalloc(&out, 64);
  hipMemset(tab, 1, (size_t)nrec * 64);
  const int blocks = 256 * 24, iters = 2000;
  const char *names[4] = {"A own 4x16B", "B quad-coop regs", "C quad-coop LDS-DMA", "D own 2x16B"};
  for (unsigned long long amask : {0xffffffffffffffffull, 0x5555555555555555ull, 0x1111111111111111ull, 0x0f0f0f0f0f0f0f0full}) {
    const int nact = __builtin_popcountll(amask);
    double ms[4] = {run<0>(tab, nrec, iters, amask, out, blocks), run<1>(tab, nrec, iters, amask, out, blocks),
                    run<2>(tab, nrec, iters, amask, out, blocks), run<3>(tab, nrec, iters, amask, out, blocks)};
    for (int m = 0; m < 4; m++) {
      const double recs = (double)blocks * nact * iters;
      printf("table %zu MB  active %2d/64 (mask %016llx)  %-22s %8.2f ms  %7.2f Grec/s  %7.2f TB/s\n", mb, nact, amask, names[m], ms[m],
             recs / ms[m] / 1e6, recs * (m == 3 ? 32 : 64) / ms[m] / 1e9);
    }
  }
  return 0;
}
