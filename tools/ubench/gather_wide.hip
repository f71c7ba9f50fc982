// Microbenchmark: random record gathers of 32 / 48 / 64 / 128 / 256 bytes per lane (own-lane dwordx4 loads) from tables of
// different sizes: is the limit past L2 a REQUEST rate (records/s flat) or a BYTE rate (TB/s flat)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int NQ>  // NQ x 16 bytes per record
__global__ void __launch_bounds__(64) gather(const char *tab, uint32_t nrec, int iters, float *out) {
  uint32_t seed = blockIdx.x * 64u + threadIdx.x;
  float acc = 0.f;
  for (int it = 0; it < iters; it++) {
    seed = hash(seed + it);
    const uint4 *p = reinterpret_cast<const uint4 *>(tab + (size_t)(seed % nrec) * (NQ * 16));
    uint4 v[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) v[q] = p[q];
#pragma unroll
    for (int q = 0; q < NQ; q++) acc += __uint_as_float(v[q].x ^ v[q].w);
  }
  if (acc == 12345.678f) out[0] = acc;
}
template <int NQ>
void run(const char *tab, size_t bytes, float *out) {
  const uint32_t nrec = (uint32_t)(bytes / (NQ * 16));
  const int blocks = 256 * 24, iters = 1000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(gather<NQ>, dim3(blocks), dim3(64), 0, 0, tab, nrec, 8, out);
  hipEventRecord(e0);
  hipLaunchKernelGGL(gather<NQ>, dim3(blocks), dim3(64), 0, 0, tab, nrec, iters, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double recs = (double)blocks * 64 * iters;
  printf("table %5zu MB  record %3d B  %8.2f ms  %7.2f Grec/s  %6.2f TB/s\n", bytes >> 20, NQ * 16, ms, recs / ms / 1e6, recs * NQ * 16 / ms / 1e9);
}
// `gather_wide pmc` (round 6): the calibration of rocprofv3's FETCH_SIZE for THIS access shape -- 64-byte records at random addresses,
// four own-lane dwordx4 loads each, what the render kernel's node and triangle fetches are -- on tables past L2 (112 MB: inside the
// Infinity Cache; 1536 MB: past it).  One dispatch per line, the bytes it gathers known exactly; tools/fetch_size_calibration.sh runs it under
// `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and divides (profiles/r06_fetch_size_calibration.txt).
static int pmc_mode(const char *tab, float *out) {
  const int blocks = 256 * 24;
  int n = 0;
  for (size_t mb : {112, 1536, 112, 1536}) {
    const int iters = n < 2 ? 400 : 1000;
    const uint32_t nrec = (uint32_t)((mb << 20) / 64);
    hipLaunchKernelGGL(gather<4>, dim3(blocks), dim3(64), 0, 0, tab, nrec, iters, out);
    hipDeviceSynchronize();
    printf("DISPATCH %d table_MB %zu record_B 64 records %.0f bytes %.0f\n", n, mb, (double)blocks * 64 * iters, (double)blocks * 64 * iters * 64);
    n++;
  }
  return 0;
}
int main(int argc, char **argv) {
  char *tab; float *out;
  const size_t cap = (size_t)2048 << 20;
  hipMalloc(&tab, cap); hipMalloc(&out, 64); hipMemset(tab, 1, cap);
  if (argc > 1 && argv[1][0] == 'p') return pmc_mode(tab, out);
  for (size_t mb : {1, 2, 16, 28, 48, 112, 512, 2048}) {  // (28 MB: the quad nodes of BASELINE C3; 48-byte records: r03, would a node of three 16-byte pieces gather faster?)
    run<2>(tab, mb << 20, out); run<3>(tab, mb << 20, out); run<4>(tab, mb << 20, out); run<8>(tab, mb << 20, out); run<16>(tab, mb << 20, out);
  }
  return 0;
}
