// Orders a rank's pixels by what they cost in the first launch of a two-launch frame (capi.cpp render_device),
// most expensive first, so that the second launch hands the cheap pixels out last and the persistent waves run
// dry together.  Scheduling only: which lane renders a pixel, and when, changes no bit of the film.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>

#include <hipcub/hipcub.hpp>

#include "device_types.h"

namespace pbrt_hip {

// sum[0] += work of all pixels, sum[1] += number of pixels rendered (both zeroed by the caller)
static __global__ void pixel_work_sum_kernel(const float4 *pixel_state, uint32_t n, unsigned long long *sum) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long w = q < n ? __float_as_uint(pixel_state[2 * (size_t)q].w) : 0u;
  unsigned long long c = w != 0ull ? 1ull : 0ull;
  for (int off = 32; off > 0; off >>= 1) {
    w += __shfl_down(w, off, 64);
    c += __shfl_down(c, off, 64);
  }
  if ((threadIdx.x & 63u) == 0u && c != 0ull) {
    atomicAdd(&sum[0], w);
    atomicAdd(&sum[1], c);
  }
}

static __global__ void pixel_keys_kernel(const float4 *pixel_state, uint32_t n, const unsigned long long *sum, uint32_t buckets,
                                         uint32_t smooth_own, uint32_t *keys, uint32_t *vals) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;  // (n is a multiple of 4096: no partial waves)
  if (q >= n) return;
  // key = the pixel's work (node steps + triangle tests + 16 per ray) in units of 1/buckets of the mean
  // pixel's, at most 255; 0 also for pixels not rendered (outside the film).  The pixels of one bucket keep their
  // spatial order (the sort is stable).  `buckets` is small when a lane renders many pixels (the frame is then
  // handed out as neighbouring pixels of mixed cost, which the waves render fastest, and only markedly
  // expensive pixels move to the front) and large when it renders few (then the order of the tail is what counts).
  // the 16 or so samples of the first launch make a noisy estimate of what a pixel costs; the cost varies smoothly
  // over the image, so the pixel's own work is averaged with the mean of its 8x8 block (64 consecutive list entries
  // = one wave here)
  const uint32_t own = __float_as_uint(pixel_state[2 * (size_t)q].w);
  unsigned long long bsum = own, bcnt = own ? 1u : 0u;
  for (int off = 32; off > 0; off >>= 1) {
    bsum += __shfl_xor(bsum, off, 64);
    bcnt += __shfl_xor(bcnt, off, 64);
  }
  const unsigned long long work = own ? (own * (unsigned long long)smooth_own + (bcnt ? bsum / bcnt : 0ull) * (unsigned long long)(4u - smooth_own)) / 4ull : 0ull;
  const unsigned long long mean = sum[1] ? sum[0] / sum[1] : 1ull;
  const unsigned long long k = ((unsigned long long)buckets * work) / (mean ? mean : 1ull);
  keys[q] = k > 255ull ? 255u : (uint32_t)k;
  vals[q] = q;
}

hipError_t launch_pixel_order(const float4 *pixel_state, uint32_t n_pixels, uint32_t buckets, unsigned long long *work_sum, uint32_t *keys,
                              uint32_t *keys_out, uint32_t *vals, uint32_t *order, void *tmp, size_t *tmp_bytes, hipStream_t stream) {
  if (tmp == nullptr) {  // size query (the sort of a small list may need no scratch at all: never report 0)
    hipError_t e = hipcub::DeviceRadixSort::SortPairsDescending(nullptr, *tmp_bytes, keys, keys_out, vals, order, (int)n_pixels, 0, 8, stream);
    if (*tmp_bytes < 256) *tmp_bytes = 256;
    return e;
  }
  if (n_pixels == 0) return hipSuccess;
  const dim3 grid((n_pixels + 255u) / 256u), block(256);
  const char *so = debug_knob("PBRT_HIP_ORDER_SMOOTH");  // weight of the pixel's own work in quarters (tuning)
  const uint32_t smooth_own = so ? (uint32_t)std::atoi(so) & 7u : 2u;
  hipError_t e = hipMemsetAsync(work_sum, 0, 2 * sizeof(unsigned long long), stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(pixel_work_sum_kernel, grid, block, 0, stream, pixel_state, n_pixels, work_sum);
  hipLaunchKernelGGL(pixel_keys_kernel, grid, block, 0, stream, pixel_state, n_pixels, work_sum, buckets, smooth_own, keys, vals);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  return hipcub::DeviceRadixSort::SortPairsDescending(tmp, *tmp_bytes, keys, keys_out, vals, order, (int)n_pixels, 0, 8, stream);
}

}  // namespace pbrt_hip
