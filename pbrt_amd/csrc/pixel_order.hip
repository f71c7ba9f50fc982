// Orders a rank's pixels by what they cost in the first launch of a two-launch frame (capi.cpp render_device),
// most expensive first, so that the second launch hands the cheap pixels out last and the persistent waves run
// dry together.  Scheduling only: which lane renders a pixel, and when, changes no bit of the film.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <hipcub/hipcub.hpp>

#include "device_types.h"

namespace pbrt_hip {

static __global__ void pixel_keys_kernel(const float4 *pixel_state, uint32_t n, uint32_t *keys, uint32_t *vals) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  // float exponent + 2 mantissa bits of the cycle count: four buckets per octave.  Coarse on purpose: pixels of
  // one bucket keep their spatial order (the sort is stable), so a wave's pixels stay close together.
  const float c = pixel_state[2 * (size_t)q].w;
  keys[q] = c > 0.f ? (__float_as_uint(c) >> 21) & 0x3ffu : 0u;
  vals[q] = q;
}

hipError_t launch_pixel_order(const float4 *pixel_state, uint32_t n_pixels, uint32_t *keys, uint32_t *keys_out, uint32_t *vals,
                              uint32_t *order, void *tmp, size_t *tmp_bytes, hipStream_t stream) {
  if (tmp == nullptr) {  // size query (the sort of a small list may need no scratch at all: never report 0)
    hipError_t e = hipcub::DeviceRadixSort::SortPairsDescending(nullptr, *tmp_bytes, keys, keys_out, vals, order, (int)n_pixels, 0, 10, stream);
    if (*tmp_bytes < 256) *tmp_bytes = 256;
    return e;
  }
  hipLaunchKernelGGL(pixel_keys_kernel, dim3((n_pixels + 255u) / 256u), dim3(256), 0, stream, pixel_state, n_pixels, keys, vals);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return hipcub::DeviceRadixSort::SortPairsDescending(tmp, *tmp_bytes, keys, keys_out, vals, order, (int)n_pixels, 0, 10, stream);
}

}  // namespace pbrt_hip
