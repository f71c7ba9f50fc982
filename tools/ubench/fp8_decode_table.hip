// What do gfx950's fp8 converts return?  Prints v_cvt_pk_f32_fp8 of all 256 codes (is it OCP E4M3 or the FNUZ form of
// gfx940?) and v_cvt_scalef32_pk_f32_fp8 with a scale of 3.0 and of 0.75 (is the scale a full multiplier or only its exponent?).
//   hipcc -O2 --offload-arch=gfx950 fp8_decode_table.hip -o fp8_decode_table && ./fp8_decode_table
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(float *out) {
  const uint32_t c = threadIdx.x, w = c | ((255u - c) << 8) | (c << 16) | (c << 24);
  f2 a, b, d;
  const float s3 = 3.0f, s075 = 0.75f;
  asm volatile("v_cvt_pk_f32_fp8 %0, %1" : "=v"(a) : "v"(w));
  asm volatile("v_cvt_scalef32_pk_f32_fp8 %0, %1, %2" : "=v"(b) : "v"(w), "v"(s3));
  asm volatile("v_cvt_scalef32_pk_f32_fp8 %0, %1, %2 op_sel:[1,0,0]" : "=v"(d) : "v"(w), "v"(s075));
  out[c * 6 + 0] = a.x; out[c * 6 + 1] = a.y; out[c * 6 + 2] = b.x; out[c * 6 + 3] = b.y; out[c * 6 + 4] = d.x; out[c * 6 + 5] = d.y;
}
int main() {
  float *d, h[256 * 6];
  (void)hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d);
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("code: cvt_pk(byte0 = code) cvt_pk(byte1 = 255 - code) | scale 3.0: byte0 byte1 | scale 0.75, word 1: byte2 byte3 (= code)\n");
  for (int c = 0; c < 256; c++) printf("%3d: %12g %12g | %12g %12g | %12g %12g\n", c, h[c * 6], h[c * 6 + 1], h[c * 6 + 2], h[c * 6 + 3], h[c * 6 + 4], h[c * 6 + 5]);
  return 0;
}
