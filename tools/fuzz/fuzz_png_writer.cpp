// <seed> <iterations>: random images of random structure through pbrt_hip_write_image (row filters + deflate, imageio.cpp) and back through
// pbrt_hip_read_image under the sanitizers; every pixel must come back as to_byte(p) / 255.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "host_math.hpp"
#include "pbrt_hip.h"
int main(int argc, char **argv) {
  if (argc < 3) return 2;
  std::mt19937 rng((unsigned)std::atoi(argv[1]));
  const int iters = std::atoi(argv[2]);
  char name[64];
  std::snprintf(name, sizeof name, "/tmp/pbrt_fuzz_writer_%s.png", argv[1]);  // runs with different seeds may share a machine
  auto unit = [&]() { return (float)(rng() >> 8) / 16777216.f; };
  for (int it = 0; it < iters; it++) {
    const int mode = rng() % 7;
    int w = 1 + rng() % 200, h = 1 + rng() % 200;
    if (rng() % 16 == 0) { w = 1 + rng() % 3000; h = 1 + rng() % 8; }
    if (rng() % 16 == 0) { w = 1 + rng() % 8; h = 1 + rng() % 3000; }
    if (rng() % 32 == 0) { w = 300 + rng() % 300; h = 300 + rng() % 300; }  // several blocks, matches across block ends
    std::vector<float> img(3 * (size_t)w * h);
    const int levels = 1 + rng() % 6, period = 1 + rng() % 97;
    const float amp = unit() * 0.2f;
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++)
        for (int c = 0; c < 3; c++) {
          float v;
          switch (mode) {
            case 0: v = unit(); break;                                                  // noise
            case 1: v = (float)(rng() % levels) / (float)levels; break;                  // few symbols
            case 2: v = (float)(((x % period) * 7 + (y % (period / 2 + 1)) * 13 + c * 5) % 256) / 255.f; break;  // tiles
            case 3: v = (float)x / (float)w * (c == 1 ? 0.f : 1.f) + (float)y / (float)h * (c == 1 ? 1.f : 0.f); break;
            case 4: v = 0.5f + amp * (unit() - 0.5f) + 0.3f * std::sin(0.05f * (float)x) * std::cos(0.07f * (float)y); break;
            case 5: v = (x / 16 + y / 16) % 2 ? 0.8f : 0.1f + amp * unit(); break;          // checkers, one colour noisy
            default: v = (rng() % 50 == 0) ? unit() * 3.f - 1.f : 0.f; break;             // sparse, out-of-range values
          }
          img[3 * ((size_t)y * w + x) + c] = v;
        }
    if (pbrt_hip_write_image(name, img.data(), w, h) != 0) { std::printf("write failed (%d x %d)\n", w, h); return 1; }
    int32_t rw = 0, rh = 0;
    std::vector<float> back(img.size());
    if (pbrt_hip_read_image(name, nullptr, &rw, &rh) != 0 || rw != w || rh != h || pbrt_hip_read_image(name, back.data(), &rw, &rh) != 0) {
      std::printf("read failed at %d (%d x %d, mode %d)\n", it, w, h, mode);
      return 1;
    }
    for (size_t i = 0; i < img.size(); i++)
      if (back[i] != (float)pbrt_hip::to_byte(img[i]) / 255.f) { std::printf("pixel %zu differs at %d (%d x %d, mode %d)\n", i, it, w, h, mode); return 1; }
  }
  std::printf("ok %d\n", iters);
}
