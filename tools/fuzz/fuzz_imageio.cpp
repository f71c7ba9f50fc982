#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include <vector>
#include "pbrt_hip.h"
static std::string slurp(const char *p) { std::ifstream f(p, std::ios::binary); std::stringstream ss; ss << f.rdbuf(); return ss.str(); }
int main(int argc, char **argv) {
  std::mt19937 rng((unsigned)std::atoi(argv[1]));
  int iters = std::atoi(argv[2]);
  std::vector<std::string> bases = {slurp(argv[3]), slurp(argv[4])};
  const char *names[2] = {"/tmp/pbrt_fuzz.png", "/tmp/pbrt_fuzz.pfm"};
  int ok = 0, err = 0;
  for (int it = 0; it < iters; it++) {
    int which = rng() % 2;
    std::string b = bases[which];
    int mode = rng() % 3;
    if (mode == 0) b.resize(rng() % b.size());
    else if (mode == 1) { for (int k = 0, m = 1 + rng() % 8; k < m; k++) b[rng() % b.size()] = (char)(rng() % 256); }
    else { size_t p = rng() % b.size(); b.erase(p, 1 + rng() % 16); }
    { std::ofstream o(names[which], std::ios::binary); o.write(b.data(), (std::streamsize)b.size()); }
    int32_t w = 0, h = 0;
    int rc = pbrt_hip_read_image(names[which], nullptr, &w, &h);   // size query
    if (rc == 0 && w > 0 && h > 0 && (long long)w * h < (1 << 22)) {
      std::vector<float> rgb(3 * (size_t)w * h);
      rc = pbrt_hip_read_image(names[which], rgb.data(), &w, &h);
    }
    if (rc == 0) ok++; else err++;
  }
  std::printf("ok %d err %d\n", ok, err);
}
