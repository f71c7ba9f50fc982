#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include "scene_parser.hpp"
int main(int argc, char **argv) {
  std::ifstream f(argv[1], std::ios::binary);
  std::stringstream ss; ss << f.rdbuf();
  std::string base = ss.str();
  std::mt19937 rng((unsigned)std::atoi(argv[2]));
  int n = std::atoi(argv[3]), ok = 0, err = 0;
  for (int it = 0; it < n; it++) {
    std::string b = base;
    int mode = rng() % 4;
    if (mode == 0) b.resize(rng() % b.size());
    else if (mode == 1) { for (int k = 0, m = 1 + rng() % 5; k < m; k++) b[rng() % b.size()] = (char)(rng() % 256); }
    else if (mode == 2) { size_t p = rng() % b.size(), q = rng() % b.size(); b.insert(p, b.substr(q, rng() % 40)); }
    else { size_t p = rng() % b.size(); b.erase(p, 1 + rng() % 30); }
    pbrt_hip::LoadedScene out; std::string msg;
    auto e = pbrt_hip::parse_scene(b.data(), b.size(), "/tmp", &out, &msg);
    if ((int)e == 0) ok++; else err++;
  }
  std::printf("ok %d err %d\n", ok, err);
}
