// <seed> <iterations> <base.ply>...: mutated PLY files (truncation, byte flips, deletions, splices of header words) through parse_ply under the
// sanitizers.  Every accepted mesh must be self-consistent: indices inside the vertex array, (u, v) for all vertices or none.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <random>
#include <sstream>
#include <string>
#include <vector>
#include "ply_reader.hpp"
static std::string slurp(const char *p) { std::ifstream f(p, std::ios::binary); std::stringstream ss; ss << f.rdbuf(); return ss.str(); }
int main(int argc, char **argv) {
  if (argc < 4) return 2;
  std::mt19937 rng((unsigned)std::atoi(argv[1]));
  const int iters = std::atoi(argv[2]);
  std::vector<std::string> bases;
  for (int i = 3; i < argc; i++) bases.push_back(slurp(argv[i]));
  static const char *words[] = {"element vertex 4000000000", "element face 99999999", "property list uchar int vertex_indices", "property list int double vertex_index",
                                "property double x", "format binary_big_endian 1.0", "format ascii 1.0", "end_header", "element vertex 0", "property list uint uint x", "-1", "1e308", "nan"};
  int ok = 0, bad = 0;
  for (int it = 0; it < iters; it++) {
    std::string b = bases[rng() % bases.size()];
    const int mode = rng() % 5;
    if (b.empty()) continue;
    if (mode == 0) b.resize(rng() % b.size());
    else if (mode == 1) { for (int k = 0, m = 1 + rng() % 8; k < m; k++) b[rng() % b.size()] = (char)(rng() % 256); }
    else if (mode == 2) { size_t p = rng() % b.size(); b.erase(p, 1 + rng() % 16); }
    else if (mode == 3) { size_t p = rng() % std::min<size_t>(b.size(), 400); b.insert(p, std::string("\n") + words[rng() % (sizeof words / sizeof *words)] + "\n"); }
    else { size_t p = rng() % std::min<size_t>(b.size(), 400); const char *w = words[rng() % (sizeof words / sizeof *words)]; b.replace(p, std::min<size_t>(b.size() - p, std::string(w).size()), w); }
    pbrt_hip::PlyMesh m;
    std::string err;
    if (pbrt_hip::parse_ply((const unsigned char *)b.data(), b.size(), &m, &err)) {
      ok++;
      const size_t nv = m.P.size() / 3;
      if (m.P.size() % 3 || m.idx.size() % 3 || !(m.uv.empty() || m.uv.size() == 2 * nv)) { std::printf("inconsistent mesh at %d\n", it); return 1; }
      for (uint32_t i : m.idx) if (i >= nv) { std::printf("index out of range at %d\n", it); return 1; }
    } else {
      bad++;
      if (err.empty()) { std::printf("refused without a message at %d\n", it); return 1; }
    }
  }
  std::printf("accepted %d refused %d\n", ok, bad);
}
