// ply_reader.hpp -- Shape "plymesh" "string filename": the triangle-mesh data format pbrt-v3 scenes keep their geometry in (PLY, the
// Stanford polygon format: ascii, binary_little_endian and binary_big_endian).  The reference has no shape code at all (its "Shape" arm
// returns NotImplemented, parser.rs:300); this is the data format on the input side of the render path, read into the same arrays a
// "trianglemesh" fills.  What pbrt-v3's reader (plymesh.cpp) takes is taken here: element "vertex" with x y z and optionally (u, v) /
// (s, t) / (texture_u, texture_v) / (texture_s, texture_t); element "face" with the list "vertex_indices" / "vertex_index", triangles and
// quads (a quad becomes (0 1 2) and (3 0 2), as there); normals and every other property or element are skipped.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace pbrt_hip {

struct PlyMesh {
  std::vector<float> P;       // 3 per vertex
  std::vector<float> uv;      // 2 per vertex, or empty
  std::vector<uint32_t> idx;  // 3 per triangle
  uint32_t skipped_faces = 0;  // faces with other than 3 or 4 vertices
};

// Parses `n` bytes of a PLY file.  false with *err set on malformed or truncated input; every count is checked against the bytes that
// are there before anything is allocated, every index against the vertex count.
bool parse_ply(const unsigned char *data, size_t n, PlyMesh *out, std::string *err);
bool read_ply(const std::string &path, PlyMesh *out, std::string *err);

}  // namespace pbrt_hip
