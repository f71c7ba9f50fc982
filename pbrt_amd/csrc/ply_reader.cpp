// ply_reader.cpp -- see ply_reader.hpp.  (Little-endian host assumed, like the PFM reader / writer of imageio.cpp.)
#include "ply_reader.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace pbrt_hip {
namespace {

enum class Ty { I8, U8, I16, U16, I32, U32, F32, F64, Bad };
size_t ty_size(Ty t) {
  switch (t) {
    case Ty::I8: case Ty::U8: return 1;
    case Ty::I16: case Ty::U16: return 2;
    case Ty::I32: case Ty::U32: case Ty::F32: return 4;
    case Ty::F64: return 8;
    default: return 0;
  }
}
Ty ty_of(const std::string &s) {
  if (s == "char" || s == "int8") return Ty::I8;
  if (s == "uchar" || s == "uint8") return Ty::U8;
  if (s == "short" || s == "int16") return Ty::I16;
  if (s == "ushort" || s == "uint16") return Ty::U16;
  if (s == "int" || s == "int32") return Ty::I32;
  if (s == "uint" || s == "uint32") return Ty::U32;
  if (s == "float" || s == "float32") return Ty::F32;
  if (s == "double" || s == "float64") return Ty::F64;
  return Ty::Bad;
}
struct Prop {
  std::string name;
  bool list = false;
  Ty count_t = Ty::Bad, t = Ty::Bad;
};
struct Elem {
  std::string name;
  uint64_t count = 0;
  std::vector<Prop> props;
};
enum class Format { None, Ascii, Little, Big };

struct Cursor {
  const unsigned char *p, *end;
  Format fmt;
  size_t left() const { return (size_t)(end - p); }
  // one scalar of type t as a double (every PLY type is exact in one); false at the end of the data or on a malformed ascii number
  bool scalar(Ty t, double *out) {
    if (fmt == Format::Ascii) {
      while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) p++;
      char buf[64];
      size_t k = 0;
      while (p < end && !(*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) {
        if (k + 1 >= sizeof buf) return false;
        buf[k++] = (char)*p++;
      }
      if (k == 0) return false;
      buf[k] = 0;
      char *e = nullptr;
      *out = std::strtod(buf, &e);
      return e != buf && *e == 0;
    }
    const size_t sz = ty_size(t);
    if (left() < sz) return false;
    unsigned char b[8];
    for (size_t i = 0; i < sz; i++) b[i] = fmt == Format::Big ? p[sz - 1 - i] : p[i];
    p += sz;
    switch (t) {
      case Ty::I8: { int8_t v; std::memcpy(&v, b, 1); *out = v; break; }
      case Ty::U8: { uint8_t v; std::memcpy(&v, b, 1); *out = v; break; }
      case Ty::I16: { int16_t v; std::memcpy(&v, b, 2); *out = v; break; }
      case Ty::U16: { uint16_t v; std::memcpy(&v, b, 2); *out = v; break; }
      case Ty::I32: { int32_t v; std::memcpy(&v, b, 4); *out = v; break; }
      case Ty::U32: { uint32_t v; std::memcpy(&v, b, 4); *out = v; break; }
      case Ty::F32: { float v; std::memcpy(&v, b, 4); *out = v; break; }
      case Ty::F64: { double v; std::memcpy(&v, b, 8); *out = v; break; }
      default: return false;
    }
    return true;
  }
};

bool fail(std::string *err, const std::string &m) {
  if (err) *err = "ply: " + m;
  return false;
}

// the least number of bytes one row of the element can take in this format (lists counted as empty): bounds `count` by the data
size_t min_row_bytes(const Elem &e, Format fmt) {
  size_t n = 0;
  for (const Prop &p : e.props) n += fmt == Format::Ascii ? 2 : (p.list ? ty_size(p.count_t) : ty_size(p.t));
  return n ? n : 1;
}

}  // namespace

bool parse_ply(const unsigned char *data, size_t n, PlyMesh *out, std::string *err) {
  *out = PlyMesh();
  // ---- header: lines up to "end_header" ----
  size_t pos = 0;
  auto line = [&](std::string *l) {
    if (pos >= n) return false;
    size_t e = pos;
    while (e < n && data[e] != '\n') e++;
    if (e - pos > 4096) return false;
    l->assign((const char *)data + pos, e - pos);
    while (!l->empty() && (l->back() == '\r' || l->back() == ' ' || l->back() == '\t')) l->pop_back();
    pos = e < n ? e + 1 : e;
    return true;
  };
  std::string l;
  if (!line(&l) || l != "ply") return fail(err, "not a PLY file (no \"ply\" magic)");
  Format fmt = Format::None;
  std::vector<Elem> elems;
  bool ended = false;
  while (line(&l)) {
    std::istringstream ss(l);
    std::string w;
    if (!(ss >> w)) continue;
    if (w == "end_header") { ended = true; break; }
    if (w == "comment" || w == "obj_info") continue;
    if (w == "format") {
      std::string f, v;
      ss >> f >> v;
      fmt = f == "ascii" ? Format::Ascii : f == "binary_little_endian" ? Format::Little : f == "binary_big_endian" ? Format::Big : Format::None;
      if (fmt == Format::None) return fail(err, "unknown format \"" + f + "\"");
    } else if (w == "element") {
      Elem e;
      std::string c;
      if (!(ss >> e.name >> c)) return fail(err, "malformed element line");
      char *end = nullptr;
      const unsigned long long cnt = std::strtoull(c.c_str(), &end, 10);
      if (c.empty() || c[0] == '-' || end == c.c_str() || *end) return fail(err, "malformed element count \"" + c + "\"");
      e.count = cnt;
      if (elems.size() >= 64) return fail(err, "too many elements");
      elems.push_back(e);
    } else if (w == "property") {
      if (elems.empty()) return fail(err, "property before any element");
      Prop p;
      std::string t;
      if (!(ss >> t)) return fail(err, "malformed property line");
      if (t == "list") {
        std::string ct, it;
        if (!(ss >> ct >> it >> p.name)) return fail(err, "malformed list property");
        p.list = true;
        p.count_t = ty_of(ct);
        p.t = ty_of(it);
        if (p.count_t == Ty::Bad || p.t == Ty::Bad) return fail(err, "unknown property type in \"" + l + "\"");
      } else {
        p.t = ty_of(t);
        if (p.t == Ty::Bad || !(ss >> p.name)) return fail(err, "unknown property type in \"" + l + "\"");
      }
      if (elems.back().props.size() >= 256) return fail(err, "too many properties");
      elems.back().props.push_back(p);
    } else {
      return fail(err, "unknown header line \"" + w + "\"");
    }
  }
  if (!ended) return fail(err, "no end_header");
  if (fmt == Format::None) return fail(err, "no format line");

  // ---- data ----
  Cursor c{data + pos, data + n, fmt};
  bool have_vertices = false;
  uint64_t n_verts = 0;
  for (const Elem &e : elems) {
    if (e.count > c.left() / min_row_bytes(e, fmt) + 1) return fail(err, "element \"" + e.name + "\": " + std::to_string(e.count) + " rows do not fit the file");
    if (e.name == "vertex") {
      if (have_vertices) return fail(err, "two vertex elements");
      have_vertices = true;
      if (e.count >= 0xffffffffull / 3) return fail(err, "too many vertices");
      n_verts = e.count;
      int ix = -1, iy = -1, iz = -1, iu = -1, iv = -1;
      for (size_t k = 0; k < e.props.size(); k++) {
        const std::string &nm = e.props[k].name;
        if (e.props[k].list) continue;
        if (nm == "x") ix = (int)k;
        else if (nm == "y") iy = (int)k;
        else if (nm == "z") iz = (int)k;
        else if (nm == "u" || nm == "s" || nm == "texture_u" || nm == "texture_s") iu = (int)k;
        else if (nm == "v" || nm == "t" || nm == "texture_v" || nm == "texture_t") iv = (int)k;
      }
      if (ix < 0 || iy < 0 || iz < 0) return fail(err, "vertex element without x, y, z");
      const bool has_uv = iu >= 0 && iv >= 0;
      out->P.resize(3 * (size_t)e.count);
      if (has_uv) out->uv.resize(2 * (size_t)e.count);
      for (uint64_t r = 0; r < e.count; r++)
        for (size_t k = 0; k < e.props.size(); k++) {
          const Prop &p = e.props[k];
          double v;
          if (p.list) {
            if (!c.scalar(p.count_t, &v) || !(v >= 0) || v > (double)c.left()) return fail(err, "truncated vertex data");
            for (uint64_t i = 0, m = (uint64_t)v; i < m; i++)
              if (!c.scalar(p.t, &v)) return fail(err, "truncated vertex data");
            continue;
          }
          if (!c.scalar(p.t, &v)) return fail(err, "truncated vertex data (vertex " + std::to_string(r) + ")");
          if ((int)k == ix) out->P[3 * r] = (float)v;
          else if ((int)k == iy) out->P[3 * r + 1] = (float)v;
          else if ((int)k == iz) out->P[3 * r + 2] = (float)v;
          else if (has_uv && (int)k == iu) out->uv[2 * r] = (float)v;
          else if (has_uv && (int)k == iv) out->uv[2 * r + 1] = (float)v;
        }
    } else {
      const bool is_face = e.name == "face";
      int ilist = -1;
      if (is_face)
        for (size_t k = 0; k < e.props.size(); k++)
          if (e.props[k].list && (e.props[k].name == "vertex_indices" || e.props[k].name == "vertex_index")) ilist = (int)k;
      if (is_face && ilist < 0) return fail(err, "face element without a vertex_indices list");
      for (uint64_t r = 0; r < e.count; r++)
        for (size_t k = 0; k < e.props.size(); k++) {
          const Prop &p = e.props[k];
          double v;
          if (!p.list) {
            if (!c.scalar(p.t, &v)) return fail(err, "truncated data in element \"" + e.name + "\"");
            continue;
          }
          if (!c.scalar(p.count_t, &v) || !(v >= 0) || v > (double)c.left()) return fail(err, "truncated data in element \"" + e.name + "\"");
          const uint64_t m = (uint64_t)v;
          double f[4] = {0, 0, 0, 0};
          for (uint64_t i = 0; i < m; i++) {
            if (!c.scalar(p.t, &v)) return fail(err, "truncated data in element \"" + e.name + "\"");
            if (i < 4) f[i] = v;
          }
          if ((int)k != ilist) continue;
          if (m != 3 && m != 4) { out->skipped_faces++; continue; }
          for (uint64_t i = 0; i < m; i++)
            if (!(f[i] >= 0) || f[i] >= 4294967295.0 || f[i] != std::floor(f[i])) return fail(err, "face " + std::to_string(r) + ": vertex index out of range");
          out->idx.push_back((uint32_t)f[0]); out->idx.push_back((uint32_t)f[1]); out->idx.push_back((uint32_t)f[2]);
          if (m == 4) { out->idx.push_back((uint32_t)f[3]); out->idx.push_back((uint32_t)f[0]); out->idx.push_back((uint32_t)f[2]); }
        }
    }
  }
  if (!have_vertices) return fail(err, "no vertex element");
  for (uint32_t i : out->idx)  // (a face element may precede the vertices: checked once everything is read)
    if (i >= n_verts) return fail(err, "vertex index " + std::to_string(i) + " out of range (" + std::to_string(n_verts) + " vertices)");
  return true;
}

bool read_ply(const std::string &path, PlyMesh *out, std::string *err) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return fail(err, "cannot read '" + path + "'");
  std::ostringstream ss;
  ss << f.rdbuf();
  const std::string s = ss.str();
  return parse_ply((const unsigned char *)s.data(), s.size(), out, err);
}

}  // namespace pbrt_hip
