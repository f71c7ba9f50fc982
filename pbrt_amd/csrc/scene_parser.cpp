// scene_parser.cpp -- see scene_parser.hpp.
#include "scene_parser.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <algorithm>
#include <array>
#include <memory>
#include <tuple>

#include "host_math.hpp"
#include "ply_reader.hpp"

namespace pbrt_hip {

// ------------------------------------------------------------------------------------------------
// Tokenizer: parser.rs:61-170.  Tokens are: quoted strings (kept with their quotes), '[' and ']',
// comments ('#' to end of line, returned so the parser can skip them, parser.rs:349) and runs of
// anything else up to white space, a quote or a bracket.
// ------------------------------------------------------------------------------------------------
bool Tokenizer::next(std::string *tok, ParseError *err) {
  *err = ParseError::None;
  for (;;) {
    if (pos == len) return false;
    const size_t start = pos;
    const char c = data[pos++];
    if (c == ' ' || c == '\n' || c == '\t' || c == '\r') continue;
    if (c == '"') {
      for (;;) {
        if (pos == len) { *err = ParseError::Eof; return true; }  // parser.rs:79
        const char b = data[pos++];
        if (b == '"') break;
        if (b == '\n') { *err = ParseError::UnterminatedString; return true; }  // parser.rs:80
        if (b == '\\') {  // escapes: the reference has unimplemented!() here (parser.rs:96); kept verbatim
          if (pos == len) { *err = ParseError::Eof; return true; }
          pos++;
        }
      }
      tok->assign(data + start, pos - start);
      return true;
    }
    if (c == '[' || c == ']') {
      tok->assign(1, c);
      return true;
    }
    if (c == '#') {
      while (pos < len && data[pos] != '\n' && data[pos] != '\r') pos++;
      tok->assign(data + start, pos - start);
      return true;
    }
    while (pos < len) {
      const char b = data[pos];
      if (b == ' ' || b == '\n' || b == '\t' || b == '\r' || b == '"' || b == '[' || b == ']') break;
      pos++;
    }
    tok->assign(data + start, pos - start);
    return true;
  }
}

// ------------------------------------------------------------------------------------------------
// ParamSet: typed name -> values bag (paramset.rs:109-111, find_one_* :237-513).  Unlike the
// reference it has array getters (the reference's `find` is private, paramset.rs:217-223, so its
// "integer indices" / "point P" are unreachable).
// ------------------------------------------------------------------------------------------------
const ParamItem *ParamSet::find(const char *name, const char *t1, const char *t2, const char *t3) const {
  for (const ParamItem &it : items) {
    if (it.name != name) continue;
    if (it.type == t1 || (t2 && it.type == t2) || (t3 && it.type == t3)) {
      it.looked_up = true;
      return &it;
    }
  }
  return nullptr;
}
// find_one_*: the FIRST value of the named parameter, the default when there is none (paramset.rs:237-513: `pl.0.first().map_or(default, ..)`
// -- pbrt-v3 itself wants exactly one value; the reference is the model here)
float ParamSet::one_float(const char *name, float dflt) const {
  const ParamItem *p = find(name, "float");
  return (p && !p->nums.empty()) ? (float)p->nums[0] : dflt;
}
int ParamSet::one_int(const char *name, int dflt) const {
  const ParamItem *p = find(name, "integer");
  return (p && !p->nums.empty()) ? (int)p->nums[0] : dflt;
}
bool ParamSet::one_bool(const char *name, bool dflt) const {
  const ParamItem *p = find(name, "bool");
  return (p && !p->strs.empty()) ? p->strs[0] == "true" : dflt;
}
std::string ParamSet::one_string(const char *name, const std::string &dflt) const {
  const ParamItem *p = find(name, "string");
  return (p && !p->strs.empty()) ? p->strs[0] : dflt;
}
bool ParamSet::point3(const char *name, float out[3]) const {
  const ParamItem *p = find(name, "point3", "vector3", "normal");
  if (!p || p->nums.size() != 3) return false;
  for (int i = 0; i < 3; i++) out[i] = (float)p->nums[i];
  return true;
}
std::vector<std::string> ParamSet::unused() const {
  std::vector<std::string> r;
  for (const ParamItem &it : items)
    if (!it.looked_up) r.push_back(it.type + " " + it.name);
  return r;
}

namespace {

// ---- transforms (transform.rs) ----
void mat_mul(const float a[16], const float b[16], float r[16]) {  // transform.rs:270-282
  float t[16];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      t[4 * i + j] = a[4 * i] * b[j] + a[4 * i + 1] * b[4 + j] + a[4 * i + 2] * b[8 + j] + a[4 * i + 3] * b[12 + j];
  std::memcpy(r, t, 64);
}
Xform xf_identity() {
  Xform x;
  mat_identity(x.m);
  mat_identity(x.inv);
  return x;
}
// Transform * Transform (transform.rs:618-626).  NOTE: the reference composes m_inv as
// self.m_inv * rhs.m_inv, which is not the inverse of the product; here m_inv = rhs.m_inv * self.m_inv.
Xform xf_mul(const Xform &a, const Xform &b) {
  Xform r;
  mat_mul(a.m, b.m, r.m);
  mat_mul(b.inv, a.inv, r.inv);
  return r;
}
Xform xf_translate(float x, float y, float z) {  // transform.rs:375-393
  Xform r = xf_identity();
  r.m[3] = x; r.m[7] = y; r.m[11] = z;
  r.inv[3] = -x; r.inv[7] = -y; r.inv[11] = -z;
  return r;
}
Xform xf_scale(float x, float y, float z) {  // transform.rs:539-558
  Xform r = xf_identity();
  r.m[0] = x; r.m[5] = y; r.m[10] = z;
  r.inv[0] = 1.f / x; r.inv[5] = 1.f / y; r.inv[10] = 1.f / z;
  return r;
}
Xform xf_rotate(float deg, float ax, float ay, float az) {  // transform.rs:444-481
  F3 a = unit3({ax, ay, az});
  const float th = deg * (3.14159265358979323846f / 180.f);
  const float s = std::sin(th), c = std::cos(th);
  Xform r = xf_identity();
  r.m[0] = a.x * a.x + (1.f - a.x * a.x) * c;
  r.m[1] = a.x * a.y * (1.f - c) - a.z * s;
  r.m[2] = a.x * a.z * (1.f - c) + a.y * s;
  r.m[4] = a.x * a.y * (1.f - c) + a.z * s;
  r.m[5] = a.y * a.y + (1.f - a.y * a.y) * c;
  r.m[6] = a.y * a.z * (1.f - c) - a.x * s;
  r.m[8] = a.x * a.z * (1.f - c) - a.y * s;
  r.m[9] = a.y * a.z * (1.f - c) + a.x * s;
  r.m[10] = a.z * a.z + (1.f - a.z * a.z) * c;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) r.inv[4 * i + j] = r.m[4 * j + i];  // transpose
  return r;
}
Xform xf_from_matrix(const float m[16]) {  // From<Matrix4x4> for Transform, transform.rs:603-610
  Xform r;
  std::memcpy(r.m, m, 64);
  mat_inverse(m, r.inv);
  return r;
}
void xf_point(const float m[16], const float p[3], float out[3]) {
  const float x = m[0] * p[0] + m[1] * p[1] + m[2] * p[2] + m[3];
  const float y = m[4] * p[0] + m[5] * p[1] + m[6] * p[2] + m[7];
  const float z = m[8] * p[0] + m[9] * p[1] + m[10] * p[2] + m[11];
  const float w = m[12] * p[0] + m[13] * p[1] + m[14] * p[2] + m[15];
  if (w == 1.f) { out[0] = x; out[1] = y; out[2] = z; }
  else { out[0] = x / w; out[1] = y / w; out[2] = z / w; }
}
void xf_vector(const float m[16], const float v[3], float out[3]) {
  out[0] = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  out[1] = m[4] * v[0] + m[5] * v[1] + m[6] * v[2];
  out[2] = m[8] * v[0] + m[9] * v[1] + m[10] * v[2];
}
bool swaps_handedness(const float m[16]) {
  const float det = m[0] * (m[5] * m[10] - m[6] * m[9]) - m[1] * (m[4] * m[10] - m[6] * m[8]) +
                    m[2] * (m[4] * m[9] - m[5] * m[8]);
  return det < 0.f;
}

// ---- spectra -> RGB ----
// CIE 1931 colour matching functions, multi-lobe Gaussian fit (Wyman, Sloan, Shirley 2013); the
// reference's spectral tables do not exist (SampledSpectrum is all todo!(), spectrum.rs:95-124).
double lobe(double l, double mu, double s1, double s2) {
  const double t = (l - mu) / (l < mu ? s1 : s2);
  return std::exp(-0.5 * t * t);
}
void cie_xyz(double l, double out[3]) {
  out[0] = 1.056 * lobe(l, 599.8, 37.9, 31.0) + 0.362 * lobe(l, 442.0, 16.0, 26.7) - 0.065 * lobe(l, 501.1, 20.4, 26.2);
  out[1] = 0.821 * lobe(l, 568.8, 46.9, 40.5) + 0.286 * lobe(l, 530.9, 16.3, 31.1);
  out[2] = 1.217 * lobe(l, 437.0, 11.8, 36.0) + 0.681 * lobe(l, 459.0, 26.0, 13.8);
}
double planck(double lambda_nm, double T) {
  const double c = 299792458.0, h = 6.62606957e-34, kb = 1.3806488e-23;
  const double l = lambda_nm * 1e-9;
  const double l5 = l * l * l * l * l;
  return (2.0 * h * c * c) / (l5 * (std::exp((h * c) / (l * kb * T)) - 1.0));
}
// pbrt-v3 BlackbodyNormalized (peak = 1) -> XYZ / integral(Y) -> RGB, times `scale`
void blackbody_rgb(double T, double scale, float rgb[3]) {
  const double lmax = 2.8977721e-3 / T * 1e9;
  const double peak = planck(lmax, T);
  double X = 0, Y = 0, Z = 0, yint = 0;
  for (int l = 360; l <= 830; l += 1) {
    double cmf[3];
    cie_xyz(l, cmf);
    const double v = planck(l, T) / peak;
    X += cmf[0] * v; Y += cmf[1] * v; Z += cmf[2] * v;
    yint += cmf[1];
  }
  const float xyz[3] = {(float)(X / yint * scale), (float)(Y / yint * scale), (float)(Z / yint * scale)};
  xyz_to_rgb(xyz, rgb);
}

bool is_quoted(const std::string &s) { return s.size() >= 2 && s.front() == '"' && s.back() == '"'; }
std::string dequote(const std::string &s) { return s.substr(1, s.size() - 2); }

// ------------------------------------------------------------------------------------------------
// The API state machine: PbrtAPI (api.rs:355-386) + the directive parser (parser.rs:205-317).
// ------------------------------------------------------------------------------------------------
struct MaterialDef {
  uint32_t type = 0;  // matte
  float k[3] = {0.5f, 0.5f, 0.5f};
  uint32_t kd_tex = 0;  // 1 + index into LoadedScene::textures when "texture Kd" names a checkerboard
};
struct GraphicsState {  // api.rs:251-289
  MaterialDef material;
  bool has_area_light = false;
  float area_le[3] = {0, 0, 0};
  bool reverse_orientation = false;
};
enum class ApiState { Uninitialized, OptionsBlock, WorldBlock };  // api.rs:190-199
constexpr int kMaxTransforms = 2;                                  // api.rs: MAX_TRANSFORMS
constexpr uint32_t kStartBit = 1, kEndBit = 2, kAllBits = 3;

struct Api {
  LoadedScene *out;
  ApiState state = ApiState::OptionsBlock;  // init() already called (api.rs:400-406)
  Xform ctm[kMaxTransforms];
  uint32_t active_bits = kAllBits;
  std::map<std::string, std::pair<Xform, Xform>> named_cs;
  GraphicsState gs;
  std::vector<GraphicsState> pushed_gs;
  std::vector<std::pair<Xform, Xform>> pushed_ctm;
  std::vector<uint32_t> pushed_bits;
  std::map<std::string, std::array<float, 3>> spectrum_textures;
  std::map<std::string, uint32_t> checker_textures;  // name -> 1 + index into out->textures
  std::map<std::string, MaterialDef> named_materials;
  std::map<std::tuple<uint32_t, float, float, float, float, float, float, uint32_t>, uint16_t> material_ids;
  bool camera_set = false;
  std::string cur_dir;  // directory of the file being parsed: resolves Shape "plymesh" "string filename" like Include
  // ObjectBegin "name" ... ObjectEnd / ObjectInstance "name" (pbrt-v3 api.cpp pbrtObjectBegin / pbrtObjectInstance; NotImplemented in the
  // reference, parser.rs:283-289).  An object's shapes are kept in world space as of their own CTM (their ObjectToWorld); an instance
  // appends a copy moved by the CTM at the ObjectInstance (InstanceToWorld): the BVH is over flattened geometry, instances cost memory.
  struct ObjectDef {
    std::vector<float> P, tri_uv;
    std::vector<uint32_t> idx;
    std::vector<uint16_t> mat_id;
    std::vector<pbrt_hip_sphere> spheres;
  };
  std::map<std::string, ObjectDef> objects;
  bool in_object = false;
  std::string object_name;
  ObjectDef stash;  // the scene's own arrays while an object is being defined (shape() always appends to out->...)
  bool warned_object_light = false;
  // instancing multiplies geometry (instances x the object's triangles): a bound on what a small file can ask for (the tests lower it)
  const size_t kMaxSceneTriangles = [] {
    const char *on = std::getenv("PBRT_HIP_DEBUG_KNOBS"), *v = std::getenv("PBRT_HIP_MAX_SCENE_TRIANGLES");
    return (on && *on && !(on[0] == '0' && !on[1]) && v && *v) ? (size_t)std::strtoull(v, nullptr, 10) : (size_t)1 << 24;  // = what pbrt_hip_scene_create takes (a leaf reference holds a 24-bit slot)
  }();

  explicit Api(LoadedScene *o) : out(o) {
    ctm[0] = ctm[1] = xf_identity();
    mat_identity(out->cam_to_world);
  }
  void warn(const std::string &s) { out->warnings.push_back(s); }
  // api.rs:897-901 warn_if_animated_transform: the two CTMs differ (ActiveTransform StartTime / EndTime); the start transform is used
  void warn_if_animated_transform(const char *what) {
    if (std::memcmp(ctm[0].m, ctm[1].m, 64) != 0)
      warn(std::string("Animated transformations set; ignoring for \"") + what + "\" and using the start transform only");
  }
  void report_unused(const char *what, const ParamSet &ps) {
    for (const std::string &u : ps.unused()) warn(std::string(what) + ": parameter \"" + u + "\" not used");
  }
  template <class F>
  void for_active(F f) {  // api.rs:871-881
    for (int i = 0; i < kMaxTransforms; i++)
      if (active_bits & (1u << i)) f(ctm[i]);
  }
  void concat(const Xform &t) { for_active([&](Xform &c) { c = xf_mul(c, t); }); }

  // a spectrum-valued parameter in any of its spellings (parser.rs:433-475 type table)
  bool spectrum(const ParamSet &ps, const char *name, float rgb[3]) {
    if (const ParamItem *p = ps.find(name, "rgb", "color")) {
      if (p->nums.size() == 3) { for (int i = 0; i < 3; i++) rgb[i] = (float)p->nums[i]; return true; }
    }
    if (const ParamItem *p = ps.find(name, "xyz")) {
      if (p->nums.size() == 3) {
        const float xyz[3] = {(float)p->nums[0], (float)p->nums[1], (float)p->nums[2]};
        xyz_to_rgb(xyz, rgb);
        return true;
      }
    }
    if (const ParamItem *p = ps.find(name, "blackbody")) {
      if (p->nums.size() == 2) { blackbody_rgb(p->nums[0], p->nums[1], rgb); return true; }
    }
    if (const ParamItem *p = ps.find(name, "spectrum")) {
      if (!p->nums.empty() && p->nums.size() % 2 == 0) {  // (lambda, value) pairs: flat average
        double s = 0;
        for (size_t i = 1; i < p->nums.size(); i += 2) s += p->nums[i];
        rgb[0] = rgb[1] = rgb[2] = (float)(s / (p->nums.size() / 2));
        warn(std::string("spectrum \"") + name + "\": sampled spectra are reduced to their mean (out of scope)");
        return true;
      }
      warn(std::string("spectrum \"") + name + "\": SPD files are not supported");
    }
    if (const ParamItem *p = ps.find(name, "texture")) {
      if (p->strs.size() == 1) {
        auto it = spectrum_textures.find(p->strs[0]);
        if (it != spectrum_textures.end()) { for (int i = 0; i < 3; i++) rgb[i] = it->second[i]; return true; }
        warn("Spectrum texture '" + p->strs[0] + "' is unknown");  // api.rs:939
      }
    }
    return false;
  }

  MaterialDef make_material(const std::string &type, const ParamSet &ps) {
    MaterialDef m;
    if (type == "matte") {
      m.type = 0;
      if (const ParamItem *p = ps.find("Kd", "texture")) {  // a checkerboard keeps its pattern; k = its mean colour (spheres, fallbacks)
        if (p->strs.size() == 1) {
          auto it = checker_textures.find(p->strs[0]);
          if (it != checker_textures.end()) m.kd_tex = it->second;
        }
      }
      spectrum(ps, "Kd", m.k);
      ps.find("sigma", "float");
    } else if (type == "mirror") {
      m.type = 1;
      m.k[0] = m.k[1] = m.k[2] = 0.9f;
      spectrum(ps, "Kr", m.k);
    } else {
      warn("Material \"" + type + "\" is not supported by this path: using matte Kd 0.5");
    }
    return m;
  }

  uint16_t material_id(const MaterialDef &m, const float le[3]) {
    auto key = std::make_tuple(m.type, m.k[0], m.k[1], m.k[2], le[0], le[1], le[2], m.kd_tex);
    auto it = material_ids.find(key);
    if (it != material_ids.end()) return it->second;
    pbrt_hip_material pm{};
    pm.type = m.type;
    pm.kd_tex = m.kd_tex;
    for (int i = 0; i < 3; i++) { pm.k[i] = m.k[i]; pm.le[i] = le[i]; }
    const uint16_t id = (uint16_t)out->mats.size();
    out->mats.push_back(pm);
    material_ids[key] = id;
    return id;
  }

  // Texture "name" "spectrum" "class" ... (api.rs:524-580 stores nothing; texture.rs is an empty marker).  On this path: a 2-D
  // CHECKERBOARD over the (u, v) mapping keeps its pattern (pbrt-v3 Checkerboard2DTexture, point-sampled: DESIGN.md 3.15) when a
  // matte material names it as Kd; every spectrum texture is also reduced to one constant colour, which is what a parameter other
  // than a matte Kd gets.
  void texture(const std::string &name, const std::string &kind, const std::string &cls, const ParamSet &ps) {
    if (kind != "spectrum" && kind != "color" && kind != "rgb") return;  // float textures: nothing on the path uses them
    std::array<float, 3> c = {0.5f, 0.5f, 0.5f};
    float a[3] = {1, 1, 1}, b[3] = {0, 0, 0};
    if (cls == "constant") {
      spectrum(ps, "value", a);
      c = {a[0], a[1], a[2]};
    } else if (cls == "checkerboard") {
      spectrum(ps, "tex1", a);
      spectrum(ps, "tex2", b);
      c = {0.5f * (a[0] + b[0]), 0.5f * (a[1] + b[1]), 0.5f * (a[2] + b[2])};
      const int dim = ps.one_int("dimension", 2);
      std::string mapping = "uv", aamode = "closedform";
      if (const ParamItem *p = ps.find("mapping", "string")) if (p->strs.size() == 1) mapping = p->strs[0];
      if (const ParamItem *p = ps.find("aamode", "string")) if (p->strs.size() == 1) aamode = p->strs[0];
      if (dim == 2 && mapping == "uv") {
        pbrt_hip_texture t{};
        t.type = 0;
        for (int i = 0; i < 3; i++) { t.tex1[i] = a[i]; t.tex2[i] = b[i]; }
        t.su = ps.one_float("uscale", 1.f); t.sv = ps.one_float("vscale", 1.f);
        t.du = ps.one_float("udelta", 0.f); t.dv = ps.one_float("vdelta", 0.f);
        out->textures.push_back(t);
        checker_textures[name] = (uint32_t)out->textures.size();
        if (aamode != "none") warn("Texture \"" + name + "\" (checkerboard): point-sampled (aamode \"none\": ray differentials are not carried)");
      } else {
        warn("Texture \"" + name + "\" (checkerboard): only dimension 2 with the \"uv\" mapping keeps its pattern, replaced by the mean of its two colours");
      }
    } else if (cls == "mix" || cls == "dots") {
      spectrum(ps, cls == "dots" ? "inside" : "tex1", a);
      spectrum(ps, cls == "dots" ? "outside" : "tex2", b);
      c = {0.5f * (a[0] + b[0]), 0.5f * (a[1] + b[1]), 0.5f * (a[2] + b[2])};
      warn("Texture \"" + name + "\" (" + cls + "): not supported, replaced by the mean of its two colours");
    } else if (cls == "scale") {
      b[0] = b[1] = b[2] = 1.f;
      spectrum(ps, "tex1", a);
      spectrum(ps, "tex2", b);
      c = {a[0] * b[0], a[1] * b[1], a[2] * b[2]};
    } else {
      warn("Texture \"" + name + "\" (" + cls + "): not supported, replaced by grey 0.5");
    }
    spectrum_textures[name] = c;
    report_unused("Texture", ps);
  }

  // vertices (object space) + triangles of one mesh shape, through the CTM, into the scene's arrays
  void add_mesh(const float *P, uint32_t nv, const uint32_t *tri, size_t n_tri, const float *uv, uint16_t mid) {
    const float *M = ctm[0].m;
    const uint32_t base = (uint32_t)(out->P.size() / 3);
    for (uint32_t v = 0; v < nv; v++) {
      float q[3];
      xf_point(M, P + 3 * (size_t)v, q);
      out->P.insert(out->P.end(), q, q + 3);
    }
    const bool flip = gs.reverse_orientation ^ swaps_handedness(M);
    for (size_t t = 0; t < n_tri; t++) {
      uint32_t a = tri[3 * t], b = tri[3 * t + 1], c = tri[3 * t + 2];
      if (a >= nv || b >= nv || c >= nv) { warn("triangle mesh index out of range: triangle skipped"); continue; }
      if (flip) std::swap(b, c);
      out->idx.push_back(base + a); out->idx.push_back(base + b); out->idx.push_back(base + c);
      out->mat_id.push_back(mid);
      // per-vertex (u, v) if the mesh has them, else Triangle::GetUVs' (0,0) (1,0) (1,1)
      const uint32_t corner[3] = {a, b, c};
      const float dflt[6] = {0.f, 0.f, 1.f, 0.f, 1.f, 1.f};
      for (int v = 0; v < 3; v++)
        for (int k = 0; k < 2; k++) out->tri_uv.push_back(uv ? uv[2 * (size_t)corner[v] + k] : dflt[2 * (flip && v ? 3 - v : v) + k]);
    }
  }

  void shape(const std::string &name, const ParamSet &ps) {
    warn_if_animated_transform("pbrt.shape");
    const float zero[3] = {0, 0, 0};
    if (in_object && gs.has_area_light && !warned_object_light) {
      warned_object_light = true;
      warn("Area lights not supported with object instancing");  // pbrt-v3 api.cpp pbrtShape: the shape is kept, its emission is not
    }
    const float *le = (gs.has_area_light && !in_object) ? gs.area_le : zero;
    if (out->mats.size() >= 65535) { warn("more than 65535 materials: shape skipped"); return; }
    const uint16_t mid = material_id(gs.material, le);
    const float *M = ctm[0].m;
    if (name == "sphere") {
      const float r = ps.one_float("radius", 1.f);
      const float o[3] = {0, 0, 0}, ex[3] = {1, 0, 0};
      float c[3], sx[3];
      xf_point(M, o, c);
      xf_vector(M, ex, sx);
      pbrt_hip_sphere s{};
      for (int i = 0; i < 3; i++) s.c[i] = c[i];
      s.r = r * std::sqrt(sx[0] * sx[0] + sx[1] * sx[1] + sx[2] * sx[2]);  // uniform scale assumed
      s.mat = mid;
      out->spheres.push_back(s);
      if (gs.has_area_light && !in_object) warn("sphere area lights emit but are not sampled by the direct-light estimate");
      if (gs.material.kd_tex) {  // (u, v) = (phi / 2 pi, 1 - theta / pi) about the WORLD's z axis: a sphere is a centre and a radius here
        bool axes_kept = true;
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++)
            if (i != j && M[4 * i + j] != 0.f) axes_kept = false;
        if (!axes_kept || M[0] < 0.f || M[5] < 0.f || M[10] < 0.f)
          warn("a textured sphere under a rotating / mirroring transform: its (u, v) follow the world's axes, not the transformed ones");
      }
    } else if (name == "trianglemesh") {
      const ParamItem *pi = ps.find("indices", "integer");
      const ParamItem *pp = ps.find("P", "point3");
      if (!pi || !pp || pi->nums.size() % 3 || pp->nums.size() % 3) { warn("trianglemesh without valid indices / P: skipped"); return; }
      const uint32_t nv = (uint32_t)(pp->nums.size() / 3);
      std::vector<float> P(pp->nums.begin(), pp->nums.end());
      // per-vertex (u, v): "uv" or "st" (pbrt-v3 CreateTriangleMeshShape; floats or point2s)
      const ParamItem *puv = ps.find("uv", "float", "point2");
      if (!puv) puv = ps.find("st", "float", "point2");
      else ps.find("st", "float", "point2");
      if (puv && puv->nums.size() != 2 * (size_t)nv) { warn("trianglemesh: \"uv\" / \"st\" does not hold two numbers per vertex: ignored"); puv = nullptr; }
      std::vector<float> uv;
      if (puv) uv.assign(puv->nums.begin(), puv->nums.end());
      std::vector<uint32_t> tri(pi->nums.size());
      for (size_t i = 0; i < tri.size(); i++)  // (a negative or huge index becomes one that is out of range)
        tri[i] = (pi->nums[i] >= 0 && pi->nums[i] < 4294967295.0) ? (uint32_t)pi->nums[i] : 0xffffffffu;
      add_mesh(P.data(), nv, tri.data(), tri.size() / 3, puv ? uv.data() : nullptr, mid);
      ps.find("N", "normal"); ps.find("S", "vector3");
    } else if (name == "plymesh") {  // pbrt-v3 CreatePLYMesh (plymesh.cpp): the mesh of a PLY file, relative names against the scene file's directory
      const std::string fn = ps.one_string("filename", "");
      if (fn.empty()) { warn("plymesh without a \"string filename\": skipped"); return; }
      const std::string path = (fn[0] == '/' || cur_dir.empty()) ? fn : cur_dir + "/" + fn;
      PlyMesh mesh;
      std::string perr;
      if (!read_ply(path, &mesh, &perr)) { warn("plymesh \"" + fn + "\": " + perr + ": shape skipped"); return; }  // (pbrt-v3 logs the error and goes on)
      if (mesh.skipped_faces) warn("plymesh \"" + fn + "\": " + std::to_string(mesh.skipped_faces) + " faces with other than 3 or 4 vertices ignored");
      if (mesh.idx.empty()) { warn("plymesh \"" + fn + "\": no faces: skipped"); return; }
      if (out->idx.size() / 3 + mesh.idx.size() / 3 > kMaxSceneTriangles) { warn("plymesh \"" + fn + "\": scene would pass 2^24 triangles: skipped"); return; }
      add_mesh(mesh.P.data(), (uint32_t)(mesh.P.size() / 3), mesh.idx.data(), mesh.idx.size() / 3, mesh.uv.empty() ? nullptr : mesh.uv.data(), mid);
      if (ps.find("alpha", "texture", "float") || ps.find("shadowalpha", "texture", "float")) warn("plymesh: alpha / shadowalpha are not supported (opaque)");
    } else {
      warn("Shape \"" + name + "\" is not supported by this path: skipped");
      return;
    }
    report_unused("Shape", ps);
  }

  void swap_geometry(ObjectDef &d) {
    out->P.swap(d.P); out->tri_uv.swap(d.tri_uv); out->idx.swap(d.idx); out->mat_id.swap(d.mat_id); out->spheres.swap(d.spheres);
  }
  void object_begin(const std::string &name) {  // pbrtObjectBegin: an AttributeBegin, then shapes go to the named object
    if (in_object) { warn("ObjectBegin called inside of instance definition"); return; }
    in_object = true;
    object_name = name;
    stash = ObjectDef();
    swap_geometry(stash);  // out->... is empty now and collects the object's shapes; stash holds the scene
  }
  void object_end() {
    if (!in_object) { warn("ObjectEnd called outside of instance definition"); return; }
    ObjectDef def;
    swap_geometry(def);    // def = the object's shapes
    swap_geometry(stash);  // the scene is back
    stash = ObjectDef();
    objects[object_name] = std::move(def);
    in_object = false;
  }
  bool object_instance(const std::string &name) {  // false: the scene would pass kMaxSceneTriangles
    if (in_object) { warn("ObjectInstance can't be called inside instance definition"); return true; }
    auto it = objects.find(name);
    if (it == objects.end()) { warn("Unable to find instance named \"" + name + "\""); return true; }
    const ObjectDef &d = it->second;
    if (out->idx.size() / 3 + d.idx.size() / 3 > kMaxSceneTriangles || out->spheres.size() + d.spheres.size() > kMaxSceneTriangles) return false;
    // ... and its VERTICES: every instance copies the object's whole P array (unreferenced vertices included), so a vertex-heavy object
    // with few faces, instanced 10^5 times, stays under the triangle bound and asks for terabytes: the same refusal
    if (out->P.size() / 3 + d.P.size() / 3 > 3 * kMaxSceneTriangles) return false;
    const float *M = ctm[0].m;
    const uint32_t base = (uint32_t)(out->P.size() / 3);
    for (size_t v = 0; v + 2 < d.P.size(); v += 3) {
      float q[3];
      xf_point(M, d.P.data() + v, q);
      out->P.insert(out->P.end(), q, q + 3);
    }
    // a mirroring instance transform turns the winding round: the geometric normal must stay the transformed normal
    const bool flip = swaps_handedness(M);
    for (size_t t = 0; t + 2 < d.idx.size(); t += 3) {
      const int o1 = flip ? 2 : 1, o2 = flip ? 1 : 2;
      out->idx.push_back(base + d.idx[t]); out->idx.push_back(base + d.idx[t + o1]); out->idx.push_back(base + d.idx[t + o2]);
      const float *uv = d.tri_uv.data() + 2 * t;
      const float c[6] = {uv[0], uv[1], uv[2 * o1], uv[2 * o1 + 1], uv[2 * o2], uv[2 * o2 + 1]};
      out->tri_uv.insert(out->tri_uv.end(), c, c + 6);
    }
    out->mat_id.insert(out->mat_id.end(), d.mat_id.begin(), d.mat_id.end());
    const float ex[3] = {1, 0, 0};
    float sx[3];
    xf_vector(M, ex, sx);
    const float scale = std::sqrt(sx[0] * sx[0] + sx[1] * sx[1] + sx[2] * sx[2]);  // uniform scale assumed, as for a sphere's own CTM
    for (pbrt_hip_sphere s : d.spheres) {
      float c[3];
      xf_point(M, s.c, c);
      for (int i = 0; i < 3; i++) s.c[i] = c[i];
      s.r *= scale;
      out->spheres.push_back(s);
    }
    return true;
  }

  void light_source(const std::string &name, const ParamSet &ps) {  // replaces make_light's todo!()s, api.rs:334-351
    warn_if_animated_transform("pbrt.light_source");  // api.rs:687
    float scale[3] = {1, 1, 1};
    spectrum(ps, "scale", scale);
    pbrt_hip_light l{};
    if (name == "point") {
      float I[3] = {1, 1, 1}, from[3] = {0, 0, 0}, p[3];
      spectrum(ps, "I", I);
      ps.point3("from", from);
      xf_point(ctm[0].m, from, p);
      l.type = 0;
      for (int i = 0; i < 3; i++) { l.p[i] = p[i]; l.c[i] = I[i] * scale[i]; }
    } else if (name == "distant") {
      float L[3] = {1, 1, 1}, from[3] = {0, 0, 0}, to[3] = {0, 0, 1}, d[3], w[3];
      spectrum(ps, "L", L);
      ps.point3("from", from);
      ps.point3("to", to);
      for (int i = 0; i < 3; i++) d[i] = from[i] - to[i];
      xf_vector(ctm[0].m, d, w);
      F3 u = unit3({w[0], w[1], w[2]});
      l.type = 1;
      l.p[0] = u.x; l.p[1] = u.y; l.p[2] = u.z;
      for (int i = 0; i < 3; i++) l.c[i] = L[i] * scale[i];
    } else if (name == "infinite" || name == "exinfinite") {
      float L[3] = {1, 1, 1};
      spectrum(ps, "L", L);
      if (ps.find("mapname", "string")) warn("infinite light: environment maps are out of scope, constant L used");
      ps.find("samples", "integer"); ps.find("nsamples", "integer");
      l.type = 2;
      for (int i = 0; i < 3; i++) l.c[i] = L[i] * scale[i];
    } else {
      warn("light_source: light type '" + name + "' unknown.");  // api.rs:692 (spot, goniometric, projection: not on this path either)
      return;
    }
    out->lights.push_back(l);
    report_unused("LightSource", ps);
  }
};

struct Parser {
  struct File {
    std::string text, dir;
    Tokenizer tok;
    File(std::string t, std::string d) : text(std::move(t)), dir(std::move(d)), tok(nullptr, 0) {}
  };
  std::vector<std::unique_ptr<File>> files;  // parser.rs:206 file_stack
  bool have_unget = false;
  std::string unget;
  ParseError err = ParseError::None;
  std::string msg;

  bool fail(ParseError e, const std::string &m) { err = e; msg = m; return false; }

  void push_file(std::string text, std::string dir) {
    files.emplace_back(new File(std::move(text), std::move(dir)));
    File &f = *files.back();
    f.tok = Tokenizer(f.text.data(), f.text.size());
  }

  // parser.rs:323-352 next_token: comments skipped, files popped at EOF
  bool next(std::string *tok, bool required) {
    if (have_unget) { have_unget = false; *tok = unget; return true; }
    for (;;) {
      if (files.empty()) {
        if (required) fail(ParseError::Eof, "");
        return false;
      }
      ParseError e;
      if (!files.back()->tok.next(tok, &e)) { files.pop_back(); continue; }
      if (e != ParseError::None) return fail(e, e == ParseError::Eof ? "premature EOF inside a quoted string" : "unterminated string");
      if (!tok->empty() && (*tok)[0] == '#') continue;
      return true;
    }
  }

  bool number(float *out) {
    std::string t;
    if (!next(&t, true)) return false;
    char *end = nullptr;
    const double v = std::strtod(t.c_str(), &end);
    if (end == t.c_str() || *end) return fail(ParseError::Syntax, "input not float: '" + t + "'");  // parser.rs:37-38 NumberErr
    *out = (float)v;
    return true;
  }
  bool numbers(float *out, int n) {
    for (int i = 0; i < n; i++)
      if (!number(out + i)) return false;
    return true;
  }
  bool quoted(std::string *out) {
    std::string t;
    if (!next(&t, true)) { if (err == ParseError::None) fail(ParseError::Unquoted, ""); return false; }
    if (!is_quoted(t)) return fail(ParseError::Unquoted, "got '" + t + "'");
    *out = dequote(t);
    return true;
  }

  // parser.rs:354-414 parse_params + :504-738 add_param (type table :433-475)
  bool params(ParamSet *ps, Api &api) {
    for (;;) {
      std::string decl;
      if (!next(&decl, false)) return err == ParseError::None;
      if (!is_quoted(decl)) { have_unget = true; unget = decl; return true; }
      ParamItem item;
      std::string d = dequote(decl);
      size_t a = d.find_first_not_of(" \t");
      size_t b = a == std::string::npos ? a : d.find_first_of(" \t", a);
      bool ok_decl = a != std::string::npos && b != std::string::npos;
      if (ok_decl) {
        item.type = d.substr(a, b - a);
        size_t c = d.find_first_not_of(" \t", b);
        if (c == std::string::npos) ok_decl = false;
        else {
          item.name = d.substr(c);
          item.name.erase(item.name.find_last_not_of(" \t") + 1);
        }
      }
      auto add = [&](const std::string &v) -> bool {
        if (is_quoted(v)) {
          if (!item.nums.empty()) return fail(ParseError::MixedParameters, "mixed string and numeric parameters");
          item.strs.push_back(dequote(v));
        } else {
          if (!item.strs.empty()) return fail(ParseError::MixedParameters, "mixed string and numeric parameters");
          if (v == "true" || v == "false") { item.strs.push_back(v); return true; }  // unquoted bools (pbrt-v3 accepts them)
          char *end = nullptr;
          const double x = std::strtod(v.c_str(), &end);
          if (end == v.c_str() || *end) return fail(ParseError::Syntax, "input not float: '" + v + "'");
          item.nums.push_back(x);
        }
        return true;
      };
      std::string v;
      if (!next(&v, true)) return false;
      if (v == "[") {
        for (;;) {
          if (!next(&v, true)) return false;
          if (v == "]") break;
          if (!add(v)) return false;
        }
      } else if (!add(v)) {
        return false;
      }
      if (!ok_decl) { api.warn("parameter \"" + d + "\" has no type or no name: ignored"); continue; }  // parser.rs:477-502
      static const std::map<std::string, std::string> canon = {
          {"float", "float"}, {"integer", "integer"}, {"bool", "bool"}, {"point2", "point2"}, {"vector2", "vector2"},
          {"point3", "point3"}, {"vector3", "vector3"}, {"point", "point3"}, {"vector", "vector3"}, {"normal", "normal"},
          {"string", "string"}, {"texture", "texture"}, {"color", "rgb"}, {"rgb", "rgb"}, {"xyz", "xyz"},
          {"blackbody", "blackbody"}, {"spectrum", "spectrum"}};
      auto it = canon.find(item.type);
      if (it == canon.end()) { api.warn("unknown parameter type '" + item.type + "': ignored"); continue; }
      item.type = it->second;
      ps->items.push_back(std::move(item));
    }
  }

  // parser.rs:416-429 basic_param_list_entrypoint
  bool named_params(std::string *name, ParamSet *ps, Api &api) { return quoted(name) && params(ps, api); }
};

std::pair<uint32_t, uint32_t> strata_for(int n) {  // spp = nx * ny, nx >= ny, as square as the factors allow
  if (n < 1) n = 1;
  int ny = (int)std::floor(std::sqrt((double)n));
  while (ny > 1 && n % ny) ny--;
  return {(uint32_t)(n / ny), (uint32_t)ny};
}

bool read_file(const std::string &path, std::string *out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::ostringstream ss;
  ss << f.rdbuf();
  *out = ss.str();
  return true;
}

}  // namespace

ParseError parse_scene(const char *text, size_t len, const std::string &base_dir, LoadedScene *out, std::string *msg) {
  Api api(out);
  Parser p;
  p.push_file(std::string(text, len), base_dir);
  auto fin = [&](bool ok) {
    std::memcpy(out->final_ctm, api.ctm[0].m, 64);
    if (!ok && msg) *msg = p.msg;
    return ok ? ParseError::None : (p.err == ParseError::None ? ParseError::Syntax : p.err);
  };
  // the reference's name for a directive's API call: "AttributeBegin" -> "pbrt.attribute_begin" (api.rs:421-814)
  auto api_name = [](const char *directive) {
    std::string r = "pbrt.";
    for (const char *c = directive; *c; c++) {
      if (*c >= 'A' && *c <= 'Z') { if (c != directive) r += '_'; r += (char)(*c - 'A' + 'a'); }
      else r += *c;
    }
    return r;
  };
  auto in_world = [&](const char *what) {  // verify_world!, api.rs:320-332: log (its wording) and ignore
    if (api.state == ApiState::WorldBlock) return true;
    api.warn("Scene description must be inside world block; \"" + api_name(what) + "\" not allowed. Ignoring.");
    return false;
  };
  auto in_options = [&](const char *what) {  // verify_options!, api.rs:304-316
    if (api.state == ApiState::OptionsBlock) return true;
    api.warn("Options cannot be set inside world block; \"" + api_name(what) + "\" not allowed. Ignoring.");
    return false;
  };
  std::string tok;
  while (p.next(&tok, false)) {
    api.cur_dir = p.files.empty() ? base_dir : p.files.back()->dir;
    std::string name;
    ParamSet ps;
    float v[16];
    if (tok == "Accelerator") {
      if (!p.named_params(&name, &ps, api)) return fin(false);
      if (in_options("Accelerator")) { out->accelerator_name = name; if (name != "bvh") api.warn("Accelerator \"" + name + "\": this path always uses its BVH"); }
    } else if (tok == "ActiveTransform") {  // api.rs:733-747
      std::string which;
      if (!p.next(&which, true)) return fin(false);
      if (which == "All") api.active_bits = kAllBits;
      else if (which == "StartTime") api.active_bits = kStartBit;
      else if (which == "EndTime") api.active_bits = kEndBit;
      else return fin(p.fail(ParseError::Syntax, "ActiveTransform " + which));
    } else if (tok == "AreaLightSource") {
      if (!p.named_params(&name, &ps, api)) return fin(false);
      if (in_world("AreaLightSource")) {
        if (name != "diffuse") api.warn("AreaLightSource \"" + name + "\": treated as \"diffuse\"");
        float L[3] = {1, 1, 1}, sc[3] = {1, 1, 1};
        api.spectrum(ps, "L", L);
        api.spectrum(ps, "scale", sc);
        if (ps.one_bool("twosided", false)) api.warn("AreaLightSource: twosided is not supported (one-sided emission)");
        ps.find("samples", "integer"); ps.find("nsamples", "integer");
        api.gs.has_area_light = true;
        for (int i = 0; i < 3; i++) api.gs.area_le[i] = L[i] * sc[i];
        api.report_unused("AreaLightSource", ps);
      }
    } else if (tok == "AttributeBegin") {  // the reference matches the misspelling "AttrbuteBegin" (parser.rs:233)
      if (in_world("AttributeBegin")) {
        api.pushed_gs.push_back(api.gs);
        api.pushed_ctm.emplace_back(api.ctm[0], api.ctm[1]);
        api.pushed_bits.push_back(api.active_bits);
      }
    } else if (tok == "AttributeEnd") {
      if (in_world("AttributeEnd")) {
        if (api.pushed_gs.empty() || api.pushed_ctm.empty()) {
          api.warn("Unmatched pbrt.attribute_end() encountered. Ignoring it.");  // api.rs:497
        } else {
          api.gs = api.pushed_gs.back(); api.pushed_gs.pop_back();
          api.ctm[0] = api.pushed_ctm.back().first; api.ctm[1] = api.pushed_ctm.back().second; api.pushed_ctm.pop_back();
          api.active_bits = api.pushed_bits.back(); api.pushed_bits.pop_back();
        }
      }
    } else if (tok == "Camera") {  // api.rs:813-820: camera_to_world = CTM^-1, named "camera"
      if (!p.named_params(&name, &ps, api)) return fin(false);
      if (in_options("Camera")) {
        out->camera_name = name;
        if (name != "perspective") api.warn("Camera \"" + name + "\": only \"perspective\" is implemented, used instead");
        out->fov = ps.one_float("fov", 90.f);
        std::memcpy(out->cam_to_world, api.ctm[0].inv, 64);
        Xform c2w;
        std::memcpy(c2w.m, api.ctm[0].inv, 64);
        std::memcpy(c2w.inv, api.ctm[0].m, 64);
        api.named_cs["camera"] = {c2w, c2w};
        api.camera_set = true;
        // what the pinhole camera of this path does not model is said, not dropped in silence
        if (ps.one_float("lensradius", 0.f) != 0.f) api.warn("Camera: \"lensradius\" ignored (pinhole camera: no depth of field)");
        if (ps.find("screenwindow", "float") || ps.find("frameaspectratio", "float"))
          api.warn("Camera: \"screenwindow\" / \"frameaspectratio\" ignored (the screen window follows the film's aspect ratio)");
        ps.find("focaldistance", "float"); ps.find("shutteropen", "float"); ps.find("shutterclose", "float");
        api.report_unused("Camera", ps);
      }
    } else if (tok == "ConcatTransform" || tok == "Transform") {
      std::string br;
      if (!p.next(&br, true)) return fin(false);
      const bool bracket = br == "[";
      if (!bracket) { p.have_unget = true; p.unget = br; }
      if (!p.numbers(v, 16)) return fin(false);
      if (bracket && (!p.next(&br, true) || br != "]")) return fin(p.fail(ParseError::Syntax, "expected ]"));
      float m[16];
      for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) m[4 * i + j] = v[4 * j + i];  // scene files are column-major (pbrt-v3)
      const Xform t = xf_from_matrix(m);
      if (tok == "Transform") api.for_active([&](Xform &c) { c = t; });
      else api.concat(t);
    } else if (tok == "CoordinateSystem") {
      if (!p.quoted(&name)) return fin(false);
      api.named_cs[name] = {api.ctm[0], api.ctm[1]};
    } else if (tok == "CoordSysTransform") {
      if (!p.quoted(&name)) return fin(false);
      auto it = api.named_cs.find(name);
      if (it == api.named_cs.end()) api.warn("Couldn\xe2\x80\x99t find named coordinate system \"" + name + "\"");  // api.rs:745 (its apostrophe is U+2019)
      else { api.ctm[0] = it->second.first; api.ctm[1] = it->second.second; }
    } else if (tok == "Film") {
      if (!p.named_params(&name, &ps, api)) return fin(false);
      if (in_options("Film")) {
        out->film_name = name;
        out->xres = ps.one_int("xresolution", 1280);
        out->yres = ps.one_int("yresolution", 720);
        out->filename = ps.one_string("filename", "pbrt.png");
        out->film_scale = ps.one_float("scale", 1.f);
        if (const ParamItem *c = ps.find("cropwindow", "float"))
          if (c->nums.size() == 4) for (int i = 0; i < 4; i++) out->crop[i] = (float)c->nums[i];
        ps.find("diagonal", "float");
        {
          const float ml = ps.one_float("maxsampleluminance", 0.f);  // (pbrt-v3's default is infinity: 0 stands for it here)
          out->max_sample_luminance = (ml > 0.f && std::isfinite(ml)) ? ml : 0.f;
        }
        api.report_unused("Film", ps);
      }
    } else if (tok == "Identity") {
      api.for_active([&](Xform &c) { c = xf_identity(); });
    } else if (tok == "Include") {
      if (!p.quoted(&name)) return fin(false);
      const std::string dir = p.files.empty() ? base_dir : p.files.back()->dir;
      const std::string path = (!name.empty() && name[0] == '/') ? name : (dir.empty() ? name : dir + "/" + name);
      // (a file that includes itself, directly or through others, would never end: the reference's file_stack,
      // parser.rs:206, has no guard either; 32 levels are more than any real scene nests)
      if (p.files.size() >= 32) return fin(p.fail(ParseError::Syntax, "Include: nesting too deep / recursive include of '" + path + "'"));
      std::string text2;
      if (!read_file(path, &text2)) return fin(p.fail(ParseError::Io, "Include: cannot read '" + path + "'"));
      const size_t slash = path.rfind('/');
      p.push_file(std::move(text2), slash == std::string::npos ? "" : path.substr(0, slash));
    } else if (tok == "Integrator") {
      if (!p.named_params(&name, &ps, api)) return fin(false);
      if (in_options("Integrator")) {
        out->integrator_name = name;
        out->max_depth = (uint32_t)std::max(0, ps.one_int("maxdepth", 5));
        if (name == "directlighting") out->integrator = PBRT_HIP_INTEGRATOR_DIRECT;
        else {
          // `Integrator "path"` in a scene file MEANS pbrt-v3's path integrator (the reference's default name, api.rs:239; its render sketch
          // is pbrt-v3's, api.rs:446-453), whose direct-light estimate is multiple-importance-sampled: integrator 2 since
          // round 6 (VERDICT r05 missing 2 / item 8).  "bool mis" "false" selects SURVEY A8's estimator without it (integrator 0: what
          // BASELINE's synthetic configs, which name their integrator through the C ABI, are measured and pinned on).
          out->integrator = ps.one_bool("mis", true) ? PBRT_HIP_INTEGRATOR_PATH_MIS : PBRT_HIP_INTEGRATOR_PATH;
          if (name != "path") api.warn("Integrator \"" + name + "\": only \"path\" and \"directlighting\" exist, \"path\" used");
        }
        if (ps.one_float("rrthreshold", 1.f) != 1.f) api.warn("Integrator: \"rrthreshold\" ignored (Russian roulette from the fourth bounce on with q = max(0.05, 1 - max beta))");
        {
          const std::string lss = ps.one_string("lightsamplestrategy", "uniform"), st = ps.one_string("strategy", "one");
          if (lss != "uniform") api.warn("Integrator: \"lightsamplestrategy\" \"" + lss + "\" ignored (one light chosen uniformly)");
          if (st != "one") api.warn("Integrator: \"strategy\" \"" + st + "\" ignored (one light per vertex, chosen uniformly, x the number of lights)");
        }
        api.report_unused("Integrator", ps);
      }
    } else if (tok == "LightSource") {
      if (!p.named_params(&name, &ps, api)) return fin(false);
      if (in_world("LightSource")) api.light_source(name, ps);
    } else if (tok == "LookAt") {  // parser.rs:250-269 -> api.rs:677-682
      if (!p.numbers(v, 9)) return fin(false);
      Xform t;
      look_at(v, v + 3, v + 6, t.m, t.inv);
      api.concat(t);
    } else if (tok == "MakeNamedMaterial") {
      if (!p.named_params(&name, &ps, api)) return fin(false);
      api.named_materials[name] = api.make_material(ps.one_string("type", "matte"), ps);
    } else if (tok == "Material") {
      if (!p.named_params(&name, &ps, api)) return fin(false);
      if (in_world("Material")) { api.gs.material = api.make_material(name, ps); api.report_unused("Material", ps); }
    } else if (tok == "NamedMaterial") {
      if (!p.quoted(&name)) return fin(false);
      auto it = api.named_materials.find(name);
      if (it == api.named_materials.end()) api.warn("NamedMaterial \"" + name + "\" is not defined");
      else api.gs.material = it->second;
    } else if (tok == "PixelFilter") {
      if (!p.named_params(&name, &ps, api)) return fin(false);
      if (in_options("PixelFilter")) {
        out->filter_name = name;
        if (name != "box") api.warn("PixelFilter \"" + name + "\": only the box filter is implemented (box.rs:57-61); its radii are used");
        const float xw = ps.one_float("xwidth", 0.5f), yw = ps.one_float("ywidth", 0.5f);
        out->filter_radius[0] = xw; out->filter_radius[1] = yw;  // handed to the render desc (radii other than 0.5: DESIGN.md 3.11)
      }
    } else if (tok == "ReverseOrientation") {
      if (in_world("ReverseOrientation")) api.gs.reverse_orientation = !api.gs.reverse_orientation;
    } else if (tok == "Rotate") {
      if (!p.numbers(v, 4)) return fin(false);
      api.concat(xf_rotate(v[0], v[1], v[2], v[3]));
    } else if (tok == "Sampler") {
      if (!p.named_params(&name, &ps, api)) return fin(false);
      if (in_options("Sampler")) {
        out->sampler_name = name;
        if (name == "stratified") {
          out->sampler = PBRT_HIP_SAMPLER_STRATIFIED;
          out->spp_x = (uint32_t)std::max(1, ps.one_int("xsamples", 4));
          out->spp_y = (uint32_t)std::max(1, ps.one_int("ysamples", 4));
          if (!ps.one_bool("jitter", true)) api.warn("Sampler \"stratified\": \"jitter\" \"false\" ignored (samples are jittered inside their strata)");
          ps.find("dimensions", "integer");
        } else {
          // "halton" (the reference's default name, api.rs:235) and "sobol" have samplers of their own; the
          // other low-discrepancy names are served by the padded (0,2)-sequence sampler of 3.10, "random" and anything unknown
          // by the stratified one; pixelsamples = spp_x * spp_y either way
          auto s = strata_for(ps.one_int("pixelsamples", 16));
          out->spp_x = s.first; out->spp_y = s.second;
          if (name == "sobol") {
            out->sampler = PBRT_HIP_SAMPLER_SOBOL_ND;  // Sobol' proper: own dimensions per request
          } else if (name == "halton") {
            out->sampler = PBRT_HIP_SAMPLER_HALTON;    // the reference's default name (api.rs:235): scrambled radical inverses
          } else if (name == "02sequence" || name == "lowdiscrepancy" || name == "zerotwosequence" || name == "maxmindist") {
            out->sampler = PBRT_HIP_SAMPLER_SOBOL;
            if (name == "maxmindist") api.warn("Sampler \"" + name + "\": served by the (0,2)-sequence (Sobol') sampler");
          } else {
            out->sampler = PBRT_HIP_SAMPLER_STRATIFIED;
            if (name != "random") api.warn("Sampler \"" + name + "\" is unknown: the stratified sampler is used");
          }
        }
        api.report_unused("Sampler", ps);
      }
    } else if (tok == "Scale") {  // parser.rs:292-299
      if (!p.numbers(v, 3)) return fin(false);
      api.concat(xf_scale(v[0], v[1], v[2]));
    } else if (tok == "Shape") {
      if (!p.named_params(&name, &ps, api)) return fin(false);
      if (in_world("Shape")) api.shape(name, ps);
    } else if (tok == "Texture") {  // Texture "name" "spectrum|color|float" "class" params (api.rs:524-580)
      std::string kind, cls;
      if (!p.quoted(&name) || !p.quoted(&kind) || !p.quoted(&cls) || !p.params(&ps, api)) return fin(false);
      if (in_world("Texture")) api.texture(name, kind, cls, ps);
    } else if (tok == "TransformBegin") {
      if (in_world("TransformBegin")) { api.pushed_ctm.emplace_back(api.ctm[0], api.ctm[1]); api.pushed_bits.push_back(api.active_bits); }  // api.rs:504-509
    } else if (tok == "TransformEnd") {
      if (in_world("TransformEnd")) {
        if (api.pushed_ctm.empty()) api.warn("Unmatched pbrt.transform_end() encountered. Ignoring it.");  // api.rs:517
        else {
          api.ctm[0] = api.pushed_ctm.back().first; api.ctm[1] = api.pushed_ctm.back().second; api.pushed_ctm.pop_back();
          api.active_bits = api.pushed_bits.back(); api.pushed_bits.pop_back();
        }
      }
    } else if (tok == "Translate") {
      if (!p.numbers(v, 3)) return fin(false);
      api.concat(xf_translate(v[0], v[1], v[2]));
    } else if (tok == "WorldBegin") {  // api.rs:420-429
      if (in_options("WorldBegin")) {
        api.state = ApiState::WorldBlock;
        api.ctm[0] = api.ctm[1] = xf_identity();
        api.active_bits = kAllBits;
        api.named_cs["world"] = {api.ctm[0], api.ctm[1]};
      }
    } else if (tok == "WorldEnd") {  // api.rs:432-473: the render call site
      if (in_world("WorldEnd")) {
        if (api.in_object) { api.warn("Missing end to ObjectBegin"); api.object_end(); }
        if (!api.pushed_gs.empty() || !api.pushed_ctm.empty()) api.warn("Missing end to AttributeBegin / TransformBegin");
        api.state = ApiState::OptionsBlock;
        out->world_ended = true;
        break;  // one render per file (the reference would continue to a next frame, api.rs:458)
      }
    } else if (tok == "ObjectBegin") {  // pbrt-v3 pbrtObjectBegin = pbrtAttributeBegin + the named instance collects the shapes
      if (!p.quoted(&name)) return fin(false);
      if (in_world("ObjectBegin")) {
        api.pushed_gs.push_back(api.gs);
        api.pushed_ctm.emplace_back(api.ctm[0], api.ctm[1]);
        api.pushed_bits.push_back(api.active_bits);
        api.object_begin(name);
      }
    } else if (tok == "ObjectEnd") {
      if (in_world("ObjectEnd")) {
        api.object_end();
        if (!api.pushed_gs.empty() && !api.pushed_ctm.empty()) {  // pbrtAttributeEnd
          api.gs = api.pushed_gs.back(); api.pushed_gs.pop_back();
          api.ctm[0] = api.pushed_ctm.back().first; api.ctm[1] = api.pushed_ctm.back().second; api.pushed_ctm.pop_back();
          api.active_bits = api.pushed_bits.back(); api.pushed_bits.pop_back();
        }
      }
    } else if (tok == "ObjectInstance") {
      if (!p.quoted(&name)) return fin(false);
      if (in_world("ObjectInstance") && !api.object_instance(name))
        return fin(p.fail(ParseError::Syntax, "ObjectInstance \"" + name + "\": the scene would pass 2^24 triangles"));
    } else if (tok == "MakeNamedMedium" || tok == "MediumInterface" || tok == "TransformTimes") {
      return fin(p.fail(ParseError::NotImplemented, tok));  // parser.rs:270-310: out of scope here too
    } else {
      return fin(p.fail(ParseError::Syntax, tok));  // parser.rs:313
    }
  }
  if (p.err != ParseError::None) return fin(false);
  if (api.in_object) { api.warn("Missing end to ObjectBegin"); api.object_end(); }  // (the file ended inside an object: the scene's arrays come back)
  {  // corner (u, v) travel only when some triangle's material is textured
    bool textured = false;
    for (uint16_t m : out->mat_id) textured = textured || out->mats[m].kd_tex != 0u;
    if (!textured) { out->tri_uv.clear(); out->tri_uv.shrink_to_fit(); }
  }
  if (!api.camera_set) mat_identity(out->cam_to_world);
  // (round 6: spheres are primitives of the BVH like triangles -- the warning that hundreds of them cost every ray a loop is gone; what
  // pbrt_hip_scene_create takes is 2^24 primitives in all)
  if (out->idx.size() / 3 + out->spheres.size() > api.kMaxSceneTriangles)
    api.warn(std::to_string(out->idx.size() / 3) + " triangles + " + std::to_string(out->spheres.size()) + " spheres: more than the 2^24 primitives pbrt_hip_scene_create takes");
  return fin(true);
}

}  // namespace pbrt_hip
