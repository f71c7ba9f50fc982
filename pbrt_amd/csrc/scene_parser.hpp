// scene_parser.hpp -- .pbrt scene ingestion for the render path: tokenizer + parameter lists + the
// API state machine, restated in C++ from /root/reference/src/core/parser.rs (tokenizer :61-170,
// directive dispatch :205-317, parse_params :354-414, type table :433-475, add_param :504-738) and
// /root/reference/src/core/api.rs (CTM ops :588-747, attribute / transform stacks :481-522, option
// setters :778-820, world_begin/world_end :420-473).  The reference stops at 12 of its 37 directive
// arms (25 return NotImplemented, "AttributeBegin" is misspelt, parser.rs:233) and stores no
// geometry (api.rs:220-223); this implementation completes the arms the render path needs and
// collects the arrays of include/pbrt_hip.h's pbrt_hip_scene_desc.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "../../include/pbrt_hip.h"
#include "../../include/pbrt_hip_debug.h"

namespace pbrt_hip {

// parser.rs:31-58 `Error`
enum class ParseError {
  None = 0,
  Eof,                 // premature EOF (inside a quoted string, or a required token is missing)
  UnterminatedString,  // newline inside a quoted string
  MixedParameters,     // strings and numbers mixed in one parameter list
  Unquoted,            // a quoted string was required
  Syntax,              // unknown directive / malformed number
  NotImplemented,      // directive the render path does not cover (media, animated transforms)
  Io,
};

struct Tokenizer {
  const char *data;
  size_t len, pos = 0;
  Tokenizer(const char *d, size_t n) : data(d), len(n) {}
  // false at EOF; on error sets *err.  Comments are returned as tokens starting with '#'.
  bool next(std::string *tok, ParseError *err);
};

struct ParamItem {
  std::string type, name;       // "float" / "fov"
  std::vector<double> nums;     // numeric values as parsed (f64, parser.rs:198)
  std::vector<std::string> strs;
  mutable bool looked_up = false;
};

struct ParamSet {
  std::vector<ParamItem> items;
  const ParamItem *find(const char *name, const char *t1, const char *t2 = nullptr, const char *t3 = nullptr) const;
  float one_float(const char *name, float dflt) const;
  int one_int(const char *name, int dflt) const;
  bool one_bool(const char *name, bool dflt) const;
  std::string one_string(const char *name, const std::string &dflt) const;
  bool point3(const char *name, float out[3]) const;
  std::vector<std::string> unused() const;  // paramset.rs:519-531 report_unused
};

struct Xform {  // Transform {m, m_inv}, transform.rs:303-306 (row-major)
  float m[16], inv[16];
};

struct LoadedScene {
  // geometry / shading tables
  std::vector<float> P;
  std::vector<uint32_t> idx;
  std::vector<uint16_t> mat_id;
  std::vector<pbrt_hip_material> mats;
  std::vector<pbrt_hip_light> lights;
  std::vector<pbrt_hip_sphere> spheres;
  std::vector<pbrt_hip_texture> textures;  // checkerboards named by a material's "texture Kd" 
  std::vector<float> tri_uv;               // 6 per triangle (corner u, v); empty when no material is textured
  // RenderOptions (api.rs:201-249) resolved to values
  float cam_to_world[16];
  float fov = 90.f;
  int xres = 1280, yres = 720;
  float crop[4] = {0.f, 1.f, 0.f, 1.f};
  std::string filename = "pbrt.png";
  std::string camera_name = "perspective", sampler_name = "halton", integrator_name = "path", filter_name = "box",
              accelerator_name = "bvh", film_name = "image";  // defaults: api.rs:231-241
  float filter_radius[2] = {0.5f, 0.5f};                       // box.rs:57-61
  uint32_t integrator = PBRT_HIP_INTEGRATOR_PATH_MIS, max_depth = 5, spp_x = 4, spp_y = 4;  // (no Integrator directive: "path" as pbrt-v3 means it, api.rs:239)
  float max_sample_luminance = 0.f;  // Film "float maxsampleluminance" (film.rs:75,279); 0 = none
  float film_scale = 1.f;  // Film "float scale" (film.rs:368-371), exported by pbrt_hip_loaded_film_scale
  uint32_t sampler = PBRT_HIP_SAMPLER_HALTON;  // the default sampler name is "halton" (api.rs:235): DESIGN.md 3.13, see scene_parser.cpp "Sampler"
  bool world_ended = false;
  float final_ctm[16];  // CTM when parsing stopped (for the state-machine tests)
  std::vector<std::string> warnings;
};

// Parses `text`; `base_dir` resolves Include.  Returns ParseError::None or the first error with a
// message in *msg.
ParseError parse_scene(const char *text, size_t len, const std::string &base_dir, LoadedScene *out, std::string *msg);

}  // namespace pbrt_hip
