// pbrt_main.cpp -- the command line of the render path, the C++ counterpart of the reference binary
// /root/reference/src/bin/pbrt.rs:24-85: same flags (-n/--nthreads, --quick, -q/--quiet, -v/--verbose,
// -o/--outfile, positional scene files), same three log levels (stderrlog verbosity 1 / 2 / 3, pbrt.rs:48-62: quiet = errors and
// WARNINGS, default = + info, verbose = + debug).  Where the reference parses and pretty-prints its state
// (pbrt.rs:72-83, WorldEnd renders nothing), this parses, renders on the GPU through the C ABI and
// writes the image named by Film "string filename" (or -o).
#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pbrt_hip.h"

namespace {
int g_level = 2;  // 1 quiet, 2 default, 3 verbose
void logf(int level, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
void logf(int level, const char *fmt, ...) {
  if (level > g_level) return;
  va_list ap;
  va_start(ap, fmt);
  std::vfprintf(stderr, fmt, ap);
  va_end(ap);
  std::fputc('\n', stderr);
}
void usage() {
  std::fprintf(stderr,
               "Parses a scene file and renders it on an MI355X.\n\n"
               "USAGE: pbrt [OPTIONS] [SCENE_FILES]...\n\n"
               "  -n, --nthreads <N>    accepted for compatibility (the GPU path ignores it; the reference never reads it)\n"
               "      --quick           quick render: a quarter of the samples per pixel\n"
               "  -q, --quiet           squelch all non-error output\n"
               "  -v, --verbose         enable extra logging output\n"
               "  -o, --outfile <FILE>  path to store the rendered output\n"
               "      --gpus <N>        GPUs of this node to render on (default: all visible ones; one process, RCCL gather)\n");
}
}  // namespace

int main(int argc, char **argv) {
  bool quick = false, quiet = false, verbose = false;
  int gpus = 0;  // 0 = all visible
  std::string outfile;
  std::vector<std::string> scenes;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    auto value = [&](const char *name) -> const char * {
      if (i + 1 >= argc) { std::fprintf(stderr, "error: %s needs a value\n", name); std::exit(2); }
      return argv[++i];
    };
    if (a == "-n" || a == "--nthreads") (void)std::atoi(value("--nthreads"));
    else if (a == "--quick") quick = true;
    else if (a == "-q" || a == "--quiet") quiet = true;
    else if (a == "-v" || a == "--verbose") verbose = true;
    else if (a == "-o" || a == "--outfile") outfile = value("--outfile");
    else if (a == "--gpus") gpus = std::atoi(value("--gpus"));
    else if (a == "-h" || a == "--help") { usage(); return 0; }
    else if (!a.empty() && a[0] == '-') { std::fprintf(stderr, "error: unknown option %s\n", a.c_str()); usage(); return 2; }
    else scenes.push_back(a);
  }
  g_level = verbose ? 3 : (quiet ? 1 : 2);
  if (scenes.empty()) { logf(1, "no scene files given"); return 1; }
  for (const std::string &path : scenes) {
    pbrt_hip_loaded *loaded = nullptr;
    if (pbrt_hip_load_file(path.c_str(), &loaded) != 0) { logf(1, "%s: %s", path.c_str(), pbrt_hip_last_error()); return 1; }
    std::vector<char> wbuf(1 << 16);
    if (pbrt_hip_loaded_warnings(loaded, wbuf.data(), wbuf.size()) > 0)
      for (char *w = std::strtok(wbuf.data(), "\n"); w; w = std::strtok(nullptr, "\n")) logf(1, "warning: %s", w);  // WARN passes every level (pbrt.rs:51-53: quiet is "only WARN and higher")
    pbrt_hip_scene_desc desc;
    pbrt_hip_render_desc rd;
    char filename[4096];
    pbrt_hip_loaded_get(loaded, &desc, &rd, filename, sizeof filename);
    const float film_scale = pbrt_hip_loaded_film_scale(loaded);  // Film "float scale", applied by Film::write_image (film.rs:368-371)
    if (quick) {
      rd.spp_x = rd.spp_x > 1 ? rd.spp_x / 2 : 1;
      rd.spp_y = rd.spp_y > 1 ? rd.spp_y / 2 : 1;
    }
    logf(2, "%s: %u triangles, %u spheres, %u lights, %dx%d, %u spp", path.c_str(), desc.n_tris, desc.n_spheres, desc.n_lights,
         desc.xres, desc.yres, rd.spp_x * rd.spp_y);
    const auto t0 = std::chrono::steady_clock::now();
    int32_t b[4];
    pbrt_hip_film_cropped_bounds(desc.xres, desc.yres, desc.crop, b);
    const int w = b[2] - b[0], h = b[3] - b[1];
    std::vector<float> film((size_t)w * h * 4), rgb((size_t)w * h * 3);
    // every GPU of the node from this one process (the reference binary is one process, bin/pbrt.rs:72-83): the scene
    // is copied device to device, GPU g renders the super-tiles t % n == g, one ncclGather assembles the film
    // (--gpus 0 = all visible devices, but no more than the film has 64x64 super-tiles: the library decides)
    const int n = std::max(1, pbrt_hip_device_count());
    std::vector<pbrt_hip_stats> per_gpu((size_t)n);  // zero-initialised: entries of GPUs that were not used stay empty
    const int rc = pbrt_hip_render_multi(&desc, &rd, gpus > 0 ? gpus : 0, film.data(), per_gpu.data());
    pbrt_hip_loaded_free(loaded);
    if (rc != 0) { logf(1, "%s: %s", path.c_str(), pbrt_hip_last_error()); return 1; }
    pbrt_hip_stats st = per_gpu[0];
    for (int g = 1; g < n; g++) { st.samples += per_gpu[g].samples; if (per_gpu[g].kernel_ms > st.kernel_ms) st.kernel_ms = per_gpu[g].kernel_ms; }
    pbrt_hip_film_to_rgb(film.data(), (int64_t)w * h, film_scale, rgb.data());  // Film::write_image, film.rs:340-372
    std::string out = outfile.empty() ? filename : outfile;
    if (out.rfind('.') == std::string::npos || out.rfind('.') < out.rfind('/') + 1) out += ".png";
    if (pbrt_hip_write_image(out.c_str(), rgb.data(), w, h) != 0) { logf(1, "cannot write '%s'", out.c_str()); return 1; }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    logf(2, "wrote %s: kernel %.1f ms (%.1f Msamples/s), total %.2f s", out.c_str(), st.kernel_ms,
         st.kernel_ms > 0 ? st.samples / st.kernel_ms / 1e3 : 0.0, secs);
  }
  return 0;
}
