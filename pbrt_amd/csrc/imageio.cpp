// imageio.cpp -- imageio::write_image of the reference (core/imageio.rs:235-283) for the two
// formats it implements: 8-bit RGB PNG through to_byte (imageio.rs:66-68, :245-271) and PFM
// (imageio.rs:186-213: "PF\n{w} {h}\n{scale}\n", rows bottom to top, host-endian floats, scale -1 on
// little-endian hosts).  The PNG stream uses stored (uncompressed) deflate blocks: the reference
// delegates compression to the `png` crate, which carries no arithmetic of the path, and any
// conforming decoder returns the same bytes.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pbrt_hip.h"
#include "host_math.hpp"

namespace {

uint32_t crc_table[256];
bool crc_ready = false;
void crc_init() {
  for (uint32_t n = 0; n < 256; n++) {
    uint32_t c = n;
    for (int k = 0; k < 8; k++) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
    crc_table[n] = c;
  }
  crc_ready = true;
}
uint32_t crc32(const uint8_t *p, size_t n, uint32_t c = 0xffffffffu) {
  if (!crc_ready) crc_init();
  for (size_t i = 0; i < n; i++) c = crc_table[(c ^ p[i]) & 0xff] ^ (c >> 8);
  return c;
}
void be32(std::vector<uint8_t> &v, uint32_t x) {
  v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x);
}
void chunk(std::vector<uint8_t> &out, const char type[4], const std::vector<uint8_t> &data) {
  be32(out, (uint32_t)data.size());
  std::vector<uint8_t> td(type, type + 4);
  td.insert(td.end(), data.begin(), data.end());
  out.insert(out.end(), td.begin(), td.end());
  be32(out, crc32(td.data(), td.size()) ^ 0xffffffffu);
}

bool write_png(const char *name, const float *rgb, int w, int h) {
  std::vector<uint8_t> raw;  // filter byte 0 + RGB8 per row
  raw.reserve((size_t)h * (3 * (size_t)w + 1));
  for (int y = 0; y < h; y++) {
    raw.push_back(0);
    for (int x = 0; x < 3 * w; x++) raw.push_back(pbrt_hip::to_byte(rgb[(size_t)y * 3 * w + x]));
  }
  std::vector<uint8_t> z = {0x78, 0x01};
  uint32_t a = 1, b = 0;
  for (uint8_t v : raw) { a = (a + v) % 65521u; b = (b + a) % 65521u; }
  size_t pos = 0;
  do {
    size_t n = raw.size() - pos < 65535 ? raw.size() - pos : 65535;
    z.push_back(pos + n == raw.size() ? 1 : 0);
    z.push_back(n & 0xff); z.push_back(n >> 8);
    z.push_back(~n & 0xff); z.push_back((~n >> 8) & 0xff);
    z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
    pos += n;
  } while (pos < raw.size());
  be32(z, (b << 16) | a);
  std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  std::vector<uint8_t> ihdr;
  be32(ihdr, (uint32_t)w); be32(ihdr, (uint32_t)h);
  ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
  chunk(out, "IHDR", ihdr);
  chunk(out, "IDAT", z);
  chunk(out, "IEND", {});
  FILE *f = std::fopen(name, "wb");
  if (!f) return false;
  bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
  return std::fclose(f) == 0 && ok;
}

bool write_pfm(const char *name, const float *rgb, int w, int h) {
  FILE *f = std::fopen(name, "wb");
  if (!f) return false;
  const uint16_t probe = 0x1234;
  const bool little = *(const uint8_t *)&probe == 0x34;
  std::fprintf(f, "PF\n%d %d\n%d\n", w, h, little ? -1 : 1);
  bool ok = true;
  for (int y = h - 1; y >= 0 && ok; y--)
    ok = std::fwrite(rgb + (size_t)y * 3 * w, 4, 3 * (size_t)w, f) == 3 * (size_t)w;
  return std::fclose(f) == 0 && ok;
}

}  // namespace

extern "C" int pbrt_hip_write_image(const char *name, const float *rgb, int32_t width, int32_t height) {
  if (!name || !rgb || width <= 0 || height <= 0) return PBRT_HIP_ERR_INVALID;
  std::string n(name);
  size_t dot = n.rfind('.');
  std::string ext = dot == std::string::npos ? "" : n.substr(dot + 1);
  for (auto &c : ext) c = (char)std::tolower((unsigned char)c);
  if (ext == "png") return write_png(name, rgb, width, height) ? PBRT_HIP_OK : PBRT_HIP_ERR_INTERNAL;
  if (ext == "pfm") return write_pfm(name, rgb, width, height) ? PBRT_HIP_OK : PBRT_HIP_ERR_INTERNAL;
  return PBRT_HIP_ERR_INVALID;  // imageio.rs:272-280: exr / tga unimplemented, unknown extension
}
