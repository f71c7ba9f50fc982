// imageio.cpp -- imageio::write_image of the reference (core/imageio.rs:235-283) for the two
// formats it implements: 8-bit RGB PNG through to_byte (imageio.rs:66-68, :245-271) and PFM
// (imageio.rs:186-213: "PF\n{w} {h}\n{scale}\n", rows bottom to top, host-endian floats, scale -1 on
// little-endian hosts).  The PNG stream uses stored (uncompressed) deflate blocks: the reference
// delegates compression to the `png` crate, which carries no arithmetic of the path, and any
// conforming decoder returns the same bytes.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
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

// ---- readers: imageio::read_image (imageio.rs:87-184) ----

// PFM, imageio.rs:87-140: header words "PF"|"Pf", width, height, scale separated by ' ', '\n' or '\t';
// scale < 0 = little-endian floats, |scale| multiplies; rows bottom to top; 1-channel images are
// replicated to RGB (RGBSpectrum::new(f), imageio.rs:129-133).
constexpr uint64_t kMaxImagePixels = 1ull << 28;  // 16384 x 16384: 3 GiB of float RGB; anything larger is refused
bool read_word(FILE *f, std::string *w) {
  w->clear();
  for (;;) {
    int c = std::fgetc(f);
    if (c == EOF) return false;
    if (c == ' ' || c == '\n' || c == '\t') return true;
    if (w->size() >= 64) return false;  // no header word is that long
    w->push_back((char)c);
  }
}
bool read_pfm(const char *name, std::vector<float> *rgb, int *w, int *h) {
  FILE *f = std::fopen(name, "rb");
  if (!f) return false;
  std::string hdr, sw, sh, ss;
  bool ok = read_word(f, &hdr) && read_word(f, &sw) && read_word(f, &sh) && read_word(f, &ss);
  const int nc = hdr == "PF" ? 3 : (hdr == "Pf" ? 1 : 0);
  if (!ok || nc == 0) { std::fclose(f); return false; }
  *w = std::atoi(sw.c_str());
  *h = std::atoi(sh.c_str());
  const float scale = (float)std::atof(ss.c_str());
  if (*w <= 0 || *h <= 0 || scale == 0.f) { std::fclose(f); return false; }
  // the header is untrusted: the pixel data it announces must actually be in the file before anything is allocated
  if ((uint64_t)*w * (uint64_t)*h > kMaxImagePixels) { std::fclose(f); return false; }
  {
    const long at = std::ftell(f);
    if (at < 0 || std::fseek(f, 0, SEEK_END) != 0) { std::fclose(f); return false; }
    const long end = std::ftell(f);
    if (end < at || (uint64_t)(end - at) < (uint64_t)*w * (uint64_t)*h * (uint64_t)nc * 4u || std::fseek(f, at, SEEK_SET) != 0) {
      std::fclose(f);
      return false;
    }
  }
  const bool file_le = scale < 0.f;
  const float mag = scale < 0.f ? -scale : scale;
  const uint16_t probe = 0x1234;
  const bool host_le = *(const uint8_t *)&probe == 0x34;
  rgb->assign((size_t)*w * *h * 3, 0.f);
  std::vector<uint8_t> row((size_t)*w * nc * 4);
  for (int y = *h - 1; y >= 0 && ok; y--) {
    ok = std::fread(row.data(), 1, row.size(), f) == row.size();
    for (int x = 0; x < *w && ok; x++)
      for (int c = 0; c < 3; c++) {
        uint8_t b[4];
        std::memcpy(b, &row[((size_t)x * nc + (nc == 3 ? c : 0)) * 4], 4);
        if (file_le != host_le) { std::swap(b[0], b[3]); std::swap(b[1], b[2]); }
        float v;
        std::memcpy(&v, b, 4);
        (*rgb)[((size_t)y * *w + x) * 3 + c] = v * mag;
      }
  }
  std::fclose(f);
  return ok;
}

// A small inflate (RFC 1951): stored, fixed and dynamic Huffman blocks; canonical codes decoded
// bit by bit from per-length counts.  The reference leaves this to the `png` crate.
struct Inflater {
  const uint8_t *in;
  size_t n, pos = 0;
  uint32_t bitbuf = 0;
  int bitcnt = 0;
  std::vector<uint8_t> out;
  size_t out_cap = ~(size_t)0;  // the caller knows how many bytes the image needs: more is a decompression bomb
  bool bad = false;
  int bits(int need) {
    uint32_t v = bitbuf;
    while (bitcnt < need) {
      if (pos >= n) { bad = true; return 0; }
      v |= (uint32_t)in[pos++] << bitcnt;
      bitcnt += 8;
    }
    bitbuf = v >> need;
    bitcnt -= need;
    return (int)(v & ((1u << need) - 1));
  }
  struct Huff { uint16_t count[16]; uint16_t symbol[288]; };
  static void build(Huff &h, const uint8_t *len, int n) {
    for (int i = 0; i < 16; i++) h.count[i] = 0;
    for (int i = 0; i < n; i++) h.count[len[i]]++;
    uint16_t offs[16];
    offs[1] = 0;
    for (int i = 1; i < 15; i++) offs[i + 1] = offs[i] + h.count[i];
    for (int i = 0; i < n; i++)
      if (len[i]) h.symbol[offs[len[i]]++] = (uint16_t)i;
  }
  int decode(const Huff &h) {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; len++) {
      code |= bits(1);
      if (bad) return -1;
      const int count = h.count[len];
      if (code - count < first) return h.symbol[index + (code - first)];
      index += count;
      first += count;
      first <<= 1;
      code <<= 1;
    }
    bad = true;
    return -1;
  }
  bool codes(const Huff &lc, const Huff &dc) {
    static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint16_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint16_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    for (;;) {
      int sym = decode(lc);
      if (bad) return false;
      if (sym < 256) { if (out.size() >= out_cap) return false; out.push_back((uint8_t)sym); }
      else if (sym == 256) return true;
      else {
        sym -= 257;
        if (sym >= 29) return false;
        const int len = lbase[sym] + bits(lext[sym]);
        const int ds = decode(dc);
        if (bad || ds < 0 || ds >= 30) return false;
        const size_t dist = dbase[ds] + (size_t)bits(dext[ds]);
        if (bad || dist > out.size() || out.size() + (size_t)len > out_cap) return false;
        for (int i = 0; i < len; i++) out.push_back(out[out.size() - dist]);
      }
    }
  }
  bool run() {
    int last;
    do {
      last = bits(1);
      const int type = bits(2);
      if (bad) return false;
      if (type == 0) {
        bitbuf = 0; bitcnt = 0;
        if (pos + 4 > n) return false;
        const unsigned len = in[pos] | (in[pos + 1] << 8), nlen = in[pos + 2] | (in[pos + 3] << 8);
        pos += 4;
        if ((len ^ 0xffffu) != nlen || pos + len > n || out.size() + len > out_cap) return false;
        out.insert(out.end(), in + pos, in + pos + len);
        pos += len;
      } else if (type == 1) {
        uint8_t l[288];
        for (int i = 0; i < 144; i++) l[i] = 8;
        for (int i = 144; i < 256; i++) l[i] = 9;
        for (int i = 256; i < 280; i++) l[i] = 7;
        for (int i = 280; i < 288; i++) l[i] = 8;
        Huff lc, dc;
        build(lc, l, 288);
        uint8_t d[30];
        for (int i = 0; i < 30; i++) d[i] = 5;
        build(dc, d, 30);
        if (!codes(lc, dc)) return false;
      } else if (type == 2) {
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        const int nlen = bits(5) + 257, ndist = bits(5) + 1, ncode = bits(4) + 4;
        if (bad || nlen > 286 || ndist > 30) return false;
        uint8_t l[320] = {0};
        for (int i = 0; i < ncode; i++) l[order[i]] = (uint8_t)bits(3);
        Huff cl;
        build(cl, l, 19);
        uint8_t lens[320] = {0};
        int idx = 0;
        while (idx < nlen + ndist) {
          int sym = decode(cl);
          if (bad) return false;
          if (sym < 16) lens[idx++] = (uint8_t)sym;
          else {
            int prev = 0, rep;
            if (sym == 16) { if (idx == 0) return false; prev = lens[idx - 1]; rep = 3 + bits(2); }
            else if (sym == 17) rep = 3 + bits(3);
            else rep = 11 + bits(7);
            if (idx + rep > nlen + ndist) return false;
            while (rep--) lens[idx++] = (uint8_t)prev;
          }
        }
        Huff lc, dc;
        build(lc, lens, nlen);
        build(dc, lens + nlen, ndist);
        if (!codes(lc, dc)) return false;
      } else {
        return false;
      }
    } while (!last);
    return !bad;
  }
};

// PNG, imageio.rs:142-178: 8-bit images only (RGB as the reference assumes; grey, grey+alpha and RGBA are
// accepted too, alpha dropped); value = byte / 255.
bool read_png(const char *name, std::vector<float> *rgb, int *w, int *h) {
  FILE *f = std::fopen(name, "rb");
  if (!f) return false;
  std::vector<uint8_t> file;
  uint8_t buf[65536];
  size_t n;
  while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + n);
  std::fclose(f);
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) return false;
  auto be = [&](size_t p) { return ((uint32_t)file[p] << 24) | ((uint32_t)file[p + 1] << 16) | ((uint32_t)file[p + 2] << 8) | file[p + 3]; };
  std::vector<uint8_t> z;
  int depth = 0, ctype = -1, interlace = 0;
  for (size_t p = 8; p + 12 <= file.size();) {
    const uint32_t len = be(p);
    if (p + 12 + (size_t)len > file.size()) return false;
    const char *type = (const char *)&file[p + 4];
    if (!std::memcmp(type, "IHDR", 4) && len >= 13) {
      *w = (int)be(p + 8); *h = (int)be(p + 12);
      depth = file[p + 16]; ctype = file[p + 17]; interlace = file[p + 20];
    } else if (!std::memcmp(type, "IDAT", 4)) {
      z.insert(z.end(), file.begin() + p + 8, file.begin() + p + 8 + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      break;
    }
    p += 12 + (size_t)len;
  }
  const int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
  if (depth != 8 || ch == 0 || interlace != 0 || *w <= 0 || *h <= 0 || z.size() < 6) return false;
  if ((uint64_t)*w * (uint64_t)*h > kMaxImagePixels) return false;
  const size_t stride = (size_t)*w * ch;
  Inflater inf;
  inf.in = z.data() + 2;  // zlib header
  inf.n = z.size() - 2;
  inf.out_cap = (stride + 1) * (size_t)*h;
  if (!inf.run()) return false;
  if (inf.out.size() < (stride + 1) * (size_t)*h) return false;
  std::vector<uint8_t> img(stride * (size_t)*h);
  for (int y = 0; y < *h; y++) {
    const uint8_t *src = &inf.out[(stride + 1) * (size_t)y];
    const int ft = src[0];
    uint8_t *dst = &img[stride * (size_t)y];
    const uint8_t *up = y ? dst - stride : nullptr;
    for (size_t i = 0; i < stride; i++) {
      const int a = i >= (size_t)ch ? dst[i - ch] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)ch) ? up[i - ch] : 0;
      int pred = 0;
      if (ft == 1) pred = a;
      else if (ft == 2) pred = b;
      else if (ft == 3) pred = (a + b) >> 1;
      else if (ft == 4) {
        const int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c);
        pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
      } else if (ft != 0) return false;
      dst[i] = (uint8_t)(src[1 + i] + pred);
    }
  }
  rgb->resize((size_t)*w * *h * 3);
  for (size_t px = 0; px < (size_t)*w * *h; px++)
    for (int c = 0; c < 3; c++) (*rgb)[3 * px + c] = (float)img[px * ch + (ch >= 3 ? c : 0)] / 255.f;
  return true;
}

}  // namespace

// imageio::read_image: two calls, the first with rgb == NULL returns the size
extern "C" int pbrt_hip_read_image(const char *name, float *rgb, int32_t *width, int32_t *height) {
  if (!name || !width || !height) return PBRT_HIP_ERR_INVALID;
  try {
  std::string n(name);
  size_t dot = n.rfind('.');
  std::string ext = dot == std::string::npos ? "" : n.substr(dot + 1);
  for (auto &c : ext) c = (char)std::tolower((unsigned char)c);
  std::vector<float> px;
  int w = 0, h = 0;
  bool ok;
  if (ext == "png") ok = read_png(name, &px, &w, &h);
  else if (ext == "pfm") ok = read_pfm(name, &px, &w, &h);
  else return PBRT_HIP_ERR_INVALID;  // imageio.rs:179-182: exr / tga not implemented, unknown extension
  if (!ok) return PBRT_HIP_ERR_INTERNAL;
  if (rgb) {
    if (*width != w || *height != h) return PBRT_HIP_ERR_INVALID;
    std::memcpy(rgb, px.data(), px.size() * 4);
  }
  *width = w;
  *height = h;
  return PBRT_HIP_OK;
  } catch (const std::exception &) {  // std::bad_alloc and friends never cross the C ABI
    return PBRT_HIP_ERR_INTERNAL;
  }
}

extern "C" int pbrt_hip_write_image(const char *name, const float *rgb, int32_t width, int32_t height) {
  if (!name || !rgb || width <= 0 || height <= 0) return PBRT_HIP_ERR_INVALID;
  if ((uint64_t)width * (uint64_t)height > kMaxImagePixels) return PBRT_HIP_ERR_LIMIT;
  try {
  std::string n(name);
  size_t dot = n.rfind('.');
  std::string ext = dot == std::string::npos ? "" : n.substr(dot + 1);
  for (auto &c : ext) c = (char)std::tolower((unsigned char)c);
  if (ext == "png") return write_png(name, rgb, width, height) ? PBRT_HIP_OK : PBRT_HIP_ERR_INTERNAL;
  if (ext == "pfm") return write_pfm(name, rgb, width, height) ? PBRT_HIP_OK : PBRT_HIP_ERR_INTERNAL;
  return PBRT_HIP_ERR_INVALID;  // imageio.rs:272-280: exr / tga unimplemented, unknown extension
  } catch (const std::exception &) {
    return PBRT_HIP_ERR_INTERNAL;
  }
}
