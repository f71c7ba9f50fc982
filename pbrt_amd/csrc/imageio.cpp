// imageio.cpp -- imageio::write_image of the reference (core/imageio.rs:235-283) for the two
// formats it implements: 8-bit RGB PNG through to_byte (imageio.rs:66-68, :245-271) and PFM
// (imageio.rs:186-213: "PF\n{w} {h}\n{scale}\n", rows bottom to top, host-endian floats, scale -1 on
// little-endian hosts).  The PNG stream is compressed here (round 5; stored blocks until then): adaptive row filters (the
// minimum-sum-of-absolute-differences heuristic over None / Sub / Up / Paeth) and deflate with LZ77 matches (32 KB window, hash
// chains, lazy matching) in blocks coded the cheapest of stored / fixed Huffman / dynamic Huffman (RFC 1951 3.2.4-3.2.7).  The reference delegates compression to the `png` crate, which carries no
// arithmetic of the path: any conforming decoder returns the same bytes, and the tests decode with this file's own inflate AND with an
// independent one (PIL).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>
#include <utility>
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

// ---- deflate (RFC 1951): LZ77 over hash chains with one-step lazy matching; per block the cheapest of stored, fixed-Huffman and
// dynamic-Huffman coding ----
struct BitWriter {
  std::vector<uint8_t> &out;
  uint64_t acc = 0;
  int n = 0;
  explicit BitWriter(std::vector<uint8_t> &o) : out(o) {}
  void bits(uint32_t v, int k) {  // k <= 16 bits, least significant first
    acc |= (uint64_t)v << n;
    n += k;
    while (n >= 8) { out.push_back((uint8_t)acc); acc >>= 8; n -= 8; }
  }
  void code(uint32_t c, int k) {  // a Huffman code: most significant bit first
    uint32_t r = 0;
    for (int i = 0; i < k; i++) r |= ((c >> i) & 1u) << (k - 1 - i);
    bits(r, k);
  }
  void flush() { if (n > 0) { out.push_back((uint8_t)acc); acc = 0; n = 0; } }
};
const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
inline int len_symbol(uint32_t len) { int i = 28; while (kLenBase[i] > len) i--; return i; }
inline int dist_symbol(uint32_t dist) { int i = 29; while (kDistBase[i] > dist) i--; return i; }

// Code lengths (<= limit) of a prefix code for `freq`: Huffman's algorithm, then the over-long codes folded back by moving leaves until
// the Kraft sum is 1 again, the longest lengths going to the rarest symbols.  Fewer than two used symbols get two codes of one bit, so
// that the code is complete for every decoder.
void huff_lengths(const uint32_t *freq, int n, int limit, uint8_t *len) {
  std::vector<int> sym;
  for (int i = 0; i < n; i++) { len[i] = 0; if (freq[i]) sym.push_back(i); }
  if (sym.size() < 2) {
    const int only = sym.empty() ? 0 : sym[0];
    len[only] = 1;
    len[only == 0 ? 1 : 0] = 1;
    return;
  }
  std::stable_sort(sym.begin(), sym.end(), [&](int x, int y) { return freq[x] < freq[y]; });
  const int m = (int)sym.size();
  std::vector<uint64_t> w(2 * m - 1);
  std::vector<int> parent(2 * m - 1, -1);
  for (int i = 0; i < m; i++) w[i] = freq[sym[i]];
  int leaf = 0, inner = m, made = m;  // two queues: the sorted leaves and the interior nodes in the order they were made
  auto take = [&]() { return (leaf < m && (inner >= made || w[leaf] <= w[inner])) ? leaf++ : inner++; };
  while (made < 2 * m - 1) {
    const int x = take(), y = take();
    w[made] = w[x] + w[y];
    parent[x] = parent[y] = made;
    made++;
  }
  std::vector<int> depth(2 * m - 1, 0), count(64, 0);
  for (int i = 2 * m - 3; i >= 0; i--) depth[i] = depth[parent[i]] + 1;
  for (int i = 0; i < m; i++) count[depth[i] > limit ? limit : depth[i]]++;
  uint64_t kraft = 0;
  for (int l = 1; l <= limit; l++) kraft += (uint64_t)count[l] << (limit - l);
  while (kraft > (1ull << limit)) {  // one code of the greatest length pairs up with a shorter one pushed down a level
    count[limit]--;
    for (int l = limit - 1; l >= 1; l--)
      if (count[l]) { count[l]--; count[l + 1] += 2; break; }
    kraft--;
  }
  int at = 0;
  for (int l = limit; l >= 1; l--)
    for (int k = 0; k < count[l]; k++) len[sym[at++]] = (uint8_t)l;
}
void canonical_codes(const uint8_t *len, int n, uint16_t *code) {  // RFC 1951 3.2.2
  int count[16] = {0}, next[16] = {0};
  for (int i = 0; i < n; i++) count[len[i]]++;
  count[0] = 0;
  for (int l = 1, c = 0; l < 16; l++) { c = (c + count[l - 1]) << 1; next[l] = c; }
  for (int i = 0; i < n; i++) code[i] = len[i] ? (uint16_t)next[len[i]]++ : 0;
}

struct Token { uint16_t len_or_lit, dist; };  // dist == 0: a literal

void write_tokens(BitWriter &bw, const Token *t, size_t n, const uint8_t *ll_len, const uint16_t *ll_code, const uint8_t *d_len, const uint16_t *d_code) {
  for (size_t i = 0; i < n; i++) {
    if (t[i].dist == 0) { bw.code(ll_code[t[i].len_or_lit], ll_len[t[i].len_or_lit]); continue; }
    const int li = len_symbol(t[i].len_or_lit), di = dist_symbol(t[i].dist);
    bw.code(ll_code[257 + li], ll_len[257 + li]);
    if (kLenExtra[li]) bw.bits(t[i].len_or_lit - kLenBase[li], kLenExtra[li]);
    bw.code(d_code[di], d_len[di]);
    if (kDistExtra[di]) bw.bits(t[i].dist - kDistBase[di], kDistExtra[di]);
  }
  bw.code(ll_code[256], ll_len[256]);
}

// one deflate block for tokens [t, t + n) that cover raw[from, to)
void write_block(BitWriter &bw, const Token *t, size_t n, const uint8_t *raw, size_t from, size_t to, bool final_block) {
  uint32_t ll_freq[288] = {0}, d_freq[30] = {0};
  uint64_t extra_bits = 0;
  for (size_t i = 0; i < n; i++) {
    if (t[i].dist == 0) { ll_freq[t[i].len_or_lit]++; continue; }
    const int li = len_symbol(t[i].len_or_lit), di = dist_symbol(t[i].dist);
    ll_freq[257 + li]++; d_freq[di]++;
    extra_bits += kLenExtra[li] + kDistExtra[di];
  }
  ll_freq[256] = 1;
  uint8_t fix_ll[288], fix_d[30], dyn_ll[288], dyn_d[30];
  for (int i = 0; i < 288; i++) fix_ll[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
  for (int i = 0; i < 30; i++) fix_d[i] = 5;
  huff_lengths(ll_freq, 286, 15, dyn_ll);
  huff_lengths(d_freq, 30, 15, dyn_d);
  int hlit = 286, hdist = 30;
  while (hlit > 257 && dyn_ll[hlit - 1] == 0) hlit--;
  while (hdist > 1 && dyn_d[hdist - 1] == 0) hdist--;
  // the code lengths, run-length coded with 16 (repeat previous 3-6), 17 (zeros 3-10), 18 (zeros 11-138)   (RFC 1951 3.2.7)
  std::vector<uint8_t> all(dyn_ll, dyn_ll + hlit);
  all.insert(all.end(), dyn_d, dyn_d + hdist);
  std::vector<std::pair<uint8_t, uint8_t>> rle;  // {symbol, extra value}
  for (size_t i = 0; i < all.size();) {
    size_t run = 1;
    while (i + run < all.size() && all[i + run] == all[i]) run++;
    if (all[i] == 0 && run >= 3) {
      const size_t r = run > 138 ? 138 : run;
      if (r >= 11) rle.push_back({18, (uint8_t)(r - 11)}); else rle.push_back({17, (uint8_t)(r - 3)});
      i += r;
    } else if (run >= 4) {
      rle.push_back({all[i], 0});
      size_t r = run - 1 > 6 ? 6 : run - 1;
      rle.push_back({16, (uint8_t)(r - 3)});
      i += 1 + r;
    } else {
      rle.push_back({all[i], 0});
      i++;
    }
  }
  uint32_t cl_freq[19] = {0};
  for (auto &e : rle) cl_freq[e.first]++;
  uint8_t cl_len[19];
  uint16_t cl_code[19];
  huff_lengths(cl_freq, 19, 7, cl_len);
  canonical_codes(cl_len, 19, cl_code);
  static const uint8_t kOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  int hclen = 19;
  while (hclen > 4 && cl_len[kOrder[hclen - 1]] == 0) hclen--;
  uint64_t head_bits = 5 + 5 + 4 + 3 * (uint64_t)hclen;
  for (auto &e : rle) head_bits += cl_len[e.first] + (e.first == 16 ? 2 : e.first == 17 ? 3 : e.first == 18 ? 7 : 0);
  uint64_t dyn_bits = head_bits + extra_bits, fix_bits = extra_bits;
  for (int i = 0; i < 286; i++) { dyn_bits += (uint64_t)ll_freq[i] * dyn_ll[i]; fix_bits += (uint64_t)ll_freq[i] * fix_ll[i]; }
  for (int i = 0; i < 30; i++) { dyn_bits += (uint64_t)d_freq[i] * dyn_d[i]; fix_bits += (uint64_t)d_freq[i] * fix_d[i]; }
  const uint64_t n_stored = (to - from + 65534) / 65535 + (to == from ? 1 : 0);
  const uint64_t stored_bits = 8 * (uint64_t)(to - from) + n_stored * 40;  // 3 header bits, up to 7 of padding, LEN and NLEN
  if (stored_bits < dyn_bits && stored_bits < fix_bits) {
    size_t at = from;
    do {
      const size_t k = to - at < 65535 ? to - at : 65535;
      bw.bits(final_block && at + k == to ? 1 : 0, 1);
      bw.bits(0, 2);
      bw.flush();
      bw.bits((uint32_t)k, 16);
      bw.bits((uint32_t)k ^ 0xffffu, 16);
      bw.out.insert(bw.out.end(), raw + at, raw + at + k);
      at += k;
    } while (at < to);
    return;
  }
  bw.bits(final_block ? 1 : 0, 1);
  if (fix_bits <= dyn_bits) {
    uint16_t ll_code[288], d_code[30];
    canonical_codes(fix_ll, 288, ll_code);
    canonical_codes(fix_d, 30, d_code);
    bw.bits(1, 2);
    write_tokens(bw, t, n, fix_ll, ll_code, fix_d, d_code);
    return;
  }
  uint16_t ll_code[288], d_code[30];
  canonical_codes(dyn_ll, 286, ll_code);
  canonical_codes(dyn_d, 30, d_code);
  bw.bits(2, 2);
  bw.bits((uint32_t)(hlit - 257), 5);
  bw.bits((uint32_t)(hdist - 1), 5);
  bw.bits((uint32_t)(hclen - 4), 4);
  for (int i = 0; i < hclen; i++) bw.bits(cl_len[kOrder[i]], 3);
  for (auto &e : rle) {
    bw.code(cl_code[e.first], cl_len[e.first]);
    if (e.first == 16) bw.bits(e.second, 2);
    else if (e.first == 17) bw.bits(e.second, 3);
    else if (e.first == 18) bw.bits(e.second, 7);
  }
  write_tokens(bw, t, n, dyn_ll, ll_code, dyn_d, d_code);
}

// zlib stream (RFC 1950) of `raw`
void zlib_deflate(const std::vector<uint8_t> &raw, std::vector<uint8_t> *z) {
  z->clear();
  z->push_back(0x78); z->push_back(0x9c);
  BitWriter bw(*z);
  const size_t n = raw.size();
  constexpr uint32_t kHashBits = 15, kWindow = 32768, kMaxChain = 64;
  constexpr uint64_t kNil = ~0ull;
  constexpr size_t kBlockTokens = 1u << 16;
  // hash chains over the last kWindow positions only (a ring: 32 K entries whatever the image's size; positions are 64-bit, a film of
  // 32 768^2 pixels is 3.2 GB of scanlines)
  std::vector<uint64_t> head(1u << kHashBits, kNil), prev(kWindow, kNil);
  auto hash3 = [&](size_t i) { return (((uint32_t)raw[i] | (uint32_t)raw[i + 1] << 8 | (uint32_t)raw[i + 2] << 16) * 0x9e3779b1u) >> (32 - kHashBits); };
  auto enter = [&](size_t i) { if (i + 3 <= n) { const uint32_t hsh = hash3(i); prev[i & (kWindow - 1)] = head[hsh]; head[hsh] = i; } };
  auto longest = [&](size_t i, uint32_t *dist) -> uint32_t {  // the longest earlier occurrence of raw[i ...] inside the window (0 if under 3 bytes)
    if (i + 3 > n) return 0;
    const size_t max_len = n - i < 258 ? n - i : 258;
    uint32_t best = 0, chain = 0;
    uint64_t cand = head[hash3(i)];
    // (a position's ring slot is its own while it is less than kWindow back: older candidates end the chain)
    while (cand != kNil && i - cand < kWindow && chain++ < kMaxChain) {
      if (raw[cand + best] == raw[i + best]) {
        size_t l = 0;
        while (l < max_len && raw[cand + l] == raw[i + l]) l++;
        if (l > best) { best = (uint32_t)l; *dist = (uint32_t)(i - cand); if (l == max_len) break; }
      }
      const uint64_t older = prev[cand & (kWindow - 1)];
      if (older != kNil && older >= cand) break;  // (the slot was taken over by a later position: not this chain any more)
      cand = older;
    }
    return best >= 3 ? best : 0;
  };
  std::vector<Token> tok;
  tok.reserve(kBlockTokens);
  size_t i = 0, block_from = 0;
  while (i < n) {
    uint32_t dist = 0, len = longest(i, &dist);
    enter(i);
    if (len && len < 32 && i + 1 < n) {  // lazy: a longer match one byte on wins, this byte goes out as a literal
      uint32_t dist2 = 0;
      if (longest(i + 1, &dist2) > len) len = 0;
    }
    if (len) {
      tok.push_back({(uint16_t)len, (uint16_t)dist});  // 32768 fits 16 bits
      for (size_t k = 1; k < len; k++) enter(i + k);
      i += len;
    } else {
      tok.push_back({raw[i], 0});
      i++;
    }
    if (tok.size() == kBlockTokens && i < n) {
      write_block(bw, tok.data(), tok.size(), raw.data(), block_from, i, false);
      tok.clear();
      block_from = i;
    }
  }
  write_block(bw, tok.data(), tok.size(), raw.data(), block_from, n, true);
  bw.flush();
  uint32_t a = 1, b = 0;
  for (uint8_t v : raw) { a = (a + v) % 65521u; b = (b + a) % 65521u; }
  be32(*z, (b << 16) | a);
}

bool write_png(const char *name, const float *rgb, int w, int h) {
  // scanlines with the row filter that leaves the smallest sum of absolute (signed-byte) residuals: None, Sub, Up or Paeth (PNG 9.2, 12.8)
  const size_t stride = 3 * (size_t)w;
  std::vector<uint8_t> raw, cur(stride), up(stride, 0), cand[4];
  for (auto &c : cand) c.resize(stride);
  raw.reserve((size_t)h * (stride + 1));
  static const uint8_t kType[4] = {0, 1, 2, 4};
  for (int y = 0; y < h; y++) {
    for (size_t x = 0; x < stride; x++) cur[x] = pbrt_hip::to_byte(rgb[(size_t)y * stride + x]);
    uint64_t cost[4] = {0, 0, 0, 0};
    for (size_t x = 0; x < stride; x++) {
      const int a = x >= 3 ? cur[x - 3] : 0, b = up[x], c = x >= 3 ? up[x - 3] : 0;
      const int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c);
      const int paeth = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
      const uint8_t v[4] = {cur[x], (uint8_t)(cur[x] - a), (uint8_t)(cur[x] - b), (uint8_t)(cur[x] - paeth)};
      for (int k = 0; k < 4; k++) { cand[k][x] = v[k]; cost[k] += (uint64_t)std::abs((int)(int8_t)v[k]); }
    }
    int best = 0;
    for (int k = 1; k < 4; k++) if (cost[k] < cost[best]) best = k;
    raw.push_back(kType[best]);
    raw.insert(raw.end(), cand[best].begin(), cand[best].end());
    up.swap(cur);
  }
  std::vector<uint8_t> z;
  zlib_deflate(raw, &z);
  std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  std::vector<uint8_t> ihdr;
  be32(ihdr, (uint32_t)w); be32(ihdr, (uint32_t)h);
  ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
  chunk(out, "IHDR", ihdr);
  // (a chunk's length field is 31 bits: a film whose compressed scanlines pass 1 GiB goes out as several IDAT chunks, which a decoder
  // reads as one stream)
  for (size_t at = 0; at < z.size() || at == 0; at += (size_t)1 << 30) {
    const size_t k = std::min(z.size() - at, (size_t)1 << 30);
    chunk(out, "IDAT", std::vector<uint8_t>(z.begin() + (ptrdiff_t)at, z.begin() + (ptrdiff_t)(at + k)));
    if (z.empty()) break;
  }
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
constexpr uint64_t kMaxImagePixels = 1ull << 28;  // READING (untrusted headers): 16384 x 16384, 3 GiB of float RGB; anything larger is refused
constexpr uint64_t kMaxWritePixels = 1ull << 30;  // WRITING (the caller's own film): 32768 x 32768, the largest film the tests render
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

// The message behind pbrt_hip_last_error(): capi.cpp's when this file is part of the library; the sanitizer harnesses link this file alone.
namespace pbrt_hip { int fail(int code, const std::string &msg) __attribute__((weak)); }
static int say(int code, const std::string &msg) { return pbrt_hip::fail ? pbrt_hip::fail(code, msg) : code; }

// imageio::read_image: two calls, the first with rgb == NULL returns the size
extern "C" int pbrt_hip_read_image(const char *name, float *rgb, int32_t *width, int32_t *height) {
  if (!name || !width || !height) return say(PBRT_HIP_ERR_INVALID, "read_image: null argument");
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
  else if (ext == "exr" || ext == "tga") return say(PBRT_HIP_ERR_INVALID, "read_image: reading ." + ext + " files is not implemented");  // imageio.rs:179-180
  else return say(PBRT_HIP_ERR_INVALID, "read_image: unknown file extension " + ext);  // imageio.rs:182
  if (!ok) return say(PBRT_HIP_ERR_INTERNAL, std::string("read_image: cannot read '") + name + "' (missing, truncated or not a valid ." + ext + " file)");
  if (rgb) {
    if (*width != w || *height != h) return say(PBRT_HIP_ERR_INVALID, "read_image: the size passed in is not the file's");
    std::memcpy(rgb, px.data(), px.size() * 4);
  }
  *width = w;
  *height = h;
  return PBRT_HIP_OK;
  } catch (const std::exception &e) {  // std::bad_alloc and friends never cross the C ABI
    return say(PBRT_HIP_ERR_INTERNAL, std::string("read_image: ") + e.what());
  }
}

extern "C" int pbrt_hip_write_image(const char *name, const float *rgb, int32_t width, int32_t height) {
  if (!name || !rgb || width <= 0 || height <= 0) return say(PBRT_HIP_ERR_INVALID, "write_image: null argument or empty image");
  if ((uint64_t)width * (uint64_t)height > kMaxWritePixels) return say(PBRT_HIP_ERR_LIMIT, "write_image: more than 2^30 pixels");
  try {
  std::string n(name);
  size_t dot = n.rfind('.');
  std::string ext = dot == std::string::npos ? "" : n.substr(dot + 1);
  for (auto &c : ext) c = (char)std::tolower((unsigned char)c);
  if (ext == "png") return write_png(name, rgb, width, height) ? PBRT_HIP_OK : say(PBRT_HIP_ERR_INTERNAL, std::string("Failed to create file '") + name + "'");  // imageio.rs:248
  if (ext == "pfm") return write_pfm(name, rgb, width, height) ? PBRT_HIP_OK : say(PBRT_HIP_ERR_INTERNAL, std::string("Failed to write PFM to '") + name + "'");  // imageio.rs:278
  if (ext == "exr" || ext == "tga") return say(PBRT_HIP_ERR_INVALID, "writing ." + ext + " files is not implemented");  // imageio.rs:272-273
  return say(PBRT_HIP_ERR_INVALID, "unknown file extension " + ext);  // imageio.rs:281
  } catch (const std::exception &e) {
    return say(PBRT_HIP_ERR_INTERNAL, std::string("write_image: ") + e.what());
  }
}
