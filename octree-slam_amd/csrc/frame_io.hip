// frame_io.hip -- recorded-sensor input.
//
// Replaces sensor::OpenNIDevice (src/sensor/openni_device.cpp:13-150) as the producer of RawFrame
// (common_types.h:65-73): depth as uint16 millimetres, colour as RGB888, both row-major on the
// device, a monotonically increasing timestamp, focal lengths from the field of view
// (openni_device.cpp:64-65: f = size / (2 tan(fov / 2))).  Frames come from a TUM-RGB-D style
// association list: each line names a depth image and a colour image with their timestamps,
//     <t_a> <file_a> <t_b> <file_b>        ('#' starts a comment; paths relative to the list)
// in either order (the 16-bit single-channel image is the depth).  Images: PNG (8-bit RGB /
// 16-bit grey, non-interlaced -- what the TUM sets contain; own inflate, no zlib/libpng
// dependency), binary PGM (P5, maxval 65535, big-endian) and PPM (P6, maxval 255).
// Host code only; the single device operation is the upload in frame_reader_next.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "common.hpp"
#include "frame_io.hpp"

namespace svoslam {

// ---------------------------------------------------------------------------------------------
// inflate (RFC 1951) + zlib wrapper (RFC 1950)
// ---------------------------------------------------------------------------------------------
namespace {

struct BitReader {
  const uint8_t *p, *end;
  uint32_t buf = 0;
  int cnt = 0;
  bool bad = false;
  int bits(int n) {
    while (cnt < n) {
      if (p >= end) { bad = true; return 0; }
      buf |= (uint32_t)(*p++) << cnt;
      cnt += 8;
    }
    const int v = (int)(buf & ((1u << n) - 1u));
    buf >>= n; cnt -= n;
    return v;
  }
};

struct Huffman {
  uint16_t count[16];
  uint16_t symbol[288];
  bool build(const uint8_t *len, int n) {
    memset(count, 0, sizeof(count));
    for (int i = 0; i < n; i++) count[len[i]]++;
    if (count[0] == n) return true;  // no codes: legal for an unused distance tree
    int left = 1;
    for (int l = 1; l < 16; l++) { left <<= 1; left -= count[l]; if (left < 0) return false; }
    uint16_t offs[16];
    offs[1] = 0;
    for (int l = 1; l < 15; l++) offs[l + 1] = offs[l] + count[l];
    for (int i = 0; i < n; i++) if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
    return true;
  }
  int decode(BitReader &br) const {
    int code = 0, first = 0, index = 0;
    for (int l = 1; l < 16; l++) {
      code |= br.bits(1);
      if (br.bad) return -1;
      const int c = count[l];
      if (code - c < first) return symbol[index + (code - first)];
      index += c; first += c; first <<= 1; code <<= 1;
    }
    return -1;
  }
};

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint16_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint16_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

bool inflate_codes(BitReader &br, std::vector<uint8_t> &out, const Huffman &lencode, const Huffman &distcode) {
  for (;;) {
    int sym = lencode.decode(br);
    if (sym < 0) return false;
    if (sym < 256) { out.push_back((uint8_t)sym); continue; }
    if (sym == 256) return true;
    sym -= 257;
    if (sym >= 29) return false;
    const int len = kLenBase[sym] + br.bits(kLenExtra[sym]);
    const int ds = distcode.decode(br);
    if (ds < 0 || ds >= 30) return false;
    const size_t dist = (size_t)kDistBase[ds] + (size_t)br.bits(kDistExtra[ds]);
    if (br.bad || dist > out.size()) return false;
    const size_t start = out.size() - dist;
    for (int i = 0; i < len; i++) out.push_back(out[start + (size_t)i]);
  }
}

bool inflate_raw(const uint8_t *src, size_t n, std::vector<uint8_t> &out) {
  BitReader br{src, src + n};
  int last;
  do {
    last = br.bits(1);
    const int type = br.bits(2);
    if (br.bad) return false;
    if (type == 0) {
      br.buf = 0; br.cnt = 0;  // to the byte boundary
      if (br.p + 4 > br.end) return false;
      const unsigned len = br.p[0] | (br.p[1] << 8), nlen = br.p[2] | (br.p[3] << 8);
      br.p += 4;
      if ((len ^ 0xFFFFu) != nlen || br.p + len > br.end) return false;
      out.insert(out.end(), br.p, br.p + len);
      br.p += len;
    } else if (type == 1) {
      uint8_t l[320];
      int i = 0;
      for (; i < 144; i++) l[i] = 8;
      for (; i < 256; i++) l[i] = 9;
      for (; i < 280; i++) l[i] = 7;
      for (; i < 288; i++) l[i] = 8;
      Huffman lc, dc;
      lc.build(l, 288);
      for (i = 0; i < 30; i++) l[i] = 5;
      dc.build(l, 30);
      if (!inflate_codes(br, out, lc, dc)) return false;
    } else if (type == 2) {
      const int nlen = br.bits(5) + 257, ndist = br.bits(5) + 1, ncode = br.bits(4) + 4;
      if (br.bad || nlen > 286 || ndist > 30) return false;
      static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
      uint8_t l[320];
      memset(l, 0, sizeof(l));
      for (int i = 0; i < ncode; i++) l[order[i]] = (uint8_t)br.bits(3);
      Huffman cl;
      if (!cl.build(l, 19)) return false;
      uint8_t lens[320];
      int idx = 0;
      while (idx < nlen + ndist) {
        const int sym = cl.decode(br);
        if (sym < 0) return false;
        if (sym < 16) { lens[idx++] = (uint8_t)sym; continue; }
        int rep, val = 0;
        if (sym == 16) { if (idx == 0) return false; val = lens[idx - 1]; rep = 3 + br.bits(2); }
        else if (sym == 17) rep = 3 + br.bits(3);
        else rep = 11 + br.bits(7);
        if (br.bad || idx + rep > nlen + ndist) return false;
        while (rep--) lens[idx++] = (uint8_t)val;
      }
      if (lens[256] == 0) return false;
      Huffman lc, dc;
      if (!lc.build(lens, nlen) || !dc.build(lens + nlen, ndist)) return false;
      if (!inflate_codes(br, out, lc, dc)) return false;
    } else {
      return false;
    }
  } while (!last);
  return !br.bad;
}

bool zlib_decompress(const std::vector<uint8_t> &z, std::vector<uint8_t> &out) {
  if (z.size() < 6) return false;
  if ((z[0] & 0x0F) != 8 || ((z[0] << 8) | z[1]) % 31 != 0 || (z[1] & 0x20)) return false;
  if (!inflate_raw(z.data() + 2, z.size() - 6, out)) return false;
  uint32_t a = 1, b = 0;  // Adler-32
  for (uint8_t c : out) { a = (a + c) % 65521u; b = (b + a) % 65521u; }
  const uint32_t want = ((uint32_t)z[z.size() - 4] << 24) | ((uint32_t)z[z.size() - 3] << 16) | ((uint32_t)z[z.size() - 2] << 8) | z[z.size() - 1];
  return ((b << 16) | a) == want;
}

uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

int decode_png(const std::vector<uint8_t> &f, HostImage &img) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (f.size() < 8 + 25 || memcmp(f.data(), sig, 8) != 0) return SVOSLAM_ERR_FORMAT;
  size_t pos = 8;
  int w = 0, h = 0, depth = 0, ctype = -1;
  std::vector<uint8_t> z;
  bool end = false;
  while (!end && pos + 12 <= f.size()) {
    const uint32_t len = be32(&f[pos]);
    const uint8_t *type = &f[pos + 4];
    if (pos + 12 + (size_t)len > f.size()) return SVOSLAM_ERR_FORMAT;
    const uint8_t *d = &f[pos + 8];
    if (!memcmp(type, "IHDR", 4)) {
      if (len < 13) return SVOSLAM_ERR_FORMAT;
      w = (int)be32(d); h = (int)be32(d + 4); depth = d[8]; ctype = d[9];
      if (d[10] != 0 || d[11] != 0 || d[12] != 0) return SVOSLAM_ERR_FORMAT;  // compression, filter, interlace
    } else if (!memcmp(type, "IDAT", 4)) {
      z.insert(z.end(), d, d + len);
    } else if (!memcmp(type, "IEND", 4)) {
      end = true;
    }
    pos += 12 + (size_t)len;
  }
  if (w <= 0 || h <= 0 || w > 16384 || h > 16384) return SVOSLAM_ERR_FORMAT;
  int channels;
  if (ctype == 0 && (depth == 16 || depth == 8)) channels = 1;
  else if (ctype == 2 && depth == 8) channels = 3;
  else if (ctype == 6 && depth == 8) channels = 4;  // alpha dropped
  else return SVOSLAM_ERR_FORMAT;
  const int bpp = channels * depth / 8;
  const size_t stride = (size_t)w * (size_t)bpp;
  std::vector<uint8_t> raw;
  raw.reserve((stride + 1) * (size_t)h);
  if (!zlib_decompress(z, raw) || raw.size() != (stride + 1) * (size_t)h) return SVOSLAM_ERR_FORMAT;
  std::vector<uint8_t> prev(stride, 0), cur(stride);
  img.width = w; img.height = h;
  img.channels = channels == 4 ? 3 : channels;
  img.bits = depth;
  img.data.assign((size_t)w * h * img.channels * (depth / 8), 0);
  for (int y = 0; y < h; y++) {
    const uint8_t *row = &raw[(size_t)y * (stride + 1)];
    const int ft = row[0];
    for (size_t i = 0; i < stride; i++) {
      const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
      int v = row[1 + i];
      switch (ft) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) >> 1; break;
        case 4: v += paeth(a, b, c); break;
        default: return SVOSLAM_ERR_FORMAT;
      }
      cur[i] = (uint8_t)v;
    }
    if (channels == 1 && depth == 16) {  // big-endian samples -> host uint16
      uint16_t *o = reinterpret_cast<uint16_t *>(img.data.data()) + (size_t)y * w;
      for (int x = 0; x < w; x++) o[x] = (uint16_t)((cur[2 * x] << 8) | cur[2 * x + 1]);
    } else if (channels == 4) {
      uint8_t *o = &img.data[(size_t)y * w * 3];
      for (int x = 0; x < w; x++) { o[3 * x] = cur[4 * x]; o[3 * x + 1] = cur[4 * x + 1]; o[3 * x + 2] = cur[4 * x + 2]; }
    } else {
      memcpy(&img.data[(size_t)y * stride], cur.data(), stride);
    }
    prev.swap(cur);
  }
  return SVOSLAM_OK;
}

// "P5"/"P6" <ws> width <ws> height <ws> maxval <single ws> binary samples; '#' comments in the header
int decode_pnm(const std::vector<uint8_t> &f, HostImage &img) {
  if (f.size() < 8 || f[0] != 'P' || (f[1] != '5' && f[1] != '6')) return SVOSLAM_ERR_FORMAT;
  size_t pos = 2;
  long vals[3];
  for (int k = 0; k < 3; k++) {
    for (;;) {
      while (pos < f.size() && isspace(f[pos])) pos++;
      if (pos < f.size() && f[pos] == '#') { while (pos < f.size() && f[pos] != '\n') pos++; continue; }
      break;
    }
    long v = 0; bool any = false;
    while (pos < f.size() && f[pos] >= '0' && f[pos] <= '9') { v = v * 10 + (f[pos++] - '0'); any = true; if (v > 1000000) return SVOSLAM_ERR_FORMAT; }
    if (!any) return SVOSLAM_ERR_FORMAT;
    vals[k] = v;
  }
  if (pos >= f.size() || !isspace(f[pos])) return SVOSLAM_ERR_FORMAT;
  pos++;
  const int w = (int)vals[0], h = (int)vals[1];
  const long maxval = vals[2];
  if (w <= 0 || h <= 0 || w > 16384 || h > 16384 || maxval <= 0 || maxval > 65535) return SVOSLAM_ERR_FORMAT;
  const int channels = f[1] == '5' ? 1 : 3, bytes = maxval > 255 ? 2 : 1;
  if (channels == 3 && bytes != 1) return SVOSLAM_ERR_FORMAT;
  const size_t need = (size_t)w * h * channels * bytes;
  if (f.size() - pos < need) return SVOSLAM_ERR_FORMAT;
  img.width = w; img.height = h; img.channels = channels; img.bits = 8 * bytes;
  img.data.resize(need);
  if (bytes == 2) {
    uint16_t *o = reinterpret_cast<uint16_t *>(img.data.data());
    for (size_t i = 0; i < (size_t)w * h; i++) o[i] = (uint16_t)((f[pos + 2 * i] << 8) | f[pos + 2 * i + 1]);
  } else {
    memcpy(img.data.data(), &f[pos], need);
  }
  return SVOSLAM_OK;
}

int read_file(const char *path, std::vector<uint8_t> &out) {
  FILE *f = fopen(path, "rb");
  if (!f) return SVOSLAM_ERR_IO;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  if (n < 0) { fclose(f); return SVOSLAM_ERR_IO; }
  out.resize((size_t)n);
  const bool ok = n == 0 || fread(out.data(), 1, (size_t)n, f) == (size_t)n;
  fclose(f);
  return ok ? SVOSLAM_OK : SVOSLAM_ERR_IO;
}

}  // namespace

int image_load(const char *path, HostImage &img) {
  if (!path) return SVOSLAM_ERR_INVALID_ARG;
  std::vector<uint8_t> f;
  SVO_TRY(read_file(path, f));
  if (f.size() >= 2 && f[0] == 'P') return decode_pnm(f, img);
  return decode_png(f, img);
}

}  // namespace svoslam

// ---------------------------------------------------------------------------------------------
// frame reader
// ---------------------------------------------------------------------------------------------
struct svoslam_frame_reader {
  struct Entry { long long stamp_us; std::string depth, color; };
  std::vector<Entry> entries;
  size_t next = 0;
  int width = 0, height = 0;
  float depth_units_per_metre = 1000.0f;
  uint16_t *h_depth = nullptr;  // pinned staging
  uint8_t *h_color = nullptr;
};

namespace svoslam {

static std::string dir_of(const std::string &p) {
  const size_t s = p.find_last_of('/');
  return s == std::string::npos ? std::string(".") : p.substr(0, s);
}

int frame_reader_open(svoslam_frame_reader **out, const char *association_file, float depth_units_per_metre) {
  if (!out || !association_file || !(depth_units_per_metre > 0.0f)) return SVOSLAM_ERR_INVALID_ARG;
  FILE *f = fopen(association_file, "r");
  if (!f) return SVOSLAM_ERR_IO;
  svoslam_frame_reader *r = new svoslam_frame_reader();
  r->depth_units_per_metre = depth_units_per_metre;
  const std::string base = dir_of(association_file);
  char line[4096];
  int rc = SVOSLAM_OK;
  bool first_col_is_depth = true;
  while (fgets(line, sizeof(line), f)) {
    char *p = line;
    while (*p == ' ' || *p == '\t') p++;
    if (*p == '#' || *p == '\n' || *p == '\r' || *p == 0) continue;
    double ta = 0, tb = 0;
    char fa[2048], fb[2048];
    if (sscanf(p, "%lf %2047s %lf %2047s", &ta, fa, &tb, fb) != 4) { rc = SVOSLAM_ERR_FORMAT; break; }
    svoslam_frame_reader::Entry e;
    auto full = [&](const char *n) { return n[0] == '/' ? std::string(n) : base + "/" + n; };
    if (r->entries.empty()) {  // which column is the depth image: decided from the first pair (16-bit single channel)
      HostImage a, b;
      if ((rc = image_load(full(fa).c_str(), a)) != SVOSLAM_OK) break;
      if ((rc = image_load(full(fb).c_str(), b)) != SVOSLAM_OK) break;
      const bool a_depth = a.channels == 1 && a.bits == 16, b_depth = b.channels == 1 && b.bits == 16;
      if (a_depth == b_depth || a.width != b.width || a.height != b.height) { rc = SVOSLAM_ERR_FORMAT; break; }
      r->width = a.width; r->height = a.height;
      first_col_is_depth = a_depth;
    }
    e.depth = full(first_col_is_depth ? fa : fb);
    e.color = full(first_col_is_depth ? fb : fa);
    e.stamp_us = (long long)llround((first_col_is_depth ? ta : tb) * 1e6);  // the depth image's timestamp
    r->entries.push_back(e);
  }
  fclose(f);
  if (rc == SVOSLAM_OK && r->entries.empty()) rc = SVOSLAM_ERR_FORMAT;
  if (rc == SVOSLAM_OK) {
    const size_t n = (size_t)r->width * r->height;
    if (hipHostMalloc((void **)&r->h_depth, n * 2, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&r->h_color, n * 3, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      // no device / no pinned memory: plain host staging still serves frame_reader_next_host
      if (r->h_depth) { (void)hipHostFree(r->h_depth); r->h_depth = nullptr; }
      r->h_color = nullptr;
    }
  }
  if (rc != SVOSLAM_OK) { delete r; return rc; }
  *out = r;
  return SVOSLAM_OK;
}

int frame_reader_close(svoslam_frame_reader *r) {
  if (!r) return SVOSLAM_OK;
  if (r->h_depth) (void)hipHostFree(r->h_depth);
  if (r->h_color) (void)hipHostFree(r->h_color);
  delete r;
  return SVOSLAM_OK;
}

int frame_reader_info(const svoslam_frame_reader *r, int *width, int *height, int *num_frames) {
  if (!r) return SVOSLAM_ERR_INVALID_ARG;
  if (width) *width = r->width;
  if (height) *height = r->height;
  if (num_frames) *num_frames = (int)r->entries.size();
  return SVOSLAM_OK;
}

// openni_device.cpp:64-65
int focal_from_fov(int width, int height, float hfov_rad, float vfov_rad, float *fx, float *fy) {
  if (!fx || !fy || width <= 0 || height <= 0) return SVOSLAM_ERR_INVALID_ARG;
  *fx = (float)width / (2.0f * tanf(0.5f * hfov_rad));
  *fy = (float)height / (2.0f * tanf(0.5f * vfov_rad));
  return SVOSLAM_OK;
}

// decodes the next pair into host buffers (depth in millimetres); 1 in *got, 0 at the end of the list
int frame_reader_next_host(svoslam_frame_reader *r, uint16_t *h_depth, uint8_t *h_color, long long *timestamp, int *got) {
  if (!r || !h_depth || !h_color || !got) return SVOSLAM_ERR_INVALID_ARG;
  *got = 0;
  if (r->next >= r->entries.size()) return SVOSLAM_OK;
  const auto &e = r->entries[r->next];
  HostImage d, c;
  SVO_TRY(image_load(e.depth.c_str(), d));
  SVO_TRY(image_load(e.color.c_str(), c));
  if (d.channels != 1 || d.bits != 16 || c.channels != 3 || c.bits != 8 || d.width != r->width || d.height != r->height ||
      c.width != r->width || c.height != r->height)
    return SVOSLAM_ERR_FORMAT;
  const size_t n = (size_t)r->width * r->height;
  const uint16_t *src = reinterpret_cast<const uint16_t *>(d.data.data());
  if (r->depth_units_per_metre == 1000.0f) {
    memcpy(h_depth, src, n * 2);
  } else {  // e.g. TUM: 5000 units per metre -> millimetres, round to nearest
    const double k = 1000.0 / (double)r->depth_units_per_metre;
    for (size_t i = 0; i < n; i++) {
      const double mm = rint((double)src[i] * k);
      h_depth[i] = (uint16_t)(mm > 65535.0 ? 65535.0 : mm);
    }
  }
  memcpy(h_color, c.data.data(), n * 3);
  if (timestamp) *timestamp = e.stamp_us;
  r->next++;
  *got = 1;
  return SVOSLAM_OK;
}

// RawFrame on the device (OpenNIDevice::readFrame, openni_device.cpp:93-150): blocking upload
int frame_reader_next(svoslam_frame_reader *r, uint16_t *d_depth, uint8_t *d_color, long long *timestamp, int *got,
                      hipStream_t stream) {
  if (!r || !d_depth || !d_color || !got) return SVOSLAM_ERR_INVALID_ARG;
  if (!r->h_depth || !r->h_color) return SVOSLAM_ERR_NO_DEVICE;
  SVO_HIP(hipStreamSynchronize(stream));  // the staging buffers may still feed the previous upload
  SVO_TRY(frame_reader_next_host(r, r->h_depth, r->h_color, timestamp, got));
  if (!*got) return SVOSLAM_OK;
  const size_t n = (size_t)r->width * r->height;
  SVO_HIP(hipMemcpyAsync(d_depth, r->h_depth, n * 2, hipMemcpyHostToDevice, stream));
  SVO_HIP(hipMemcpyAsync(d_color, r->h_color, n * 3, hipMemcpyHostToDevice, stream));
  return SVOSLAM_OK;
}

int frame_reader_rewind(svoslam_frame_reader *r) {
  if (!r) return SVOSLAM_ERR_INVALID_ARG;
  r->next = 0;
  return SVOSLAM_OK;
}

}  // namespace svoslam
