#include "image_size.h"

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace ptzcalib {
namespace {
uint32_t Be32(const unsigned char* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }
uint32_t Be16(const unsigned char* p) { return (uint32_t(p[0]) << 8) | p[1]; }
uint32_t Le32(const unsigned char* p) { return (uint32_t(p[3]) << 24) | (uint32_t(p[2]) << 16) | (uint32_t(p[1]) << 8) | p[0]; }
uint32_t Le16(const unsigned char* p) { return (uint32_t(p[1]) << 8) | p[0]; }

bool JpegSize(FILE* f, Size& size)
{
  // walk the marker segments up to the first start-of-frame (SOF0..SOF15 except DHT 0xC4, JPG 0xC8, DAC 0xCC)
  if (fseek(f, 2, SEEK_SET) != 0) return false;
  for (int guard = 0; guard < 4096; ++guard) {
    int c;
    do { c = fgetc(f); } while (c != EOF && c != 0xFF);
    if (c == EOF) return false;
    do { c = fgetc(f); } while (c == 0xFF);
    if (c == EOF) return false;
    const int marker = c;
    if (marker == 0xD8 || marker == 0x01 || (marker >= 0xD0 && marker <= 0xD7)) continue;  // no payload
    if (marker == 0xD9 || marker == 0xDA) return false;                                     // EOI / SOS before any SOF
    unsigned char len[2];
    if (fread(len, 1, 2, f) != 2) return false;
    const uint32_t seg = Be16(len);
    if (seg < 2) return false;
    if (marker >= 0xC0 && marker <= 0xCF && marker != 0xC4 && marker != 0xC8 && marker != 0xCC) {
      unsigned char b[5];
      if (fread(b, 1, 5, f) != 5) return false;
      size.height = static_cast<int>(Be16(b + 1));
      size.width = static_cast<int>(Be16(b + 3));
      return size.width > 0 && size.height > 0;
    }
    if (fseek(f, static_cast<long>(seg) - 2, SEEK_CUR) != 0) return false;
  }
  return false;
}

bool TiffSize(FILE* f, const unsigned char* head, Size& size)
{
  const bool le = head[0] == 'I';
  auto r16 = [&](const unsigned char* p) { return le ? Le16(p) : Be16(p); };
  auto r32 = [&](const unsigned char* p) { return le ? Le32(p) : Be32(p); };
  if (r16(head + 2) != 42) return false;
  const uint32_t ifd = r32(head + 4);
  if (fseek(f, static_cast<long>(ifd), SEEK_SET) != 0) return false;
  unsigned char nb[2];
  if (fread(nb, 1, 2, f) != 2) return false;
  const uint32_t n = r16(nb);
  int w = 0, h = 0;
  for (uint32_t i = 0; i < n; ++i) {
    unsigned char e[12];
    if (fread(e, 1, 12, f) != 12) return false;
    const uint32_t tag = r16(e), type = r16(e + 2);
    const uint32_t val = (type == 3) ? r16(e + 8) : r32(e + 8);
    if (tag == 256) w = static_cast<int>(val);
    if (tag == 257) h = static_cast<int>(val);
  }
  size.width = w; size.height = h;
  return w > 0 && h > 0;
}
}  // namespace

bool ReadImageSize(const std::string& path, Size& size)
{
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  unsigned char head[32];
  const size_t got = fread(head, 1, sizeof(head), f);
  bool ok = false;
  static const unsigned char kPng[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (got >= 24 && !memcmp(head, kPng, 8) && !memcmp(head + 12, "IHDR", 4)) {
    size.width = static_cast<int>(Be32(head + 16));
    size.height = static_cast<int>(Be32(head + 20));
    ok = size.width > 0 && size.height > 0;
  }
  else if (got >= 4 && head[0] == 0xFF && head[1] == 0xD8) ok = JpegSize(f, size);
  else if (got >= 26 && head[0] == 'B' && head[1] == 'M') {
    const uint32_t hdr = Le32(head + 14);
    if (hdr == 12) { size.width = static_cast<int>(Le16(head + 18)); size.height = static_cast<int>(Le16(head + 20)); }
    else {
      size.width = static_cast<int>(static_cast<int32_t>(Le32(head + 18)));
      const int32_t h = static_cast<int32_t>(Le32(head + 22));
      size.height = h < 0 ? -h : h;  // negative height = top-down rows
    }
    ok = size.width > 0 && size.height > 0;
  }
  else if (got >= 8 && ((head[0] == 'I' && head[1] == 'I') || (head[0] == 'M' && head[1] == 'M'))) ok = TiffSize(f, head, size);
  fclose(f);
  return ok;
}

}  // namespace ptzcalib
