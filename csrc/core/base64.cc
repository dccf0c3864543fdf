#include "tca/base64.h"

namespace tca {

namespace {
constexpr char kAlphabet[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";

struct DecodeTable {
  int8_t v[256];
  constexpr DecodeTable() : v() {
    for (int i = 0; i < 256; ++i) v[i] = -1;
    for (int i = 0; i < 64; ++i) v[static_cast<unsigned char>(kAlphabet[i])] = static_cast<int8_t>(i);
    v[static_cast<unsigned char>('-')] = 62;  // url-safe variants
    v[static_cast<unsigned char>('_')] = 63;
  }
};
constexpr DecodeTable kDecode;
}  // namespace

std::string Base64Encode(const void* data, size_t n) {
  const uint8_t* p = static_cast<const uint8_t*>(data);
  std::string out;
  out.resize(Base64EncodedSize(n));
  char* o = &out[0];
  size_t i = 0;
  for (; i + 3 <= n; i += 3) {
    uint32_t w = (p[i] << 16) | (p[i + 1] << 8) | p[i + 2];
    *o++ = kAlphabet[(w >> 18) & 63];
    *o++ = kAlphabet[(w >> 12) & 63];
    *o++ = kAlphabet[(w >> 6) & 63];
    *o++ = kAlphabet[w & 63];
  }
  if (i < n) {
    uint32_t w = p[i] << 16;
    if (i + 1 < n) w |= p[i + 1] << 8;
    *o++ = kAlphabet[(w >> 18) & 63];
    *o++ = kAlphabet[(w >> 12) & 63];
    *o++ = (i + 1 < n) ? kAlphabet[(w >> 6) & 63] : '=';
    *o++ = '=';
  }
  return out;
}

bool Base64Decode(const char* text, size_t n, std::string* out) {
  out->clear();
  out->reserve(n / 4 * 3 + 3);
  uint32_t acc = 0;
  int bits = 0;
  bool seen_pad = false;
  for (size_t i = 0; i < n; ++i) {
    unsigned char c = static_cast<unsigned char>(text[i]);
    if (c == '\n' || c == '\r' || c == ' ' || c == '\t') continue;
    if (c == '=') { seen_pad = true; continue; }
    if (seen_pad) return false;
    int8_t d = kDecode.v[c];
    if (d < 0) return false;
    acc = (acc << 6) | static_cast<uint32_t>(d);
    bits += 6;
    if (bits >= 8) {
      bits -= 8;
      out->push_back(static_cast<char>((acc >> bits) & 0xFF));
    }
  }
  // 6 leftover bits would mean a dangling single symbol
  return bits < 6;
}

}  // namespace tca
