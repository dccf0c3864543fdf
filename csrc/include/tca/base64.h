// Base64 encode + decode (RFC 4648, no line wrapping).
//
// The reference ships only an encoder (libb64, `src/c++/library/cencode.c:30-109`)
// that inserts a newline every 72 characters; servers tolerate but do not need
// those, so this codec emits none.  The decoder exists because the in-repo v2
// server receives HIP IPC handles and model files in base64 over HTTP.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>

namespace tca {

// Number of characters produced for `n` input bytes (with '=' padding).
inline size_t Base64EncodedSize(size_t n) { return (n + 2) / 3 * 4; }

std::string Base64Encode(const void* data, size_t n);
inline std::string Base64Encode(const std::string& s) { return Base64Encode(s.data(), s.size()); }

// Decodes `text`; whitespace is skipped.  Returns false on an invalid symbol
// or a truncated quantum.
bool Base64Decode(const char* text, size_t n, std::string* out);
inline bool Base64Decode(const std::string& s, std::string* out) {
  return Base64Decode(s.data(), s.size(), out);
}

}  // namespace tca
