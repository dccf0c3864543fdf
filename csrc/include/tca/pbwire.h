// Protocol-buffers wire format, reader and writer (header only).  The image
// has no protobuf C++ runtime; the gRPC client / server front encode and decode
// the `inference.*` messages with these primitives (hand-written for the
// inference hot path, table-driven for everything else, see tca/pbjson.h).
#pragma once

#include <cstdint>
#include <cstring>
#include <string>

namespace tca {
namespace pb {

enum WireType : uint32_t { kVarint = 0, kFixed64 = 1, kLen = 2, kFixed32 = 5 };

class Writer {
 public:
  explicit Writer(std::string* out) : out_(out) {}

  void Varint(uint64_t v) {
    while (v >= 0x80) {
      out_->push_back(static_cast<char>((v & 0x7F) | 0x80));
      v >>= 7;
    }
    out_->push_back(static_cast<char>(v));
  }
  void Tag(uint32_t field, WireType wt) { Varint((static_cast<uint64_t>(field) << 3) | wt); }
  void UInt(uint32_t field, uint64_t v) { Tag(field, kVarint); Varint(v); }
  void Int(uint32_t field, int64_t v) { UInt(field, static_cast<uint64_t>(v)); }
  void Bool(uint32_t field, bool v) { UInt(field, v ? 1 : 0); }
  void Double(uint32_t field, double v) {
    Tag(field, kFixed64);
    uint64_t bits;
    std::memcpy(&bits, &v, 8);
    for (int i = 0; i < 8; ++i) out_->push_back(static_cast<char>(bits >> (8 * i)));
  }
  void Float(uint32_t field, float v) {
    Tag(field, kFixed32);
    uint32_t bits;
    std::memcpy(&bits, &v, 4);
    for (int i = 0; i < 4; ++i) out_->push_back(static_cast<char>(bits >> (8 * i)));
  }
  void Bytes(uint32_t field, const void* data, size_t len) {
    Tag(field, kLen);
    Varint(len);
    out_->append(static_cast<const char*>(data), len);
  }
  void String(uint32_t field, const std::string& s) { Bytes(field, s.data(), s.size()); }
  // Header of a length-delimited field whose `len` payload bytes the caller appends itself.
  void LenHeader(uint32_t field, size_t len) { Tag(field, kLen); Varint(len); }
  // Nested message built in a scratch string.
  void Message(uint32_t field, const std::string& body) { String(field, body); }
  std::string* out() { return out_; }

  static size_t VarintSize(uint64_t v) {
    size_t n = 1;
    while (v >= 0x80) { v >>= 7; ++n; }
    return n;
  }

 private:
  std::string* out_;
};

class Reader {
 public:
  Reader(const void* data, size_t len) : p_(static_cast<const uint8_t*>(data)), end_(p_ + len) {}
  explicit Reader(const std::string& s) : Reader(s.data(), s.size()) {}

  bool Done() const { return p_ >= end_ || failed_; }
  bool Failed() const { return failed_; }

  // Reads the next tag; false at the end of input or on malformed data.
  bool Next(uint32_t* field, uint32_t* wire_type) {
    if (p_ >= end_ || failed_) return false;
    uint64_t tag = 0;
    if (!Varint(&tag)) return false;
    *field = static_cast<uint32_t>(tag >> 3);
    *wire_type = static_cast<uint32_t>(tag & 7);
    return true;
  }
  bool Varint(uint64_t* v) {
    uint64_t r = 0;
    for (int shift = 0; shift < 64 && p_ < end_; shift += 7) {
      const uint8_t b = *p_++;
      r |= static_cast<uint64_t>(b & 0x7F) << shift;
      if (!(b & 0x80)) { *v = r; return true; }
    }
    failed_ = true;
    return false;
  }
  bool Fixed64(uint64_t* v) {
    if (end_ - p_ < 8) { failed_ = true; return false; }
    uint64_t r = 0;
    for (int i = 0; i < 8; ++i) r |= static_cast<uint64_t>(p_[i]) << (8 * i);
    p_ += 8;
    *v = r;
    return true;
  }
  bool Fixed32(uint32_t* v) {
    if (end_ - p_ < 4) { failed_ = true; return false; }
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) r |= static_cast<uint32_t>(p_[i]) << (8 * i);
    p_ += 4;
    *v = r;
    return true;
  }
  // Length-delimited payload as a view into the input.
  bool Len(const uint8_t** data, size_t* len) {
    uint64_t n = 0;
    if (!Varint(&n) || n > static_cast<uint64_t>(end_ - p_)) { failed_ = true; return false; }
    *data = p_;
    *len = static_cast<size_t>(n);
    p_ += n;
    return true;
  }
  bool String(std::string* s) {
    const uint8_t* d;
    size_t n;
    if (!Len(&d, &n)) return false;
    s->assign(reinterpret_cast<const char*>(d), n);
    return true;
  }
  bool Skip(uint32_t wire_type) {
    uint64_t v;
    uint32_t w;
    const uint8_t* d;
    size_t n;
    switch (wire_type) {
      case kVarint: return Varint(&v);
      case kFixed64: return Fixed64(&v);
      case kLen: return Len(&d, &n);
      case kFixed32: return Fixed32(&w);
      default: failed_ = true; return false;
    }
  }

 private:
  const uint8_t* p_;
  const uint8_t* end_;
  bool failed_ = false;
};

}  // namespace pb
}  // namespace tca
