// Small self-contained JSON document model, parser and writer.
//
// The reference links an external rapidjson wrapper (`triton_json.h`) and a
// `ParseJson` helper with NaN/Inf enabled (`src/c++/library/json_utils.cc:33-45`).
// None of those headers exist in this image, so the native client, the perf
// driver and the C++ examples share this implementation instead.
//
// Design: `Value` is a tagged union holding int64/uint64/double separately so
// that 64-bit tensor data round-trips exactly; objects keep insertion order
// (protocol bodies are small, linear lookup is faster than hashing here).
#pragma once

#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "tca/error.h"

namespace tca {
namespace json {

class Value;
using Member = std::pair<std::string, Value>;

class Value {
 public:
  enum class Kind : uint8_t { Null, Bool, Int, UInt, Double, String, Array, Object };

  Value() : kind_(Kind::Null) {}
  Value(std::nullptr_t) : kind_(Kind::Null) {}
  Value(bool b) : kind_(Kind::Bool) { u_.b = b; }
  Value(int v) : kind_(Kind::Int) { u_.i = v; }
  Value(long v) : kind_(Kind::Int) { u_.i = v; }
  Value(long long v) : kind_(Kind::Int) { u_.i = v; }
  Value(unsigned v) : kind_(Kind::UInt) { u_.u = v; }
  Value(unsigned long v) : kind_(Kind::UInt) { u_.u = v; }
  Value(unsigned long long v) : kind_(Kind::UInt) { u_.u = v; }
  Value(double v) : kind_(Kind::Double) { u_.d = v; }
  Value(float v) : kind_(Kind::Double) { u_.d = v; }
  Value(const char* s) : kind_(Kind::String), str_(s) {}
  Value(std::string s) : kind_(Kind::String), str_(std::move(s)) {}

  static Value Array() {
    Value v;
    v.kind_ = Kind::Array;
    return v;
  }
  static Value Object() {
    Value v;
    v.kind_ = Kind::Object;
    return v;
  }

  Kind kind() const { return kind_; }
  bool IsNull() const { return kind_ == Kind::Null; }
  bool IsBool() const { return kind_ == Kind::Bool; }
  bool IsNumber() const {
    return kind_ == Kind::Int || kind_ == Kind::UInt || kind_ == Kind::Double;
  }
  bool IsInteger() const { return kind_ == Kind::Int || kind_ == Kind::UInt; }
  bool IsString() const { return kind_ == Kind::String; }
  bool IsArray() const { return kind_ == Kind::Array; }
  bool IsObject() const { return kind_ == Kind::Object; }

  bool AsBool() const { return kind_ == Kind::Bool ? u_.b : AsInt() != 0; }
  int64_t AsInt() const {
    switch (kind_) {
      case Kind::Int: return u_.i;
      case Kind::UInt: return static_cast<int64_t>(u_.u);
      case Kind::Double: return static_cast<int64_t>(u_.d);
      case Kind::Bool: return u_.b ? 1 : 0;
      default: return 0;
    }
  }
  uint64_t AsUInt() const {
    switch (kind_) {
      case Kind::Int: return static_cast<uint64_t>(u_.i);
      case Kind::UInt: return u_.u;
      case Kind::Double: return static_cast<uint64_t>(u_.d);
      case Kind::Bool: return u_.b ? 1 : 0;
      default: return 0;
    }
  }
  double AsDouble() const {
    switch (kind_) {
      case Kind::Int: return static_cast<double>(u_.i);
      case Kind::UInt: return static_cast<double>(u_.u);
      case Kind::Double: return u_.d;
      case Kind::Bool: return u_.b ? 1.0 : 0.0;
      default: return 0.0;
    }
  }
  const std::string& AsString() const { return str_; }

  // ---- array ----
  size_t Size() const {
    return kind_ == Kind::Array ? arr_.size() : kind_ == Kind::Object ? obj_.size() : 0;
  }
  Value& operator[](size_t i) { return arr_[i]; }
  const Value& operator[](size_t i) const { return arr_[i]; }
  // int overloads: a literal 0 would otherwise be ambiguous with the const char* key form
  Value& operator[](int i) { return arr_[static_cast<size_t>(i)]; }
  const Value& operator[](int i) const { return arr_[static_cast<size_t>(i)]; }
  Value& Append(Value v) {
    if (kind_ != Kind::Array) { *this = Array(); }
    arr_.push_back(std::move(v));
    return arr_.back();
  }
  void Reserve(size_t n) { if (kind_ == Kind::Array) arr_.reserve(n); }
  const std::vector<Value>& Items() const { return arr_; }
  std::vector<Value>& Items() { return arr_; }

  // ---- object ----
  const Value* Find(const std::string& key) const {
    if (kind_ != Kind::Object) return nullptr;
    for (const auto& m : obj_) if (m.first == key) return &m.second;
    return nullptr;
  }
  Value* Find(const std::string& key) {
    return const_cast<Value*>(static_cast<const Value*>(this)->Find(key));
  }
  bool Has(const std::string& key) const { return Find(key) != nullptr; }
  Value& operator[](const std::string& key) {
    if (kind_ != Kind::Object) { *this = Object(); }
    for (auto& m : obj_) if (m.first == key) return m.second;
    obj_.emplace_back(key, Value());
    return obj_.back().second;
  }
  Value& operator[](const char* key) { return (*this)[std::string(key)]; }
  const std::vector<Member>& Members() const { return obj_; }
  bool Erase(const std::string& key) {
    for (auto it = obj_.begin(); it != obj_.end(); ++it)
      if (it->first == key) { obj_.erase(it); return true; }
    return false;
  }

  // ---- typed getters with defaults ----
  std::string GetString(const std::string& key, const std::string& dflt = "") const {
    const Value* v = Find(key);
    return (v && v->IsString()) ? v->AsString() : dflt;
  }
  int64_t GetInt(const std::string& key, int64_t dflt = 0) const {
    const Value* v = Find(key);
    return (v && (v->IsNumber() || v->IsBool())) ? v->AsInt() : dflt;
  }
  bool GetBool(const std::string& key, bool dflt = false) const {
    const Value* v = Find(key);
    return (v && (v->IsBool() || v->IsNumber())) ? v->AsBool() : dflt;
  }

  // ---- serialisation ----
  std::string Dump() const {
    std::string out;
    Write(&out);
    return out;
  }
  void Write(std::string* out) const;

 private:
  Kind kind_;
  union {
    bool b;
    int64_t i;
    uint64_t u;
    double d;
  } u_{};
  std::string str_;
  std::vector<Value> arr_;
  std::vector<Member> obj_;
};

namespace detail {

inline void WriteEscaped(const std::string& s, std::string* out) {
  out->push_back('"');
  for (unsigned char c : s) {
    switch (c) {
      case '"': out->append("\\\""); break;
      case '\\': out->append("\\\\"); break;
      case '\b': out->append("\\b"); break;
      case '\f': out->append("\\f"); break;
      case '\n': out->append("\\n"); break;
      case '\r': out->append("\\r"); break;
      case '\t': out->append("\\t"); break;
      default:
        if (c < 0x20) {
          char buf[8];
          std::snprintf(buf, sizeof(buf), "\\u%04x", c);
          out->append(buf);
        } else {
          out->push_back(static_cast<char>(c));
        }
    }
  }
  out->push_back('"');
}

inline void WriteDouble(double d, std::string* out) {
  if (std::isnan(d)) { out->append("NaN"); return; }
  if (std::isinf(d)) { out->append(d < 0 ? "-Infinity" : "Infinity"); return; }
  char buf[40];
  // shortest representation that round-trips
  for (int prec = 15; prec <= 17; ++prec) {
    std::snprintf(buf, sizeof(buf), "%.*g", prec, d);
    if (std::strtod(buf, nullptr) == d) break;
  }
  out->append(buf);
  if (!std::strpbrk(buf, ".eEnN")) out->append(".0");
}

class Parser {
 public:
  Parser(const char* begin, const char* end) : p_(begin), end_(end), begin_(begin) {}

  Error Run(Value* out) {
    SkipWs();
    TCA_RETURN_IF_ERROR(ParseValue(out, 0));
    SkipWs();
    if (p_ != end_) return Fail("trailing characters after JSON value");
    return Error::Success();
  }

 private:
  static constexpr int kMaxDepth = 256;

  Error Fail(const std::string& what) const {
    return Error("failed to parse JSON at offset " + std::to_string(p_ - begin_) + ": " + what, 400);
  }
  void SkipWs() {
    while (p_ < end_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\r' || *p_ == '\t')) ++p_;
  }
  bool Match(const char* lit) {
    size_t n = std::strlen(lit);
    if (static_cast<size_t>(end_ - p_) >= n && std::memcmp(p_, lit, n) == 0) {
      p_ += n;
      return true;
    }
    return false;
  }

  Error ParseValue(Value* out, int depth) {
    if (depth > kMaxDepth) return Fail("nesting too deep");
    if (p_ >= end_) return Fail("unexpected end of input");
    switch (*p_) {
      case '{': return ParseObject(out, depth);
      case '[': return ParseArray(out, depth);
      case '"': {
        std::string s;
        TCA_RETURN_IF_ERROR(ParseString(&s));
        *out = Value(std::move(s));
        return Error::Success();
      }
      case 't': if (Match("true")) { *out = Value(true); return Error::Success(); } break;
      case 'f': if (Match("false")) { *out = Value(false); return Error::Success(); } break;
      case 'n': if (Match("null")) { *out = Value(); return Error::Success(); } break;
      case 'N': if (Match("NaN")) { *out = Value(std::numeric_limits<double>::quiet_NaN()); return Error::Success(); } break;
      case 'I': if (Match("Infinity")) { *out = Value(std::numeric_limits<double>::infinity()); return Error::Success(); } break;
      default: break;
    }
    if (*p_ == '-' || (*p_ >= '0' && *p_ <= '9')) return ParseNumber(out);
    return Fail(std::string("unexpected character '") + *p_ + "'");
  }

  Error ParseNumber(Value* out) {
    const char* start = p_;
    bool neg = false, is_float = false;
    if (*p_ == '-') {
      neg = true;
      ++p_;
      if (Match("Infinity")) { *out = Value(-std::numeric_limits<double>::infinity()); return Error::Success(); }
    }
    if (p_ >= end_ || !(*p_ >= '0' && *p_ <= '9')) return Fail("malformed number");
    while (p_ < end_ && *p_ >= '0' && *p_ <= '9') ++p_;
    if (p_ < end_ && *p_ == '.') { is_float = true; ++p_; while (p_ < end_ && *p_ >= '0' && *p_ <= '9') ++p_; }
    if (p_ < end_ && (*p_ == 'e' || *p_ == 'E')) {
      is_float = true;
      ++p_;
      if (p_ < end_ && (*p_ == '+' || *p_ == '-')) ++p_;
      while (p_ < end_ && *p_ >= '0' && *p_ <= '9') ++p_;
    }
    std::string tok(start, p_);
    if (!is_float) {
      errno = 0;
      if (neg) {
        long long v = std::strtoll(tok.c_str(), nullptr, 10);
        if (errno == 0) { *out = Value(v); return Error::Success(); }
      } else {
        unsigned long long v = std::strtoull(tok.c_str(), nullptr, 10);
        if (errno == 0) {
          if (v <= static_cast<unsigned long long>(std::numeric_limits<int64_t>::max()))
            *out = Value(static_cast<long long>(v));
          else
            *out = Value(v);
          return Error::Success();
        }
      }
    }
    *out = Value(std::strtod(tok.c_str(), nullptr));
    return Error::Success();
  }

  static void AppendUtf8(uint32_t cp, std::string* s) {
    if (cp < 0x80) {
      s->push_back(static_cast<char>(cp));
    } else if (cp < 0x800) {
      s->push_back(static_cast<char>(0xC0 | (cp >> 6)));
      s->push_back(static_cast<char>(0x80 | (cp & 0x3F)));
    } else if (cp < 0x10000) {
      s->push_back(static_cast<char>(0xE0 | (cp >> 12)));
      s->push_back(static_cast<char>(0x80 | ((cp >> 6) & 0x3F)));
      s->push_back(static_cast<char>(0x80 | (cp & 0x3F)));
    } else {
      s->push_back(static_cast<char>(0xF0 | (cp >> 18)));
      s->push_back(static_cast<char>(0x80 | ((cp >> 12) & 0x3F)));
      s->push_back(static_cast<char>(0x80 | ((cp >> 6) & 0x3F)));
      s->push_back(static_cast<char>(0x80 | (cp & 0x3F)));
    }
  }

  bool Hex4(uint32_t* v) {
    if (end_ - p_ < 4) return false;
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) {
      char c = p_[i];
      r <<= 4;
      if (c >= '0' && c <= '9') r |= c - '0';
      else if (c >= 'a' && c <= 'f') r |= c - 'a' + 10;
      else if (c >= 'A' && c <= 'F') r |= c - 'A' + 10;
      else return false;
    }
    p_ += 4;
    *v = r;
    return true;
  }

  Error ParseString(std::string* s) {
    ++p_;  // opening quote
    while (true) {
      if (p_ >= end_) return Fail("unterminated string");
      const char* run = p_;
      while (p_ < end_ && *p_ != '"' && *p_ != '\\') ++p_;
      s->append(run, p_);
      if (p_ >= end_) return Fail("unterminated string");
      if (*p_ == '"') { ++p_; return Error::Success(); }
      ++p_;  // backslash
      if (p_ >= end_) return Fail("unterminated escape");
      char e = *p_++;
      switch (e) {
        case '"': s->push_back('"'); break;
        case '\\': s->push_back('\\'); break;
        case '/': s->push_back('/'); break;
        case 'b': s->push_back('\b'); break;
        case 'f': s->push_back('\f'); break;
        case 'n': s->push_back('\n'); break;
        case 'r': s->push_back('\r'); break;
        case 't': s->push_back('\t'); break;
        case 'u': {
          uint32_t cp;
          if (!Hex4(&cp)) return Fail("bad \\u escape");
          if (cp >= 0xD800 && cp <= 0xDBFF && end_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u') {
            const char* save = p_;
            p_ += 2;
            uint32_t lo;
            if (Hex4(&lo) && lo >= 0xDC00 && lo <= 0xDFFF) {
              cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
            } else {
              p_ = save;
            }
          }
          AppendUtf8(cp, s);
          break;
        }
        default: return Fail("unknown escape");
      }
    }
  }

  Error ParseArray(Value* out, int depth) {
    ++p_;
    *out = Value::Array();
    SkipWs();
    if (p_ < end_ && *p_ == ']') { ++p_; return Error::Success(); }
    while (true) {
      Value item;
      SkipWs();
      TCA_RETURN_IF_ERROR(ParseValue(&item, depth + 1));
      out->Append(std::move(item));
      SkipWs();
      if (p_ >= end_) return Fail("unterminated array");
      if (*p_ == ',') { ++p_; continue; }
      if (*p_ == ']') { ++p_; return Error::Success(); }
      return Fail("expected ',' or ']'");
    }
  }

  Error ParseObject(Value* out, int depth) {
    ++p_;
    *out = Value::Object();
    SkipWs();
    if (p_ < end_ && *p_ == '}') { ++p_; return Error::Success(); }
    while (true) {
      SkipWs();
      if (p_ >= end_ || *p_ != '"') return Fail("expected object key");
      std::string key;
      TCA_RETURN_IF_ERROR(ParseString(&key));
      SkipWs();
      if (p_ >= end_ || *p_ != ':') return Fail("expected ':'");
      ++p_;
      SkipWs();
      Value item;
      TCA_RETURN_IF_ERROR(ParseValue(&item, depth + 1));
      (*out)[key] = std::move(item);
      SkipWs();
      if (p_ >= end_) return Fail("unterminated object");
      if (*p_ == ',') { ++p_; continue; }
      if (*p_ == '}') { ++p_; return Error::Success(); }
      return Fail("expected ',' or '}'");
    }
  }

  const char* p_;
  const char* end_;
  const char* begin_;
};

}  // namespace detail

inline void Value::Write(std::string* out) const {
  switch (kind_) {
    case Kind::Null: out->append("null"); break;
    case Kind::Bool: out->append(u_.b ? "true" : "false"); break;
    case Kind::Int: out->append(std::to_string(u_.i)); break;
    case Kind::UInt: out->append(std::to_string(u_.u)); break;
    case Kind::Double: detail::WriteDouble(u_.d, out); break;
    case Kind::String: detail::WriteEscaped(str_, out); break;
    case Kind::Array: {
      out->push_back('[');
      bool first = true;
      for (const auto& v : arr_) {
        if (!first) out->push_back(',');
        first = false;
        v.Write(out);
      }
      out->push_back(']');
      break;
    }
    case Kind::Object: {
      out->push_back('{');
      bool first = true;
      for (const auto& m : obj_) {
        if (!first) out->push_back(',');
        first = false;
        detail::WriteEscaped(m.first, out);
        out->push_back(':');
        m.second.Write(out);
      }
      out->push_back('}');
      break;
    }
  }
}

// Parse `len` bytes of JSON text.  NaN / Infinity / -Infinity literals are
// accepted, matching the reference's kParseNanAndInfFlag behaviour.
inline Error Parse(const char* data, size_t len, Value* out) {
  detail::Parser p(data, data + len);
  return p.Run(out);
}
inline Error Parse(const std::string& text, Value* out) {
  return Parse(text.data(), text.size(), out);
}

}  // namespace json
}  // namespace tca
