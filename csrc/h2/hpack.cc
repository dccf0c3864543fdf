// HPACK (RFC 7541): static table, dynamic table, integer / string primitives
// and the Huffman code.
//
// The Huffman code of RFC 7541 Appendix B is canonical: sorted by (code length,
// symbol value) the codes are consecutive integers, each length starting at
// (last code of the previous length + 1) shifted left by the length difference.
// So only the LENGTH of every symbol is tabulated here and the codes are
// generated at start-up; the generator checks that the lengths form a complete
// prefix code (Kraft sum exactly 1), which any mistyped length would break.
#include <algorithm>
#include <cstring>
#include <mutex>
#include <vector>

#include "tca/h2.h"

namespace tca {
namespace h2 {
namespace {

struct StaticEntry {
  const char* name;
  const char* value;
};

const StaticEntry kStatic[] = {
    {"", ""},  // index 0 is unused
    {":authority", ""}, {":method", "GET"}, {":method", "POST"}, {":path", "/"}, {":path", "/index.html"},
    {":scheme", "http"}, {":scheme", "https"}, {":status", "200"}, {":status", "204"}, {":status", "206"},
    {":status", "304"}, {":status", "400"}, {":status", "404"}, {":status", "500"}, {"accept-charset", ""},
    {"accept-encoding", "gzip, deflate"}, {"accept-language", ""}, {"accept-ranges", ""}, {"accept", ""},
    {"access-control-allow-origin", ""}, {"age", ""}, {"allow", ""}, {"authorization", ""}, {"cache-control", ""},
    {"content-disposition", ""}, {"content-encoding", ""}, {"content-language", ""}, {"content-length", ""},
    {"content-location", ""}, {"content-range", ""}, {"content-type", ""}, {"cookie", ""}, {"date", ""}, {"etag", ""},
    {"expect", ""}, {"expires", ""}, {"from", ""}, {"host", ""}, {"if-match", ""}, {"if-modified-since", ""},
    {"if-none-match", ""}, {"if-range", ""}, {"if-unmodified-since", ""}, {"last-modified", ""}, {"link", ""},
    {"location", ""}, {"max-forwards", ""}, {"proxy-authenticate", ""}, {"proxy-authorization", ""}, {"range", ""},
    {"referer", ""}, {"refresh", ""}, {"retry-after", ""}, {"server", ""}, {"set-cookie", ""},
    {"strict-transport-security", ""}, {"transfer-encoding", ""}, {"user-agent", ""}, {"vary", ""}, {"via", ""},
    {"www-authenticate", ""},
};
constexpr size_t kStaticCount = sizeof(kStatic) / sizeof(kStatic[0]) - 1;  // 61

// --- Huffman -----------------------------------------------------------------
struct LengthGroup {
  int bits;
  std::vector<int> symbols;
};

std::vector<int> Range(int first, int last) {
  std::vector<int> v;
  for (int i = first; i <= last; ++i) v.push_back(i);
  return v;
}

std::vector<int> Cat(std::initializer_list<std::vector<int>> parts) {
  std::vector<int> v;
  for (const auto& p : parts) v.insert(v.end(), p.begin(), p.end());
  return v;
}

std::vector<int> Chars(const char* s) {
  std::vector<int> v;
  for (; *s; ++s) v.push_back(static_cast<unsigned char>(*s));
  return v;
}

struct HuffmanTables {
  uint32_t code[257];
  uint8_t bits[257];
  // decoding tree: node = {child0, child1}; leaves are stored as -(symbol + 1)
  std::vector<std::pair<int, int>> tree;
  bool valid = false;
};

const HuffmanTables& Huffman() {
  static HuffmanTables t;
  static std::once_flag once;
  std::call_once(once, [] {
    const std::vector<LengthGroup> groups = {
        {5, Chars("012aceiost")},
        {6, Cat({Chars(" %-./3456789=A_bdfghlmnpru")})},
        {7, Cat({Chars(":BCDEFGHIJKLMNOPQRSTUVWYjkqvwxyz")})},
        {8, Chars("&*,;XZ")},
        {10, Chars("!\"()?")},
        {11, Chars("'+|")},
        {12, Chars("#>")},
        {13, Cat({{0}, Chars("$@[]~")})},
        {14, Chars("^}")},
        {15, Chars("<`{")},
        {19, {92, 195, 208}},
        {20, {128, 130, 131, 162, 184, 194, 224, 226}},
        {21, {153, 161, 167, 172, 176, 177, 179, 209, 216, 217, 227, 229, 230}},
        {22, {129, 132, 133, 134, 136, 146, 154, 156, 160, 163, 164, 169, 170, 173, 178, 181, 185, 186, 187, 189, 190,
              196, 198, 228, 232, 233}},
        {23, {1, 135, 137, 138, 139, 140, 141, 143, 147, 149, 150, 151, 152, 155, 157, 158, 165, 166, 168, 174, 175, 180,
              182, 183, 188, 191, 197, 231, 239}},
        {24, {9, 142, 144, 145, 148, 159, 171, 206, 215, 225, 236, 237}},
        {25, {199, 207, 234, 235}},
        {26, {192, 193, 200, 201, 202, 205, 210, 213, 218, 219, 238, 240, 242, 243, 255}},
        {27, {203, 204, 211, 212, 214, 221, 222, 223, 241, 244, 245, 246, 247, 248, 250, 251, 252, 253, 254}},
        {28, Cat({Range(2, 8), {11, 12}, Range(14, 21), Range(23, 31), {127, 220, 249}})},
        {30, {10, 13, 22, 256}},
    };
    bool seen[257] = {false};
    uint64_t code = 0;
    int prev_bits = groups.front().bits;
    size_t count = 0;
    for (const auto& g : groups) {
      code <<= (g.bits - prev_bits);
      prev_bits = g.bits;
      std::vector<int> ordered = g.symbols;
      std::sort(ordered.begin(), ordered.end());
      for (int sym : ordered) {
        if (sym < 0 || sym > 256 || seen[sym]) return;  // table typo: leave invalid
        seen[sym] = true;
        t.code[sym] = static_cast<uint32_t>(code++);
        t.bits[sym] = static_cast<uint8_t>(g.bits);
        ++count;
      }
    }
    // complete prefix code: the code after the last one must be exactly 2^30
    if (count != 257 || code != (1ull << 30)) return;
    t.tree.emplace_back(0, 0);
    for (int sym = 0; sym <= 256; ++sym) {
      int node = 0;
      for (int b = t.bits[sym] - 1; b >= 0; --b) {
        const int bit = (t.code[sym] >> b) & 1;
        int child = bit ? t.tree[node].second : t.tree[node].first;
        if (b == 0) {
          child = -(sym + 1);
        } else if (child == 0) {  // the root (index 0) is never a child, so 0 means "unset"
          t.tree.emplace_back(0, 0);
          child = static_cast<int>(t.tree.size() - 1);
        }
        (bit ? t.tree[node].second : t.tree[node].first) = child;
        node = child;
      }
    }
    t.valid = true;
  });
  return t;
}

// --- primitives --------------------------------------------------------------
bool DecodeInt(const uint8_t*& p, const uint8_t* end, int prefix_bits, uint64_t* out) {
  if (p >= end) return false;
  const uint64_t mask = (1u << prefix_bits) - 1;
  uint64_t v = *p++ & mask;
  if (v < mask) { *out = v; return true; }
  int shift = 0;
  while (p < end) {
    const uint8_t b = *p++;
    v += static_cast<uint64_t>(b & 0x7F) << shift;
    shift += 7;
    if (!(b & 0x80)) { *out = v; return true; }
    if (shift > 56) return false;
  }
  return false;
}

void EncodeInt(uint64_t v, int prefix_bits, uint8_t first_byte_flags, std::string* out) {
  const uint64_t mask = (1u << prefix_bits) - 1;
  if (v < mask) {
    out->push_back(static_cast<char>(first_byte_flags | v));
    return;
  }
  out->push_back(static_cast<char>(first_byte_flags | mask));
  v -= mask;
  while (v >= 128) {
    out->push_back(static_cast<char>((v & 0x7F) | 0x80));
    v >>= 7;
  }
  out->push_back(static_cast<char>(v));
}

Error DecodeString(const uint8_t*& p, const uint8_t* end, std::string* out) {
  if (p >= end) return Error("HPACK: truncated string");
  const bool huffman = (*p & 0x80) != 0;
  uint64_t len = 0;
  if (!DecodeInt(p, end, 7, &len) || len > static_cast<uint64_t>(end - p)) return Error("HPACK: bad string length");
  if (huffman) {
    TCA_RETURN_IF_ERROR(HuffmanDecode(p, len, out));
  } else {
    out->assign(reinterpret_cast<const char*>(p), len);
  }
  p += len;
  return Error::Success();
}

void EncodeString(const std::string& s, std::string* out) {
  EncodeInt(s.size(), 7, 0x00, out);
  out->append(s);
}

}  // namespace

Error HuffmanDecode(const uint8_t* data, size_t len, std::string* out) {
  const HuffmanTables& t = Huffman();
  if (!t.valid) return Error("HPACK: Huffman table failed its self-check");
  out->clear();
  int node = 0;
  int depth = 0;       // bits consumed since the last symbol
  bool all_ones = true;  // of those bits
  for (size_t i = 0; i < len; ++i) {
    for (int b = 7; b >= 0; --b) {
      const int bit = (data[i] >> b) & 1;
      const int next = bit ? t.tree[node].second : t.tree[node].first;
      ++depth;
      all_ones = all_ones && bit;
      if (next < 0) {
        const int sym = -next - 1;
        if (sym == 256) return Error("HPACK: EOS symbol inside a Huffman string");
        out->push_back(static_cast<char>(sym));
        node = 0;
        depth = 0;
        all_ones = true;
      } else if (next == 0) {
        return Error("HPACK: invalid Huffman code");
      } else {
        node = next;
      }
    }
  }
  // padding: at most 7 bits, all ones (a prefix of EOS)
  if (depth > 7 || !all_ones) return Error("HPACK: invalid Huffman padding");
  return Error::Success();
}

void HuffmanEncode(const std::string& in, std::string* out) {
  const HuffmanTables& t = Huffman();
  out->clear();
  uint64_t acc = 0;
  int nbits = 0;
  for (unsigned char c : in) {
    acc = (acc << t.bits[c]) | t.code[c];
    nbits += t.bits[c];
    while (nbits >= 8) {
      out->push_back(static_cast<char>((acc >> (nbits - 8)) & 0xFF));
      nbits -= 8;
    }
  }
  if (nbits) out->push_back(static_cast<char>(((acc << (8 - nbits)) | ((1u << (8 - nbits)) - 1)) & 0xFF));
}

// --- decoder -------------------------------------------------------------------
Error HpackDecoder::Lookup(uint64_t index, HeaderField* out) const {
  if (index == 0) return Error("HPACK: index 0");
  if (index <= kStaticCount) {
    out->name = kStatic[index].name;
    out->value = kStatic[index].value;
    return Error::Success();
  }
  const uint64_t d = index - kStaticCount - 1;
  if (d >= table_.size()) return Error("HPACK: index beyond the dynamic table");
  *out = table_[d];
  return Error::Success();
}

void HpackDecoder::Evict() {
  while (table_bytes_ > table_limit_ && !table_.empty()) {
    table_bytes_ -= table_.back().name.size() + table_.back().value.size() + 32;
    table_.pop_back();
  }
}

void HpackDecoder::Insert(HeaderField field) {
  const size_t sz = field.name.size() + field.value.size() + 32;
  if (sz > table_limit_) {  // larger than the table: empties it
    table_.clear();
    table_bytes_ = 0;
    return;
  }
  table_.push_front(std::move(field));
  table_bytes_ += sz;
  Evict();
}

Error HpackDecoder::Decode(const uint8_t* p, size_t len, HeaderList* out) {
  const uint8_t* end = p + len;
  while (p < end) {
    const uint8_t b = *p;
    if (b & 0x80) {  // indexed
      uint64_t idx = 0;
      if (!DecodeInt(p, end, 7, &idx)) return Error("HPACK: bad index");
      HeaderField f;
      TCA_RETURN_IF_ERROR(Lookup(idx, &f));
      out->push_back(std::move(f));
    } else if ((b & 0xE0) == 0x20) {  // dynamic table size update
      uint64_t sz = 0;
      if (!DecodeInt(p, end, 5, &sz)) return Error("HPACK: bad table size update");
      if (sz > settings_limit_) return Error("HPACK: table size update above the advertised limit");
      table_limit_ = sz;
      Evict();
    } else {
      const bool incremental = (b & 0xC0) == 0x40;
      const int prefix = incremental ? 6 : 4;
      uint64_t idx = 0;
      if (!DecodeInt(p, end, prefix, &idx)) return Error("HPACK: bad literal");
      HeaderField f;
      if (idx) {
        HeaderField named;
        TCA_RETURN_IF_ERROR(Lookup(idx, &named));
        f.name = std::move(named.name);
      } else {
        TCA_RETURN_IF_ERROR(DecodeString(p, end, &f.name));
      }
      TCA_RETURN_IF_ERROR(DecodeString(p, end, &f.value));
      if (incremental) Insert(f);
      out->push_back(std::move(f));
    }
  }
  return Error::Success();
}

// --- encoder -------------------------------------------------------------------
void HpackEncoder::Encode(const HeaderList& headers, std::string* out) {
  for (const auto& h : headers) {
    size_t name_idx = 0;
    size_t full_idx = 0;
    for (size_t i = 1; i <= kStaticCount; ++i) {
      if (h.name == kStatic[i].name) {
        if (!name_idx) name_idx = i;
        if (kStatic[i].value[0] != '\0' && h.value == kStatic[i].value) { full_idx = i; break; }
      }
    }
    if (full_idx) {
      EncodeInt(full_idx, 7, 0x80, out);
    } else if (name_idx) {
      EncodeInt(name_idx, 4, 0x00, out);  // literal without indexing, indexed name
      EncodeString(h.value, out);
    } else {
      out->push_back(0x00);  // literal without indexing, new name
      EncodeString(h.name, out);
      EncodeString(h.value, out);
    }
  }
}

}  // namespace h2
}  // namespace tca
