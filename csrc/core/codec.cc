#include "tca/codec.h"

#include <cstring>

namespace tca {

namespace {
struct DtypeRow {
  const char* name;
  DataType dt;
  size_t size;
};
constexpr DtypeRow kRows[] = {
    {"BOOL", DataType::BOOL, 1},     {"UINT8", DataType::UINT8, 1},   {"UINT16", DataType::UINT16, 2},
    {"UINT32", DataType::UINT32, 4}, {"UINT64", DataType::UINT64, 8}, {"INT8", DataType::INT8, 1},
    {"INT16", DataType::INT16, 2},   {"INT32", DataType::INT32, 4},   {"INT64", DataType::INT64, 8},
    {"FP16", DataType::FP16, 2},     {"FP32", DataType::FP32, 4},     {"FP64", DataType::FP64, 8},
    {"BYTES", DataType::BYTES, 0},   {"BF16", DataType::BF16, 2},
};
}  // namespace

DataType ParseDataType(const std::string& name) {
  for (const auto& r : kRows)
    if (name == r.name) return r.dt;
  return DataType::INVALID;
}

const char* DataTypeName(DataType dt) {
  for (const auto& r : kRows)
    if (dt == r.dt) return r.name;
  return "INVALID";
}

size_t DataTypeByteSize(DataType dt) {
  for (const auto& r : kRows)
    if (dt == r.dt) return r.size;
  return 0;
}

int64_t ElementCount(const std::vector<int64_t>& shape) {
  int64_t n = 1;
  for (int64_t d : shape) {
    if (d < 0) return -1;
    n *= d;
  }
  return n;
}

size_t BytesSerializedSize(const uint32_t* lens, size_t n) {
  size_t total = 4 * n;
  for (size_t i = 0; i < n; ++i) total += lens[i];
  return total;
}

void BytesAppend(const void* data, uint32_t len, std::string* out) {
  const uint8_t prefix[4] = {static_cast<uint8_t>(len), static_cast<uint8_t>(len >> 8),
                             static_cast<uint8_t>(len >> 16), static_cast<uint8_t>(len >> 24)};
  out->append(reinterpret_cast<const char*>(prefix), 4);
  out->append(static_cast<const char*>(data), len);
}

std::string BytesSerialize(const std::vector<std::string>& items) {
  size_t total = 4 * items.size();
  for (const auto& s : items) total += s.size();
  std::string out;
  out.reserve(total);
  for (const auto& s : items) BytesAppend(s.data(), static_cast<uint32_t>(s.size()), &out);
  return out;
}

Error BytesIndex(const uint8_t* buf, size_t size, int64_t expected,
                 std::vector<uint64_t>* offsets, std::vector<uint32_t>* lens) {
  offsets->clear();
  lens->clear();
  size_t pos = 0;
  while (expected < 0 ? pos < size : static_cast<int64_t>(lens->size()) < expected) {
    if (pos + 4 > size)
      return Error("malformed BYTES tensor: truncated length prefix at offset " + std::to_string(pos), 400);
    uint32_t len = static_cast<uint32_t>(buf[pos]) | (static_cast<uint32_t>(buf[pos + 1]) << 8) |
                   (static_cast<uint32_t>(buf[pos + 2]) << 16) | (static_cast<uint32_t>(buf[pos + 3]) << 24);
    pos += 4;
    if (len > size - pos)
      return Error("malformed BYTES tensor: element of " + std::to_string(len) +
                       " bytes overruns the buffer at offset " + std::to_string(pos), 400);
    offsets->push_back(pos);
    lens->push_back(len);
    pos += len;
  }
  return Error::Success();
}

Error BytesDeserialize(const uint8_t* buf, size_t size, int64_t expected,
                       std::vector<std::string>* items) {
  std::vector<uint64_t> offs;
  std::vector<uint32_t> lens;
  TCA_RETURN_IF_ERROR(BytesIndex(buf, size, expected, &offs, &lens));
  items->clear();
  items->reserve(offs.size());
  for (size_t i = 0; i < offs.size(); ++i)
    items->emplace_back(reinterpret_cast<const char*>(buf + offs[i]), lens[i]);
  return Error::Success();
}

void Fp32ToBf16Truncate(const float* src, uint16_t* dst, size_t n) {
  const uint32_t* bits = reinterpret_cast<const uint32_t*>(src);
  for (size_t i = 0; i < n; ++i) {
    uint32_t w;
    std::memcpy(&w, bits + i, 4);
    dst[i] = static_cast<uint16_t>(w >> 16);
  }
}

void Bf16ToFp32(const uint16_t* src, float* dst, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    uint32_t w = static_cast<uint32_t>(src[i]) << 16;
    std::memcpy(dst + i, &w, 4);
  }
}

float HalfToFloat(uint16_t h) {
  uint32_t sign = static_cast<uint32_t>(h & 0x8000) << 16;
  uint32_t exp = (h >> 10) & 0x1F;
  uint32_t man = h & 0x3FF;
  uint32_t w;
  if (exp == 0) {
    if (man == 0) {
      w = sign;
    } else {  // subnormal: renormalise
      int shift = 0;
      while (!(man & 0x400)) { man <<= 1; ++shift; }
      man &= 0x3FF;
      w = sign | ((127 - 15 - shift + 1) << 23) | (man << 13);
    }
  } else if (exp == 31) {
    w = sign | 0x7F800000u | (man << 13);
  } else {
    w = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  std::memcpy(&f, &w, 4);
  return f;
}

uint16_t FloatToHalf(float f) {
  uint32_t w;
  std::memcpy(&w, &f, 4);
  uint32_t sign = (w >> 16) & 0x8000;
  int32_t exp = static_cast<int32_t>((w >> 23) & 0xFF) - 127 + 15;
  uint32_t man = w & 0x7FFFFF;
  if (((w >> 23) & 0xFF) == 0xFF) return static_cast<uint16_t>(sign | 0x7C00 | (man ? 0x200 | (man >> 13) : 0));
  if (exp >= 31) return static_cast<uint16_t>(sign | 0x7C00);
  if (exp <= 0) {
    if (exp < -10) return static_cast<uint16_t>(sign);
    man |= 0x800000;
    uint32_t shift = static_cast<uint32_t>(14 - exp);
    uint32_t half = man >> shift;
    uint32_t rem = man & ((1u << shift) - 1);
    uint32_t mid = 1u << (shift - 1);
    if (rem > mid || (rem == mid && (half & 1))) ++half;
    return static_cast<uint16_t>(sign | half);
  }
  uint32_t half = (static_cast<uint32_t>(exp) << 10) | (man >> 13);
  uint32_t rem = man & 0x1FFF;
  if (rem > 0x1000 || (rem == 0x1000 && (half & 1))) ++half;  // RNE, may carry into exponent
  return static_cast<uint16_t>(sign | half);
}

}  // namespace tca
