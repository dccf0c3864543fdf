// Tensor codec: datatype table, BYTES (length-prefixed) and BF16 helpers.
//
// Wire formats (reference `src/python/library/tritonclient/utils/__init__.py:193-348`,
// `src/c++/library/common.cc:168-183`):
//   BYTES : for each element in row-major order, u32 little-endian byte count
//           followed by the payload (no terminator);
//   BF16  : upper 16 bits of the IEEE-754 binary32 pattern (truncation).
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "tca/error.h"

namespace tca {

enum class DataType : uint8_t {
  INVALID = 0, BOOL, UINT8, UINT16, UINT32, UINT64, INT8, INT16, INT32, INT64,
  FP16, FP32, FP64, BYTES, BF16
};

// "INT32" -> DataType::INT32 ; unknown names give DataType::INVALID.
DataType ParseDataType(const std::string& name);
const char* DataTypeName(DataType dt);
// Bytes per element; 0 for BYTES / INVALID (variable or unknown).
size_t DataTypeByteSize(DataType dt);
inline size_t DataTypeByteSize(const std::string& name) { return DataTypeByteSize(ParseDataType(name)); }

// Element count of a shape ([] -> 1); negative dims are rejected with -1.
int64_t ElementCount(const std::vector<int64_t>& shape);

// ---- BYTES ----
// Total wire size of `n` elements with the given payload lengths.
size_t BytesSerializedSize(const uint32_t* lens, size_t n);
// Append one element to `out`.
void BytesAppend(const void* data, uint32_t len, std::string* out);
// Serialise a vector of strings.
std::string BytesSerialize(const std::vector<std::string>& items);
// Walk a serialised buffer.  `offsets[i]` is the payload start of element i,
// `lens[i]` its length.  If `expected >= 0` exactly that many elements are
// read (trailing bytes ignored, as needed for shm regions larger than their
// payload), otherwise the whole buffer must be consumed.
Error BytesIndex(const uint8_t* buf, size_t size, int64_t expected,
                 std::vector<uint64_t>* offsets, std::vector<uint32_t>* lens);
Error BytesDeserialize(const uint8_t* buf, size_t size, int64_t expected,
                       std::vector<std::string>* items);

// ---- BF16 ----
void Fp32ToBf16Truncate(const float* src, uint16_t* dst, size_t n);
void Bf16ToFp32(const uint16_t* src, float* dst, size_t n);

// ---- FP16 (for JSON conversion of half tensors on the host) ----
float HalfToFloat(uint16_t h);
uint16_t FloatToHalf(float f);

}  // namespace tca
