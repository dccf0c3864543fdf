// gRPC message compression ("gzip" and "deflate" = zlib stream) on zlib.
#pragma once

#include <string>

#include "tca/error.h"

namespace tca {
namespace grpcz {

// `gzip` selects the gzip container, otherwise a zlib stream ("deflate").
Error Compress(const void* data, size_t len, bool gzip, std::string* out);
// Accepts either container (auto-detected); `limit` bounds the output size.
Error Decompress(const void* data, size_t len, std::string* out, size_t limit = (size_t(1) << 31));

}  // namespace grpcz
}  // namespace tca
