// gRPC message compression helpers (see tca/grpc_compress.h).
#include "tca/grpc_compress.h"

#include <zlib.h>

#include <cstring>

namespace tca {
namespace grpcz {

Error Compress(const void* data, size_t len, bool gzip, std::string* out) {
  z_stream zs;
  std::memset(&zs, 0, sizeof(zs));
  if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, gzip ? 15 + 16 : 15, 8, Z_DEFAULT_STRATEGY) != Z_OK)
    return Error("zlib: deflateInit2 failed");
  out->resize(deflateBound(&zs, len));
  zs.next_in = reinterpret_cast<Bytef*>(const_cast<void*>(data));
  zs.avail_in = static_cast<uInt>(len);
  zs.next_out = reinterpret_cast<Bytef*>(&(*out)[0]);
  zs.avail_out = static_cast<uInt>(out->size());
  const int rc = deflate(&zs, Z_FINISH);
  const size_t produced = zs.total_out;
  deflateEnd(&zs);
  if (rc != Z_STREAM_END) return Error("zlib: deflate failed");
  out->resize(produced);
  return Error::Success();
}

Error Decompress(const void* data, size_t len, std::string* out, size_t limit) {
  z_stream zs;
  std::memset(&zs, 0, sizeof(zs));
  if (inflateInit2(&zs, 15 + 32) != Z_OK) return Error("zlib: inflateInit2 failed");  // +32: gzip or zlib header
  out->clear();
  zs.next_in = reinterpret_cast<Bytef*>(const_cast<void*>(data));
  zs.avail_in = static_cast<uInt>(len);
  char buf[64 * 1024];
  int rc = Z_OK;
  while (rc != Z_STREAM_END) {
    zs.next_out = reinterpret_cast<Bytef*>(buf);
    zs.avail_out = sizeof(buf);
    rc = inflate(&zs, Z_NO_FLUSH);
    if (rc != Z_OK && rc != Z_STREAM_END) {
      inflateEnd(&zs);
      return Error("zlib: corrupt compressed message");
    }
    out->append(buf, sizeof(buf) - zs.avail_out);
    if (out->size() > limit) {
      inflateEnd(&zs);
      return Error("decompressed message exceeds the size limit");
    }
    if (rc == Z_OK && zs.avail_in == 0 && zs.avail_out != 0) {
      inflateEnd(&zs);
      return Error("zlib: truncated compressed message");
    }
  }
  inflateEnd(&zs);
  return Error::Success();
}

}  // namespace grpcz
}  // namespace tca
