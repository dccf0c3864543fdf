// C ABI over the host-side core (system shm + codec), consumed by Python via
// ctypes.  Plays the role of the reference's `libcshm.so`
// (`tritonclient/utils/shared_memory/shared_memory.cc:45-149`) with two
// differences: a region can also be *attached* (what the in-repo server needs)
// and reads/writes are bounds checked here rather than in Python.
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <thread>

#include "tca/base64.h"
#include "tca/codec.h"
#include "tca/image_io.h"
#include "tca/shm_utils.h"

namespace {

enum : int {
  kOk = 0,
  kErrBadArg = -1,
  kErrOpen = -2,      // shm_open failed
  kErrSize = -3,      // ftruncate failed / object too small
  kErrMap = -4,       // mmap failed
  kErrUnlink = -5,    // shm_unlink failed
  kErrUnmap = -6,     // munmap failed
  kErrRange = -7,     // offset + size outside the region
  kErrFormat = -8,    // malformed BYTES stream
};

struct Region {
  std::string name;
  std::string key;
  void* map_base = nullptr;
  size_t map_size = 0;
  uint8_t* addr = nullptr;  // map_base + bias
  uint64_t offset = 0;      // offset inside the shm object
  uint64_t size = 0;
  bool creator = false;
};

}  // namespace

extern "C" {

const char* tca_strerror(int code) {
  switch (code) {
    case kOk: return "success";
    case kErrBadArg: return "invalid argument";
    case kErrOpen: return "unable to get shared memory descriptor";
    case kErrSize: return "unable to initialize the size";
    case kErrMap: return "unable to read/mmap the shared memory region";
    case kErrUnlink: return "unable to unlink the shared memory region";
    case kErrUnmap: return "unable to munmap the shared memory region";
    case kErrRange: return "access outside of the shared memory region";
    case kErrFormat: return "malformed BYTES tensor";
    default: return "unknown error";
  }
}

int tca_shm_create(const char* name, const char* key, uint64_t byte_size, void** handle) {
  if (!name || !key || !handle) return kErrBadArg;
  int fd = shm_open(key, O_RDWR | O_CREAT, S_IRUSR | S_IWUSR);
  if (fd == -1) return kErrOpen;
  if (ftruncate(fd, static_cast<off_t>(byte_size)) == -1) {
    close(fd);
    return kErrSize;
  }
  void* p = byte_size ? mmap(nullptr, byte_size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0) : nullptr;
  close(fd);  // the mapping keeps the object alive
  if (p == MAP_FAILED) return kErrMap;
  Region* r = new (std::nothrow) Region();
  if (!r) { if (p) munmap(p, byte_size); return kErrBadArg; }
  r->name = name;
  r->key = key;
  r->map_base = p;
  r->map_size = byte_size;
  r->addr = static_cast<uint8_t*>(p);
  r->size = byte_size;
  r->creator = true;
  *handle = r;
  return kOk;
}

int tca_shm_attach(const char* name, const char* key, uint64_t offset, uint64_t byte_size, void** handle) {
  if (!name || !key || !handle) return kErrBadArg;
  int fd = shm_open(key, O_RDWR, S_IRUSR | S_IWUSR);
  if (fd == -1) return kErrOpen;
  struct stat st;
  if (fstat(fd, &st) == -1 || static_cast<uint64_t>(st.st_size) < offset + byte_size) {
    close(fd);
    return kErrSize;
  }
  const uint64_t page = static_cast<uint64_t>(sysconf(_SC_PAGESIZE));
  const uint64_t aligned = offset / page * page;
  const uint64_t bias = offset - aligned;
  void* p = mmap(nullptr, byte_size + bias, PROT_READ | PROT_WRITE, MAP_SHARED, fd, static_cast<off_t>(aligned));
  close(fd);
  if (p == MAP_FAILED) return kErrMap;
  Region* r = new (std::nothrow) Region();
  if (!r) { munmap(p, byte_size + bias); return kErrBadArg; }
  r->name = name;
  r->key = key;
  r->map_base = p;
  r->map_size = byte_size + bias;
  r->addr = static_cast<uint8_t*>(p) + bias;
  r->offset = offset;
  r->size = byte_size;
  *handle = r;
  return kOk;
}

int tca_shm_write(void* handle, uint64_t offset, uint64_t n, const void* src) {
  Region* r = static_cast<Region*>(handle);
  if (!r || (!src && n)) return kErrBadArg;
  if (offset > r->size || n > r->size - offset) return kErrRange;
  if (n) std::memcpy(r->addr + offset, src, n);
  return kOk;
}

int tca_shm_read(void* handle, uint64_t offset, uint64_t n, void* dst) {
  Region* r = static_cast<Region*>(handle);
  if (!r || (!dst && n)) return kErrBadArg;
  if (offset > r->size || n > r->size - offset) return kErrRange;
  if (n) std::memcpy(dst, r->addr + offset, n);
  return kOk;
}

int tca_shm_info(void* handle, void** addr, const char** name, const char** key, uint64_t* offset, uint64_t* byte_size) {
  Region* r = static_cast<Region*>(handle);
  if (!r) return kErrBadArg;
  if (addr) *addr = r->addr;
  if (name) *name = r->name.c_str();
  if (key) *key = r->key.c_str();
  if (offset) *offset = r->offset;
  if (byte_size) *byte_size = r->size;
  return kOk;
}

// unlink_key != 0 also removes the shm object (what a creator normally wants).
int tca_shm_destroy(void* handle, int unlink_key) {
  Region* r = static_cast<Region*>(handle);
  if (!r) return kErrBadArg;
  int rc = kOk;
  if (r->map_base && munmap(r->map_base, r->map_size) == -1) rc = kErrUnmap;
  if (unlink_key && shm_unlink(r->key.c_str()) == -1 && rc == kOk) rc = kErrUnlink;
  delete r;
  return rc;
}

// ---- codec ---------------------------------------------------------------
// Index a serialised BYTES buffer.  Returns the element count (>= 0) or a
// negative error.  When `cap` is smaller than the count only the first `cap`
// entries are written (call again with a larger buffer).
int64_t tca_bytes_index(const uint8_t* buf, uint64_t size, int64_t expected, uint64_t* offsets, uint32_t* lens, uint64_t cap) {
  std::vector<uint64_t> o;
  std::vector<uint32_t> l;
  tca::Error e = tca::BytesIndex(buf, size, expected, &o, &l);
  if (!e.IsOk()) return kErrFormat;
  const uint64_t n = o.size() < cap ? o.size() : cap;
  if (offsets && n) std::memcpy(offsets, o.data(), n * sizeof(uint64_t));
  if (lens && n) std::memcpy(lens, l.data(), n * sizeof(uint32_t));
  return static_cast<int64_t>(o.size());
}

void tca_fp32_to_bf16(const float* src, uint16_t* dst, uint64_t n) { tca::Fp32ToBf16Truncate(src, dst, n); }
void tca_bf16_to_fp32(const uint16_t* src, float* dst, uint64_t n) { tca::Bf16ToFp32(src, dst, n); }

uint64_t tca_base64_encode(const void* src, uint64_t n, char* dst, uint64_t cap) {
  std::string s = tca::Base64Encode(src, n);
  if (dst && s.size() <= cap) std::memcpy(dst, s.data(), s.size());
  return s.size();
}

int64_t tca_base64_decode(const char* src, uint64_t n, void* dst, uint64_t cap) {
  std::string out;
  if (!tca::Base64Decode(src, n, &out)) return kErrFormat;
  if (dst && out.size() <= cap) std::memcpy(dst, out.data(), out.size());
  return static_cast<int64_t>(out.size());
}

// ---- image files -> pixels ------------------------------------------------------------------
// A batch of encoded image files (PNG, sequential JPEG, binary PNM - by signature) decoded side by side on
// native threads: the caller (a Python model behind a serving front) holds no interpreter lock while this
// runs.  `channels`: 0 keeps what the file has, 1 / 3 converts.  Every entry of `out` is filled: pixels
// (malloc'ed, HWC, rows top to bottom; freed by tca_free_images) and geometry, or a message in `error`.
struct TcaImage {
  int32_t height, width, channels;
  uint8_t* pixels;
  char error[128];
};

int tca_decode_images(int n, const void* const* data, const uint64_t* sizes, int channels, int threads, TcaImage* out) {
  if (n < 0 || (n > 0 && (!data || !sizes || !out)) || (channels != 0 && channels != 1 && channels != 3)) return kErrBadArg;
  for (int i = 0; i < n; ++i) {
    out[i].height = out[i].width = out[i].channels = 0;
    out[i].pixels = nullptr;
    out[i].error[0] = 0;
  }
  std::atomic<int> next{0}, failed{0};
  auto work = [&] {
    for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
      tca::DecodedImage img, converted;
      tca::Error e = tca::DecodeImage(static_cast<const uint8_t*>(data[i]), static_cast<size_t>(sizes[i]), &img);
      if (e.IsOk() && channels != 0 && img.channels != channels) {
        tca::ConvertChannels(img, channels, &converted);
        img = std::move(converted);
      }
      if (e.IsOk()) {
        out[i].pixels = static_cast<uint8_t*>(std::malloc(img.pixels.size() ? img.pixels.size() : 1));
        if (out[i].pixels == nullptr) e = tca::Error("out of memory");
      }
      if (!e.IsOk()) {
        std::snprintf(out[i].error, sizeof(out[i].error), "%s", e.Message().c_str());
        ++failed;
        continue;
      }
      std::memcpy(out[i].pixels, img.pixels.data(), img.pixels.size());
      out[i].height = img.height;
      out[i].width = img.width;
      out[i].channels = img.channels;
    }
  };
  const int workers = std::max(1, std::min(threads, n));
  std::vector<std::thread> pool;
  for (int t = 1; t < workers; ++t) pool.emplace_back(work);
  work();
  for (auto& t : pool) t.join();
  return failed.load();
}

void tca_free_images(int n, TcaImage* images) {
  for (int i = 0; i < n; ++i) {
    std::free(images[i].pixels);
    images[i].pixels = nullptr;
  }
}

}  // extern "C"
