// hipshm: device shared-memory regions on HIP IPC.
//
// Replaces the reference's CUDA shared-memory module
// (`tritonclient/utils/cuda_shared_memory/__init__.py:107-429`, which drives
// cudaMalloc / cudaIpcGetMemHandle / cudaMemcpyAsync through cuda-python) with
// a native MI355X implementation:
//
//   client side   hipMalloc + hipIpcGetMemHandle  -> 64-byte handle on the wire
//   server side   hipIpcOpenMemHandle             -> zero-copy view of the region
//
// Differences by design:
//  * host<->device traffic goes through a persistent, double-buffered PINNED
//    staging ring per device (hipHostMalloc) so repeated small transfers never
//    pay the runtime's pin/unpin path and large ones overlap memcpy with DMA;
//  * reads copy only [offset, offset+n) (the reference always copies the whole
//    region, `__init__.py:265-276`);
//  * device-to-device ingestion (DLPack sources) uses hipMemcpyAsync with
//    hipMemcpyDefault, which routes over xGMI when the source lives on a peer;
//  * region sizes are validated against hipMemGetInfo (288 GB HBM3E per GPU);
//  * regions are destroyed explicitly (RAII in this file), not by Python GC.
//
// A *host-backed* mode (env TCA_HIPSHM_FAKE=1) keeps the same ABI but places
// regions in POSIX shared memory.  It exists so that protocol tests can run on
// machines without a GPU; it is never selected implicitly.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

thread_local std::string g_last_error;

int Fail(const std::string& what, hipError_t e) {
  g_last_error = what + ": " + hipGetErrorString(e);
  return -static_cast<int>(e) - 1000;
}
int FailMsg(const std::string& what, int code = -1) {
  g_last_error = what;
  return code;
}

#define HIP_TRY(expr, what)                           \
  do {                                                \
    hipError_t _e = (expr);                           \
    if (_e != hipSuccess) return Fail((what), _e);    \
  } while (0)

bool FakeMode() {
  static const bool fake = [] {
    const char* v = std::getenv("TCA_HIPSHM_FAKE");
    return v && v[0] == '1';
  }();
  return fake;
}

constexpr char kFakeMagic[8] = {'T', 'C', 'A', 'F', 'A', 'K', 'E', 0};

// ---- per-device copy engine state ------------------------------------------
struct DeviceCtx {
  hipStream_t stream = nullptr;
  void* pinned[2] = {nullptr, nullptr};
  size_t pinned_size = 0;
  hipEvent_t ev[2] = {nullptr, nullptr};
  std::mutex mu;
  uint64_t waits = 0;  // completed WaitEvent calls on this engine (under mu)
};

constexpr size_t kStageBytes = 8u << 20;       // per half of the ring, first engine of a device (bulk copies)
constexpr size_t kSmallStageBytes = 1u << 20;  // the other engines: small copies of concurrent requests
constexpr int kEnginesPerDevice = 4;
constexpr size_t kSpinBytes = 256u << 10;      // copies up to this size poll for completion instead of sleeping

// Copy engines of one device.  A serving process moves many small tensors for
// concurrent requests (BYTES tensors in device regions are read and written through
// the host): one engine would serialise them behind one mutex and one stream.
struct DevicePool {
  DeviceCtx* engine[kEnginesPerDevice] = {nullptr};
  std::atomic<unsigned> next{0};
};

std::mutex g_ctx_mu;
std::unordered_map<int, DevicePool*> g_ctx;

class DeviceGuard {
 public:
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev_) != hipSuccess) prev_ = -1;
    ok_ = hipSetDevice(device) == hipSuccess;
  }
  ~DeviceGuard() {
    if (prev_ >= 0) (void)hipSetDevice(prev_);
  }
  bool ok() const { return ok_; }

 private:
  int prev_ = -1;
  bool ok_ = false;
};

int NewEngine(int device, size_t stage_bytes, DeviceCtx** out) {
  DeviceGuard guard(device);
  if (!guard.ok()) return FailMsg("hipSetDevice(" + std::to_string(device) + ") failed");
  auto* ctx = new DeviceCtx();
  HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking), "hipStreamCreateWithFlags");
  for (int i = 0; i < 2; ++i) {
    HIP_TRY(hipHostMalloc(&ctx->pinned[i], stage_bytes, hipHostMallocDefault), "hipHostMalloc(staging)");
    HIP_TRY(hipEventCreateWithFlags(&ctx->ev[i], hipEventDisableTiming), "hipEventCreate");
  }
  ctx->pinned_size = stage_bytes;
  *out = ctx;
  return 0;
}

// A locked copy engine of `device`: the first idle one (engines are created on demand,
// so a single-threaded client only ever owns one), else the next in turn once it frees up.
class EngineLease {
 public:
  int Acquire(int device) {
    DevicePool* pool = nullptr;
    {
      std::lock_guard<std::mutex> lk(g_ctx_mu);
      auto it = g_ctx.find(device);
      if (it == g_ctx.end()) it = g_ctx.emplace(device, new DevicePool()).first;
      pool = it->second;
      for (int i = 0; i < kEnginesPerDevice; ++i) {
        if (pool->engine[i] == nullptr) {
          int rc = NewEngine(device, i == 0 ? kStageBytes : kSmallStageBytes, &pool->engine[i]);
          if (rc) return rc;
        }
        if (pool->engine[i]->mu.try_lock()) {
          ctx_ = pool->engine[i];
          return 0;
        }
      }
    }
    ctx_ = pool->engine[pool->next.fetch_add(1) % kEnginesPerDevice];
    ctx_->mu.lock();
    return 0;
  }
  ~EngineLease() {
    if (ctx_) ctx_->mu.unlock();
  }
  DeviceCtx* get() const { return ctx_; }

 private:
  DeviceCtx* ctx_ = nullptr;
};

// Waits for `ev`.  A small copy finishes within microseconds: polling sees it at once,
// where a blocking wait pays the wake-up of a sleeping thread (tens of microseconds).
int WaitEvent(hipEvent_t ev, bool small) {
  if (small) {
    const auto give_up = std::chrono::steady_clock::now() + std::chrono::microseconds(200);
    do {
      const hipError_t e = hipEventQuery(ev);
      if (e == hipSuccess) return 0;
      if (e != hipErrorNotReady) return Fail("hipEventQuery", e);
    } while (std::chrono::steady_clock::now() < give_up);
  }
  HIP_TRY(hipEventSynchronize(ev), "hipEventSynchronize");
  return 0;
}

// The last wait of a staged copy: the engine's stream is idle afterwards.  Every 64th time the stream itself is
// synchronised - the runtime keeps about a kilobyte per host-to-device hipMemcpyAsync until then, whatever
// events were waited for.
int WaitLast(DeviceCtx* ctx, hipEvent_t ev, bool small) {
  if (int rc = WaitEvent(ev, small)) return rc;
  if ((++ctx->waits & 63) == 0) HIP_TRY(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
  return 0;
}


struct Region {
  std::string name;
  void* ptr = nullptr;
  uint64_t size = 0;
  int device = 0;
  uint8_t handle[64] = {0};
  bool fake = false;
  bool opened = false;  // true: obtained through open(), false: owner
  std::string fake_key;
};

std::atomic<uint64_t> g_fake_seq{0};

int FakeCreate(Region* r) {
  char key[48];
  std::snprintf(key, sizeof(key), "/tca_hipshm_%d_%llu", static_cast<int>(getpid()),
                static_cast<unsigned long long>(g_fake_seq.fetch_add(1)));
  int fd = shm_open(key, O_RDWR | O_CREAT | O_EXCL, 0600);
  if (fd < 0) return FailMsg(std::string("shm_open(fake hipshm) failed: ") + std::strerror(errno));
  if (ftruncate(fd, static_cast<off_t>(r->size ? r->size : 1)) != 0) {
    close(fd);
    shm_unlink(key);
    return FailMsg("ftruncate(fake hipshm) failed");
  }
  void* p = mmap(nullptr, r->size ? r->size : 1, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) {
    shm_unlink(key);
    return FailMsg("mmap(fake hipshm) failed");
  }
  r->ptr = p;
  r->fake = true;
  r->fake_key = key;
  std::memset(r->handle, 0, 64);
  std::memcpy(r->handle, kFakeMagic, 8);
  std::strncpy(reinterpret_cast<char*>(r->handle) + 8, key, 55);
  return 0;
}

// Chunked H2D through the pinned ring: memcpy of chunk k+1 overlaps DMA of k.
int StagedH2D(DeviceCtx* ctx, void* dst, const void* src, size_t n) {
  const char* s = static_cast<const char*>(src);
  char* d = static_cast<char*>(dst);
  int slot = 0;
  bool used[2] = {false, false};
  const bool small = n <= kSpinBytes;
  while (n) {
    size_t chunk = n < ctx->pinned_size ? n : ctx->pinned_size;
    if (used[slot]) HIP_TRY(hipEventSynchronize(ctx->ev[slot]), "hipEventSynchronize");
    std::memcpy(ctx->pinned[slot], s, chunk);
    HIP_TRY(hipMemcpyAsync(d, ctx->pinned[slot], chunk, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync(H2D)");
    HIP_TRY(hipEventRecord(ctx->ev[slot], ctx->stream), "hipEventRecord");
    used[slot] = true;
    s += chunk;
    d += chunk;
    n -= chunk;
    slot ^= 1;
  }
  return WaitLast(ctx, ctx->ev[slot ^ 1], small);  // the event of the last chunk: the stream is in order
}

int StagedD2H(DeviceCtx* ctx, void* dst, const void* src, size_t n) {
  const char* s = static_cast<const char*>(src);
  char* d = static_cast<char*>(dst);
  // two DMA chunks in flight; drain in order
  size_t issued = 0, drained = 0;
  size_t sizes[2] = {0, 0};
  int head = 0, tail = 0, inflight = 0;
  const bool small = n <= kSpinBytes;
  while (drained < n) {
    while (inflight < 2 && issued < n) {
      size_t chunk = (n - issued) < ctx->pinned_size ? (n - issued) : ctx->pinned_size;
      HIP_TRY(hipMemcpyAsync(ctx->pinned[head], s + issued, chunk, hipMemcpyDeviceToHost, ctx->stream),
              "hipMemcpyAsync(D2H)");
      HIP_TRY(hipEventRecord(ctx->ev[head], ctx->stream), "hipEventRecord");
      sizes[head] = chunk;
      issued += chunk;
      head ^= 1;
      ++inflight;
    }
    if (int rc = WaitEvent(ctx->ev[tail], small)) return rc;
    std::memcpy(d + drained, ctx->pinned[tail], sizes[tail]);
    drained += sizes[tail];
    tail ^= 1;
    --inflight;
  }
  // device-to-host copies accumulate runtime bookkeeping like host-to-device ones (see WaitLast)
  if ((++ctx->waits & 63) == 0) HIP_TRY(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
  return 0;
}

int CheckRange(const Region* r, uint64_t offset, uint64_t n) {
  if (offset > r->size || n > r->size - offset)
    return FailMsg("access of " + std::to_string(n) + " bytes at offset " + std::to_string(offset) +
                   " is outside of region '" + r->name + "' (" + std::to_string(r->size) + " bytes)", -7);
  return 0;
}

}  // namespace

extern "C" {

const char* tca_hip_last_error() { return g_last_error.c_str(); }

int tca_hipshm_fake_mode() { return FakeMode() ? 1 : 0; }

int tca_hip_device_count(int* n) {
  if (FakeMode()) { *n = 1; return 0; }
  hipError_t e = hipGetDeviceCount(n);
  if (e != hipSuccess) { *n = 0; return Fail("hipGetDeviceCount", e); }
  return 0;
}

int tca_hip_mem_info(int device, uint64_t* free_bytes, uint64_t* total_bytes) {
  if (FakeMode()) { *free_bytes = *total_bytes = 288ull << 30; return 0; }
  DeviceGuard guard(device);
  if (!guard.ok()) return FailMsg("hipSetDevice failed");
  size_t f = 0, t = 0;
  HIP_TRY(hipMemGetInfo(&f, &t), "hipMemGetInfo");
  *free_bytes = f;
  *total_bytes = t;
  return 0;
}

int tca_hipshm_create_ex(const char* name, uint64_t byte_size, int device, int fine_grained, void** out);

int tca_hipshm_create(const char* name, uint64_t byte_size, int device, void** out) {
  return tca_hipshm_create_ex(name, byte_size, device, 0, out);
}

// fine_grained != 0: memory that stays coherent between GPUs (and processes) while kernels
// run - for flags that one GPU sets and another polls; zero-filled.  Plain regions are the
// default: their contents are exchanged at kernel boundaries.
int tca_hipshm_create_ex(const char* name, uint64_t byte_size, int device, int fine_grained, void** out) {
  if (!name || !out) return FailMsg("invalid argument");
  auto* r = new Region();
  r->name = name;
  r->size = byte_size;
  r->device = device;
  if (FakeMode()) {
    int rc = FakeCreate(r);
    if (rc != 0) { delete r; return rc; }
    *out = r;
    return 0;
  }
  DeviceGuard guard(device);
  if (!guard.ok()) { delete r; return FailMsg("unable to select device " + std::to_string(device)); }
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && byte_size > free_b) {
    delete r;
    return FailMsg("region of " + std::to_string(byte_size) + " bytes does not fit in the " +
                   std::to_string(free_b) + " bytes free on device " + std::to_string(device));
  }
  hipError_t e;
  if (fine_grained) {
    e = hipExtMallocWithFlags(&r->ptr, byte_size ? byte_size : 1, hipDeviceMallocFinegrained);
    if (e == hipSuccess) {
      e = hipMemset(r->ptr, 0, byte_size ? byte_size : 1);
      if (e == hipSuccess) e = hipDeviceSynchronize();
      if (e != hipSuccess) (void)hipFree(r->ptr);
    }
  } else {
    e = hipMalloc(&r->ptr, byte_size ? byte_size : 1);
  }
  if (e != hipSuccess) { delete r; return Fail(fine_grained ? "hipExtMallocWithFlags" : "hipMalloc", e); }
  hipIpcMemHandle_t h;
  e = hipIpcGetMemHandle(&h, r->ptr);
  if (e != hipSuccess) {
    (void)hipFree(r->ptr);
    delete r;
    return Fail("hipIpcGetMemHandle", e);
  }
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "HIP IPC handle is expected to be 64 bytes on the wire");
  std::memcpy(r->handle, &h, 64);
  *out = r;
  return 0;
}

int tca_hipshm_info(void* handle, void** dev_ptr, uint8_t* raw_handle, int* device, uint64_t* byte_size, int* is_host) {
  auto* r = static_cast<Region*>(handle);
  if (!r) return FailMsg("null region");
  if (dev_ptr) *dev_ptr = r->ptr;
  if (raw_handle) std::memcpy(raw_handle, r->handle, 64);
  if (device) *device = r->device;
  if (byte_size) *byte_size = r->size;
  if (is_host) *is_host = r->fake ? 1 : 0;
  return 0;
}

// Host -> region.
int tca_hipshm_write(void* handle, uint64_t offset, const void* src, uint64_t n) {
  auto* r = static_cast<Region*>(handle);
  if (!r) return FailMsg("null region");
  int rc = CheckRange(r, offset, n);
  if (rc) return rc;
  if (n == 0) return 0;
  if (r->fake) { std::memcpy(static_cast<char*>(r->ptr) + offset, src, n); return 0; }
  DeviceGuard guard(r->device);
  EngineLease lease;
  rc = lease.Acquire(r->device);
  if (rc) return rc;
  return StagedH2D(lease.get(), static_cast<char*>(r->ptr) + offset, src, n);
}

// Region -> host; only the requested range moves.
int tca_hipshm_read(void* handle, uint64_t offset, void* dst, uint64_t n) {
  auto* r = static_cast<Region*>(handle);
  if (!r) return FailMsg("null region");
  int rc = CheckRange(r, offset, n);
  if (rc) return rc;
  if (n == 0) return 0;
  if (r->fake) { std::memcpy(dst, static_cast<char*>(r->ptr) + offset, n); return 0; }
  DeviceGuard guard(r->device);
  EngineLease lease;
  rc = lease.Acquire(r->device);
  if (rc) return rc;
  return StagedD2H(lease.get(), dst, static_cast<char*>(r->ptr) + offset, n);
}

// Several small region -> host reads (the input tensors of one request) as ONE trip:
// every copy is enqueued, then one wait.  Falls back to one read at a time when the
// ranges are on different devices, host-backed, or larger than a staging buffer.
int tca_hipshm_read_many(int count, void* const* handles, const uint64_t* offsets, void* const* dsts,
                         const uint64_t* sizes) {
  if (count <= 0) return 0;
  uint64_t total = 0;
  bool batched = true;
  for (int i = 0; i < count; ++i) {
    auto* r = static_cast<Region*>(handles[i]);
    if (!r) return FailMsg("null region");
    int rc = CheckRange(r, offsets[i], sizes[i]);
    if (rc) return rc;
    if (r->fake || r->device != static_cast<Region*>(handles[0])->device) batched = false;
    total += (sizes[i] + 63) & ~63ull;
  }
  if (!batched || total > kSmallStageBytes) {
    for (int i = 0; i < count; ++i) {
      int rc = tca_hipshm_read(handles[i], offsets[i], dsts[i], sizes[i]);
      if (rc) return rc;
    }
    return 0;
  }
  const int device = static_cast<Region*>(handles[0])->device;
  DeviceGuard guard(device);
  EngineLease lease;
  int rc = lease.Acquire(device);
  if (rc) return rc;
  DeviceCtx* ctx = lease.get();
  char* stage = static_cast<char*>(ctx->pinned[0]);
  uint64_t cursor = 0;
  for (int i = 0; i < count; ++i) {
    if (sizes[i] == 0) continue;
    auto* r = static_cast<Region*>(handles[i]);
    HIP_TRY(hipMemcpyAsync(stage + cursor, static_cast<char*>(r->ptr) + offsets[i], sizes[i], hipMemcpyDeviceToHost,
                           ctx->stream), "hipMemcpyAsync(D2H)");
    cursor += (sizes[i] + 63) & ~63ull;
  }
  HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream), "hipEventRecord");
  rc = WaitLast(ctx, ctx->ev[0], total <= kSpinBytes);
  if (rc) return rc;
  cursor = 0;
  for (int i = 0; i < count; ++i) {
    std::memcpy(dsts[i], stage + cursor, sizes[i]);
    cursor += (sizes[i] + 63) & ~63ull;
  }
  return 0;
}

// The counterpart for the output tensors of one request: host -> regions, one wait.
int tca_hipshm_write_many(int count, void* const* handles, const uint64_t* offsets, const void* const* srcs,
                          const uint64_t* sizes) {
  if (count <= 0) return 0;
  uint64_t total = 0;
  bool batched = true;
  for (int i = 0; i < count; ++i) {
    auto* r = static_cast<Region*>(handles[i]);
    if (!r) return FailMsg("null region");
    int rc = CheckRange(r, offsets[i], sizes[i]);
    if (rc) return rc;
    if (r->fake || r->device != static_cast<Region*>(handles[0])->device) batched = false;
    total += (sizes[i] + 63) & ~63ull;
  }
  if (!batched || total > kSmallStageBytes) {
    for (int i = 0; i < count; ++i) {
      int rc = tca_hipshm_write(handles[i], offsets[i], srcs[i], sizes[i]);
      if (rc) return rc;
    }
    return 0;
  }
  const int device = static_cast<Region*>(handles[0])->device;
  DeviceGuard guard(device);
  EngineLease lease;
  int rc = lease.Acquire(device);
  if (rc) return rc;
  DeviceCtx* ctx = lease.get();
  char* stage = static_cast<char*>(ctx->pinned[0]);
  uint64_t cursor = 0;
  for (int i = 0; i < count; ++i) {
    if (sizes[i] == 0) continue;
    auto* r = static_cast<Region*>(handles[i]);
    std::memcpy(stage + cursor, srcs[i], sizes[i]);
    HIP_TRY(hipMemcpyAsync(static_cast<char*>(r->ptr) + offsets[i], stage + cursor, sizes[i], hipMemcpyHostToDevice,
                           ctx->stream), "hipMemcpyAsync(H2D)");
    cursor += (sizes[i] + 63) & ~63ull;
  }
  HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream), "hipEventRecord");
  return WaitLast(ctx, ctx->ev[0], total <= kSpinBytes);
}

// Any address the runtime can classify (device, pinned host, peer device) ->
// region.  `src_device` < 0 means pageable/unknown host memory.
int tca_hipshm_copy_from(void* handle, uint64_t offset, const void* src, uint64_t n, int src_device) {
  auto* r = static_cast<Region*>(handle);
  if (!r) return FailMsg("null region");
  int rc = CheckRange(r, offset, n);
  if (rc) return rc;
  if (n == 0) return 0;
  if (r->fake) { std::memcpy(static_cast<char*>(r->ptr) + offset, src, n); return 0; }
  if (src_device < 0) return tca_hipshm_write(handle, offset, src, n);
  DeviceGuard guard(r->device);
  EngineLease lease;
  rc = lease.Acquire(r->device);
  if (rc) return rc;
  DeviceCtx* ctx = lease.get();
  if (src_device != r->device) {
    int can = 0;
    (void)hipDeviceCanAccessPeer(&can, r->device, src_device);
    if (can) (void)hipDeviceEnablePeerAccess(src_device, 0);  // already-enabled is fine
    (void)hipGetLastError();
  }
  HIP_TRY(hipMemcpyAsync(static_cast<char*>(r->ptr) + offset, src, n, hipMemcpyDefault, ctx->stream),
          "hipMemcpyAsync(D2D)");
  HIP_TRY(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
  return 0;
}

// Same as tca_hipshm_copy_from for device sources, but only ENQUEUES the copy on the
// caller's stream (created on the SOURCE device): several of these towards different
// peer GPUs run concurrently, one xGMI link each.  The caller synchronises the stream.
int tca_hipshm_copy_from_async(void* handle, uint64_t offset, const void* src, uint64_t n, int src_device,
                               void* stream) {
  auto* r = static_cast<Region*>(handle);
  if (!r) return FailMsg("null region");
  int rc = CheckRange(r, offset, n);
  if (rc) return rc;
  if (n == 0) return 0;
  if (r->fake) { std::memcpy(static_cast<char*>(r->ptr) + offset, src, n); return 0; }
  if (src_device < 0) return FailMsg("asynchronous copies need a device (or pinned) source");
  DeviceGuard guard(src_device);
  if (src_device != r->device) {
    int can = 0;
    (void)hipDeviceCanAccessPeer(&can, src_device, r->device);
    if (can) (void)hipDeviceEnablePeerAccess(r->device, 0);  // already-enabled is fine
    (void)hipGetLastError();
  }
  HIP_TRY(hipMemcpyAsync(static_cast<char*>(r->ptr) + offset, src, n, hipMemcpyDefault, static_cast<hipStream_t>(stream)),
          "hipMemcpyAsync(peer)");
  return 0;
}

int tca_hipshm_destroy(void* handle) {
  auto* r = static_cast<Region*>(handle);
  if (!r) return FailMsg("null region");
  int rc = 0;
  if (r->fake) {
    munmap(r->ptr, r->size ? r->size : 1);
    if (!r->opened) shm_unlink(r->fake_key.c_str());
  } else if (r->opened) {
    DeviceGuard guard(r->device);
    hipError_t e = hipIpcCloseMemHandle(r->ptr);
    if (e != hipSuccess) rc = Fail("hipIpcCloseMemHandle", e);
  } else {
    DeviceGuard guard(r->device);
    hipError_t e = hipFree(r->ptr);
    if (e != hipSuccess) rc = Fail("hipFree", e);
  }
  delete r;
  return rc;
}

// Server side: map a region exported by ANOTHER process.
int tca_hipshm_open(const char* name, const uint8_t* raw_handle, uint64_t byte_size, int device, void** out) {
  if (!raw_handle || !out) return FailMsg("invalid argument");
  auto* r = new Region();
  r->name = name ? name : "";
  r->size = byte_size;
  r->device = device;
  r->opened = true;
  std::memcpy(r->handle, raw_handle, 64);
  if (std::memcmp(raw_handle, kFakeMagic, 8) == 0) {
    if (!FakeMode()) { delete r; return FailMsg("host-backed hipshm handle received but TCA_HIPSHM_FAKE is not set"); }
    r->fake = true;
    r->fake_key.assign(reinterpret_cast<const char*>(raw_handle) + 8);
    int fd = shm_open(r->fake_key.c_str(), O_RDWR, 0600);
    if (fd < 0) { delete r; return FailMsg("unable to open host-backed hipshm region"); }
    struct stat st;
    if (fstat(fd, &st) != 0 || static_cast<uint64_t>(st.st_size) < byte_size) {
      close(fd);
      delete r;
      return FailMsg("registered byte_size exceeds the size of the exported region");
    }
    void* p = mmap(nullptr, byte_size ? byte_size : 1, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete r; return FailMsg("mmap of host-backed hipshm region failed"); }
    r->ptr = p;
    *out = r;
    return 0;
  }
  if (FakeMode()) { delete r; return FailMsg("TCA_HIPSHM_FAKE is set but a real HIP IPC handle was received"); }
  DeviceGuard guard(device);
  if (!guard.ok()) { delete r; return FailMsg("unable to select device " + std::to_string(device)); }
  hipIpcMemHandle_t h;
  std::memcpy(&h, raw_handle, 64);
  hipError_t e = hipIpcOpenMemHandle(&r->ptr, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) { delete r; return Fail("hipIpcOpenMemHandle", e); }
  *out = r;
  return 0;
}

int tca_hip_device_synchronize(int device) {
  if (FakeMode()) return 0;
  DeviceGuard guard(device);
  HIP_TRY(hipDeviceSynchronize(), "hipDeviceSynchronize");
  return 0;
}

}  // extern "C"
