// Wire-tensor staging for the serving path: tensors that arrived inside request
// messages (pageable host memory) -> rows of the static device buffer a captured
// graph reads.
//
//   host:   one memcpy per tensor into a small ring of page-locked buffers (a copy
//           from pageable memory would be staged by the runtime in small synchronous
//           chunks on the calling thread); the ring is deliberately small so it stays
//           in the CPU's last-level cache, and an entry is recycled as soon as the
//           H2D copy that read it is done
//   copy stream:    asynchronous H2D copies into one of two device staging buffers, so
//           the upload of the next batch overlaps the forward of the current one
//           instead of queueing behind it on the compute stream
//   compute stream: waits for the upload's event, then moves the rows into place with
//           device-to-device copies (microseconds), merging neighbouring row ranges
//
// Everything runs inside ONE call without the interpreter lock (the Python batcher
// thread only decides what goes where), which is what this file is for: per tensor the
// Python form of this loop held the lock for tens of microseconds of bookkeeping.
// All calls on one stager must come from one thread at a time.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_err;

int Fail(const char* what, hipError_t e) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return -1000 - static_cast<int>(e);
}
#define ST_TRY(expr, what)                       \
  do {                                           \
    hipError_t _e = (expr);                      \
    if (_e != hipSuccess) return Fail(what, _e); \
  } while (0)

// Splits one large host copy over a few persistent helper threads (plus the caller): one core
// moves 6-7 GB/s into page-locked memory, which made this copy the bottleneck of the wire path
// (a batch of sixteen 1.2 MB requests took 2.9 ms); four cores keep up with the H2D engine.
class CopyPool {
 public:
  explicit CopyPool(int helpers) {
    for (int i = 0; i < helpers; ++i) threads_.emplace_back([this] { Run(); });
  }
  ~CopyPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
      ++generation_;
    }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
  }

  void Copy(uint8_t* dst, const uint8_t* src, size_t n) {
    if (threads_.empty() || n < 2 * kChunk) {
      std::memcpy(dst, src, n);
      return;
    }
    {
      std::lock_guard<std::mutex> lk(mu_);
      dst_ = dst;
      src_ = src;
      total_ = n;
      next_.store(0, std::memory_order_relaxed);
      working_ = static_cast<int>(threads_.size());
      ++generation_;
    }
    cv_.notify_all();
    Drain();
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [&] { return working_ == 0; });
  }

 private:
  static constexpr size_t kChunk = 128 * 1024;

  void Drain() {
    for (;;) {
      const size_t off = next_.fetch_add(kChunk, std::memory_order_relaxed);
      if (off >= total_) return;
      std::memcpy(dst_ + off, src_ + off, std::min(kChunk, total_ - off));
    }
  }

  void Run() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return generation_ != seen; });
        seen = generation_;
        if (stop_) return;
      }
      Drain();
      {
        std::lock_guard<std::mutex> lk(mu_);
        --working_;
      }
      done_cv_.notify_one();
    }
  }

  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  uint8_t* dst_ = nullptr;
  const uint8_t* src_ = nullptr;
  size_t total_ = 0;
  std::atomic<size_t> next_{0};
  int working_ = 0;
  uint64_t generation_ = 0;
  bool stop_ = false;
};

int HelperThreads() {
  if (const char* env = std::getenv("TCA_STAGER_THREADS")) return std::max(0, std::min(15, std::atoi(env) - 1));
  const unsigned cores = std::thread::hardware_concurrency();
  return cores >= 16 ? 3 : (cores >= 8 ? 1 : 0);
}

struct RingEntry {
  uint8_t* host = nullptr;
  hipEvent_t read = nullptr;  // after the last H2D copy that read this entry
  bool in_use = false;        // `read` has been recorded since the entry was last reset
  uint64_t used = 0;
};

struct Stager {
  int device = 0;
  hipStream_t copy_stream = nullptr;
  std::vector<RingEntry> ring;
  uint64_t entry_bytes = 0;
  size_t cursor = 0;
  uint8_t* dev[2] = {nullptr, nullptr};
  uint64_t dev_bytes = 0;
  hipEvent_t dev_free[2] = {nullptr, nullptr};  // after the compute stream's last read of each staging buffer
  bool dev_used[2] = {false, false};
  hipEvent_t landed = nullptr;
  int next = 0;
  uint64_t staged_bytes = 0, staged_tensors = 0, ring_waits = 0, stage_calls = 0;
  CopyPool pool{HelperThreads()};

  int FreeRing() {
    for (auto& e : ring) {
      if (e.read) hipEventDestroy(e.read);
      if (e.host) hipHostFree(e.host);
    }
    ring.clear();
    return 0;
  }

  int AllocRing(uint64_t bytes, int entries) {
    // nothing may still read the old buffers
    ST_TRY(hipStreamSynchronize(copy_stream), "hipStreamSynchronize");
    FreeRing();
    ring.resize(static_cast<size_t>(entries));
    for (auto& e : ring) {
      ST_TRY(hipHostMalloc(reinterpret_cast<void**>(&e.host), bytes, hipHostMallocDefault), "hipHostMalloc");
      ST_TRY(hipEventCreateWithFlags(&e.read, hipEventDisableTiming), "hipEventCreate");
    }
    entry_bytes = bytes;
    cursor = 0;
    return 0;
  }

  int AllocDevice(uint64_t bytes) {
    ST_TRY(hipDeviceSynchronize(), "hipDeviceSynchronize");
    for (int i = 0; i < 2; ++i) {
      if (dev[i]) hipFree(dev[i]);
      dev[i] = nullptr;
      ST_TRY(hipMalloc(reinterpret_cast<void**>(&dev[i]), bytes), "hipMalloc");
      dev_used[i] = false;
    }
    dev_bytes = bytes;
    return 0;
  }
};

}  // namespace

extern "C" {

const char* tca_stager_error() { return g_err.c_str(); }

// `ring_entry_bytes` / `ring_entries`: page-locked ring (grown when a tensor does not fit an entry);
// `dev_bytes`: size of each of the two device staging buffers (>= the largest batch staged at once).
int tca_stager_create(int device, uint64_t ring_entry_bytes, int ring_entries, uint64_t dev_bytes, void** out) {
  if (out == nullptr || ring_entries < 2 || ring_entries > 64 || ring_entry_bytes == 0 || dev_bytes == 0) {
    g_err = "tca_stager_create: bad arguments";
    return -1;
  }
  ST_TRY(hipSetDevice(device), "hipSetDevice");
  Stager* s = new Stager();
  s->device = device;
  int rc = 0;
  do {
    hipError_t e = hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { rc = Fail("hipStreamCreateWithFlags", e); break; }
    if ((rc = s->AllocRing(ring_entry_bytes, ring_entries)) != 0) break;
    if ((rc = s->AllocDevice(dev_bytes)) != 0) break;
    for (int i = 0; i < 2 && rc == 0; ++i) {
      e = hipEventCreateWithFlags(&s->dev_free[i], hipEventDisableTiming);
      if (e != hipSuccess) rc = Fail("hipEventCreate", e);
    }
    if (rc != 0) break;
    e = hipEventCreateWithFlags(&s->landed, hipEventDisableTiming);
    if (e != hipSuccess) rc = Fail("hipEventCreate", e);
  } while (false);
  if (rc != 0) {
    delete s;  // leaks what was created on this rare path rather than risking a double free
    return rc;
  }
  *out = s;
  return 0;
}

// Stages `count` host tensors: tensor i (`srcs[i]`, `sizes[i]` bytes) lands at `dst + offsets[i]`.
// Offsets must lie inside [0, dev_bytes) and not overlap; the copies are ordered before later work
// on `compute_stream`.  Returns 0 or a negative error (tca_stager_error()).
int tca_stager_stage(void* handle, int count, const void* const* srcs, const uint64_t* sizes, const uint64_t* offsets,
                     void* dst, void* compute_stream) {
  Stager* s = static_cast<Stager*>(handle);
  if (s == nullptr || count < 0 || (count > 0 && (srcs == nullptr || sizes == nullptr || offsets == nullptr || dst == nullptr))) {
    g_err = "tca_stager_stage: bad arguments";
    return -1;
  }
  if (count == 0) return 0;
  ST_TRY(hipSetDevice(s->device), "hipSetDevice");
  hipStream_t compute = static_cast<hipStream_t>(compute_stream);
  uint64_t need = 0, extent = 0;
  for (int i = 0; i < count; ++i) {
    need = std::max(need, sizes[i]);
    if (offsets[i] + sizes[i] < offsets[i]) {
      g_err = "tca_stager_stage: offset overflow";
      return -1;
    }
    extent = std::max(extent, offsets[i] + sizes[i]);
  }
  if (need > s->entry_bytes) {
    const int rc = s->AllocRing(need, static_cast<int>(s->ring.size()));
    if (rc != 0) return rc;
  }
  if (extent > s->dev_bytes) {
    const int rc = s->AllocDevice(extent);
    if (rc != 0) return rc;
  }
  // The runtime keeps about a kilobyte per host-to-device hipMemcpyAsync until the stream itself is synchronised
  // (the events below do not release it): drain the copy stream now and then - the previous upload has long
  // landed, its batch has been running since.
  if ((++s->stage_calls & 31) == 0) ST_TRY(hipStreamSynchronize(s->copy_stream), "hipStreamSynchronize");
  const int slot = s->next;
  s->next ^= 1;
  uint8_t* dev = s->dev[slot];
  // GPU-side wait: the batch that used this staging buffer two uploads ago must have been read
  if (s->dev_used[slot]) ST_TRY(hipStreamWaitEvent(s->copy_stream, s->dev_free[slot], 0), "hipStreamWaitEvent");
  for (int i = 0; i < count; ++i) {
    if (sizes[i] == 0) continue;
    RingEntry* e = &s->ring[s->cursor];
    if (e->in_use && e->used + sizes[i] > s->entry_bytes) {
      // full: move on and wait for the next entry's last reader
      s->cursor = (s->cursor + 1) % s->ring.size();
      e = &s->ring[s->cursor];
      if (e->in_use) {
        if (hipEventQuery(e->read) != hipSuccess) {
          ++s->ring_waits;
          ST_TRY(hipEventSynchronize(e->read), "hipEventSynchronize");
        }
        e->in_use = false;
      }
      e->used = 0;
    } else if (!e->in_use) {
      e->used = 0;
    }
    uint8_t* host = e->host + e->used;
    e->used = (e->used + sizes[i] + 255) & ~uint64_t(255);
    s->pool.Copy(host, static_cast<const uint8_t*>(srcs[i]), sizes[i]);
    ST_TRY(hipMemcpyAsync(dev + offsets[i], host, sizes[i], hipMemcpyHostToDevice, s->copy_stream), "hipMemcpyAsync(H2D)");
    ST_TRY(hipEventRecord(e->read, s->copy_stream), "hipEventRecord");  // the latest reader of this entry
    e->in_use = true;
    s->staged_bytes += sizes[i];
    ++s->staged_tensors;
  }
  ST_TRY(hipEventRecord(s->landed, s->copy_stream), "hipEventRecord");
  ST_TRY(hipStreamWaitEvent(compute, s->landed, 0), "hipStreamWaitEvent");
  // merge neighbouring ranges: a batch made only of wire requests is one copy
  std::vector<std::pair<uint64_t, uint64_t>> runs;
  runs.reserve(static_cast<size_t>(count));
  for (int i = 0; i < count; ++i)
    if (sizes[i]) runs.emplace_back(offsets[i], offsets[i] + sizes[i]);
  std::sort(runs.begin(), runs.end());
  size_t k = 0;
  while (k < runs.size()) {
    uint64_t begin = runs[k].first, end = runs[k].second;
    while (++k < runs.size() && runs[k].first == end) end = runs[k].second;
    ST_TRY(hipMemcpyAsync(static_cast<uint8_t*>(dst) + begin, dev + begin, end - begin, hipMemcpyDeviceToDevice, compute),
           "hipMemcpyAsync(D2D)");
  }
  ST_TRY(hipEventRecord(s->dev_free[slot], compute), "hipEventRecord");
  s->dev_used[slot] = true;
  return 0;
}

// Counters: bytes and tensors staged, times the host had to wait for a ring entry.
int tca_stager_stats(void* handle, uint64_t* bytes, uint64_t* tensors, uint64_t* ring_waits) {
  Stager* s = static_cast<Stager*>(handle);
  if (s == nullptr) return -1;
  if (bytes) *bytes = s->staged_bytes;
  if (tensors) *tensors = s->staged_tensors;
  if (ring_waits) *ring_waits = s->ring_waits;
  return 0;
}

int tca_stager_destroy(void* handle) {
  Stager* s = static_cast<Stager*>(handle);
  if (s == nullptr) return 0;
  hipSetDevice(s->device);
  hipDeviceSynchronize();
  s->FreeRing();
  for (int i = 0; i < 2; ++i) {
    if (s->dev[i]) hipFree(s->dev[i]);
    if (s->dev_free[i]) hipEventDestroy(s->dev_free[i]);
  }
  if (s->landed) hipEventDestroy(s->landed);
  if (s->copy_stream) hipStreamDestroy(s->copy_stream);
  delete s;
  return 0;
}

}  // extern "C"
