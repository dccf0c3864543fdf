// Thin C ABI over the few HIP runtime calls the Python ops need when no torch
// is around on the client side: streams, pinned host buffers, raw device
// buffers and async copies.  Errors are returned as negative codes; the text
// is available through tca_hip_rt_error().
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

namespace {
thread_local std::string g_err;
int Fail(const char* what, hipError_t e) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return -1000 - static_cast<int>(e);
}
#define RT_TRY(expr, what)                   \
  do {                                       \
    hipError_t _e = (expr);                  \
    if (_e != hipSuccess) return Fail(what, _e); \
  } while (0)
}  // namespace

extern "C" {

const char* tca_hip_rt_error() { return g_err.c_str(); }

int tca_hip_set_device(int device) {
  RT_TRY(hipSetDevice(device), "hipSetDevice");
  return 0;
}

int tca_hip_stream_create(int device, void** stream) {
  RT_TRY(hipSetDevice(device), "hipSetDevice");
  hipStream_t s;
  RT_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreateWithFlags");
  *stream = s;
  return 0;
}

int tca_hip_stream_sync(void* stream) {
  RT_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)), "hipStreamSynchronize");
  return 0;
}

// 0: everything queued on the stream has completed, 1: not yet.  (Like a synchronisation, a query that finds
// the stream drained lets the runtime release the bookkeeping of the commands behind it.)
int tca_hip_stream_query(void* stream) {
  const hipError_t e = hipStreamQuery(static_cast<hipStream_t>(stream));
  if (e == hipErrorNotReady) return 1;
  RT_TRY(e, "hipStreamQuery");
  return 0;
}

int tca_hip_stream_destroy(void* stream) {
  RT_TRY(hipStreamDestroy(static_cast<hipStream_t>(stream)), "hipStreamDestroy");
  return 0;
}

int tca_hip_host_alloc(uint64_t bytes, void** ptr) {
  RT_TRY(hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault), "hipHostMalloc");
  return 0;
}

int tca_hip_host_free(void* ptr) {
  RT_TRY(hipHostFree(ptr), "hipHostFree");
  return 0;
}

int tca_hip_malloc(int device, uint64_t bytes, void** ptr) {
  RT_TRY(hipSetDevice(device), "hipSetDevice");
  RT_TRY(hipMalloc(ptr, bytes ? bytes : 1), "hipMalloc");
  return 0;
}

int tca_hip_free(void* ptr) {
  RT_TRY(hipFree(ptr), "hipFree");
  return 0;
}

// kind: 0 = default (runtime classifies the pointers), 1 = H2D, 2 = D2H, 3 = D2D
int tca_hip_memcpy_async(void* dst, const void* src, uint64_t bytes, int kind, void* stream) {
  static const hipMemcpyKind kinds[] = {hipMemcpyDefault, hipMemcpyHostToDevice, hipMemcpyDeviceToHost,
                                        hipMemcpyDeviceToDevice};
  if (kind < 0 || kind > 3) kind = 0;
  RT_TRY(hipMemcpyAsync(dst, src, bytes, kinds[kind], static_cast<hipStream_t>(stream)), "hipMemcpyAsync");
  return 0;
}

// Captured graphs and events for callers that only hold raw handles (the native serving path replays
// graphs that PyTorch captured: `exec` is a hipGraphExec_t).
int tca_hip_graph_launch(void* exec, void* stream) {
  RT_TRY(hipGraphLaunch(static_cast<hipGraphExec_t>(exec), static_cast<hipStream_t>(stream)), "hipGraphLaunch");
  return 0;
}

int tca_hip_event_create(void** event) {
  hipEvent_t e;
  RT_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreateWithFlags");
  *event = e;
  return 0;
}

int tca_hip_event_record(void* event, void* stream) {
  RT_TRY(hipEventRecord(static_cast<hipEvent_t>(event), static_cast<hipStream_t>(stream)), "hipEventRecord");
  return 0;
}

int tca_hip_event_sync(void* event) {
  RT_TRY(hipEventSynchronize(static_cast<hipEvent_t>(event)), "hipEventSynchronize");
  return 0;
}

int tca_hip_event_destroy(void* event) {
  RT_TRY(hipEventDestroy(static_cast<hipEvent_t>(event)), "hipEventDestroy");
  return 0;
}

int tca_hip_memset_async(void* dst, int value, uint64_t bytes, void* stream) {
  RT_TRY(hipMemsetAsync(dst, value, bytes, static_cast<hipStream_t>(stream)), "hipMemsetAsync");
  return 0;
}

}  // extern "C"
