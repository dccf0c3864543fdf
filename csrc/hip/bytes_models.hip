// BYTES tensors served without leaving the GPU: the string models of the test server
// (decimal add / subtract, identity) for requests whose tensors sit in device shared-memory
// regions.  A batch of requests is three launches, whatever its size:
//
//   BytesIndexBatchKernel   one workgroup per input stream: the stream is staged into LDS with
//                           coalesced 16-byte loads, one lane walks the length prefixes there
//                           (the wire form: <u32 little-endian length><bytes> per element,
//                           reference `tritonclient/utils/__init__.py:219-276`), every element's
//                           payload offset and length land in the workspace
//   DecimalAddSubKernel     one thread per element: parse both operands, add / subtract,
//                           format the results as decimal text
//   BytesPackBatchKernel    one workgroup per output stream: exclusive scan of 4 + length in
//                           LDS, then one wavefront per element writes prefix and payload into
//                           the caller's output region; the last bytes of the output's range
//                           get an index trailer {magic, count, offsets[count], end} when the stream
//                           leaves them free (a reader that finds it, and finds it consistent
//                           with the prefixes, fetches exactly the stream and skips the walk)
//
// Request descriptors and status words live in page-locked host memory that the kernels
// access directly: no copies besides the tensors' own bytes.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <new>
#include <string>

namespace {

constexpr int kThreads = 256;
constexpr uint32_t kMaxStream = 48 * 1024;  // bytes of one input stream (staged in LDS whole)
constexpr int kTextSlot = 24;               // bytes per formatted result (int64 fits in 20 + sign)
constexpr uint32_t kTrailerMagic = 0x58444942u;  // "BIDX"

enum : uint32_t { kOk = 0, kMalformed = 1, kWrongCount = 2, kOutputTooSmall = 3, kNotANumber = 4 };

struct StreamIn {  // one BYTES input tensor of one request
  const uint8_t* data;
  uint64_t bytes;
};

struct StreamOut {  // one BYTES output tensor of one request
  uint8_t* data;
  uint64_t capacity;
};

struct RequestStatus {
  uint32_t code;         // first failure of the request (kOk: none)
  uint32_t reserved;
  uint64_t out_bytes[2];
};

thread_local std::string g_err;

int Fail(const char* what, hipError_t e) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return -1000 - static_cast<int>(e);
}

// the status words are host memory: plain stores (any failing code will do when several race)
__device__ void Raise(RequestStatus* st, uint32_t code) {
  volatile uint32_t* word = &st->code;
  if (*word == kOk) *word = code;
}

// offsets / lens: [streams][count]
__global__ __launch_bounds__(kThreads) void BytesIndexBatchKernel(const StreamIn* __restrict__ in, int inputs_per_request,
                                                                  uint32_t count, uint32_t* __restrict__ offsets,
                                                                  uint32_t* __restrict__ lens,
                                                                  RequestStatus* __restrict__ status) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[kMaxStream + 16];
  const int s = blockIdx.x;
  const uint8_t* __restrict__ src = in[s].data;
  const uint32_t size = static_cast<uint32_t>(in[s].bytes);
  RequestStatus* st = status + s / inputs_per_request;
  const bool aligned = (reinterpret_cast<uintptr_t>(src) & 15) == 0;
  for (uint32_t off = threadIdx.x * 16u; off < size; off += kThreads * 16u) {
    if (aligned && off + 16 <= size) {
      *reinterpret_cast<uint4*>(lds + off) = *reinterpret_cast<const uint4*>(src + off);
    } else {
      for (uint32_t j = off; j < size && j < off + 16; ++j) lds[j] = src[j];
    }
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  uint32_t cur = 0;
  for (uint32_t e = 0; e < count; ++e) {
    if (cur + 4 > size) {
      Raise(st, cur == size ? kWrongCount : kMalformed);
      return;
    }
    const uint32_t* words = reinterpret_cast<const uint32_t*>(lds + (cur & ~3u));
    const uint32_t lo = words[0], hi = words[1], shift = (cur & 3u) * 8u;
    const uint32_t len = shift ? (lo >> shift) | (hi << (32u - shift)) : lo;
    if (len > size - (cur + 4)) {
      Raise(st, kMalformed);
      return;
    }
    offsets[static_cast<size_t>(s) * count + e] = cur + 4;
    lens[static_cast<size_t>(s) * count + e] = len;
    cur += 4 + len;
  }
  if (cur != size) Raise(st, kWrongCount);
}

__device__ bool ParseDecimal(const uint8_t* p, uint32_t len, long long* out) {
  // what Python's int() takes from these clients: optional blanks, optional sign, digits
  while (len && (*p == ' ' || *p == '\t' || *p == '\n')) { ++p; --len; }
  while (len && (p[len - 1] == ' ' || p[len - 1] == '\t' || p[len - 1] == '\n' || p[len - 1] == 0)) --len;
  bool neg = false;
  if (len && (*p == '-' || *p == '+')) { neg = *p == '-'; ++p; --len; }
  if (len == 0 || len > 18) return false;
  long long v = 0;
  for (uint32_t i = 0; i < len; ++i) {
    const uint32_t d = static_cast<uint32_t>(p[i]) - '0';
    if (d > 9) return false;
    v = v * 10 + d;
  }
  *out = neg ? -v : v;
  return true;
}

__device__ uint32_t FormatDecimal(long long v, uint8_t* dst) {
  char tmp[kTextSlot];
  int n = 0;
  unsigned long long u = v < 0 ? 0ull - static_cast<unsigned long long>(v) : static_cast<unsigned long long>(v);
  do {
    tmp[n++] = static_cast<char>('0' + u % 10);
    u /= 10;
  } while (u);
  uint32_t w = 0;
  if (v < 0) dst[w++] = '-';
  while (n) dst[w++] = static_cast<uint8_t>(tmp[--n]);
  return w;
}

// text / text_lens: [requests][2][count] slots of kTextSlot bytes
__global__ __launch_bounds__(kThreads) void DecimalAddSubKernel(const StreamIn* __restrict__ in, uint32_t count,
                                                                uint32_t total, const uint32_t* __restrict__ offsets,
                                                                const uint32_t* __restrict__ lens,
                                                                uint8_t* __restrict__ text, uint32_t* __restrict__ text_lens,
                                                                RequestStatus* __restrict__ status) {
  const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= total) return;
  const uint32_t r = i / count, e = i - r * count;
  RequestStatus* st = status + r;
  if (st->code != kOk) return;
  const size_t a_at = static_cast<size_t>(2 * r) * count + e, b_at = a_at + count;
  long long a, b;
  if (!ParseDecimal(in[2 * r].data + offsets[a_at], lens[a_at], &a) ||
      !ParseDecimal(in[2 * r + 1].data + offsets[b_at], lens[b_at], &b)) {
    Raise(st, kNotANumber);
    return;
  }
  text_lens[a_at] = FormatDecimal(a + b, text + a_at * kTextSlot);
  text_lens[b_at] = FormatDecimal(a - b, text + b_at * kTextSlot);
}

// Output stream o (= request o / outputs_per_request, output o % outputs_per_request): element e is
// `src_lens[o*count + e]` bytes at `src_base(o) + src_offsets[o*count + e]`.  `slots` != 0: the elements are
// text slots (base = text, offset = index * kTextSlot); otherwise they are payloads of input stream o.
__global__ __launch_bounds__(kThreads) void BytesPackBatchKernel(const StreamOut* __restrict__ out, int outputs_per_request,
                                                                 uint32_t count, const StreamIn* __restrict__ in,
                                                                 const uint8_t* __restrict__ text,
                                                                 const uint32_t* __restrict__ src_offsets,
                                                                 const uint32_t* __restrict__ src_lens, int slots,
                                                                 RequestStatus* __restrict__ status) {
  extern __shared__ uint32_t starts[];  // [count + 1] byte position of every element's prefix
  const int o = blockIdx.x;
  const int r = o / outputs_per_request;
  RequestStatus* st = status + r;
  if (st->code != kOk) return;
  const uint32_t* __restrict__ lens = src_lens + static_cast<size_t>(o) * count;
  // exclusive scan of (4 + len): one wavefront, 64 elements a round (element counts are small next to byte volume)
  if (threadIdx.x < 64) {
    uint32_t carry = 0;
    for (uint32_t base = 0; base < count; base += 64) {
      const uint32_t e = base + threadIdx.x;
      const uint32_t v = e < count ? 4u + lens[e] : 0u;
      uint32_t incl = v;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if (static_cast<int>(threadIdx.x) >= d) incl += up;
      }
      if (e < count) starts[e] = carry + incl - v;
      carry += __shfl(incl, 63, 64);
    }
    if (threadIdx.x == 0) starts[count] = carry;
  }
  __syncthreads();
  const uint32_t total = starts[count];
  uint8_t* __restrict__ dst = out[o].data;
  if (total > out[o].capacity) {
    if (threadIdx.x == 0) Raise(st, kOutputTooSmall);
    return;
  }
  const uint8_t* __restrict__ base = slots ? text + static_cast<size_t>(o) * count * kTextSlot : in[o].data;
  const int lane = threadIdx.x & 63;
  for (uint32_t e = threadIdx.x >> 6; e < count; e += kThreads / 64) {
    const uint32_t len = lens[e];
    const uint8_t* __restrict__ src = base + (slots ? e * kTextSlot : src_offsets[static_cast<size_t>(o) * count + e]);
    uint8_t* __restrict__ at = dst + starts[e];
    if (lane < 4) at[lane] = static_cast<uint8_t>(len >> (8 * lane));
    for (uint32_t j = lane; j < len; j += 64) at[4 + j] = src[j];
  }
  // index trailer {magic, count, start of every payload, end of the stream} in the last bytes of the range the
  // request gave this output, if the stream leaves them free
  const uint64_t need = 12 + 4ull * count;
  const uintptr_t range_end = reinterpret_cast<uintptr_t>(dst) + out[o].capacity;
  const uintptr_t at = (range_end - need) & ~static_cast<uintptr_t>(3);
  const bool room = out[o].capacity >= need + 3 && at >= reinterpret_cast<uintptr_t>(dst) + total;
  if (room) {
    uint32_t* t = reinterpret_cast<uint32_t*>(at);
    if (threadIdx.x == 0) {
      t[0] = kTrailerMagic;
      t[1] = count;
    }
    for (uint32_t e = threadIdx.x; e < count; e += kThreads) t[2 + e] = starts[e] + 4;
    if (threadIdx.x == 0) t[2 + count] = total;
  }
  if (threadIdx.x == 0) st->out_bytes[o % outputs_per_request] = total;
}

// Everything one batch needs, allocated once: descriptors and status in page-locked host memory
// (mapped for the device), scratch on the device.
struct Pipe {
  int device = 0;
  int max_requests = 0;
  uint32_t max_count = 0;
  StreamIn* in = nullptr;        // [max_requests * 2]
  StreamOut* out = nullptr;      // [max_requests * 2]
  RequestStatus* status = nullptr;
  uint32_t* offsets = nullptr;   // [max_requests * 2 * max_count]
  uint32_t* lens = nullptr;
  uint8_t* text = nullptr;
  uint32_t* text_lens = nullptr;
  hipEvent_t done = nullptr;
};

template <typename T>
hipError_t HostArray(T** p, size_t n) {
  return hipHostMalloc(reinterpret_cast<void**>(p), n * sizeof(T), hipHostMallocMapped | hipHostMallocPortable);
}

}  // namespace

extern "C" {

const char* tca_bytes_pipe_error() { return g_err.c_str(); }

int tca_bytes_pipe_max_stream_bytes() { return static_cast<int>(kMaxStream); }

int tca_bytes_pipe_create(int device, int max_requests, int max_count, void** out) {
  if (!out || max_requests <= 0 || max_count <= 0) return -1;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return Fail("hipSetDevice", e);
  auto* p = new (std::nothrow) Pipe();
  if (!p) return -2;
  p->device = device;
  p->max_requests = max_requests;
  p->max_count = static_cast<uint32_t>(max_count);
  const size_t streams = static_cast<size_t>(max_requests) * 2;
  const size_t elems = streams * static_cast<size_t>(max_count);
  if ((e = HostArray(&p->in, streams)) != hipSuccess || (e = HostArray(&p->out, streams)) != hipSuccess ||
      (e = HostArray(&p->status, static_cast<size_t>(max_requests))) != hipSuccess ||
      (e = hipMalloc(reinterpret_cast<void**>(&p->offsets), elems * 4)) != hipSuccess ||
      (e = hipMalloc(reinterpret_cast<void**>(&p->lens), elems * 4)) != hipSuccess ||
      (e = hipMalloc(reinterpret_cast<void**>(&p->text), elems * kTextSlot)) != hipSuccess ||
      (e = hipMalloc(reinterpret_cast<void**>(&p->text_lens), elems * 4)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&p->done, hipEventDisableTiming)) != hipSuccess) {
    const int rc = Fail("bytes pipe allocation", e);
    // partial allocations are released by destroy
    *out = p;
    return rc;
  }
  *out = p;
  return 0;
}

void tca_bytes_pipe_destroy(void* pipe) {
  auto* p = static_cast<Pipe*>(pipe);
  if (!p) return;
  (void)hipSetDevice(p->device);
  if (p->done) (void)hipEventDestroy(p->done);
  if (p->in) (void)hipHostFree(p->in);
  if (p->out) (void)hipHostFree(p->out);
  if (p->status) (void)hipHostFree(p->status);
  (void)hipFree(p->offsets);
  (void)hipFree(p->lens);
  (void)hipFree(p->text);
  (void)hipFree(p->text_lens);
  delete p;
}

// op 0: OUTPUT0 = INPUT0 + INPUT1, OUTPUT1 = INPUT0 - INPUT1 (decimal strings; two inputs, two outputs per request)
// op 1: OUTPUT0 = INPUT0 (one input, one output per request)
// in_ptrs / in_bytes: [n * inputs], out_ptrs / out_caps: [n * outputs] device addresses, request-major.
// Enqueues on `stream`; tca_bytes_pipe_wait collects the verdicts.  One batch per pipe at a time.
int tca_bytes_pipe_submit(void* pipe, int op, int n, int count, const uint64_t* in_ptrs, const uint64_t* in_bytes,
                          const uint64_t* out_ptrs, const uint64_t* out_caps, void* stream) {
  auto* p = static_cast<Pipe*>(pipe);
  if (!p || n <= 0 || n > p->max_requests || count <= 0 || static_cast<uint32_t>(count) > p->max_count || (op != 0 && op != 1))
    return -1;
  const int ins = op == 0 ? 2 : 1, outs = op == 0 ? 2 : 1;
  for (int i = 0; i < n * ins; ++i) {
    if (in_bytes[i] > kMaxStream) return -3;  // the caller takes the host path for such streams
    p->in[i] = StreamIn{reinterpret_cast<const uint8_t*>(in_ptrs[i]), in_bytes[i]};
  }
  for (int i = 0; i < n * outs; ++i) p->out[i] = StreamOut{reinterpret_cast<uint8_t*>(out_ptrs[i]), out_caps[i]};
  std::memset(p->status, 0, sizeof(RequestStatus) * static_cast<size_t>(n));
  hipError_t e = hipSetDevice(p->device);
  if (e != hipSuccess) return Fail("hipSetDevice", e);
  auto st = static_cast<hipStream_t>(stream);
  const uint32_t c = static_cast<uint32_t>(count);
  hipLaunchKernelGGL(BytesIndexBatchKernel, dim3(n * ins), dim3(kThreads), 0, st, p->in, ins, c, p->offsets, p->lens,
                     p->status);
  const size_t scan_bytes = (static_cast<size_t>(count) + 1) * sizeof(uint32_t);
  if (op == 0) {
    const uint32_t total = static_cast<uint32_t>(n) * c;
    hipLaunchKernelGGL(DecimalAddSubKernel, dim3((total + kThreads - 1) / kThreads), dim3(kThreads), 0, st, p->in, c, total,
                       p->offsets, p->lens, p->text, p->text_lens, p->status);
    hipLaunchKernelGGL(BytesPackBatchKernel, dim3(n * outs), dim3(kThreads), scan_bytes, st, p->out, outs, c, p->in, p->text,
                       p->offsets, p->text_lens, 1, p->status);
  } else {
    hipLaunchKernelGGL(BytesPackBatchKernel, dim3(n * outs), dim3(kThreads), scan_bytes, st, p->out, outs, c, p->in, p->text,
                       p->offsets, p->lens, 0, p->status);
  }
  e = hipGetLastError();
  if (e != hipSuccess) return Fail("bytes pipe launch", e);
  e = hipEventRecord(p->done, st);
  return e == hipSuccess ? 0 : Fail("hipEventRecord", e);
}

// Blocks until the batch has finished; codes[i] = verdict of request i, out_bytes[2*i + k] = bytes of its output k.
int tca_bytes_pipe_wait(void* pipe, int n, uint32_t* codes, uint64_t* out_bytes) {
  auto* p = static_cast<Pipe*>(pipe);
  if (!p || n <= 0 || n > p->max_requests) return -1;
  const hipError_t e = hipEventSynchronize(p->done);
  if (e != hipSuccess) return Fail("hipEventSynchronize", e);
  for (int i = 0; i < n; ++i) {
    codes[i] = p->status[i].code;
    out_bytes[2 * i] = p->status[i].out_bytes[0];
    out_bytes[2 * i + 1] = p->status[i].out_bytes[1];
  }
  return 0;
}

}  // extern "C"
