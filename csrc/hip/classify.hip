// Classification extension on the GPU (gfx950):
//
//   topk     logits [rows, cols] (fp32 / fp16 / bf16) -> k best (value, index)
//            per row.  One workgroup per row; the row is staged once in LDS as
//            fp32 and k rounds of a wavefront-shuffle arg-max pick the winners
//            (ties resolve to the lowest index), so HBM is read exactly once.
//   strings  (value, index) -> the protocol's "value:index[:label]" BYTES
//            elements, already length-prefixed, ready to be the response
//            payload or to land in a hipshm output region.  The strings the
//            reference's image_client parses come from the server
//            (`src/c++/examples/image_client.cc:256-276`,
//            `src/python/examples/grpc_image_client.py:227-243`); value is
//            printed like printf("%f").
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cfloat>
#include <cstdint>

#include "kernels.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxCols = 12288;  // 48 KiB of fp32 in LDS

template <typename T> __device__ __forceinline__ float ToFloat(T v);
template <> __device__ __forceinline__ float ToFloat<float>(float v) { return v; }
template <> __device__ __forceinline__ float ToFloat<__half>(__half v) { return __half2float(v); }
template <> __device__ __forceinline__ float ToFloat<uint16_t>(uint16_t v) {  // bf16 bits
  return __uint_as_float(static_cast<uint32_t>(v) << 16);
}

template <typename T>
__global__ __launch_bounds__(kThreads) void TopKKernel(const T* __restrict__ logits, int cols, int k,
                                                       float* __restrict__ values, int* __restrict__ indices) {
  __shared__ float row[kMaxCols];
  __shared__ float wave_val[kThreads / 64];
  __shared__ int wave_idx[kThreads / 64];
  const T* __restrict__ src = logits + static_cast<size_t>(blockIdx.x) * cols;
  for (int i = threadIdx.x; i < cols; i += kThreads) {
    float v = ToFloat<T>(src[i]);
    row[i] = (v != v) ? -FLT_MAX : v;  // NaN sorts last
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  for (int round = 0; round < k; ++round) {
    float best = -INFINITY;
    int best_i = 0x7FFFFFFF;
    for (int i = threadIdx.x; i < cols; i += kThreads) {
      const float v = row[i];
      if (v > best || (v == best && i < best_i)) { best = v; best_i = i; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const float ov = __shfl_xor(best, d, 64);
      const int oi = __shfl_xor(best_i, d, 64);
      if (ov > best || (ov == best && oi < best_i)) { best = ov; best_i = oi; }
    }
    if (lane == 0) { wave_val[wave] = best; wave_idx[wave] = best_i; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < kThreads / 64; ++w)
        if (wave_val[w] > best || (wave_val[w] == best && wave_idx[w] < best_i)) { best = wave_val[w]; best_i = wave_idx[w]; }
      values[static_cast<size_t>(blockIdx.x) * k + round] = best;
      indices[static_cast<size_t>(blockIdx.x) * k + round] = best_i;
      if (best_i < cols) row[best_i] = -INFINITY;  // exclude from later rounds
    }
    __syncthreads();
  }
}

// printf("%f") of a finite float into buf; returns the length.
__device__ int FormatFixed6(float value, char* buf) {
  int n = 0;
  if (value != value) { buf[0] = 'n'; buf[1] = 'a'; buf[2] = 'n'; return 3; }
  if (value < 0.f || (value == 0.f && (__float_as_uint(value) >> 31))) { buf[n++] = '-'; value = -value; }
  if (value > 3.0e38f) { buf[n++] = 'i'; buf[n++] = 'n'; buf[n++] = 'f'; return n; }
  // exact: a float is an integer multiple of 2^-149; double holds value * 1e6 to < 1 ulp for the ranges a
  // classifier emits, rint gives the round-half-even decimal
  const double scaled = rint(static_cast<double>(value) * 1.0e6);
  double ipart_d = floor(scaled / 1.0e6);
  uint32_t frac = static_cast<uint32_t>(scaled - ipart_d * 1.0e6);
  char tmp[48];
  int t = 0;
  if (ipart_d < 1.8e19) {
    unsigned long long ip = static_cast<unsigned long long>(ipart_d);
    do { tmp[t++] = static_cast<char>('0' + ip % 10); ip /= 10; } while (ip);
  } else {  // beyond 64 bits: peel digits in floating point
    while (ipart_d >= 1.0 && t < 40) {
      const double q = floor(ipart_d / 10.0);
      tmp[t++] = static_cast<char>('0' + static_cast<int>(ipart_d - q * 10.0));
      ipart_d = q;
    }
  }
  while (t) buf[n++] = tmp[--t];
  buf[n++] = '.';
  for (int d = 100000; d >= 1; d /= 10) { buf[n++] = static_cast<char>('0' + (frac / d) % 10); }
  return n;
}

__device__ int FormatInt(int v, char* buf) {
  char tmp[12];
  int t = 0, n = 0;
  unsigned u = v < 0 ? static_cast<unsigned>(-static_cast<long long>(v)) : static_cast<unsigned>(v);
  if (v < 0) buf[n++] = '-';
  do { tmp[t++] = static_cast<char>('0' + u % 10); u /= 10; } while (u);
  while (t) buf[n++] = tmp[--t];
  return n;
}

constexpr int kChunk = 512;        // entries formatted per pass of the single formatting workgroup
constexpr int kHeadBytes = 64;     // "value:index" always fits

// One workgroup walks the entries in chunks of kChunk: format heads into LDS,
// serial scan of the chunk's lengths (<= 512 adds, cheaper than a barrier tree),
// then every thread copies its elements.  `base` carries the running output
// offset between chunks, so element order and offsets match a serial encoder.
// Elements that would cross `out_cap` are not written; `out_size` always gets
// the full size so the caller can tell.
__global__ __launch_bounds__(kThreads) void ClassifyStringsKernel(
    const float* __restrict__ values, const int* __restrict__ indices, int entries, const uint8_t* __restrict__ labels,
    const uint32_t* __restrict__ label_offsets, int num_labels, uint8_t* __restrict__ out, uint64_t out_cap,
    uint64_t* __restrict__ out_size) {
  __shared__ char heads[kChunk][kHeadBytes];
  __shared__ uint32_t head_len[kChunk];
  __shared__ uint32_t total_len[kChunk];
  __shared__ uint64_t start[kChunk + 1];
  __shared__ uint64_t base;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  for (int first = 0; first < entries; first += kChunk) {
    const int count = min(kChunk, entries - first);
    for (int e = threadIdx.x; e < count; e += kThreads) {
      char* h = heads[e];
      int n = FormatFixed6(values[first + e], h);
      h[n++] = ':';
      const int idx = indices[first + e];
      n += FormatInt(idx, h + n);
      uint32_t lab = 0;
      if (labels != nullptr && idx >= 0 && idx < num_labels) lab = 1 + (label_offsets[idx + 1] - label_offsets[idx]);
      head_len[e] = static_cast<uint32_t>(n);
      total_len[e] = static_cast<uint32_t>(n) + lab;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint64_t acc = base;
      for (int e = 0; e < count; ++e) { start[e] = acc; acc += 4 + total_len[e]; }
      start[count] = acc;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < count; e += kThreads) {
      const uint32_t len = total_len[e];
      if (start[e] + 4 + len > out_cap) continue;
      uint8_t* p = out + start[e];
      p[0] = static_cast<uint8_t>(len);
      p[1] = static_cast<uint8_t>(len >> 8);
      p[2] = static_cast<uint8_t>(len >> 16);
      p[3] = static_cast<uint8_t>(len >> 24);
      p += 4;
      for (uint32_t j = 0; j < head_len[e]; ++j) p[j] = static_cast<uint8_t>(heads[e][j]);
      if (len > head_len[e]) {
        p += head_len[e];
        *p++ = ':';
        const int idx = indices[first + e];
        const uint8_t* l = labels + label_offsets[idx];
        for (uint32_t j = 0; j + 1 < len - head_len[e]; ++j) p[j] = l[j];
      }
    }
    __syncthreads();  // heads / start are rewritten by the next chunk
    if (threadIdx.x == 0) base = start[count];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out_size = base;
}

}  // namespace

extern "C" {

int tca_topk_launch(const void* logits, int dtype, int rows, int cols, int k, float* values, int* indices,
                    void* stream) {
  if (!logits || !values || !indices || rows <= 0 || cols <= 0 || k <= 0) return -1;
  if (cols > kMaxCols || k > cols) return -2;
  auto st = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case tca::kF32:
      hipLaunchKernelGGL(TopKKernel<float>, dim3(rows), dim3(kThreads), 0, st, static_cast<const float*>(logits), cols,
                         k, values, indices);
      break;
    case tca::kF16:
      hipLaunchKernelGGL(TopKKernel<__half>, dim3(rows), dim3(kThreads), 0, st, static_cast<const __half*>(logits),
                         cols, k, values, indices);
      break;
    case tca::kBF16:
      hipLaunchKernelGGL(TopKKernel<uint16_t>, dim3(rows), dim3(kThreads), 0, st,
                         static_cast<const uint16_t*>(logits), cols, k, values, indices);
      break;
    default:
      return -3;
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -1000 - static_cast<int>(e);
}

int tca_classify_strings_launch(const float* values, const int* indices, int rows, int k, const uint8_t* labels,
                                const uint32_t* label_offsets, int num_labels, uint8_t* out, uint64_t out_cap,
                                uint64_t* out_size, void* stream) {
  const long long entries = static_cast<long long>(rows) * k;
  if (!values || !indices || !out || !out_size || entries <= 0) return -1;
  if (entries > (1 << 20)) return -2;
  if (labels && !label_offsets) return -1;
  hipLaunchKernelGGL(ClassifyStringsKernel, dim3(1), dim3(kThreads), 0, static_cast<hipStream_t>(stream), values,
                     indices, static_cast<int>(entries), labels, label_offsets, num_labels, out, out_cap, out_size);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -1000 - static_cast<int>(e);
}

}  // extern "C"
