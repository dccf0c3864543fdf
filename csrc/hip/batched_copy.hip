// Batched device-to-device copy: up to 64 (source, destination, size) triples
// in ONE launch.  The dynamic batcher gathers the inputs of the merged requests
// (each in its own hipshm region) into the model's static batch buffer and
// scatters the output rows back into the clients' regions; doing that with one
// copy kernel per request costs a launch (and a Python call) per request and
// shows up as idle gaps in front of every forward.  The triples travel in the
// kernel-argument block, so there is no descriptor upload either.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxEntries = 64;
constexpr uint32_t kChunkBytes = 64 * 1024;  // bytes one workgroup moves

struct CopyBatch {
  const uint8_t* src[kMaxEntries];
  uint8_t* dst[kMaxEntries];
  uint64_t bytes[kMaxEntries];
};

__global__ __launch_bounds__(kThreads) void BatchedCopyKernel(CopyBatch batch) {
  const int entry = blockIdx.y;
  const uint64_t total = batch.bytes[entry];
  const uint64_t begin = static_cast<uint64_t>(blockIdx.x) * kChunkBytes;
  if (begin >= total) return;
  const uint64_t end = begin + kChunkBytes < total ? begin + kChunkBytes : total;
  const uint8_t* __restrict__ src = batch.src[entry];
  uint8_t* __restrict__ dst = batch.dst[entry];
  const bool aligned = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
  if (aligned) {
    const uint64_t vec_end = begin + ((end - begin) & ~uint64_t(15));
    for (uint64_t off = begin + static_cast<uint64_t>(threadIdx.x) * 16; off < vec_end; off += kThreads * 16)
      *reinterpret_cast<uint4*>(dst + off) = *reinterpret_cast<const uint4*>(src + off);
    for (uint64_t off = vec_end + threadIdx.x; off < end; off += kThreads) dst[off] = src[off];
  } else {
    for (uint64_t off = begin + threadIdx.x; off < end; off += kThreads) dst[off] = src[off];
  }
}

}  // namespace

extern "C" int tca_batched_copy_launch(const void* const* srcs, void* const* dsts, const uint64_t* bytes, int n,
                                       void* stream) {
  if (n < 0 || (n > 0 && (!srcs || !dsts || !bytes))) return -1;
  auto st = static_cast<hipStream_t>(stream);
  for (int first = 0; first < n; first += kMaxEntries) {
    const int count = n - first < kMaxEntries ? n - first : kMaxEntries;
    CopyBatch batch;
    uint64_t longest = 0;
    for (int i = 0; i < count; ++i) {
      if ((!srcs[first + i] || !dsts[first + i]) && bytes[first + i] != 0) return -1;
      batch.src[i] = static_cast<const uint8_t*>(srcs[first + i]);
      batch.dst[i] = static_cast<uint8_t*>(dsts[first + i]);
      batch.bytes[i] = bytes[first + i];
      if (bytes[first + i] > longest) longest = bytes[first + i];
    }
    if (longest == 0) continue;
    const uint64_t chunks = (longest + kChunkBytes - 1) / kChunkBytes;
    if (chunks > 0x7FFFFFFF) return -2;
    hipLaunchKernelGGL(BatchedCopyKernel, dim3(static_cast<unsigned>(chunks), count), dim3(kThreads), 0, st, batch);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return -1000 - static_cast<int>(e);
  }
  return 0;
}
