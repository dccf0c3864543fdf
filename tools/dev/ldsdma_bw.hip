// How many bytes per clock one CU takes in from L2 / the last-level cache / HBM, by staging form:
//   dma   buffer_load_dwordx4 ... lds (1 KiB per wave instruction, no registers)
//   reg   global_load_dwordx4 into registers, then ds_write_b128
//   load  global_load_dwordx4 into registers only
// Every block streams `tile` bytes per step from a window of `window` bytes that all blocks share
// (window <= a few MB: L2 hits, as the weights and the re-read activation rows of a convolution tile;
// large: HBM).  Prints GB/s and bytes per clock per CU at 2.4 GHz.
//
//   hipcc --offload-arch=gfx950 -O3 -o ldsdma_bw ldsdma_bw.hip && ./ldsdma_bw
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using lds_ptr_t = __attribute__((address_space(3))) void*;
using uint4v = __attribute__((ext_vector_type(4))) unsigned;

template <int kMode, int kWaves, int kPieces>
__global__ __launch_bounds__(64 * kWaves) void Stream(const char* __restrict__ src, unsigned window_mask, int steps,
                                                      unsigned* __restrict__ sink) {
  constexpr int pieces = kPieces;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, static_cast<int>(window_mask + 1u), 0x00020000);
  // pieces per wave and step; a step of the block is kWaves * pieces KiB
  const unsigned step_bytes = static_cast<unsigned>(kWaves * pieces) * 1024u;
  unsigned pos = (blockIdx.x * 2654435761u) & window_mask & ~(step_bytes - 1u);
  uint4v acc = {0, 0, 0, 0};
  for (int s = 0; s < steps; ++s) {
    uint4v v[kPieces];
#pragma unroll
    for (int p = 0; p < pieces; ++p) {
      const unsigned off = (pos + static_cast<unsigned>((p * kWaves + wave) * 1024)) & window_mask;
      char* dst = lds + ((p * kWaves + wave) & 31) * 1024;
      if constexpr (kMode == 0) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, reinterpret_cast<lds_ptr_t>(reinterpret_cast<uintptr_t>(dst)), 16,
                                                 lane * 16, static_cast<int>(off), 0, 0);
      } else {
        v[p] = *reinterpret_cast<const uint4v*>(src + off + lane * 16);
      }
    }
    if constexpr (kMode != 0) {
#pragma unroll
      for (int p = 0; p < pieces; ++p) {
        char* dst = lds + ((p * kWaves + wave) & 31) * 1024;
        if constexpr (kMode == 1) *reinterpret_cast<uint4v*>(dst + lane * 16) = v[p];
        else acc ^= v[p];
      }
    }
    pos = (pos + step_bytes * 977u) & window_mask & ~(step_bytes - 1u);
    if constexpr (kMode == 0) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (kMode != 0 || steps < 0) acc ^= *reinterpret_cast<uint4v*>(lds + lane * 16);
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345679u) sink[0] = 1;  // keeps the loads alive
}

template <int kMode, int kWaves, int kPieces>
void Run(const char* name, const char* src, size_t window, int blocks, int steps, unsigned* sink) {
  const int pieces = kPieces;
  const size_t lds_bytes = 32 * 1024;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((Stream<kMode, kWaves, kPieces>), dim3(blocks), dim3(64 * kWaves), lds_bytes, 0, src,
                       static_cast<unsigned>(window - 1), steps, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double bytes = static_cast<double>(blocks) * steps * pieces * kWaves * 1024.0;
  const double gbs = bytes / (ms * 1e-3) / 1e9;
  const int cus = blocks < 256 ? blocks : 256;
  std::printf("%-5s waves %2d blocks %4d pieces/wave/step %2d window %7.1f MB: %8.1f GB/s  %6.1f B/clk/CU  (%.3f ms)\n", name,
              kWaves, blocks, pieces, window / 1048576.0, gbs, gbs * 1e9 / 2.4e9 / cus, ms);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) std::printf("  error: %s\n", hipGetErrorString(e));
}

int main() {
  const size_t total = 1ull << 30;
  char* src = nullptr;
  unsigned* sink = nullptr;
  if (hipMalloc(&src, total) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
  hipMemset(src, 1, total);
  hipMemset(sink, 0, 64);
  for (size_t window : {size_t(2) << 20, size_t(64) << 20, size_t(1) << 30}) {
    const int steps = 400;
    for (int blocks : {256, 512}) {
      Run<0, 4, 8>("dma", src, window, blocks, steps, sink);
      Run<0, 8, 4>("dma", src, window, blocks, steps, sink);
      Run<0, 16, 2>("dma", src, window, blocks, steps, sink);
      Run<1, 4, 8>("reg", src, window, blocks, steps, sink);
      Run<1, 8, 4>("reg", src, window, blocks, steps, sink);
      Run<1, 16, 2>("reg", src, window, blocks, steps, sink);
      Run<2, 4, 8>("load", src, window, blocks, steps, sink);
      Run<2, 8, 4>("load", src, window, blocks, steps, sink);
    }
  }
  hipFree(src);
  hipFree(sink);
  return 0;
}
