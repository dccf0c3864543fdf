// Classifier tail of a convolutional network on gfx950: global average pooling and the fully
// connected layer with bias, writing fp32 logits - two small kernels in place of the library's
// reduction, GEMM and cast (the reference has no model code; its models run in the server it
// talks to, e.g. `src/python/examples/image_client.py:59-150` only inspects their metadata).
//
//   pool  y [B][P][C] fp16 (channels-last)  ->  feats [B][C] fp16, mean over the P pixels in fp32.
//         A workgroup covers 256 channels of one image: 32 groups of 8 channels (16-byte loads) x 8
//         slices of the pixels, summed through LDS.
//   fc    logits[b][n] = sum_k feats[b][k] * W[n][k] + bias[n], fp32 out.  One workgroup per
//         16 classes x 16 images; its 4 waves split K and feed v_mfma_f32_16x16x32_f16 straight from
//         global memory (both operands are K-contiguous rows, which IS the MFMA operand layout),
//         partial sums meet in LDS.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels.h"

namespace {

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using float4v = __attribute__((ext_vector_type(4))) float;

constexpr int kThreads = 256;

// One workgroup: 32 groups of 8 channels x 8 slices of the pixels; the slices meet in LDS.
__global__ __launch_bounds__(kThreads) void AvgPoolKernel(const _Float16* __restrict__ y, _Float16* __restrict__ feats,
                                                          int pixels, int channels, float inv_pixels) {
  __shared__ float sums[8][32][8];
  const int b = blockIdx.y;
  const int group = threadIdx.x & 31;   // which 8 channels of the block's 256
  const int slice = threadIdx.x >> 5;   // pixels slice, slice + 8, ...
  const int c0 = blockIdx.x * 256 + group * 8;
  const bool live = c0 < channels;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (live) {
    const _Float16* src = y + (static_cast<long long>(b) * pixels) * channels + c0;
    for (int p = slice; p < pixels; p += 8) {
      const half8 v = *reinterpret_cast<const half8*>(src + static_cast<long long>(p) * channels);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += static_cast<float>(v[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) sums[slice][group][j] = acc[j];
  __syncthreads();
  if (slice != 0 || !live) return;
  half8 out;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float t = 0.f;
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) t += sums[sl][group][j];
    out[j] = static_cast<_Float16>(t * inv_pixels);
  }
  *reinterpret_cast<half8*>(feats + static_cast<long long>(b) * channels + c0) = out;
}

__global__ __launch_bounds__(kThreads) void LinearKernel(const _Float16* __restrict__ x, const _Float16* __restrict__ w,
                                                         const _Float16* __restrict__ bias, float* __restrict__ out,
                                                         int batch, int n_out, int k_total) {
  __shared__ float4v partial[kThreads / 64][64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int n0 = blockIdx.x * 16;
  const int b0 = blockIdx.y * 16;
  // operand rows past the end: the last valid row (their products are never stored)
  const int n_row = n0 + (lane & 15) < n_out ? n0 + (lane & 15) : n_out - 1;
  const int b_row = b0 + (lane & 15) < batch ? b0 + (lane & 15) : batch - 1;
  const int k_per_wave = k_total / (kThreads / 64);
  const int k_begin = wave * k_per_wave + (lane >> 4) * 8;
  const _Float16* wp = w + static_cast<long long>(n_row) * k_total + k_begin;
  const _Float16* xp = x + static_cast<long long>(b_row) * k_total + k_begin;
  float4v acc = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < k_per_wave; k += 128) {  // eight 16-byte loads in flight per lane, then four MFMAs
    half8 wf[4], xf[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      wf[u] = *reinterpret_cast<const half8*>(wp + k + u * 32);
      xf[u] = *reinterpret_cast<const half8*>(xp + k + u * 32);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[u], xf[u], acc, 0, 0, 0);
  }
  partial[wave][lane] = acc;
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int wv = 1; wv < kThreads / 64; ++wv) acc += partial[wv][lane];
  // lane: image b0 + (lane & 15), classes n0 + 4 * (lane >> 4) .. + 3
  const int b = b0 + (lane & 15);
  const int n = n0 + 4 * (lane >> 4);
  if (b >= batch) return;
  float* dst = out + static_cast<long long>(b) * n_out + n;
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (n + r < n_out) dst[r] = acc[r] + (bias != nullptr ? static_cast<float>(bias[n + r]) : 0.f);
}

int Status() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -1000 - static_cast<int>(e);
}

}  // namespace

extern "C" {

// y: [batch][pixels][channels] fp16 -> feats: [batch][channels] fp16 (channels % 8 == 0, 16-byte aligned)
int tca_avgpool_launch(const void* y, void* feats, int batch, int pixels, int channels, void* stream) {
  if (!y || !feats || batch <= 0 || pixels <= 0 || channels <= 0) return -1;
  if (channels % 8 != 0 || (reinterpret_cast<uintptr_t>(y) & 15) || (reinterpret_cast<uintptr_t>(feats) & 15)) return -2;
  hipLaunchKernelGGL(AvgPoolKernel, dim3((channels + 255) / 256, batch), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream), static_cast<const _Float16*>(y), static_cast<_Float16*>(feats), pixels,
                     channels, 1.0f / static_cast<float>(pixels));
  return Status();
}

// out[b][n] = sum_k x[b][k] * w[n][k] + bias[n]; x, w fp16 rows of k_total (a multiple of 512) halfs, out fp32
int tca_linear_launch(const void* x, const void* w, const void* bias, void* out, int batch, int n_out, int k_total,
                      void* stream) {
  if (!x || !w || !out || batch <= 0 || n_out <= 0 || k_total <= 0) return -1;
  if (k_total % 512 != 0 || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(w) & 15)) return -2;
  hipLaunchKernelGGL(LinearKernel, dim3((n_out + 15) / 16, (batch + 15) / 16), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream), static_cast<const _Float16*>(x), static_cast<const _Float16*>(w),
                     static_cast<const _Float16*>(bias), static_cast<float*>(out), batch, n_out, k_total);
  return Status();
}

}  // extern "C"
