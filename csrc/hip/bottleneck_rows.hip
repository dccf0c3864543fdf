// A whole ResNet bottleneck tail with 64 middle channels in one kernel, organised by ROWS:
//
//     t   = relu(conv3x3(x, W1) + b1)                       x: [M][64]   (stride 1, pad 1)
//     y   = relu(t x W2^T (+ xd x Wd^T) + b2 (+ residual))  y: [M][256]
//     out = relu(y x W3^T + b3)                             out: [M][N3] (the NEXT block's first 1x1)
//
// A block owns 128 consecutive output pixels and every one of its 8 wavefronts owns 16 of them for
// ALL channels of all three products.  That choice removes every workgroup barrier behind the 3x3
// convolution:
//
//   * the 3x3 result of a wave's pixels (16 x 64, in the MFMA accumulator layout: lane = pixel, four
//     consecutive channels per 16-channel fragment) is packed to fp16 IN REGISTERS and is, as it
//     stands, the activation operand of the second product - the summation index may be fed to the
//     matrix core in any order as long as both operands agree, so the host stores W2 with its K
//     columns permuted to the order the accumulator registers come in (PermuteMidK in ops/conv.py);
//   * the 256-channel result is transposed 64 channels at a time through a 2 KiB LDS tile that
//     belongs to the wave alone (LDS operations of one wave execute in order: no barrier), read
//     back as 16-byte vectors that are at once the coalesced store / residual layout AND the
//     activation operand of the third product;
//   * W2 and W3 sit in LDS for the whole phase, so a wave never waits for another one: while one
//     wave is in its epilogue the other waves of the SIMD issue MFMAs.
//
// The kernel it replaces for these shapes walks the 256 output channels in four chunks with three
// workgroup barriers each; at two blocks per CU those barriers, not bytes or MFMA rate, set its
// pace.  Here the residual rows of a wave are requested two chunks ahead (before the 3x3 loop for
// the first two), which is what keeps enough bytes in flight per CU to approach the memory rate.
//
// The 3x3 part is the halo form of conv_gemm.hip: the 128 + 2 W + 2 input rows the nine taps of
// the tile touch are brought into LDS once by LDS-DMA, the K loop streams the nine 8 KiB weight
// taps through three LDS stages (counted vmcnt, one raw barrier per tap).
//
// LDS (72.5 KiB, two blocks per CU): K loop:  [0,32K) halo   [32K,56K) weight taps
//                                    after :  [0,32K) W2     [32K,64K) W3      [64K,72K) wave tiles, biases
// With the projection shortcut inside (K2 = 128) W2 exceeds its 32 KiB: the 256 output channels run in two
// phases of 128 and the weight space is re-filled in between (two barriers).  With N3 = 128 the third
// product runs 64 of its outputs at a time: the second 64 rows of W3 replace the first ones behind the
// chunk loop and are multiplied with the output chunks, which stay in registers for that.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "kernels.h"

namespace {

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half4 = __attribute__((ext_vector_type(4))) _Float16;
using float4v = __attribute__((ext_vector_type(4))) float;
using uint4v = __attribute__((ext_vector_type(4))) unsigned;
using lds_ptr_t = __attribute__((address_space(3))) void*;

constexpr int kBM = 128;        // pixels per block
constexpr int kThreads = 512;   // 8 wavefronts x 16 pixels
constexpr int kN2 = 256;        // output channels of the block
constexpr int kHaloRows = 256;  // LDS rows of the halo (128 + 2 W + 2 used, at least one spare row of zeros)
constexpr unsigned kOutOfRange = 0x80000000u;  // buffer offset past every descriptor: loads give 0, stores are dropped

__device__ __forceinline__ void DmaToLds(__amdgpu_buffer_rsrc_t rsrc, unsigned voffset, int soffset, _Float16* lds_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, reinterpret_cast<lds_ptr_t>(reinterpret_cast<uintptr_t>(lds_base)), 16,
                                           static_cast<int>(voffset), soffset, 0, 0);
}

// Workgroup barrier that orders LDS traffic only (vector memory operations stay in flight).
__device__ __forceinline__ void LdsBarrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

__device__ __forceinline__ half4 PackRelu(const float4v& a, bool relu) {
  half4 p;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float v = relu ? (a[r] > 0.f ? a[r] : 0.f) : a[r];
    p[r] = static_cast<_Float16>(v);
  }
  return p;
}

__device__ __forceinline__ float4v Widen(const half4& b) {
  return float4v{static_cast<float>(b[0]), static_cast<float>(b[1]), static_cast<float>(b[2]), static_cast<float>(b[3])};
}

struct RowsArgs {
  const _Float16* x;         // [M][64]  input of the 3x3 convolution
  const _Float16* w1;        // [64][3][3][64]
  const _Float16* bias1;     // [64]
  const _Float16* w2;        // [256][K2]: the first 64 K columns in accumulator order, then (kXS) 64 natural ones
  const _Float16* bias2;     // [256] or null
  const _Float16* residual;  // [M][256] or null
  _Float16* y;               // [M][256]
  const _Float16* w3;        // [N3][256]
  const _Float16* bias3;     // [N3] or null
  _Float16* out3;            // [M][N3]
  const _Float16* xd;        // kXS: [M][64] second activation operand (the block input)
  int M;
  int H;
  int W;
  int xcd_chunk;
  int ablate;  // measurements only (TCA_ROWS_ABLATE): 1 = no y stores, 2 = no residual loads, 4 = no out3 stores,
               // 8 = stop after the 3x3 part, 16 = no MFMAs in the 3x3 part, 32 = none behind it
};

template <int kN3, int kXS>
__global__ __launch_bounds__(kThreads, 4) void BottleneckRowsKernel(RowsArgs a) {
  constexpr int kK2 = 64 + 64 * kXS;                       // K of the second product
  constexpr int kPhases = kXS > 0 ? 2 : 1;                 // output-channel phases (the weights of one phase fit LDS)
  constexpr int kPasses3 = kN3 / 64;                       // the third product runs 64 of its outputs at a time
  constexpr int kCPP = 4 / kPhases;                        // 64-channel chunks per phase
  constexpr int kRows2 = kN2 / kPhases;                    // W2 rows per phase
  constexpr int kW2Vecs = kRows2 * (kK2 / 8) / kThreads;   // 16-byte vectors per thread and phase
  constexpr int kW3Vecs = 64 * (kCPP * 8) / kThreads;
  constexpr int kNI3 = 4;                                  // 16-channel fragments of one pass
  static_assert(kW2Vecs >= 1 && kW3Vecs >= 1, "weight parts must cover the block");
  static_assert(kRows2 * kK2 <= 16384, "a phase's weights exceed their 32 KiB");
  static_assert(kPasses3 == 1 || kPhases == 1, "two passes of the third product need every output chunk at hand");
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
  _Float16* const halo = lds;
  _Float16* const w1s = lds + 16384;   // three stages of [64][64]
  _Float16* const w2s = lds;           // [kK2/64][kRows2][64]
  _Float16* const w3s = lds + 16384;   // [kCPP][64][64]: 64 rows of W3, the K columns of the phase
  _Float16* const tiles = lds + 32768;        // 8 x [16 pixels][32 channels]
  _Float16* const b2s = lds + 32768 + 8 * 512;  // the second product's 256 biases

  int bid = blockIdx.x;
  if (a.xcd_chunk > 0) bid = (bid & 7) * a.xcd_chunk + (bid >> 3);
  const int m0 = bid * kBM;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int px = lane & 15;
  const int q = lane >> 4;
  const int i_local = wave * 16 + px;
  const int m = m0 + i_local;
  const bool valid = m < a.M;
  const int W = a.W;

  // ---- per-lane constants -----------------------------------------------------------------
  unsigned taps = 0;
  if (valid) {
    const int hw = a.H * W;
    const int img = m / hw;
    const int rem = m - img * hw;
    const int oh = rem / W;
    const int ow = rem - oh * W;
    unsigned rows_ok = 0, cols_ok = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      rows_ok |= static_cast<unsigned>(static_cast<unsigned>(oh - 1 + k) < static_cast<unsigned>(a.H)) << k;
      cols_ok |= static_cast<unsigned>(static_cast<unsigned>(ow - 1 + k) < static_cast<unsigned>(W)) << k;
    }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
      if ((rows_ok >> kh) & 1u) taps |= cols_ok << (kh * 3);
  }
  const unsigned row_off = valid ? static_cast<unsigned>(m) : 0u;
  const unsigned y_off = valid ? (row_off * kN2 + q * 8) * 2u : kOutOfRange;
  const unsigned o3_off = valid ? (row_off * kN3 + q * 8) * 2u : kOutOfRange;
  const unsigned xd_off = valid ? (row_off * 64 + q * 8) * 2u : kOutOfRange;
  const int swz = (px >> 1) & 7;
  const int frag_off = px * 64 + ((q ^ swz) * 8);  // fragment row px, chunk q of a [rows][64] operand slab (+ XOR 32: chunk q + 4)

  const unsigned m_bytes = static_cast<unsigned>(a.M);
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.x), 0, static_cast<int>(m_bytes * 128u), 0x00020000);
  const __amdgpu_buffer_rsrc_t w1_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.w1), 0, 64 * 576 * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t w2_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.w2), 0, kN2 * kK2 * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t w3_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.w3), 0, kN3 * kN2 * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t y_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (a.ablate & 1) ? 0 : static_cast<int>(m_bytes * (kN2 * 2u)), 0x00020000);
  const __amdgpu_buffer_rsrc_t o3_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.out3, 0, (a.ablate & 4) ? 0 : static_cast<int>(m_bytes * (kN3 * 2u)), 0x00020000);
  // a null residual / xd gets an empty descriptor: every load returns zeros
  const __amdgpu_buffer_rsrc_t res_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<_Float16*>(a.residual != nullptr ? a.residual : a.x), 0,
      a.residual != nullptr && !(a.ablate & 2) ? static_cast<int>(m_bytes * (kN2 * 2u)) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t xd_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<_Float16*>(kXS > 0 ? a.xd : a.x), 0, kXS > 0 ? static_cast<int>(m_bytes * 128u) : 0, 0x00020000);

  // ---- everything the later products need, requested first ----------------------------------
  // (older than every LDS DMA below: complete, in issue order, before the K loop's first wait returns)
  unsigned w2_off[kW2Vecs], w3_off[kW3Vecs];
  uint4v w2r[kW2Vecs], w3r[kW3Vecs];
#pragma unroll
  for (int i = 0; i < kW2Vecs; ++i) {
    const int v = i * kThreads + tid;
    const int ks = v / (kRows2 * 8);
    const int r = (v >> 3) % kRows2;
    const int chunk = (v & 7) ^ ((r >> 1) & 7);
    w2_off[i] = static_cast<unsigned>((r * kK2 + ks * 64 + chunk * 8) * 2);
  }
#pragma unroll
  for (int i = 0; i < kW3Vecs; ++i) {
    const int v = i * kThreads + tid;
    const int cc = v / (64 * 8);
    const int r = (v >> 3) % 64;
    const int chunk = (v & 7) ^ ((r >> 1) & 7);
    w3_off[i] = static_cast<unsigned>((r * kN2 + cc * 64 + chunk * 8) * 2);
  }
  auto load_w3 = [&](int phase, int pass) {
#pragma unroll
    for (int i = 0; i < kW3Vecs; ++i)
      w3r[i] = __builtin_amdgcn_raw_buffer_load_b128(w3_rsrc, static_cast<int>(w3_off[i]),
                                                     (pass * 64 * kN2 + phase * kCPP * 64) * 2, 0);
  };
  auto load_weights = [&](int phase) {
#pragma unroll
    for (int i = 0; i < kW2Vecs; ++i)
      w2r[i] = __builtin_amdgcn_raw_buffer_load_b128(w2_rsrc, static_cast<int>(w2_off[i]), phase * kRows2 * kK2 * 2, 0);
    load_w3(phase, 0);
  };
  auto store_w3 = [&]() {
#pragma unroll
    for (int i = 0; i < kW3Vecs; ++i) *reinterpret_cast<uint4v*>(w3s + (i * kThreads + tid) * 8) = w3r[i];
  };
  auto store_weights = [&]() {
#pragma unroll
    for (int i = 0; i < kW2Vecs; ++i) *reinterpret_cast<uint4v*>(w2s + (i * kThreads + tid) * 8) = w2r[i];
    store_w3();
  };
  // Residual rows of all four chunks.  EVERY load of this kernel is issued before its first store: the
  // vector-memory counter retires in order, so a wait for a load issued behind a store would also wait
  // for that store's acknowledgement - the one round trip the kernel has no need to wait for.
  half8 res[4][2];  // [chunk][32-channel half]
  auto load_res = [&](int chunk, half8 (&dst)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
      dst[s] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(res_rsrc, static_cast<int>(y_off), (chunk * 64 + s * 32) * 2, 0));
  };
  half8 bxd[2];
  load_weights(0);
  if constexpr (kXS > 0) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
      bxd[s] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(xd_rsrc, static_cast<int>(xd_off), s * 64, 0));
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) load_res(c, res[c]);
  uint4v b2v{0u, 0u, 0u, 0u};  // 8 of the second product's biases per thread of the first half-wave, parked in LDS later
  if (a.bias2 != nullptr && tid < 32) b2v = *reinterpret_cast<const uint4v*>(a.bias2 + tid * 8);
  float4v acc1[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) acc1[f] = Widen(*reinterpret_cast<const half4*>(a.bias1 + f * 16 + q * 4));

  // ---- 3x3 convolution: halo + nine weight taps ------------------------------------------------
  {
    const long long first = static_cast<long long>(m0) - W - 1;  // input pixel of halo row 0
    const int used_rows = kBM + 2 * W + 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = i * 8 + wave;
      const int row = j * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ (row & 7);
      const long long p = first + row;
      const bool ok = p >= 0 && p < a.M && row < used_rows;
      DmaToLds(x_rsrc, ok ? static_cast<unsigned>((p * 64 + chunk * 8) * 2) : kOutOfRange, 0, halo + j * 512);
    }
  }
  unsigned w1_off;
  {
    const int v = wave * 64 + lane;
    const int row = v >> 3;
    const int chunk = (v & 7) ^ ((row >> 1) & 7);
    w1_off = static_cast<unsigned>((row * 576 + chunk * 8) * 2);
  }
  auto issue_tap = [&](int tap) { DmaToLds(w1_rsrc, w1_off, tap * 128, w1s + (tap % 3) * 4096 + wave * 512); };
  issue_tap(0);
  issue_tap(1);
  const int zero_off = (kBM + 2 * W + 2) * 64 + q * 8;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    if (tap < 8) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tap + 2 < 9) issue_tap(tap + 2);
    const int kh = tap / 3, kw = tap - kh * 3;
    const int hr = i_local + kh * W + kw;
    const int x_off = ((taps >> tap) & 1u) ? hr * 64 + (((q ^ hr) & 7) * 8) : zero_off;
    const _Float16* ws = w1s + (tap % 3) * 4096;
    if (a.ablate & 16) continue;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const half8 xf = *reinterpret_cast<const half8*>(halo + (x_off ^ (kk * 32)));
      half8 wf[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) wf[f] = *reinterpret_cast<const half8*>(ws + f * 1024 + (frag_off ^ (kk * 32)));
#pragma unroll
      for (int f = 0; f < 4; ++f) acc1[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[f], xf, acc1[f], 0, 0, 0);
    }
  }
  // the activation operand of the second product, straight from the accumulators: element j of
  // lane quarter q in K step s is middle channel 16 (2 s + j / 4) + 4 q + j % 4
  half8 bmid[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const half4 lo = PackRelu(acc1[2 * s], true), hi = PackRelu(acc1[2 * s + 1], true);
    bmid[s] = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
  if (a.ablate & 8) {
    if (bmid[0][0] == static_cast<_Float16>(12345.f)) a.y[tid] = bmid[1][3];  // keeps the K loop alive
    return;
  }
  LdsBarrier();  // every wave is done with the halo and the weight taps
  store_weights();
  if (tid < 32) *reinterpret_cast<uint4v*>(b2s + tid * 8) = b2v;
  if constexpr (kPhases > 1) load_weights(1);
  if constexpr (kPasses3 > 1) load_w3(0, 1);  // the second 64 rows of W3: before the first store, like every load
  LdsBarrier();

  // ---- second and third product, wave by wave ----------------------------------------------------
  // the wave's transposing tile: [16 pixels][32 channels], 64-byte rows whose 16-byte chunks are XOR-ed with
  // (pixel / 2) % 4 (pixels two apart share their banks on the write side)
  _Float16* const tile = tiles + wave * 512;
  const int tswz = (px >> 1) & 3;
  const int tw0 = px * 32 + (((q >> 1) ^ tswz) * 8) + (q & 1) * 4;        // channels 4 q .. + 3 of the half
  const int tw1 = px * 32 + (((2 + (q >> 1)) ^ tswz) * 8) + (q & 1) * 4;  // channels 16 + 4 q .. + 3
  const int tr = px * 32 + ((q ^ tswz) * 8);                              // channels 8 q .. + 7
  float4v acc3[kNI3];
  half4 b3[kPasses3 * kNI3];  // added at the very end: requested here, long there by then
#pragma unroll
  for (int f = 0; f < kNI3; ++f) acc3[f] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int f = 0; f < kPasses3 * kNI3; ++f)
    b3[f] = a.bias3 != nullptr ? *reinterpret_cast<const half4*>(a.bias3 + f * 16 + q * 4) : half4{0, 0, 0, 0};
  half8 outs[4][2];  // the finished output chunks: activation operand of the third product
#pragma unroll
  for (int ph = 0; ph < kPhases; ++ph) {
    if (ph > 0) {
      LdsBarrier();  // every wave has finished the previous phase's weights
      store_weights();
      LdsBarrier();
    }
#pragma unroll
    for (int cc = 0; cc < kCPP; ++cc) {
      const int c = ph * kCPP + cc;
      float4v acc2[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) acc2[f] = Widen(*reinterpret_cast<const half4*>(b2s + c * 64 + f * 16 + q * 4));
#pragma unroll
      for (int ks = 0; ks < kK2 / 64; ++ks) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          half8 wf[4];
#pragma unroll
          for (int f = 0; f < 4; ++f)
            wf[f] = *reinterpret_cast<const half8*>(w2s + (ks * kRows2 + cc * 64 + f * 16) * 64 + (frag_off ^ (s * 32)));
          const half8 b = ks == 0 ? bmid[s] : bxd[s];
          if (a.ablate & 32) continue;
#pragma unroll
          for (int f = 0; f < 4; ++f) acc2[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[f], b, acc2[f], 0, 0, 0);
        }
      }
      // transposition through the wave's own tile, 32 channels at a time: no barrier, LDS operations of a
      // wave run in order
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        *reinterpret_cast<half4*>(tile + tw0) = PackRelu(acc2[2 * s], false);
        *reinterpret_cast<half4*>(tile + tw1) = PackRelu(acc2[2 * s + 1], false);
        half8 v = *reinterpret_cast<const half8*>(tile + tr);
        v = v + res[c][s];  // packed fp16 adds (zeros without a residual)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = v[j] > static_cast<_Float16>(0) ? v[j] : static_cast<_Float16>(0);
        outs[c][s] = v;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4v, v), y_rsrc, static_cast<int>(y_off), (c * 64 + s * 32) * 2, 0);
      }
      // third product: the finished values are its activation operand as they stand
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (a.ablate & 32) break;
#pragma unroll
        for (int f = 0; f < kNI3; ++f) {
          const half8 wf = *reinterpret_cast<const half8*>(w3s + (cc * 64 + f * 16) * 64 + (frag_off ^ (s * 32)));
          acc3[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, outs[c][s], acc3[f], 0, 0, 0);
        }
      }
    }
  }
  // third result: + bias, ReLU, transposed 32 channels at a time through the wave's tile, 16-byte stores
  auto third_epilogue = [&](int pass) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      *reinterpret_cast<half4*>(tile + tw0) = PackRelu(acc3[2 * h] + Widen(b3[pass * kNI3 + 2 * h]), true);
      *reinterpret_cast<half4*>(tile + tw1) = PackRelu(acc3[2 * h + 1] + Widen(b3[pass * kNI3 + 2 * h + 1]), true);
      const half8 v = *reinterpret_cast<const half8*>(tile + tr);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4v, v), o3_rsrc, static_cast<int>(o3_off),
                                             (pass * 64 + h * 32) * 2, 0);
    }
  };
  third_epilogue(0);
  if constexpr (kPasses3 > 1) {
    // the other 64 outputs of the third product: its weights replace the first pass's, the operand is the
    // output chunks kept in registers
    LdsBarrier();  // every wave is done with the first 64 rows of W3
    store_w3();
    LdsBarrier();
#pragma unroll
    for (int f = 0; f < kNI3; ++f) acc3[f] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int f = 0; f < kNI3; ++f) {
          const half8 wf = *reinterpret_cast<const half8*>(w3s + (c * 64 + f * 16) * 64 + (frag_off ^ (s * 32)));
          acc3[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, outs[c][s], acc3[f], 0, 0, 0);
        }
      }
    }
    third_epilogue(1);
  }
}

template <int kN3, int kXS>
int LaunchRows(const RowsArgs& args, hipStream_t st) {
  constexpr size_t kLdsBytes = 72 * 1024 + 512;  // weights of a phase, eight 1 KiB tiles, 256 biases
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&BottleneckRowsKernel<kN3, kXS>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
  if (attr != hipSuccess) return -1000 - static_cast<int>(attr);
  const int blocks = (args.M + kBM - 1) / kBM;
  RowsArgs a = args;
  a.xcd_chunk = (blocks % 8 == 0) ? blocks / 8 : 0;
  hipLaunchKernelGGL((BottleneckRowsKernel<kN3, kXS>), dim3(blocks), dim3(kThreads), kLdsBytes, st, a);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -1000 - static_cast<int>(e);
}

}  // namespace

extern "C" {

// y = relu(conv1x1(relu(conv3x3(x, w1) + bias1), w2) (+ conv1x1(xd, wd)) + bias2 (+ residual)) and
// out3 = relu(conv1x1(y, w3) + bias3) for 64 input and middle channels, 256 output channels, stride 1,
// pad 1, image width up to 62.  x: [batch,h,w,64], w1: [64][3][3][64], w2: [256][64 (+64 with xd)] with
// its first 64 K columns in accumulator order, residual / y: [batch,h,w,256], w3: [n3][256] (n3 = 64 or
// 128), out3: [batch,h,w,n3], xd: [batch,h,w,64] or null (then no residual, n3 = 64).
int tca_bottleneck_rows_launch(const void* x, const void* w1, const void* bias1, const void* w2, const void* bias2,
                               const void* residual, void* y, int batch, int h, int w, int cout, const void* w3,
                               const void* bias3, void* out3, int n3, const void* xd, void* stream) {
  if (!x || !w1 || !bias1 || !w2 || !y || !w3 || !out3 || batch <= 0 || h <= 0 || w <= 0) return -1;
  if (cout != kN2 || (n3 != 64 && n3 != 128)) return -2;
  if (kBM + 2 * w + 2 > kHaloRows - 1) return -2;  // the halo and its row of zeros must fit the LDS rows
  const long long m_ll = static_cast<long long>(batch) * h * w;
  if (m_ll * kN2 * 2 >= 0x7FFFFFFFll) return -2;   // 32-bit buffer offsets
  if (xd != nullptr && (residual != nullptr || n3 != 64)) return -2;
  for (const void* p : {x, w1, w2, residual, static_cast<const void*>(y), w3, static_cast<const void*>(out3), xd})
    if (reinterpret_cast<uintptr_t>(p) & 15) return -2;
  RowsArgs a;
  a.x = static_cast<const _Float16*>(x);
  a.w1 = static_cast<const _Float16*>(w1);
  a.bias1 = static_cast<const _Float16*>(bias1);
  a.w2 = static_cast<const _Float16*>(w2);
  a.bias2 = static_cast<const _Float16*>(bias2);
  a.residual = static_cast<const _Float16*>(residual);
  a.y = static_cast<_Float16*>(y);
  a.w3 = static_cast<const _Float16*>(w3);
  a.bias3 = static_cast<const _Float16*>(bias3);
  a.out3 = static_cast<_Float16*>(out3);
  a.xd = static_cast<const _Float16*>(xd);
  a.M = static_cast<int>(m_ll);
  a.H = h;
  a.W = w;
  a.xcd_chunk = 0;
  static const int ablate = [] {
    const char* v = std::getenv("TCA_ROWS_ABLATE");
    return v ? std::atoi(v) : 0;
  }();
  a.ablate = ablate;
  auto st = static_cast<hipStream_t>(stream);
  if (xd != nullptr) return LaunchRows<64, 1>(a, st);
  return n3 == 64 ? LaunchRows<64, 0>(a, st) : LaunchRows<128, 0>(a, st);
}

}  // extern "C"
