// Convolution building blocks for NHWC fp16 inference on gfx950 (CDNA4).
//
//   ConvGemmKernel  a 1x1 convolution in NHWC IS a GEMM:
//                       Y[m][n] = act( sum_k X[row(m)][k] * W[n][k] + bias[n] (+ R[m][n]) )
//                   with m over the N*OH*OW output pixels, k over input and n
//                   over output channels.  Both operands are K-contiguous, so
//                   MFMA fragments are plain 16-byte reads.  Bias, the residual
//                   add and ReLU run in the epilogue on the fp32 accumulator:
//                   the activation is written to HBM exactly once instead of
//                   being re-read and re-written by two or three elementwise
//                   passes.  Stride-2 (down-sampling shortcut) only changes
//                   row(m).
//                   With kSpatial the same kernel is an implicit GEMM for RxR
//                   convolutions (3x3): K runs over (kh, kw, cin) - exactly the
//                   memory order of a channels-last weight - and every K step
//                   of 64 channels belongs to ONE tap, so staging the
//                   activation tile only shifts the source pixel per step and
//                   zero-fills rows whose tap falls into the padding.
//   BiasActKernel   the same epilogue as a stand-alone, 16-byte vectorised
//                   in-place pass for convolutions a library computed (3x3,
//                   7x7): one pass instead of bias-add, residual-add and ReLU
//                   passes.
//   BiasReluMaxPoolKernel  the stem's epilogue folded into its 3x3/2 max pooling.
//
// GEMM structure: 4 wavefronts (2x2), 8 (2x4) or 16 (4x4, 2x8: the 256-wide
// tiles, one block per CU), block tile BM x BN, K step
// 64, v_mfma_f32_16x16x32_f16.  The 8-wavefront 128x128 tile (64x32 per wave)
// keeps twice the waves per SIMD in flight for the same LDS footprint.  The weight tile is the MFMA A operand and the
// activation tile the B operand, i.e. the wave computes Y^T fragments: each
// lane then owns 4 CONSECUTIVE output channels of one pixel, which packs into
// one 8-byte LDS store for the epilogue transpose.  Tiles are staged by
// direct-to-LDS DMA (`buffer_load_dwordx4 ... lds`): no staging registers, no
// ds_write, zero fill of masked rows / padding taps by the buffer range check,
// and - because every per-lane quantity is precomputed and the per-step
// displacement is a scalar offset - three VALU instructions per activation
// vector and none per weight vector inside the K loop.  Two or three LDS
// stages, one raw s_barrier per K step, counted vmcnt.  A register-staged
// variant (global -> VGPR -> ds_write) is kept for A/B measurements and for
// tensors beyond the DMA path's 2 GiB offsets.  LDS rows are 128 bytes; the
// 16-byte chunk index is XOR-ed with (row>>1)&7, which makes every
// ds_read_b128 of a fragment (16 rows x 4 chunks per wave instruction)
// bank-conflict free for the instruction's four 16-lane groups.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <hip/hip_fp16.h>

#include <cstdint>

#include "kernels.h"

namespace {

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using half4 = __attribute__((ext_vector_type(4))) _Float16;
using ushort8 = __attribute__((ext_vector_type(8))) unsigned short;
using float4v = __attribute__((ext_vector_type(4))) float;

constexpr int kThreads = 256;
constexpr int kBK = 64;  // halfs per K step = 128-byte LDS rows

struct RowMap {  // output pixel index -> input pixel(s)
  int identity;  // 1: 1x1 stride 1, row(m) = m
  int ohw;       // OH*OW
  int ow;        // OW
  int in_h;
  int in_w;
  int stride;
  int pad;
  int ksize;     // R = S
  int cin;
};

// n / d for 0 <= n < 2^24 and d >= 7 (pixel indices by image / row sizes): a float estimate and two
// fix-ups, about ten vector instructions where the compiler's exact 32-bit division takes forty.
// The estimate is off by at most one: float(n) is exact, the quotient stays below 2^22.
__device__ __forceinline__ int SmallDiv(int n, int d, float inv_d) {
  int q = static_cast<int>(static_cast<float>(n) * inv_d);
  int r = n - q * d;
  if (r < 0) {
    --q;
    r += d;
  }
  if (r >= d) ++q;
  return q;
}

// (m / ohw, (m % ohw) / ow, (m % ohw) % ow)
__device__ __forceinline__ void SplitPixel(const RowMap& map, int m, bool small, int& img, int& oh, int& ow) {
  if (small) {
    img = SmallDiv(m, map.ohw, 1.0f / static_cast<float>(map.ohw));
    const int rem = m - img * map.ohw;
    oh = SmallDiv(rem, map.ow, 1.0f / static_cast<float>(map.ow));
    ow = rem - oh * map.ow;
  } else {
    img = m / map.ohw;
    const int rem = m - img * map.ohw;
    oh = rem / map.ow;
    ow = rem - oh * map.ow;
  }
}

__device__ __forceinline__ long long InRow(const RowMap& map, int m) {
  if (map.identity) return m;
  int img, oh, ow;
  SplitPixel(map, m, m < (1 << 24) && map.ow >= 7, img, oh, ow);
  return static_cast<long long>(img) * map.in_h * map.in_w + static_cast<long long>(oh) * map.stride * map.in_w +
         ow * map.stride;
}

// 16-byte chunk position inside a 128-byte LDS row
__device__ __forceinline__ int Swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

using lds_ptr_t = __attribute__((address_space(3))) void*;

// Byte offset no activation descriptor reaches (the host keeps them below
// 2 GiB): a lane that must contribute zeros - a row past M, a tap in the
// padding - issues its DMA with this offset and the buffer range check
// writes zeros into LDS for it.
constexpr unsigned kOutOfRange = 0x80000000u;

// One wave-wide 16-byte-per-lane copy global -> LDS without passing through
// registers: lane l's bytes land at `lds_base` (wave-uniform) + 16 * l and come
// from descriptor base + voffset (per lane) + soffset (wave-uniform).
__device__ __forceinline__ void DmaToLds(__amdgpu_buffer_rsrc_t rsrc, unsigned voffset, int soffset, _Float16* lds_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, reinterpret_cast<lds_ptr_t>(reinterpret_cast<uintptr_t>(lds_base)), 16,
                                           static_cast<int>(voffset), soffset, 0, 0);
}

// Workgroup barrier that orders LDS traffic only: this wave's LDS reads / writes have completed, vector
// memory operations (LDS DMA, loads, output stores) stay in flight across it.  __syncthreads() would wait
// for every one of them - store acknowledgements included - because a pending LDS DMA counts as an
// LDS write it has to fence.
__device__ __forceinline__ void LdsBarrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// Second, back-to-back GEMM of the fused form (kFuse): the convolution's output tile never
// leaves the CU - it is multiplied by a 1x1 convolution's weights straight from LDS.
struct Pointwise2 {
  const _Float16* w;     // [n][BN (+ 64 with xd)] : the 1x1 convolution that follows (its K is the first convolution's Cout)
  const _Float16* bias;  // [n] or null
  int n;                 // its output channels (multiple of 64)
  // kXS = 1: a second activation operand [M][64] (the block's INPUT) whose 64 channels extend the second
  // GEMM's K - the projection shortcut computed in the same accumulator instead of being read as a residual
  const _Float16* xd;
};

// Third GEMM of the fused form (kN3 > 0): the NEXT block's first 1x1 convolution, fed by the
// finished output chunks while they are still on the CU.
struct Pointwise3 {
  const _Float16* w;     // [kN3][n of Pointwise2]
  const _Float16* bias;  // [kN3] or null
  _Float16* out;         // [M][kN3] = relu(y x w^T + bias)
};

template <int BM, int BN, int kWN, bool kSpatial, bool kGlds, bool kFuse = false, int kN3 = 0, int kXS = 0, bool kHalo = false, int kWM = 2>
__global__ __launch_bounds__(64 * kWM * kWN, kFuse ? kWN : 1) void ConvGemmKernel(const _Float16* __restrict__ X, const _Float16* __restrict__ W,
                                                          const _Float16* __restrict__ bias,
                                                          const _Float16* __restrict__ residual,
                                                          _Float16* __restrict__ Y, int M, int N, int K, int relu,
                                                          RowMap map, int tiles_n, int xcd_chunk, unsigned x_bytes,
                                                          int stages, int splits, float* __restrict__ partials,
                                                          unsigned* __restrict__ arrivals, Pointwise2 pw2, Pointwise3 pw3) {
  constexpr int kW = kWM * kWN;              // wavefronts: kWM (2, or 4 for the 16-wave tiles) along m x kWN along n
  constexpr int kT = 64 * kW;                // threads
  constexpr int kAVecs = BM * 8 / kT;        // 16-byte vectors per thread and stage (activation tile)
  constexpr int kWVecs = BN * 8 / kT;        // (weight tile)
  constexpr int kMI = BM / (16 * kWM);       // 16-wide fragments per wave along m
  constexpr int kNI = BN / (16 * kWN);       // along n
  static_assert(kAVecs >= 1 && kWVecs >= 1 && kNI >= 1, "tile too small for the wave layout");
  // kHalo (3x3, stride 1, pad 1): the activation rows a tile's nine taps touch - the tile's own pixels plus
  // one image row and one pixel on either side, as consecutive rows of the flattened [pixels][cin] input -
  // are brought into LDS ONCE; tap (kh, kw) of output pixel m reads halo row m + kh*W + kw.  The K loop then
  // only streams weights.  Instead of nine activation tiles per block from L2 (and nine round trips) it is
  // one, 1.9x the tile at 128 pixels of a 56-wide image.
  static_assert(!kHalo || (kSpatial && kGlds), "the halo form is the DMA-staged implicit GEMM");
  constexpr int kStageHalfs = kHalo ? BN * kBK : (BM + BN) * kBK;
  constexpr int kCPitch = BN + 8;  // halfs; 16-byte aligned rows, spreads the transposed stores over banks
  extern __shared__ __attribute__((aligned(16))) _Float16 lds[];

  // XCD-aware tile order: consecutive block ids land on different XCDs, so give
  // each XCD a contiguous run of tiles (neighbouring n tiles share the X rows in L2)
  int bid = blockIdx.x;
  if (xcd_chunk > 0) bid = (bid & 7) * xcd_chunk + (bid >> 3);
  const int tile_m = bid / tiles_n;
  const int tile_n = bid - tile_m * tiles_n;
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  // the wave index as a scalar: LDS-DMA destinations and fragment bases stay in SGPR arithmetic
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / kWN;  // wave row (m)
  const int wn = wave % kWN;  // wave column (n)

  // ---- direct-to-LDS staging state -------------------------------------------------
  // The LDS image of one wave instruction is linear (lane l lands at base + 16*l), so
  // lane l LOADS the 16-byte chunk whose swizzled place it fills.  Everything per
  // lane is computed once: a byte offset per vector and, for RxR kernels, one bit per
  // tap saying whether that tap reads inside the image.  A K step then costs a mask
  // test and a select per activation vector and nothing per weight vector; the step's
  // tap / channel displacement is wave-uniform and rides in the scalar offset.
  unsigned a_off[kAVecs], a_taps[kAVecs], w_off[kWVecs];
  // halo form: LDS rows of the halo, validity bits of the nine taps per fragment row, the operand stages behind it
  const int halo_rows = kHalo ? ((BM + 2 * map.in_w + 2) / (8 * kW) + 1) * (8 * kW) : 0;  // >= 1 spare row: reads as zeros
  _Float16* const stage0 = lds + (kHalo ? (map.cin / kBK) * halo_rows * kBK : 0);
  unsigned h_taps[kMI];
  const bool small_pixels = M + BM < (1 << 24) && map.ow >= 7;  // wave-uniform: the cheap pixel-index division applies
  // elements the activation descriptor's base sits BEFORE X, so the offset of the
  // (possibly negative) top-left tap coordinate is never negative
  const int lead = kSpatial ? (map.pad * map.in_w + map.pad) * map.cin : 0;
  if constexpr (kHalo) {
#pragma unroll
    for (int mi = 0; mi < kMI; ++mi) {
      const int m = m0 + wm * (BM / kWM) + mi * 16 + (lane & 15);
      int img, oh, ow;
      SplitPixel(map, m, small_pixels, img, oh, ow);
      unsigned rows_ok = 0, cols_ok = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        rows_ok |= static_cast<unsigned>(static_cast<unsigned>(oh - 1 + k) < static_cast<unsigned>(map.in_h)) << k;
        cols_ok |= static_cast<unsigned>(static_cast<unsigned>(ow - 1 + k) < static_cast<unsigned>(map.in_w)) << k;
      }
      unsigned taps = 0;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
        if ((rows_ok >> kh) & 1u) taps |= cols_ok << (kh * 3);
      h_taps[mi] = m < M ? taps : 0u;
    }
  }
  if constexpr (kGlds) {
#pragma unroll
    for (int i = 0; i < kAVecs; ++i) {
      if constexpr (kHalo) break;
      const int v = (i * kW + wave) * 64 + lane;
      const int row = v >> 3;
      const int chunk = (v & 7) ^ ((row >> 1) & 7);
      const int m = m0 + row;
      if constexpr (kSpatial) {
        int img, oh, ow;
        SplitPixel(map, m, small_pixels, img, oh, ow);
        const int ih0 = oh * map.stride - map.pad;
        const int iw0 = ow * map.stride - map.pad;
        const long long pixel = (static_cast<long long>(img) * map.in_h + ih0) * map.in_w + iw0;
        a_off[i] = static_cast<unsigned>((pixel * map.cin + lead + chunk * 8) * 2);
        unsigned rows_ok = 0, cols_ok = 0;
        for (int k = 0; k < map.ksize; ++k) {
          rows_ok |= static_cast<unsigned>(static_cast<unsigned>(ih0 + k) < static_cast<unsigned>(map.in_h)) << k;
          cols_ok |= static_cast<unsigned>(static_cast<unsigned>(iw0 + k) < static_cast<unsigned>(map.in_w)) << k;
        }
        unsigned taps = 0;
        for (int kh = 0; kh < map.ksize; ++kh)
          if ((rows_ok >> kh) & 1u) taps |= cols_ok << (kh * map.ksize);
        a_taps[i] = m < M ? taps : 0u;
      } else {
        a_off[i] = m < M ? static_cast<unsigned>((InRow(map, m) * K + chunk * 8) * 2) : kOutOfRange;
        a_taps[i] = 0;
      }
    }
#pragma unroll
    for (int i = 0; i < kWVecs; ++i) {
      const int v = (i * kW + wave) * 64 + lane;
      const int row = v >> 3;
      const int chunk = (v & 7) ^ ((row >> 1) & 7);
      w_off[i] = static_cast<unsigned>(((n0 + row) * K + chunk * 8) * 2);  // N % BN == 0 (host checked)
    }
  }
  const __amdgpu_buffer_rsrc_t x_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(X) - lead, 0, static_cast<int>(x_bytes + 2u * lead), 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(W), 0, N * K * 2, 0x00020000);
  // Split-K: blockIdx.y takes a contiguous run of the K steps (host: every run is non-empty)
  int kt_begin = 0;
  int ktiles = K / kBK;
  if (splits > 1) {
    const int per = (ktiles + splits - 1) / splits;
    kt_begin = static_cast<int>(blockIdx.y) * per;
    ktiles = ktiles - kt_begin < per ? ktiles - kt_begin : per;
  }
  // position of the NEXT K step to issue (steps are issued strictly in order)
  int next_k0 = kt_begin * kBK, next_c0 = 0, next_kh = 0, next_kw = 0, next_tap = 0;
  if constexpr (kSpatial) {
    const int per_tap = map.cin / kBK;
    next_tap = kt_begin / per_tap;
    next_c0 = (kt_begin - next_tap * per_tap) * kBK;
    next_kh = next_tap / map.ksize;
    next_kw = next_tap - next_kh * map.ksize;
  }
  auto issue_tile = [&](int stage) {
    _Float16* base = stage0 + stage * kStageHalfs;
    if constexpr (kSpatial) {
      if constexpr (!kHalo) {
        const int soff = ((next_kh * map.in_w + next_kw) * map.cin + next_c0) * 2;
        const unsigned bit = 1u << next_tap;
#pragma unroll
        for (int i = 0; i < kAVecs; ++i)
          DmaToLds(x_rsrc, (a_taps[i] & bit) ? a_off[i] : kOutOfRange, soff, base + (i * kW + wave) * 512);
      }
      next_c0 += kBK;
      if (next_c0 == map.cin) {
        next_c0 = 0;
        ++next_tap;
        if (++next_kw == map.ksize) {
          next_kw = 0;
          ++next_kh;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < kAVecs; ++i) DmaToLds(x_rsrc, a_off[i], next_k0 * 2, base + (i * kW + wave) * 512);
    }
#pragma unroll
    for (int i = 0; i < kWVecs; ++i)
      DmaToLds(w_rsrc, w_off[i], next_k0 * 2, base + (kHalo ? 0 : BM * kBK) + (i * kW + wave) * 512);
    next_k0 += kBK;
  };
  // the halo: wave instruction j fills LDS rows 8j .. 8j+7 of a 64-channel slice (lane l: row 8j + l/8, the
  // 16-byte chunk whose place l%8 is under the halo's swizzle, chunk ^ (row & 7): conflict-free fragment
  // reads at every row shift); rows outside the tensor, and the spare rows, get zeros from the range check
  auto issue_halo = [&]() {
    const int slices = map.cin / kBK;
    const int first = m0 - map.in_w - 1;  // flattened input pixel of halo row 0
    const long long pixels = static_cast<long long>(M);  // stride 1, pad 1: as many input as output pixels
    for (int sl = 0; sl < slices; ++sl)
      for (int j = wave; j < halo_rows / 8; j += kW) {
        const int row = j * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (row & 7);
        const long long p = static_cast<long long>(first) + row;
        const bool ok = p >= 0 && p < pixels && row < BM + 2 * map.in_w + 2;
        DmaToLds(x_rsrc, ok ? static_cast<unsigned>((p * map.cin + lead + sl * kBK + chunk * 8) * 2) : kOutOfRange, 0,
                 lds + (sl * halo_rows + j * 8) * kBK);
      }
  };

  // ---- register staging (global -> VGPR -> LDS): the A/B alternative ---------------
  // vector v -> (row v/8, chunk v%8), stored swizzled
  const _Float16* a_src[kAVecs];  // 1x1: fixed row pointer; spatial: image base (+ chunk)
  int a_dst[kAVecs];
  int a_ih0[kAVecs], a_iw0[kAVecs];  // spatial only: top-left input coordinate of the output pixel
  const _Float16* w_src[kWVecs];
  int w_dst[kWVecs];
  const int ctiles = kSpatial ? map.cin / kBK : 1;  // K steps per tap
  if constexpr (!kGlds) {
#pragma unroll
    for (int i = 0; i < kAVecs; ++i) {
      const int v = tid + i * kT;
      const int row = v >> 3;
      const int chunk = v & 7;
      const int m = m0 + row;
      a_dst[i] = row * kBK + Swz(row, chunk) * 8;
      if constexpr (kSpatial) {
        const int img = m / map.ohw;
        const int rem = m - img * map.ohw;
        const int oh = rem / map.ow;
        const int ow = rem - oh * map.ow;
        a_src[i] = (m < M) ? X + static_cast<long long>(img) * map.in_h * map.in_w * map.cin + chunk * 8 : nullptr;
        a_ih0[i] = oh * map.stride - map.pad;
        a_iw0[i] = ow * map.stride - map.pad;
      } else {
        a_src[i] = (m < M) ? X + InRow(map, m) * K + chunk * 8 : nullptr;
        a_ih0[i] = a_iw0[i] = 0;
      }
    }
#pragma unroll
    for (int i = 0; i < kWVecs; ++i) {
      const int v = tid + i * kT;
      const int row = v >> 3;
      const int chunk = v & 7;
      w_src[i] = W + static_cast<long long>(n0 + row) * K + chunk * 8;  // N % BN == 0 (host checked)
      w_dst[i] = BM * kBK + row * kBK + Swz(row, chunk) * 8;
    }
  }

  half8 a_reg[kAVecs], w_reg[kWVecs];
  auto load_tile = [&](int kt) {
    const int k0 = kt * kBK;
    if constexpr (kSpatial) {
      const int tap = kt / ctiles;  // wave-uniform
      const int c0 = (kt - tap * ctiles) * kBK;
      const int kh = tap / map.ksize;
      const int kw = tap - kh * map.ksize;
#pragma unroll
      for (int i = 0; i < kAVecs; ++i) {
        const int ih = a_ih0[i] + kh, iw = a_iw0[i] + kw;
        const bool ok = a_src[i] != nullptr && static_cast<unsigned>(ih) < static_cast<unsigned>(map.in_h) &&
                        static_cast<unsigned>(iw) < static_cast<unsigned>(map.in_w);
        if (ok) a_reg[i] = *reinterpret_cast<const half8*>(a_src[i] + (static_cast<long long>(ih) * map.in_w + iw) * map.cin + c0);
        else a_reg[i] = half8{0, 0, 0, 0, 0, 0, 0, 0};
      }
    } else {
#pragma unroll
      for (int i = 0; i < kAVecs; ++i) {
        if (a_src[i] != nullptr) a_reg[i] = *reinterpret_cast<const half8*>(a_src[i] + k0);
        else a_reg[i] = half8{0, 0, 0, 0, 0, 0, 0, 0};
      }
    }
#pragma unroll
    for (int i = 0; i < kWVecs; ++i) w_reg[i] = *reinterpret_cast<const half8*>(w_src[i] + k0);
  };
  auto store_tile = [&](int stage) {
    _Float16* base = lds + stage * kStageHalfs;
#pragma unroll
    for (int i = 0; i < kAVecs; ++i) *reinterpret_cast<half8*>(base + a_dst[i]) = a_reg[i];
#pragma unroll
    for (int i = 0; i < kWVecs; ++i) *reinterpret_cast<half8*>(base + w_dst[i]) = w_reg[i];
  };

  float4v acc[kNI][kMI];
#pragma unroll
  for (int ni = 0; ni < kNI; ++ni)
#pragma unroll
    for (int mi = 0; mi < kMI; ++mi) acc[ni][mi] = float4v{0.f, 0.f, 0.f, 0.f};

  // fragment addressing: lane -> (row lane&15, 16-byte chunk lane>>4 [+4 for the second
  // half of the K step]); the row swizzle only depends on lane&15 because every
  // fragment starts at a multiple of 16 rows, and "+4 chunks" is an XOR of 32 halfs
  const int frag_row = lane & 15;
  const int frag_chunk = (lane >> 4) ^ ((frag_row >> 1) & 7);
  const int xf_off = (wm * (BM / kWM) + frag_row) * kBK + frag_chunk * 8;
  const int wf_off = BM * kBK + (wn * (BN / kWN) + frag_row) * kBK + frag_chunk * 8;
  // halo form: position of the K step being computed
  int cur_c0 = 0, cur_tap = 0, cur_toff = 0, cur_kw = 0;
  const int h_row = wm * (BM / kWM) + (lane & 15);
  auto compute_tile = [&](int stage) {
    const _Float16* base = stage0 + stage * kStageHalfs;
    if constexpr (kHalo) {
      const _Float16* slice = lds + (cur_c0 / kBK) * halo_rows * kBK;
      const unsigned bit = 1u << cur_tap;
      // a tap outside the image reads the spare (zero) row
      const int zero_off = (BM + 2 * map.in_w + 2) * kBK + (lane >> 4) * 8;
      // rows 16 apart share the swizzle term (row & 7): one address, then constant steps
      const int hr0 = h_row + cur_toff;
      const int off0 = hr0 * kBK + (((lane >> 4) ^ hr0) & 7) * 8;
      int x_off[kMI];
#pragma unroll
      for (int mi = 0; mi < kMI; ++mi) x_off[mi] = (h_taps[mi] & bit) ? off0 + mi * 16 * kBK : zero_off;
      const int w_off2 = (wn * (BN / kWN) + frag_row) * kBK + frag_chunk * 8;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        half8 xf[kMI], wf[kNI];
#pragma unroll
        for (int mi = 0; mi < kMI; ++mi) xf[mi] = *reinterpret_cast<const half8*>(slice + (x_off[mi] ^ (kk * 32)));
#pragma unroll
        for (int ni = 0; ni < kNI; ++ni) wf[ni] = *reinterpret_cast<const half8*>(base + ((w_off2 ^ (kk * 32)) + ni * 16 * kBK));
#pragma unroll
        for (int ni = 0; ni < kNI; ++ni)
#pragma unroll
          for (int mi = 0; mi < kMI; ++mi)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
      }
      cur_c0 += kBK;
      if (cur_c0 == map.cin) {
        cur_c0 = 0;
        ++cur_tap;
        ++cur_toff;
        if (++cur_kw == 3) {
          cur_kw = 0;
          cur_toff += map.in_w - 3;
        }
      }
      return;
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const _Float16* xs = base + (xf_off ^ (kk * 32));
      const _Float16* ws = base + (wf_off ^ (kk * 32));
      half8 xf[kMI], wf[kNI];
#pragma unroll
      for (int mi = 0; mi < kMI; ++mi) xf[mi] = *reinterpret_cast<const half8*>(xs + mi * 16 * kBK);
#pragma unroll
      for (int ni = 0; ni < kNI; ++ni) wf[ni] = *reinterpret_cast<const half8*>(ws + ni * 16 * kBK);
#pragma unroll
      for (int ni = 0; ni < kNI; ++ni)
#pragma unroll
        for (int mi = 0; mi < kMI; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
    }
  };

  // ---- fused form: everything the second / third GEMM will need first, requested before the K loop ----
  // Plain register loads (no LDS DMA in the fused phase): they complete in issue order beside the K loop's
  // DMA and are long there when the loop ends, and the compiler's counted waits see all of them.
  using uint4v = __attribute__((ext_vector_type(4))) unsigned;
  constexpr int kSlabs = kFuse ? BN / kBK + kXS : 1;  // K steps of the second GEMM: the first result's channels (+ the extra operand's)
  constexpr int kBN2 = 64;
  constexpr int kNI2 = kFuse ? kBN2 / (16 * kWN) : 1;
  constexpr int kW2Vecs = kFuse ? kBN2 * 8 * kSlabs / kT : 1;
  constexpr int kNI3 = kN3 > 0 ? kN3 / (16 * kWN) : 1;
  constexpr int kW3Vecs = kN3 > 0 ? kN3 * 8 / kT : 1;
  constexpr int kRowVecs2 = kBN2 / 8;
  constexpr int kOutVecs2 = kFuse ? BM * kRowVecs2 / kT : 1;
  static_assert(!kFuse || (kNI2 >= 1 && BN % kBK == 0 && kW2Vecs >= 1 && kOutVecs2 >= 1),
                "wave layout too wide for the 64-channel chunks of the fused GEMM");
  static_assert(kN3 == 0 || (kN3 % (16 * kWN) == 0 && kN3 * 8 % kT == 0), "third GEMM: channel count does not fit the wave layout");
  const int N2 = pw2.n;
  unsigned w2_off[kW2Vecs], w3_off[kW3Vecs];
  uint4v w2r[kW2Vecs], w3r[kW3Vecs];  // the next weight chunks, on their way to LDS
  half8 res_r[kOutVecs2];            // residual rows of the next chunk
  half4 b1r[kNI], b2r[kNI2], b3r[kNI3];
  __amdgpu_buffer_rsrc_t w2_rsrc = w_rsrc, w3_rsrc = w_rsrc;
  auto load_w2 = [&](int chunk_index) {
#pragma unroll
    for (int i = 0; i < kW2Vecs; ++i)
      w2r[i] = __builtin_amdgcn_raw_buffer_load_b128(w2_rsrc, static_cast<int>(w2_off[i]),
                                                     chunk_index * kBN2 * (kSlabs * kBK) * 2, 0);
  };
  auto load_w3 = [&](int chunk_index) {
#pragma unroll
    for (int i = 0; i < kW3Vecs; ++i)
      w3r[i] = __builtin_amdgcn_raw_buffer_load_b128(w3_rsrc, static_cast<int>(w3_off[i]), chunk_index * kBN2 * 2, 0);
  };
  auto load_res = [&](int chunk_index) {  // rows past M: the last row, unused
#pragma unroll
    for (int i = 0; i < kOutVecs2; ++i) {
      const int v = tid + i * kT;
      const int row = v / kRowVecs2;
      const int cv = v - row * kRowVecs2;
      const int m = m0 + row < M ? m0 + row : M - 1;
      res_r[i] = *reinterpret_cast<const half8*>(residual + static_cast<long long>(m) * N2 + chunk_index * kBN2 + cv * 8);
    }
  };
  auto load_bias2 = [&](int chunk_index) {
#pragma unroll
    for (int ni = 0; ni < kNI2; ++ni)
      b2r[ni] = *reinterpret_cast<const half4*>(pw2.bias + chunk_index * kBN2 + wn * (kBN2 / kWN) + ni * 16 + (lane >> 4) * 4);
  };
  // With the 128x128 first tile the K loop has no registers to spare for them (it would cost a block per
  // CU): that form asks right after the K loop instead.
  constexpr bool kEarlyPrefetch = BM * BN < 128 * 128;
  auto prefetch_first = [&]() {
    w2_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(pw2.w), 0, N2 * (kSlabs * kBK) * 2, 0x00020000);
#pragma unroll
    for (int i = 0; i < kW2Vecs; ++i) {
      const int v = (i * kW + wave) * 64 + lane;
      const int slab = v >> 9;  // 512 vectors (64 rows x 8 chunks) per slab
      const int row = (v & 511) >> 3;
      const int chunk = (v & 7) ^ ((row >> 1) & 7);
      w2_off[i] = static_cast<unsigned>((row * (kSlabs * kBK) + slab * kBK + chunk * 8) * 2);
    }
    load_w2(0);
    if constexpr (kN3 > 0) {
      w3_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(pw3.w), 0, kN3 * N2 * 2, 0x00020000);
#pragma unroll
      for (int i = 0; i < kW3Vecs; ++i) {
        const int v = (i * kW + wave) * 64 + lane;
        const int row = v >> 3;
        const int chunk = (v & 7) ^ ((row >> 1) & 7);
        w3_off[i] = static_cast<unsigned>((row * N2 + chunk * 8) * 2);
      }
      load_w3(0);
#pragma unroll
      for (int ni = 0; ni < kNI3; ++ni)
        b3r[ni] = pw3.bias != nullptr
                      ? *reinterpret_cast<const half4*>(pw3.bias + wn * (kN3 / kWN) + ni * 16 + (lane >> 4) * 4)
                      : half4{0, 0, 0, 0};
    }
    if (residual != nullptr) load_res(0);
    if (pw2.bias != nullptr) load_bias2(0);
#pragma unroll
    for (int ni = 0; ni < kNI; ++ni)
      b1r[ni] = bias != nullptr ? *reinterpret_cast<const half4*>(bias + wn * (BN / kWN) + ni * 16 + (lane >> 4) * 4)
                                : half4{0, 0, 0, 0};
  };
  if constexpr (kFuse && kEarlyPrefetch) prefetch_first();

  if constexpr (kGlds) {
    // `stages` LDS stages, the DMA runs stages-1 K steps ahead.  Per step: wait until
    // THIS step's DMA has landed (with three stages a counted vmcnt leaves the next
    // step's loads in flight), one raw barrier (a plain __syncthreads() would drain the
    // DMA queue), then refill the stage every wave finished reading one step ago.
    constexpr int kPerTile = kHalo ? kWVecs : kAVecs + kWVecs;  // DMA instructions a thread issues per K step
    const int ahead = stages > 1 ? stages - 1 : 1;
    if constexpr (kHalo) issue_halo();  // older than every weight tile: landed when the first of them has
    for (int p = 0; p < ahead && p < ktiles; ++p) issue_tile(p);
    int stage = 0;
    int refill = stages > 1 ? ahead : 0;
    for (int kt = 0; kt < ktiles; ++kt) {
      // K steps issued beyond this one (they may stay in flight: loads complete in issue order)
      const int issued = kt + ahead < ktiles ? kt + ahead : ktiles;
      const int beyond = issued - (kt + 1);
      if (beyond >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * kPerTile) : "memory");
      else if (beyond == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * kPerTile) : "memory");
      else if (beyond == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPerTile) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + ahead < ktiles) {
        issue_tile(refill);
        refill = refill + 1 == stages ? 0 : refill + 1;
      }
      compute_tile(stage);
      stage = stage + 1 == stages ? 0 : stage + 1;
    }
  } else {
    load_tile(kt_begin);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < ktiles; ++kt) {
      const bool more = kt + 1 < ktiles;
      if (more) load_tile(kt_begin + kt + 1);  // in flight while the MFMAs below run
      compute_tile(stages == 2 ? (kt & 1) : 0);
      if (more) {
        store_tile((kt + 1) & 1);  // the other stage: last read one iteration ago, behind a barrier
        __syncthreads();
      }
    }
  }
  __syncthreads();  // every wave is done reading the operand stages: reuse them for the output tile

  if (splits > 1) {
    // Split-K fix-up in the same launch: every block parks its fp32 partial tile in the
    // register layout (coalesced 8-byte stores per lane and fragment), then takes a ticket
    // for its output tile; the block that draws the last ticket adds the partials up in
    // split order (bit-reproducible whichever block that is) and runs the epilogue.  The
    // ticket counter is back at zero when the launch ends.
    //
    // Blocks of one tile run on different XCDs, whose L2s are not coherent with each other
    // for ordinary accesses.  A device-scope fence would make them so by writing back the
    // whole L2 - tens of microseconds with the previous layer's output still dirty in it.
    // Instead the partials themselves move with device-scope relaxed atomic stores and
    // loads (write-through / cache-bypassing accesses), ordered against the ticket by
    // waiting for the stores' acknowledgements.
    constexpr int kFrags = kNI * kMI;
    using u64 = unsigned long long;
    u64* tile_partials = reinterpret_cast<u64*>(partials) + static_cast<size_t>(bid) * splits * (kFrags * kT * 2);
    u64* mine = tile_partials + static_cast<size_t>(blockIdx.y) * (kFrags * kT * 2);
#pragma unroll
    for (int ni = 0; ni < kNI; ++ni)
#pragma unroll
      for (int mi = 0; mi < kMI; ++mi) {
        const float4v a = acc[ni][mi];
        u64* dst = mine + ((ni * kMI + mi) * 2) * kT + tid;
        __hip_atomic_store(dst, (static_cast<u64>(__float_as_uint(a[1])) << 32) | __float_as_uint(a[0]), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + kT, (static_cast<u64>(__float_as_uint(a[3])) << 32) | __float_as_uint(a[2]),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this lane's partial has reached device-coherent memory
    __syncthreads();                                  // ... and so has every lane's
    // the verdict travels through the first word of the (now idle) operand stages: a static
    // __shared__ variable would count against the 160 KiB the largest tiles ask for
    int* verdict = reinterpret_cast<int*>(lds);
    if (tid == 0) {
      const unsigned ticket = __hip_atomic_fetch_add(&arrivals[bid], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == static_cast<unsigned>(splits - 1);
      *verdict = last;
      // every block of this tile has drawn: ready for the next launch
      if (last) __hip_atomic_store(&arrivals[bid], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int is_last = *verdict;
    __syncthreads();  // everyone has read it before the epilogue below overwrites the stages
    if (!is_last) return;
#pragma unroll
    for (int ni = 0; ni < kNI; ++ni)
#pragma unroll
      for (int mi = 0; mi < kMI; ++mi) acc[ni][mi] = float4v{0.f, 0.f, 0.f, 0.f};
    for (int sp = 0; sp < splits; ++sp) {
      const u64* part = tile_partials + static_cast<size_t>(sp) * (kFrags * kT * 2);
#pragma unroll
      for (int ni = 0; ni < kNI; ++ni)
#pragma unroll
        for (int mi = 0; mi < kMI; ++mi) {
          const u64* src = part + ((ni * kMI + mi) * 2) * kT + tid;
          const u64 lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const u64 hi = __hip_atomic_load(src + kT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          acc[ni][mi][0] += __uint_as_float(static_cast<unsigned>(lo));
          acc[ni][mi][1] += __uint_as_float(static_cast<unsigned>(lo >> 32));
          acc[ni][mi][2] += __uint_as_float(static_cast<unsigned>(hi));
          acc[ni][mi][3] += __uint_as_float(static_cast<unsigned>(hi >> 32));
        }
    }
  }

  if constexpr (kFuse) {
    if constexpr (!kEarlyPrefetch) prefetch_first();
    // ---- fused 1x1 convolution: Y = act(relu(conv + bias) x W2^T + bias2 (+ residual)) ----
    // The first result tile [BM pixels][BN channels] (BN = all of its channels: tiles_n == 1)
    // goes to LDS in the operand layout of the K loop above - one [BM][64] swizzled slab per
    // 64 channels - and serves as the activation operand of a second GEMM whose K is BN.
    // Its weights come 64 output channels at a time: each chunk is loaded into registers one
    // chunk early and written to ONE LDS buffer as soon as the previous chunk's MFMAs are done
    // with it (a second buffer would push the 128-channel form past the LDS budget of two
    // blocks per CU); every 64-channel chunk runs the usual transposing epilogue.  The
    // intermediate activation is never written to memory and never read back.  Nothing in the
    // chunk loop waits for a memory round trip it has just started: weights, residual rows and
    // biases of chunk c + 1 are requested while chunk c computes, the output stores are never
    // waited for (the barriers only order LDS traffic).
    _Float16* ts = lds;                              // [kSlabs][BM][kBK]
    _Float16* w2s = lds + BM * kBK * kSlabs;         // [kSlabs][kBN2][kBK]
    // [BM][kBK] in the operand-slab layout (128-byte rows, 16-byte chunks XOR (row >> 1) & 7): the
    // transposing tile of a chunk's epilogue and, once its owner threads have put the finished values
    // back in place, the activation operand of the third GEMM - one buffer, no extra pass
    _Float16* cs2 = w2s + kSlabs * kBN2 * kBK;
    _Float16* ys = cs2;
    _Float16* w3s = cs2 + BM * kBK;                  // [kN3][kBK]
    float4v acc3[kNI3][kMI];
    if constexpr (kN3 > 0) {
#pragma unroll
      for (int ni = 0; ni < kNI3; ++ni)
#pragma unroll
        for (int mi = 0; mi < kMI; ++mi) acc3[ni][mi] = float4v{0.f, 0.f, 0.f, 0.f};
    }
    // weight vector v of a chunk sits at 16 * v bytes of its buffer; the thread that loaded it stores it
    auto store_w2 = [&]() {
#pragma unroll
      for (int i = 0; i < kW2Vecs; ++i) *reinterpret_cast<uint4v*>(w2s + (i * kW + wave) * 512 + lane * 8) = w2r[i];
    };
    auto store_w3 = [&]() {
#pragma unroll
      for (int i = 0; i < kW3Vecs; ++i) *reinterpret_cast<uint4v*>(w3s + (i * kW + wave) * 512 + lane * 8) = w3r[i];
    };
    if constexpr (kXS > 0) {
      // the extra operand slab: rows m0.. of the block input, 64 channels, straight into the slab after the first result's
      const __amdgpu_buffer_rsrc_t xd_rsrc = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<_Float16*>(pw2.xd), 0, static_cast<int>(static_cast<unsigned>(M) * kBK * 2u), 0x00020000);
#pragma unroll
      for (int i = 0; i < kAVecs; ++i) {
        const int v = (i * kW + wave) * 64 + lane;
        const int row = v >> 3;
        const int chunk = (v & 7) ^ ((row >> 1) & 7);
        const int m = m0 + row;
        DmaToLds(xd_rsrc, m < M ? static_cast<unsigned>((m * kBK + chunk * 8) * 2) : kOutOfRange, 0,
                 ts + (BN / kBK) * (BM * kBK) + (i * kW + wave) * 512);
      }
    }
    // first result (+bias, ReLU) -> operand slabs
#pragma unroll
    for (int ni = 0; ni < kNI; ++ni) {
      const int n_local = wn * (BN / kWN) + ni * 16 + (lane >> 4) * 4;
      float b[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) b[r] = static_cast<float>(b1r[ni][r]);
#pragma unroll
      for (int mi = 0; mi < kMI; ++mi) {
        const int m_local = wm * (BM / kWM) + mi * 16 + (lane & 15);
        half4 packed;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[ni][mi][r] + b[r];
          packed[r] = static_cast<_Float16>(v > 0.f ? v : 0.f);
        }
        const int chunk = ((n_local & 63) >> 3) ^ ((m_local >> 1) & 7);
        *reinterpret_cast<half4*>(ts + (n_local >> 6) * (BM * kBK) + m_local * kBK + chunk * 8 + (n_local & 7)) = packed;
      }
    }
    store_w2();  // chunk 0's weights (requested before the K loop)
    if constexpr (kXS > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the extra operand slab has landed
    const int w2f_off = (wn * (kBN2 / kWN) + frag_row) * kBK + frag_chunk * 8;
    const int chunks = N2 / kBN2;
    if (chunks > 1) load_w2(1);
    // Output stores go through a buffer descriptor: a row past M gets an out-of-range offset and its
    // store is dropped, so every wave issues the same store instructions per chunk, without branches.
    const __amdgpu_buffer_rsrc_t y_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(Y, 0, static_cast<int>(static_cast<unsigned>(M) * N2 * 2u), 0x00020000);
    for (int c = 0; c < chunks; ++c) {
      LdsBarrier();  // chunk c's weights (and, the first time, the operand slabs) are in LDS; the third GEMM is done with its buffers
      if constexpr (kN3 > 0) {
        // K columns c*64.. of the next convolution's weights; consumed after this chunk's epilogue
        store_w3();
        if (c + 1 < chunks) load_w3(c + 1);
      }
      float4v acc2[kNI2][kMI];
#pragma unroll
      for (int ni = 0; ni < kNI2; ++ni)
#pragma unroll
        for (int mi = 0; mi < kMI; ++mi) acc2[ni][mi] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int sl = 0; sl < kSlabs; ++sl) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const _Float16* xs = ts + sl * (BM * kBK) + (xf_off ^ (kk * 32));
          const _Float16* ws = w2s + sl * (kBN2 * kBK) + (w2f_off ^ (kk * 32));
          half8 xf[kMI], wf[kNI2];
#pragma unroll
          for (int mi = 0; mi < kMI; ++mi) xf[mi] = *reinterpret_cast<const half8*>(xs + mi * 16 * kBK);
#pragma unroll
          for (int ni = 0; ni < kNI2; ++ni) wf[ni] = *reinterpret_cast<const half8*>(ws + ni * 16 * kBK);
#pragma unroll
          for (int ni = 0; ni < kNI2; ++ni)
#pragma unroll
            for (int mi = 0; mi < kMI; ++mi)
              acc2[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni], xf[mi], acc2[ni][mi], 0, 0, 0);
        }
      }
      // chunk epilogue: +bias2 -> cs2 (transposed) -> residual, ReLU, coalesced 16-byte stores
#pragma unroll
      for (int ni = 0; ni < kNI2; ++ni) {
        const int n_local = wn * (kBN2 / kWN) + ni * 16 + (lane >> 4) * 4;
        float b[4] = {0.f, 0.f, 0.f, 0.f};
        if (pw2.bias != nullptr) {
#pragma unroll
          for (int r = 0; r < 4; ++r) b[r] = static_cast<float>(b2r[ni][r]);
        }
#pragma unroll
        for (int mi = 0; mi < kMI; ++mi) {
          const int m_local = wm * (BM / kWM) + mi * 16 + (lane & 15);
          half4 packed;
#pragma unroll
          for (int r = 0; r < 4; ++r) packed[r] = static_cast<_Float16>(acc2[ni][mi][r] + b[r]);
          *reinterpret_cast<half4*>(cs2 + m_local * kBK + ((((n_local >> 3) ^ ((m_local >> 1) & 7)) << 3) + (n_local & 7))) = packed;
        }
      }
      LdsBarrier();  // the output chunk is in cs2, and every wave is done reading the weight buffer
      if (c + 1 < chunks) {
        store_w2();  // requested one chunk ago
        if (c + 2 < chunks) load_w2(c + 2);
      }
      half8 vals[kOutVecs2];
      unsigned y_off[kOutVecs2];
#pragma unroll
      for (int i = 0; i < kOutVecs2; ++i) {
        const int v = tid + i * kT;
        const int row = v / kRowVecs2;
        const int cv = v - row * kRowVecs2;
        const int m = m0 + row;
        _Float16* slot = cs2 + row * kBK + ((cv ^ ((row >> 1) & 7)) * 8);
        half8 val = *reinterpret_cast<const half8*>(slot);
        if (residual != nullptr) val = val + res_r[i];  // packed fp16 adds: the same rounding as adding in fp32
        if (relu) {
#pragma unroll
          for (int j = 0; j < 8; ++j) val[j] = val[j] > static_cast<_Float16>(0) ? val[j] : static_cast<_Float16>(0);
        }
        vals[i] = val;
        y_off[i] = m < M ? (static_cast<unsigned>(m) * N2 + c * kBN2 + cv * 8) * 2u : kOutOfRange;
        if constexpr (kN3 > 0) *reinterpret_cast<half8*>(slot) = val;  // finished value back in its own slot (rows past M: unused)
      }
      if (c + 1 < chunks) {  // what the next chunk's epilogue adds: on its way while that chunk's MFMAs run
        if (residual != nullptr) load_res(c + 1);
        if (pw2.bias != nullptr) load_bias2(c + 1);
      }
      if constexpr (kN3 > 0) {
        LdsBarrier();  // the slab is complete (the third GEMM's weights were stored two barriers ago)
        const int w3f_off = (wn * (kN3 / kWN) + frag_row) * kBK + frag_chunk * 8;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const _Float16* xs = ys + (xf_off ^ (kk * 32));
          const _Float16* ws = w3s + (w3f_off ^ (kk * 32));
          half8 xf[kMI], wf[kNI3];
#pragma unroll
          for (int mi = 0; mi < kMI; ++mi) xf[mi] = *reinterpret_cast<const half8*>(xs + mi * 16 * kBK);
#pragma unroll
          for (int ni = 0; ni < kNI3; ++ni) wf[ni] = *reinterpret_cast<const half8*>(ws + ni * 16 * kBK);
#pragma unroll
          for (int ni = 0; ni < kNI3; ++ni)
#pragma unroll
            for (int mi = 0; mi < kMI; ++mi)
              acc3[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni], xf[mi], acc3[ni][mi], 0, 0, 0);
        }
      }
      // last in the chunk, and nothing ever waits for them
#pragma unroll
      for (int i = 0; i < kOutVecs2; ++i)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4v, vals[i]), y_rsrc, static_cast<int>(y_off[i]), 0, 0);
      // the next iteration's barrier orders these cs2 / slab reads before its writes
    }
    if constexpr (kN3 > 0) {
      // third result: +bias, ReLU -> transposed through LDS (everything above is idle now) -> coalesced stores
      constexpr int kCPitch3 = kN3 + 8;
      _Float16* cs3 = lds;
      LdsBarrier();
#pragma unroll
      for (int ni = 0; ni < kNI3; ++ni) {
        const int n_local = wn * (kN3 / kWN) + ni * 16 + (lane >> 4) * 4;
        float b[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) b[r] = static_cast<float>(b3r[ni][r]);
#pragma unroll
        for (int mi = 0; mi < kMI; ++mi) {
          const int m_local = wm * (BM / kWM) + mi * 16 + (lane & 15);
          half4 packed;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = acc3[ni][mi][r] + b[r];
            packed[r] = static_cast<_Float16>(v > 0.f ? v : 0.f);
          }
          *reinterpret_cast<half4*>(cs3 + m_local * kCPitch3 + n_local) = packed;
        }
      }
      LdsBarrier();
      constexpr int kRowVecs3 = kN3 / 8;
      constexpr int kOutVecs3 = BM * kRowVecs3 / kT;
#pragma unroll
      for (int i = 0; i < kOutVecs3; ++i) {
        const int v = tid + i * kT;
        const int row = v / kRowVecs3;
        const int cv = v - row * kRowVecs3;
        const int m = m0 + row;
        if (m >= M) continue;
        *reinterpret_cast<half8*>(pw3.out + static_cast<long long>(m) * kN3 + cv * 8) =
            *reinterpret_cast<const half8*>(cs3 + row * kCPitch3 + cv * 8);
      }
    }
    return;
  }

  // accumulator (+bias, fp32) -> LDS as fp16 [BM][kCPitch]; D^T layout: lane holds
  // pixel m = lane&15 and channels n = 4*(lane>>4) .. +3 of each fragment
  _Float16* cs = lds;
#pragma unroll
  for (int ni = 0; ni < kNI; ++ni) {
    const int n_local = wn * (BN / kWN) + ni * 16 + (lane >> 4) * 4;
    float b[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias != nullptr) {
      const half4 bv = *reinterpret_cast<const half4*>(bias + n0 + n_local);
#pragma unroll
      for (int r = 0; r < 4; ++r) b[r] = static_cast<float>(bv[r]);
    }
#pragma unroll
    for (int mi = 0; mi < kMI; ++mi) {
      const int m_local = wm * (BM / kWM) + mi * 16 + (lane & 15);
      half4 packed;
#pragma unroll
      for (int r = 0; r < 4; ++r) packed[r] = static_cast<_Float16>(acc[ni][mi][r] + b[r]);
      *reinterpret_cast<half4*>(cs + m_local * kCPitch + n_local) = packed;
    }
  }
  __syncthreads();

  // coalesced 16-byte epilogue: residual add, ReLU, store
  constexpr int kRowVecs = BN / 8;
  constexpr int kOutVecs = BM * kRowVecs / kT;
#pragma unroll
  for (int i = 0; i < kOutVecs; ++i) {
    const int v = tid + i * kT;
    const int row = v / kRowVecs;
    const int cv = v - row * kRowVecs;
    const int m = m0 + row;
    if (m >= M) continue;
    half8 val = *reinterpret_cast<const half8*>(cs + row * kCPitch + cv * 8);
    const long long off = static_cast<long long>(m) * N + n0 + cv * 8;
    if (residual != nullptr) {
      const half8 res = *reinterpret_cast<const half8*>(residual + off);
#pragma unroll
      for (int j = 0; j < 8; ++j) val[j] = static_cast<_Float16>(static_cast<float>(val[j]) + static_cast<float>(res[j]));
    }
    if (relu) {
#pragma unroll
      for (int j = 0; j < 8; ++j) val[j] = val[j] > static_cast<_Float16>(0) ? val[j] : static_cast<_Float16>(0);
    }
    *reinterpret_cast<half8*>(Y + off) = val;
  }
}

// y[m][c] = act(y[m][c] + bias[c] (+ residual[m][c])) in place, 8 halfs per thread.
__global__ __launch_bounds__(kThreads) void BiasActKernel(_Float16* __restrict__ y, const _Float16* __restrict__ bias,
                                                          const _Float16* __restrict__ residual, long long vecs,
                                                          int channel_vecs, int relu) {
  const long long stride = static_cast<long long>(gridDim.x) * kThreads;
  for (long long v = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x; v < vecs; v += stride) {
    half8 val = *reinterpret_cast<const half8*>(y + v * 8);
    const int cv = static_cast<int>(v % channel_vecs);
    const half8 b = *reinterpret_cast<const half8*>(bias + cv * 8);
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = static_cast<float>(val[j]) + static_cast<float>(b[j]);
    if (residual != nullptr) {
      const half8 r = *reinterpret_cast<const half8*>(residual + v * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] += static_cast<float>(r[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (relu && f[j] < 0.f) f[j] = 0.f;
      val[j] = static_cast<_Float16>(f[j]);
    }
    *reinterpret_cast<half8*>(y + v * 8) = val;
  }
}

// out[n][oh][ow][c] = relu(max over the 3x3 stride-2 pad-1 window of in[n][.][.][c] + bias[c]).
// Adding the bias and ReLU commute with the window maximum (both are monotonic), so
// the stem's activation is read ONCE at full resolution and only the pooled quarter
// is written - instead of an epilogue pass (read + write) followed by the pooling
// pass.  One thread per output pixel and 8 channels; 16-byte accesses; the
// overlapping window columns of neighbouring threads hit in L1/L2.
__global__ __launch_bounds__(kThreads) void BiasReluMaxPoolKernel(const _Float16* __restrict__ x,
                                                                  const _Float16* __restrict__ bias,
                                                                  _Float16* __restrict__ y, int in_h, int in_w,
                                                                  int out_h, int out_w, int channel_vecs,
                                                                  long long out_vecs) {
  const long long stride = static_cast<long long>(gridDim.x) * kThreads;
  for (long long v = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x; v < out_vecs; v += stride) {
    const int cv = static_cast<int>(v % channel_vecs);
    long long pix = v / channel_vecs;
    const int ow = static_cast<int>(pix % out_w);
    pix /= out_w;
    const int oh = static_cast<int>(pix % out_h);
    const long long img = pix / out_h;
    const _Float16* base = x + (img * in_h * in_w) * channel_vecs * 8 + cv * 8;
    half8 best;
#pragma unroll
    for (int j = 0; j < 8; ++j) best[j] = static_cast<_Float16>(-65504.f);
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {
      const int ih = oh * 2 - 1 + dh;
      if (static_cast<unsigned>(ih) >= static_cast<unsigned>(in_h)) continue;
#pragma unroll
      for (int dw = 0; dw < 3; ++dw) {
        const int iw = ow * 2 - 1 + dw;
        if (static_cast<unsigned>(iw) >= static_cast<unsigned>(in_w)) continue;
        const half8 val = *reinterpret_cast<const half8*>(base + (static_cast<long long>(ih) * in_w + iw) * channel_vecs * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) best[j] = val[j] > best[j] ? val[j] : best[j];
      }
    }
    const half8 b = *reinterpret_cast<const half8*>(bias + cv * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = static_cast<float>(best[j]) + static_cast<float>(b[j]);
      best[j] = static_cast<_Float16>(f > 0.f ? f : 0.f);
    }
    *reinterpret_cast<half8*>(y + v * 8) = best;
  }
}

// ---------------------------------------------------------------------------
// Network stem in ONE kernel: 7x7 stride-2 pad-3 convolution over 3 input
// channels + bias + ReLU + 3x3 stride-2 pad-1 max pooling.
//
//   x   [B][3][H][W]   fp16, plain NCHW - the layout clients send, so no layout
//                      conversion pass in front of the network
//   wk  [64][192]      fp16 weights with K ordered (c, kh, kw8): kw padded 7 -> 8
//                      with the zero in FRONT (kw8 = kw + 1) and K padded 168 -> 192
//                      with zeros, so that the 8 K values one MFMA lane feeds are 8
//                      CONSECUTIVE input pixels of one input row (one (c, kh) pair)
//                      starting at an EVEN column: aligned dword reads end to end
//   y   [B][PH][PW][64] fp16 NHWC, the pooled activation
//
// A block produces an 8x8 tile of pooled pixels for all 64 channels.  That needs
// a 17x17 tile of convolution outputs, which needs a 39x39 patch of each input
// plane: the patch is staged in LDS once (zero filled outside the image), the
// whole weight matrix lives in registers as MFMA A fragments (24 fragments per
// lane), and every group of 16 convolution pixels is 6 K steps x 4 channel
// fragments of v_mfma_f32_16x16x32_f16 whose B fragments are read straight from
// the patch.  Bias + ReLU are applied on the accumulators, the convolution tile
// goes to LDS as fp16 and the pooled maxima are written with 16-byte stores.
// The full-resolution convolution output (205 MB at batch 128) never exists in
// memory.  Post-ReLU values are >= 0, so pooling's padding can be treated as 0.
constexpr int kStemPool = 8;                       // pooled tile edge
constexpr int kStemConv = 2 * kStemPool + 1;       // 17 convolution rows / columns
constexpr int kStemPixels = kStemConv * kStemConv; // 289
constexpr int kStemGroups = (kStemPixels + 15) / 16;  // 19 groups of 16 pixels
constexpr int kStemPatch = 2 * (kStemConv - 1) + 7;   // 39 input rows / columns
constexpr int kStemPatchCols = 40;                 // halfs per patch row that hold pixels (column 39 is the kw padding)
// The patch is kept TWICE in LDS, the second copy shifted left by two pixels: the 8 input pixels a lane
// feeds start at column 2 * cx, so in copy (cx & 1) they start at a multiple of 4 halfs and are two aligned
// 8-byte reads (ds_read_b64: 2 LDS cycles each) instead of four 4-byte ones (ds_read2_b32: 8 cycles per pair).
// Rows are 128 bytes and the copies 64 bytes (mod 256) apart: the four 16-lane sets of a half-wave - even and
// odd columns of two consecutive K chunks, i.e. two consecutive patch rows - fall on four disjoint groups of
// 16 banks.
constexpr int kStemPatchPitch = 64;                // halfs per patch row in LDS
constexpr int kStemPatchHalfs = 3 * kStemPatch * kStemPatchPitch;
constexpr int kStemCopyHalfs = kStemPatchHalfs + 96;  // copy 1 starts here; the 16 zero halfs of the K padding sit between
static_assert((kStemCopyHalfs * 2) % 256 == 64, "the second patch copy must sit 16 banks from the first");
constexpr int kStemConvPitch = 72;                 // halfs per convolution pixel in LDS (16-byte aligned rows)
constexpr int kStemK = 192;

__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void StemConvPoolKernel(const _Float16* __restrict__ x,
                                                               const _Float16* __restrict__ wk,
                                                               const _Float16* __restrict__ bias,
                                                               _Float16* __restrict__ y, int H, int W, int OH, int OW,
                                                               int PH, int PW, int tiles_w, int tiles_per_image,
                                                               int total_tiles, unsigned x_bytes) {
  __shared__ __attribute__((aligned(256))) _Float16 patch[2 * kStemCopyHalfs];  // two copies; zero halfs for the K padding after the first
  __shared__ __attribute__((aligned(16))) _Float16 conv[kStemGroups * 16 * kStemConvPitch];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // Loaded ONCE per block (blocks are persistent over tiles): the weight matrix as MFMA A
  // fragments, lane -> (channel ni*16 + lane&15, k = 32 s + 8 (lane>>4) ..+7)
  half8 wf[4][6];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int s = 0; s < 6; ++s)
      wf[ni][s] = *reinterpret_cast<const half8*>(wk + (ni * 16 + (lane & 15)) * kStemK + s * 32 + (lane >> 4) * 8);
  // K chunk (8 values) of this lane in step s: chunk = 4 s + lane>>4 -> (c, kh); chunks 21..23 are padding
  int chunk_off[6];
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    const int chunk = 4 * s + (lane >> 4);
    const int c = chunk / 7, kh = chunk - c * 7;
    chunk_off[s] = chunk < 21 ? (c * kStemPatch + kh) * kStemPatchPitch : -1;
  }
  half4 b[4];  // kept packed (8 registers), widened at use
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) b[ni] = *reinterpret_cast<const half4*>(bias + ni * 16 + (lane >> 4) * 4);
  if (tid < 16) patch[kStemPatchHalfs + tid] = static_cast<_Float16>(0);

  constexpr int kRowDwords = kStemPatchCols / 2;     // 20 aligned dwords per patch row
  constexpr int kPitchDwords = kStemPatchPitch / 2;  // dwords per patch row in LDS
  constexpr int kRowStep = kThreads / kRowDwords;    // 12 rows per pass of the block
  constexpr int kLoadIters = (3 * kStemPatch + kRowStep - 1) / kRowStep;  // 10 passes cover the 117 rows

  // Patch rows of tile `tile_id` as aligned dwords into registers (zeros outside the image;
  // W is even, so a dword is inside or outside as a whole).  Issued one tile AHEAD, so the
  // loads are in flight while the current tile computes.
  // Thread -> (patch row row0 + kRowStep * it, dword column j): the column is fixed per
  // thread and the row advances by a constant, so a tile costs a handful of VALU
  // instructions per load; out-of-image dwords are zero-filled by the buffer range check
  // (no divergent branches around the loads).
  const __amdgpu_buffer_rsrc_t x_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(x), 0, static_cast<int>(x_bytes), 0x00020000);
  uint32_t staged[kLoadIters];
  const int load_j = tid % kRowDwords;
  const int load_row0 = tid / kRowDwords;  // 0 .. kRowStep (threads past 20 * kRowStep idle along)
  auto fetch_patch = [&](int tile_id) {
    const int img = tile_id / tiles_per_image;
    const int tile = tile_id - img * tiles_per_image;
    const int ty = tile / tiles_w;
    const int tx = tile - ty * tiles_w;
    const int ih0 = 4 * ty * kStemPool - 5;  // input row at patch row 0
    const int iw0 = 4 * tx * kStemPool - 6;  // input column at patch column 0: EVEN (one left of the first tap;
                                             // that column meets the zero weight of the kw padding)
    const int img_bytes = img * 3 * H * W * 2;  // wave-uniform: rides in the scalar offset of the load
    const int iw = iw0 + 2 * load_j;
    // validity as integer masks and a branch-free select: the compiler must not wrap the
    // loads in divergent control flow
    const unsigned col_ok = static_cast<unsigned>(static_cast<unsigned>(iw) < static_cast<unsigned>(W)) &
                            static_cast<unsigned>(load_row0 < kRowStep);
#pragma unroll
    for (int it = 0; it < kLoadIters; ++it) {
      const int rc = load_row0 + it * kRowStep;  // c * 39 + r
      const int c = static_cast<int>(rc >= kStemPatch) + static_cast<int>(rc >= 2 * kStemPatch);
      const int r = rc - c * kStemPatch;
      const int ih = ih0 + r;
      const unsigned ok = col_ok & static_cast<unsigned>(rc < 3 * kStemPatch) &
                          static_cast<unsigned>(static_cast<unsigned>(ih) < static_cast<unsigned>(H));
      // out-of-image dwords get an all-ones offset, past the descriptor: the range check returns zero
      const unsigned off = static_cast<unsigned>(((c * H + ih) * W + iw) * 2) | (ok - 1u);
      staged[it] = static_cast<uint32_t>(__builtin_amdgcn_raw_buffer_load_b32(x_rsrc, static_cast<int>(off), img_bytes, 0));
    }
  };

  if (static_cast<int>(blockIdx.x) < total_tiles) fetch_patch(blockIdx.x);
  for (int tile_id = blockIdx.x; tile_id < total_tiles; tile_id += gridDim.x) {
    const int img = tile_id / tiles_per_image;
    const int tile = tile_id - img * tiles_per_image;
    const int ty = tile / tiles_w;
    const int tx = tile - ty * tiles_w;
    const int ph0 = ty * kStemPool, pw0 = tx * kStemPool;
    const int oh0 = 2 * ph0 - 1, ow0 = 2 * pw0 - 1;  // convolution pixel at tile position (0, 0)
    if (load_row0 < kRowStep) {
#pragma unroll
      for (int it = 0; it < kLoadIters; ++it) {
        const int rc = load_row0 + it * kRowStep;
        if (rc < 3 * kStemPatch) {
          uint32_t* row32 = reinterpret_cast<uint32_t*>(patch) + rc * kPitchDwords;
          row32[load_j] = staged[it];
          if (load_j > 0) row32[kStemCopyHalfs / 2 + load_j - 1] = staged[it];  // the copy shifted by two pixels
        }
      }
    }
    __syncthreads();
    if (tile_id + static_cast<int>(gridDim.x) < total_tiles) fetch_patch(tile_id + gridDim.x);

    for (int g = wave; g < kStemGroups; g += kThreads / 64) {
      const int q = g * 16 + (lane & 15);
      const int qc = q < kStemPixels ? q : kStemPixels - 1;  // the last group is ragged: recompute a valid pixel
      const int r = qc / kStemConv;
      const int cx = qc - r * kStemConv;
      const int pix_off = 2 * r * kStemPatchPitch + 4 * (cx >> 1) + (cx & 1) * kStemCopyHalfs;
      float4v acc[4];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[ni] = float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        const int off = chunk_off[s] >= 0 ? chunk_off[s] + pix_off : kStemPatchHalfs;
        // 8 consecutive input pixels starting at a multiple of 4 halfs in the lane's copy: two 8-byte reads
        // (volatile: kept as two ds_read_b64, never merged into one slower ds_read2_b64)
        using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
        using lds_u2_ptr = const volatile __attribute__((address_space(3))) u32x2*;
        lds_u2_ptr src = reinterpret_cast<lds_u2_ptr>(reinterpret_cast<uintptr_t>(patch + off));
        union {
          u32x2 u[2];
          half8 h;
        } frag;
        frag.u[0] = src[0];
        frag.u[1] = src[1];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni][s], frag.h, acc[ni], 0, 0, 0);
      }
      // D^T layout: lane holds pixel q and channels ni*16 + 4 (lane>>4) .. +3
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        half4 packed;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float f = acc[ni][k] + static_cast<float>(b[ni][k]);
          packed[k] = static_cast<_Float16>(f > 0.f ? f : 0.f);
        }
        *reinterpret_cast<half4*>(conv + q * kStemConvPitch + ni * 16 + (lane >> 4) * 4) = packed;
      }
    }
    __syncthreads();

    // 3x3 / stride 2 / pad 1 max pooling of the tile: one thread per pooled pixel and 8
    // channels.  Window positions outside the convolution output are clamped onto the
    // nearest valid row / column, which lies inside the same window, so the maximum is
    // unchanged and there is no divergent control flow.
#pragma unroll
    for (int pass = 0; pass < kStemPool * kStemPool * 8 / kThreads; ++pass) {
      const int v = tid + pass * kThreads;
      const int cv = v & 7;
      const int pp = v >> 3;
      const int py = pp / kStemPool, px = pp - py * kStemPool;
      const int ph = ph0 + py, pw = pw0 + px;
      // post-ReLU values are non-negative (+0 included), and non-negative fp16 values order
      // like their bit patterns: an unsigned 16-bit maximum needs no NaN canonicalisation
      ushort8 best = ushort8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        int oh = oh0 + 2 * py + dy;
        oh = oh < 0 ? 0 : (oh > OH - 1 ? OH - 1 : oh);
        const int r = oh - oh0;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          int ow = ow0 + 2 * px + dx;
          ow = ow < 0 ? 0 : (ow > OW - 1 ? OW - 1 : ow);
          const int cx = ow - ow0;
          const ushort8 val = *reinterpret_cast<const ushort8*>(conv + (r * kStemConv + cx) * kStemConvPitch + cv * 8);
          best = __builtin_elementwise_max(best, val);
        }
      }
      if (ph < PH && pw < PW)
        *reinterpret_cast<ushort8*>(y + ((static_cast<long long>(img) * PH + ph) * PW + pw) * 64 + cv * 8) = best;
    }
    // the next tile's patch may be staged right away (its readers are behind the barrier above);
    // the convolution tile is only rewritten after the barrier that follows that staging
  }
}

struct SplitK {
  int splits;                        // 1 = off, 0 = let the launcher decide, n = exactly n runs of the K loop
  float* partials;                   // workspace of the partial tiles
  unsigned long long partial_bytes;
  unsigned* arrivals;                // one zero-initialised counter per output tile
  int arrival_slots;
};

template <int BM, int BN, int kWN, bool kSpatial, bool kGlds, bool kFuse = false, int kN3 = 0, int kXS = 0, bool kHalo = false, int kWM = 2>
int LaunchConvGemm(const _Float16* x, const _Float16* w, const _Float16* bias, const _Float16* residual, _Float16* y,
                  int M, int N, int K, int relu, const RowMap& map, unsigned x_bytes, int stages, const SplitK& sk,
                  hipStream_t st, const Pointwise2& pw2 = Pointwise2{nullptr, nullptr, 0, nullptr},
                  const Pointwise3& pw3 = Pointwise3{nullptr, nullptr, nullptr}) {
  const int tiles_m = (M + BM - 1) / BM;
  const int tiles_n = N / BN;
  const int blocks = tiles_m * tiles_n;
  const int xcd_chunk = (blocks % 8 == 0) ? blocks / 8 : 0;  // the remap must be a bijection
  const int ktiles_all = K / kBK;
  // split-K: requested, or (0) chosen here - when the tiles alone leave half of the 256 CUs
  // idle and the K loop is long enough (32 steps measured: the fix-up costs 3-4 us) to
  // share out; bounded by the caller's workspace
  int splits = sk.splits;
  if (splits == 0) {
    splits = 1;
    if (sk.partials != nullptr && blocks <= 128 && ktiles_all >= 32) {
      splits = 256 / blocks;
      if (splits > ktiles_all / 4) splits = ktiles_all / 4;
      if (splits > 8) splits = 8;
    }
  }
  if (splits > ktiles_all) splits = ktiles_all;
  while (splits > 1 && (splits - 1) * ((ktiles_all + splits - 1) / splits) >= ktiles_all) --splits;  // no empty run
  if (splits > 1) {
    const unsigned long long need = static_cast<unsigned long long>(blocks) * splits * BM * BN * sizeof(float);
    if (sk.partials == nullptr || sk.arrivals == nullptr || need > sk.partial_bytes || blocks > sk.arrival_slots) {
      if (sk.splits > 1) return -4;  // asked for explicitly: the workspace must hold it
      splits = 1;
    }
  }
  const int ktiles = (ktiles_all + splits - 1) / splits;
  if (!kGlds) stages = 2;
  if (stages > ktiles) stages = ktiles;
  while (stages > 2 && static_cast<size_t>(stages) * (BM + BN) * kBK * sizeof(_Float16) > 160 * 1024) --stages;  // LDS of one CU
  size_t lds_bytes = static_cast<size_t>(stages) * (BM + BN) * kBK * sizeof(_Float16);
  if (kHalo) {
    if (map.ksize != 3 || map.stride != 1 || map.pad != 1 || splits != 1) return -2;
    const size_t halo_rows = static_cast<size_t>((BM + 2 * map.in_w + 2) / (8 * kWM * kWN) + 1) * (8 * kWM * kWN);
    auto halo_lds = [&](int st) {
      return (static_cast<size_t>(map.cin / kBK) * halo_rows + static_cast<size_t>(st) * BN) * kBK * sizeof(_Float16);
    };
    while (stages > 2 && halo_lds(stages) > 160 * 1024) --stages;
    lds_bytes = halo_lds(stages);
    if (lds_bytes > 160 * 1024) return -2;
  }
  size_t c_bytes = static_cast<size_t>(BM) * (BN + 8) * sizeof(_Float16);
  if (kFuse) {
    if (tiles_n != 1 || splits != 1 || pw2.w == nullptr || pw2.n <= 0 || pw2.n % 64 != 0) return -2;
    // operand slabs of the first result + one weight chunk + the 64-channel output tile
    if ((kXS > 0) != (pw2.xd != nullptr)) return -2;
    constexpr size_t k2 = BN + 64 * kXS;  // K of the second GEMM
    c_bytes = (static_cast<size_t>(BM) * k2 + k2 * 64 + static_cast<size_t>(BM) * 64) * sizeof(_Float16);
    if (kN3 > 0) {
      if (pw3.w == nullptr || pw3.out == nullptr) return -2;
      c_bytes += static_cast<size_t>(kN3) * 64 * sizeof(_Float16);  // the third GEMM's weights (its operand slab shares cs2's space)
      const size_t c3 = static_cast<size_t>(BM) * (kN3 + 8) * sizeof(_Float16);
      if (c3 > c_bytes) c_bytes = c3;
    }
  }
  if (c_bytes > lds_bytes) lds_bytes = c_bytes;
  if (lds_bytes > 64 * 1024) {
    // above the default dynamic-LDS limit: raise it once per kernel instantiation
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&ConvGemmKernel<BM, BN, kWN, kSpatial, kGlds, kFuse, kN3, kXS, kHalo, kWM>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) return -1000 - static_cast<int>(attr);
  }
  hipLaunchKernelGGL((ConvGemmKernel<BM, BN, kWN, kSpatial, kGlds, kFuse, kN3, kXS, kHalo, kWM>), dim3(blocks, splits), dim3(64 * kWM * kWN), lds_bytes, st,
                     x, w, bias, residual, y, M, N, K, relu, map, tiles_n, xcd_chunk, x_bytes, stages, splits, sk.partials,
                     sk.arrivals, pw2, pw3);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -1000 - static_cast<int>(e);
}

}  // namespace

extern "C" {

// x: [batch,in_h,in_w,cin] fp16, w: [cout][ksize][ksize][cin] fp16 (the memory of
// a channels-last [cout,cin,ksize,ksize] weight), y: [batch,oh,ow,cout].
// variant: low 4 bits 0 auto, 1 = 128x128, 2 = 128x64, 3 = 64x64 block tile (4 wavefronts),
// 4 = 128x128, 5 = 256x128, 6 = 128x64, 7 = 64x64, 8 = 128x256 (8 wavefronts);
// 9 = 256x128, 10 = 256x256, 11 = 128x256 (16 wavefronts); 12 = 128x256, 13 = 128x128 in the halo form (3x3 stride 1 pad 1);
// bit 4 selects register staging; bits 5-6 the number of LDS stages (2 or 3).
// splits: 1 = no split-K, 0 = automatic, n > 1 = exactly n.  partials / arrivals: the split-K
// workspace (device memory; `arrivals` zero-filled once, the kernel leaves it zeroed) -
// one workspace per stream that launches concurrently.
int tca_conv_gemm_splitk_launch(const void* x, const void* w, const void* bias, const void* residual, void* y, int batch,
                                int in_h, int in_w, int ksize, int stride, int pad, int cin, int cout, int relu,
                                int variant, int splits, void* partials, unsigned long long partial_bytes,
                                void* arrivals, int arrival_slots, void* stream) {
  if (!x || !w || !y || batch <= 0 || in_h <= 0 || in_w <= 0 || stride <= 0 || ksize <= 0 || pad < 0) return -1;
  if (cin % kBK != 0 || cout % 64 != 0) return -2;  // whole K steps inside one tap, whole n tiles
  const int oh = (in_h + 2 * pad - ksize) / stride + 1;
  const int ow = (in_w + 2 * pad - ksize) / stride + 1;
  if (oh <= 0 || ow <= 0) return -2;
  const long long m_ll = static_cast<long long>(batch) * oh * ow;
  const long long k_ll = static_cast<long long>(ksize) * ksize * cin;
  if (m_ll > 0x7FFFFFFF || k_ll > 0x7FFFFFFF) return -2;
  const int M = static_cast<int>(m_ll);
  const int K = static_cast<int>(k_ll);
  const bool spatial = ksize != 1 || pad != 0;
  RowMap map;
  map.identity = !spatial && stride == 1;
  map.ohw = oh * ow;
  map.ow = ow;
  map.in_h = in_h;
  map.in_w = in_w;
  map.stride = stride;
  map.pad = pad;
  map.ksize = ksize;
  map.cin = cin;
  int staging = (variant >> 4) & 1;  // bit 4: register staging instead of the LDS DMA
  int stages = (variant >> 5) & 3;   // bits 5-6: LDS stages of the DMA pipeline (2 or 3), 0 = auto
  variant &= 15;
  // the DMA path addresses activations with 32-bit byte offsets below kOutOfRange, and
  // keeps one validity bit per tap in a 32-bit mask
  const long long x_bytes_ll = static_cast<long long>(batch) * in_h * in_w * cin * 2;
  const long long lead_bytes = (static_cast<long long>(pad) * in_w + pad) * cin * 2;
  if (x_bytes_ll + lead_bytes >= 0x7FFFFFFFll || k_ll * cout * 2 >= 0x7FFFFFFFll || ksize * ksize > 32) staging = 1;
  const unsigned x_bytes = staging ? 0u : static_cast<unsigned>(x_bytes_ll);
  if (variant == 0) {
    // thresholds from per-layer timings (tools/conv_bench.py, batches 8..128): the
    // 128x128 tile on 8 wavefronts wins as soon as it yields about one block per CU
    // and the K loop is long enough to amortise its prologue; below that, enough
    // 64x64 blocks to fill 256 CUs several times over first, then 128x64
    const long long blocks128 = static_cast<long long>((M + 127) / 128) * (cout / 128);
    const long long blocks64 = static_cast<long long>((M + 63) / 64) * (cout / 64);
    if (cout % 128 == 0 && ((K >= 128 && blocks128 >= 196) || blocks128 >= 4096)) variant = 4;
    else if (blocks64 <= 2048) variant = (K >= 512 && blocks64 <= 256) ? 7 : 3;  // few blocks, long K: more waves per block
    else variant = (cout % 128 != 0 && K >= 256) ? 7 : 2;
  }
  if ((variant == 1 || variant == 4 || variant == 5) && cout % 128 != 0) return -2;
  if ((variant == 8 || variant == 10 || variant == 11 || variant == 12) && cout % 256 != 0) return -2;
  if ((variant == 9 || variant == 13) && cout % 128 != 0) return -2;
  if (variant == 12 || variant == 13) {
    // one block per CU whatever the grid: the halo takes most of the LDS; no split-K in this form
    if (ksize != 3 || stride != 1 || pad != 1) return -2;
    splits = 1;
    stages = 3;
  }
  if (variant >= 9 && staging) return -2;  // the 16-wavefront tiles exist in the DMA-staged form only
  if (stages != 2 && stages != 3) {
    // a third stage hides more DMA latency but costs LDS: take it only when the grid is
    // too small for the LDS limit on resident blocks to bind
    const int bm = (variant == 3 || variant == 7) ? 64 : ((variant == 5 || variant == 9 || variant == 10) ? 256 : 128);
    const int bn = (variant == 2 || variant == 3 || variant == 6 || variant == 7)
                       ? 64
                       : ((variant == 8 || variant == 10 || variant == 11 || variant == 12) ? 256 : 128);
    const long long blocks = static_cast<long long>((M + bm - 1) / bm) * (cout / bn);
    const long long per_cu_wanted = (blocks + 255) / 256;
    const long long per_cu_with_3 = (160 * 1024) / (3 * (bm + bn) * kBK * 2);
    stages = per_cu_wanted <= per_cu_with_3 ? 3 : 2;
  }
  if (splits < 0 || (reinterpret_cast<uintptr_t>(partials) & 15)) return -1;
  const SplitK sk{splits, static_cast<float*>(partials), partial_bytes, static_cast<unsigned*>(arrivals), arrival_slots};
  auto st = static_cast<hipStream_t>(stream);
  auto xp = static_cast<const _Float16*>(x);
  auto wp = static_cast<const _Float16*>(w);
  auto bp = static_cast<const _Float16*>(bias);
  auto rp = static_cast<const _Float16*>(residual);
  auto yp = static_cast<_Float16*>(y);
  // staging: direct-to-LDS DMA by default; `staging` = 1 keeps the register-staged path (A/B testing)
#define TCA_DISPATCH(BM_, BN_, WN_)                                                                                      \
  (spatial ? (staging ? LaunchConvGemm<BM_, BN_, WN_, true, false>(xp, wp, bp, rp, yp, M, cout, K, relu, map, x_bytes, stages, sk, st)  \
                      : LaunchConvGemm<BM_, BN_, WN_, true, true>(xp, wp, bp, rp, yp, M, cout, K, relu, map, x_bytes, stages, sk, st))  \
           : (staging ? LaunchConvGemm<BM_, BN_, WN_, false, false>(xp, wp, bp, rp, yp, M, cout, K, relu, map, x_bytes, stages, sk, st) \
                      : LaunchConvGemm<BM_, BN_, WN_, false, true>(xp, wp, bp, rp, yp, M, cout, K, relu, map, x_bytes, stages, sk, st)))
#define TCA_DISPATCH16(BM_, BN_, WN_, WM_)                                                                                   \
  (spatial ? LaunchConvGemm<BM_, BN_, WN_, true, true, false, 0, 0, false, WM_>(xp, wp, bp, rp, yp, M, cout, K, relu, map, x_bytes, \
                                                                                stages, sk, st)                               \
           : LaunchConvGemm<BM_, BN_, WN_, false, true, false, 0, 0, false, WM_>(xp, wp, bp, rp, yp, M, cout, K, relu, map, x_bytes, \
                                                                                 stages, sk, st))
  switch (variant) {
    case 1: return TCA_DISPATCH(128, 128, 2);
    case 2: return TCA_DISPATCH(128, 64, 2);
    case 3: return TCA_DISPATCH(64, 64, 2);
    case 4: return TCA_DISPATCH(128, 128, 4);  // 8 wavefronts, 64x32 per wave
    case 5: return TCA_DISPATCH(256, 128, 4);  // 8 wavefronts, 128x32 per wave
    case 6: return TCA_DISPATCH(128, 64, 4);   // 8 wavefronts, 64x16 per wave
    case 7: return TCA_DISPATCH(64, 64, 4);    // 8 wavefronts, 32x16 per wave
    case 8: return TCA_DISPATCH(128, 256, 4);  // 8 wavefronts, 64x64 per wave
    // 16 wavefronts (4 per SIMD from ONE block): the operand bytes a CU takes in per FLOP are those of the
    // large tile, the waves in flight those of two co-resident 128x128 blocks
    case 9: return TCA_DISPATCH16(256, 128, 4, 4);   // 4x4 waves, 64x32 per wave
    case 10: return TCA_DISPATCH16(256, 256, 4, 4);  // 4x4 waves, 64x64 per wave
    case 11: return TCA_DISPATCH16(128, 256, 8, 2);  // 2x8 waves, 64x32 per wave
    // Halo form of a plain 3x3 / stride 1 / pad 1 convolution: the tile's input rows are brought into LDS once
    // and the K loop streams weights only (about half the operand bytes per FLOP of the tap-by-tap form);
    // layers whose halo does not fit LDS are refused (-2)
    case 12:
      return spatial && !staging ? LaunchConvGemm<128, 256, 4, true, true, false, 0, 0, true>(xp, wp, bp, rp, yp, M, cout, K, relu,
                                                                                          map, x_bytes, stages, sk, st)
                                 : -2;
    case 13:
      return spatial && !staging ? LaunchConvGemm<128, 128, 4, true, true, false, 0, 0, true>(xp, wp, bp, rp, yp, M, cout, K, relu,
                                                                                          map, x_bytes, stages, sk, st)
                                 : -2;
    default: return -3;
  }
#undef TCA_DISPATCH
#undef TCA_DISPATCH16
}

// y = act(conv1x1(relu(conv(x, w1) + bias1), w2) + bias2 (+ residual)) in ONE kernel: the middle
// activation ([pixels][mid] fp16) stays in LDS.  x: [batch,in_h,in_w,cin], w1: [mid][ksize][ksize][cin]
// (mid = 64 or 128: one tile holds all of its channels), w2: [cout][mid], residual / y: [batch,oh,ow,cout].
// tile_m: 0 (automatic), 64 or 128 pixels per tile.  With w3 / out3 (n3 = 64 or 128; with mid = 128 only 128): also out3 = relu(conv1x1(y, w3) + bias3), the first convolution of
// the NEXT block (w3: [n3][cout], n3 = 64 or 128; out3: [batch,oh,ow,n3]), fed from the output chunks
// while they are still on the CU.
int tca_conv_pointwise_launch(const void* x, const void* w1, const void* bias1, const void* w2, const void* bias2,
                              const void* residual, void* y, int batch, int in_h, int in_w, int ksize, int stride, int pad,
                              int cin, int mid, int cout, int relu, const void* w3, const void* bias3, void* out3, int n3,
                              int tile_m, const void* xd, void* stream) {
  if (!x || !w1 || !w2 || !y || batch <= 0 || in_h <= 0 || in_w <= 0 || stride <= 0 || ksize <= 0 || pad < 0) return -1;
  if (cin % kBK != 0 || (mid != 64 && mid != 128) || cout % 64 != 0 || cout <= 0) return -2;
  if (ksize == 1 && pad == 0) return -2;  // the fused form is instantiated for spatial kernels
  const int oh = (in_h + 2 * pad - ksize) / stride + 1;
  const int ow = (in_w + 2 * pad - ksize) / stride + 1;
  if (oh <= 0 || ow <= 0) return -2;
  const long long m_ll = static_cast<long long>(batch) * oh * ow;
  const long long k_ll = static_cast<long long>(ksize) * ksize * cin;
  const long long x_bytes_ll = static_cast<long long>(batch) * in_h * in_w * cin * 2;
  const long long lead_bytes = (static_cast<long long>(pad) * in_w + pad) * cin * 2;
  if (m_ll > 0x7FFFFFFF || k_ll > 0x7FFFFFFF || m_ll * cout * 2 >= 0x7FFFFFFFll) return -2;  // output addressed by 32-bit byte offsets
  if (x_bytes_ll + lead_bytes >= 0x7FFFFFFFll || k_ll * mid * 2 >= 0x7FFFFFFFll || ksize * ksize > 32 ||
      static_cast<long long>(cout) * mid * 2 >= 0x7FFFFFFFll)
    return -2;
  const int M = static_cast<int>(m_ll);
  const int K = static_cast<int>(k_ll);
  RowMap map;
  map.identity = 0;
  map.ohw = oh * ow;
  map.ow = ow;
  map.in_h = in_h;
  map.in_w = in_w;
  map.stride = stride;
  map.pad = pad;
  map.ksize = ksize;
  map.cin = cin;
  const SplitK sk{1, nullptr, 0, nullptr, 0};
  // xd: the block input [pixels][64] for a projection shortcut inside the kernel (mid = 64, stride 1, no
  // residual): w2 is then [cout][mid + 64] = [expansion | projection] and bias2 the sum of both biases
  if (xd != nullptr && (mid != 64 || stride != 1 || residual != nullptr || (n3 != 0 && n3 != 64) ||
                        (reinterpret_cast<uintptr_t>(xd) & 15)))
    return -2;
  const Pointwise2 pw2{static_cast<const _Float16*>(w2), static_cast<const _Float16*>(bias2), cout,
                       static_cast<const _Float16*>(xd)};
  // tile_m: pixels per tile, 64 or 128; 0 = 64 while 128-pixel tiles would leave CUs without a block
  if (tile_m != 0 && tile_m != 64 && tile_m != 128) return -2;
  const long long blocks128 = (M + 127) / 128;
  const int bm = tile_m ? tile_m : (blocks128 < 256 ? 64 : 128);
  const long long blocks = (M + bm - 1) / bm;
  const long long per_cu_wanted = (blocks + 255) / 256;
  const long long per_cu_with_3 = (160 * 1024) / (3 * (bm + mid) * kBK * 2);
  int stages = per_cu_wanted <= per_cu_with_3 ? 3 : 2;
  static const int forced_stages = [] {
    const char* v = std::getenv("TCA_FUSED_STAGES");  // experiments: 2, 3 or 4 stages of the first GEMM's DMA pipeline
    return v ? std::atoi(v) : 0;
  }();
  if (forced_stages >= 2 && forced_stages <= 4) stages = forced_stages;
  auto st = static_cast<hipStream_t>(stream);
  auto xp = static_cast<const _Float16*>(x);
  auto wp = static_cast<const _Float16*>(w1);
  auto bp = static_cast<const _Float16*>(bias1);
  auto rp = static_cast<const _Float16*>(residual);
  auto yp = static_cast<_Float16*>(y);
  const unsigned xb = static_cast<unsigned>(x_bytes_ll);
  Pointwise3 pw3{nullptr, nullptr, nullptr};
  if (w3 != nullptr) {
    if (out3 == nullptr || !relu || (n3 != 64 && n3 != 128) || (mid == 128 && n3 != 128) ||
        static_cast<long long>(cout) * n3 * 2 >= 0x7FFFFFFFll)
      return -2;
    pw3 = Pointwise3{static_cast<const _Float16*>(w3), static_cast<const _Float16*>(bias3), static_cast<_Float16*>(out3)};
  } else {
    n3 = 0;
  }
  // Halo form of the first GEMM (3x3, stride 1, pad 1): the tile's input rows go to LDS once and the K loop
  // streams weights only, three stages deep where two blocks per CU still fit (A/B switch: TCA_CONV_HALO=0).
  static const bool halo_enabled = [] {
    const char* v = std::getenv("TCA_CONV_HALO");
    return !(v && v[0] == '0');
  }();
  bool halo = halo_enabled && ksize == 3 && stride == 1 && pad == 1;
  if (halo) {
    const long long halo_rows = ((bm + 2 * in_w + 2) / 64 + 1) * 64;
    const long long halo_bytes = (cin / kBK) * halo_rows * kBK * 2;
    const long long second = (static_cast<long long>(bm) * (mid + (xd ? 64 : 0)) + (mid + (xd ? 64 : 0)) * 64 + bm * 64 + n3 * 64) * 2;
    const long long budget = second > 80 * 1024 ? second : 80 * 1024;  // never fewer blocks per CU than the fused phase allows
    if (halo_bytes + 3ll * mid * kBK * 2 <= budget) stages = 3;
    else if (halo_bytes + 2ll * mid * kBK * 2 <= budget) stages = 2;
    else halo = false;
    if (forced_stages >= 2 && forced_stages <= 4 && halo) stages = forced_stages;
  }
#define TCA_FUSED(BM_, BN_, N3_)                                                                                             \
  (halo ? LaunchConvGemm<BM_, BN_, 4, true, true, true, N3_, 0, true>(xp, wp, bp, rp, yp, M, mid, K, relu, map, xb, stages, sk, st, \
                                                                      pw2, pw3)                                              \
        : LaunchConvGemm<BM_, BN_, 4, true, true, true, N3_>(xp, wp, bp, rp, yp, M, mid, K, relu, map, xb, stages, sk, st, pw2, pw3))
#define TCA_FUSED_XS(BM_, N3_)                                                                                               \
  (halo ? LaunchConvGemm<BM_, 64, 4, true, true, true, N3_, 1, true>(xp, wp, bp, rp, yp, M, mid, K, relu, map, xb, stages, sk, st, \
                                                                     pw2, pw3)                                               \
        : LaunchConvGemm<BM_, 64, 4, true, true, true, N3_, 1>(xp, wp, bp, rp, yp, M, mid, K, relu, map, xb, stages, sk, st, pw2, pw3))
  static const bool four_waves = [] {
    const char* v = std::getenv("TCA_FUSED_WAVES");  // experiment: 4-wave blocks (64 x 32 per wave) for the 128 x 64 form
    return v && v[0] == '4';
  }();
  if (four_waves && xd == nullptr && bm == 128 && mid == 64 && n3 == 64 && halo)
    return LaunchConvGemm<128, 64, 2, true, true, true, 64, 0, true>(xp, wp, bp, rp, yp, M, mid, K, relu, map, xb, stages, sk, st, pw2, pw3);
  if (xd != nullptr) {
    if (bm == 64) return n3 == 64 ? TCA_FUSED_XS(64, 64) : TCA_FUSED_XS(64, 0);
    return n3 == 64 ? TCA_FUSED_XS(128, 64) : TCA_FUSED_XS(128, 0);
  }
  if (bm == 64) {
    if (mid == 128) return n3 == 128 ? TCA_FUSED(64, 128, 128) : TCA_FUSED(64, 128, 0);
    return n3 == 64 ? TCA_FUSED(64, 64, 64) : n3 == 128 ? TCA_FUSED(64, 64, 128) : TCA_FUSED(64, 64, 0);
  }
  if (mid == 128) return n3 == 128 ? TCA_FUSED(128, 128, 128) : TCA_FUSED(128, 128, 0);
  return n3 == 64 ? TCA_FUSED(128, 64, 64) : n3 == 128 ? TCA_FUSED(128, 64, 128) : TCA_FUSED(128, 64, 0);
#undef TCA_FUSED
#undef TCA_FUSED_XS
}

int tca_conv_gemm_launch(const void* x, const void* w, const void* bias, const void* residual, void* y, int batch,
                         int in_h, int in_w, int ksize, int stride, int pad, int cin, int cout, int relu, int variant,
                         void* stream) {
  return tca_conv_gemm_splitk_launch(x, w, bias, residual, y, batch, in_h, in_w, ksize, stride, pad, cin, cout, relu,
                                     variant, 1, nullptr, 0, nullptr, 0, stream);
}

int tca_conv1x1_launch(const void* x, const void* w, const void* bias, const void* residual, void* y, int batch,
                       int in_h, int in_w, int stride, int cin, int cout, int relu, int variant, void* stream) {
  return tca_conv_gemm_launch(x, w, bias, residual, y, batch, in_h, in_w, 1, stride, 0, cin, cout, relu, variant,
                              stream);
}

int tca_bias_act_launch(void* y, const void* bias, const void* residual, long long rows, int channels, int relu,
                        void* stream) {
  if (!y || !bias || rows <= 0 || channels <= 0) return -1;
  if (channels % 8 != 0) return -2;
  const long long vecs = rows * (channels / 8);
  long long blocks = (vecs + kThreads - 1) / kThreads;
  if (blocks > 256 * 16) blocks = 256 * 16;  // grid-stride beyond 16 blocks per CU
  hipLaunchKernelGGL(BiasActKernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream), static_cast<_Float16*>(y), static_cast<const _Float16*>(bias),
                     static_cast<const _Float16*>(residual), vecs, channels / 8, relu);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -1000 - static_cast<int>(e);
}

int tca_bias_relu_maxpool_launch(const void* x, const void* bias, void* y, int batch, int in_h, int in_w, int channels,
                                 void* stream) {
  if (!x || !bias || !y || batch <= 0 || in_h <= 0 || in_w <= 0 || channels <= 0) return -1;
  if (channels % 8 != 0) return -2;
  const int out_h = (in_h - 1) / 2 + 1;  // (H + 2*1 - 3) / 2 + 1
  const int out_w = (in_w - 1) / 2 + 1;
  const long long out_vecs = static_cast<long long>(batch) * out_h * out_w * (channels / 8);
  long long blocks = (out_vecs + kThreads - 1) / kThreads;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(BiasReluMaxPoolKernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream), static_cast<const _Float16*>(x),
                     static_cast<const _Float16*>(bias), static_cast<_Float16*>(y), in_h, in_w, out_h, out_w,
                     channels / 8, out_vecs);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -1000 - static_cast<int>(e);
}

// Stem: y[batch][PH][PW][64] = maxpool3x3/2(relu(conv7x7/2(x[batch][3][in_h][in_w]) + bias)); wk as documented at
// StemConvPoolKernel ([64][192], K order (c, kh, kw padded to 8)).
int tca_stem_conv_pool_launch(const void* x, const void* wk, const void* bias, void* y, int batch, int in_h, int in_w,
                              void* stream) {
  if (!x || !wk || !bias || !y || batch <= 0 || in_h < 7 || in_w < 7) return -1;
  if (in_w % 2 != 0 || (reinterpret_cast<uintptr_t>(x) & 3) != 0) return -2;  // rows are read as aligned dwords
  const long long x_bytes = static_cast<long long>(batch) * 3 * in_h * in_w * 2;
  if (x_bytes >= 0x7FFFFFFFll) return -2;  // 32-bit buffer offsets
  const int oh = (in_h + 6 - 7) / 2 + 1, ow = (in_w + 6 - 7) / 2 + 1;
  const int ph = (oh - 1) / 2 + 1, pw = (ow - 1) / 2 + 1;
  const int tiles_h = (ph + kStemPool - 1) / kStemPool, tiles_w = (pw + kStemPool - 1) / kStemPool;
  const long long tiles = static_cast<long long>(batch) * tiles_h * tiles_w;
  if (tiles > 0x7FFFFFFF) return -2;
  // persistent blocks: two per CU (the LDS limit: two patch copies + the convolution tile = 72 KiB), each walks
  // tiles with a grid stride and keeps the weight fragments in registers throughout
  const long long blocks = tiles < 256 * 2 ? tiles : 256 * 2;
  hipLaunchKernelGGL(StemConvPoolKernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream), static_cast<const _Float16*>(x),
                     static_cast<const _Float16*>(wk), static_cast<const _Float16*>(bias), static_cast<_Float16*>(y),
                     in_h, in_w, oh, ow, ph, pw, tiles_w, tiles_h * tiles_w, static_cast<int>(tiles),
                     static_cast<unsigned>(x_bytes));
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -1000 - static_cast<int>(e);
}

}  // extern "C"
