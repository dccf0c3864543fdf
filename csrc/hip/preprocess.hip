// Fused image preprocessing for gfx950 (MI355X): one launch per batch does
//
//   colour convert (gray<->RGB, BGR(A)->RGB)  ->  separable resize
//   ->  per-channel scale/bias  ->  cast  ->  NCHW / NHWC store at
//   batch_index * image_stride inside the destination (a hipshm region).
//
// It replaces the CPU pipeline of the reference's image_client
// (`src/python/examples/image_client.py:153-192`: PIL convert + resize(BILINEAR)
// + astype + scaling + transpose + np.stack; `src/c++/examples/image_client.cc:85-188`
// is the OpenCV twin).
//
// Resize is table driven: for every output column/row the host supplies the
// first contributing source index, the tap count and fixed-point (22 bit)
// weights.  With tables built like Pillow's `precompute_coeffs` the kernel
// reproduces `Image.resize(..., BILINEAR)` bit for bit (horizontal pass rounded
// to u8, then vertical pass rounded to u8 - the same two-pass integer scheme);
// 2-tap tables give the OpenCV INTER_LINEAR flavour.
//
// Work decomposition (64-wide wavefronts, 256 threads = 4 waves per block):
//   block  = one TH x 64 tile of the OUTPUT image of one batch element
//   stage  = the source bytes the tile needs (its rows x the columns its 64 outputs
//            draw from) go to LDS with coalesced, aligned dword loads - every source
//            byte crosses the memory system once per tile instead of once per tap
//            in scattered byte loads; rows have an odd dword pitch
//   pass 1 = horizontal filter out of the staged rows with lane = source row and
//            wave = output column group: window and weights of a column are wave
//            uniform (scalar loads) and the 64 lanes read one byte column of 64 rows,
//            one bank each; results go to LDS as planar u8 [row][channel][64] packed
//            four to a dword, rows again at an odd dword pitch
//   pass 2 = vertical filter out of LDS + scale/bias + cast; each thread owns 2
//            adjacent output pixels of one output row and reads them as one u16:
//            the 32 lanes of a half-wave read 16 consecutive dwords of one row
//            (pairs of lanes share a dword), conflict-free whatever the row step;
//            NCHW stores are 4/8-byte pieces of one contiguous run per half-wave,
//            so the HWC->CHW transpose never touches HBM
// Without room in LDS for the source tile (extreme down-scaling) pass 1 reads the
// source from memory directly.
// The grid is (tiles_x, tiles_y, batch): 8 images of 224x224 already give
// 4*14*8 = 448 blocks (> 256 CUs).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdint>

#include "kernels.h"

namespace {

constexpr int kTileW = 64;
constexpr int kThreads = 256;
constexpr int kPrecisionBits = 22;  // Pillow: 32 - 8 - 2

__device__ __forceinline__ uint32_t Clip8(int v) {
  v >>= kPrecisionBits;
  return static_cast<uint32_t>(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// Pillow's RGB -> L: (R*19595 + G*38470 + B*7471 + 0x8000) >> 16
__device__ __forceinline__ uint32_t Luma(uint32_t r, uint32_t g, uint32_t b) {
  return (r * 19595u + g * 38470u + b * 7471u + 0x8000u) >> 16;
}

// Loads one source pixel and converts it to the kernel's channel count.
template <int C_OUT>
__device__ __forceinline__ void LoadPixel(const uint8_t* __restrict__ p, int src_channels, int bgr, uint32_t (&px)[C_OUT]) {
  if (src_channels == 1) {
    const uint32_t v = p[0];
#pragma unroll
    for (int c = 0; c < C_OUT; ++c) px[c] = v;
  } else {
    uint32_t r = p[0], g = p[1], b = p[2];
    if (bgr) { const uint32_t t = r; r = b; b = t; }
    if constexpr (C_OUT == 1) {
      px[0] = Luma(r, g, b);
    } else {
      px[0] = r;
      px[1] = g;
      px[2] = b;
    }
  }
}

template <typename T> struct Caster;
template <> struct Caster<float>    { static __device__ __forceinline__ float    To(float v) { return v; } };
template <> struct Caster<double>   { static __device__ __forceinline__ double   To(float v) { return static_cast<double>(v); } };
template <> struct Caster<__half>   { static __device__ __forceinline__ __half   To(float v) { return __float2half_rn(v); } };
template <> struct Caster<uint16_t> { static __device__ __forceinline__ uint16_t To(float v) { v = rintf(v); v = v < 0.f ? 0.f : (v > 65535.f ? 65535.f : v); return static_cast<uint16_t>(v); } };
template <> struct Caster<int16_t>  { static __device__ __forceinline__ int16_t  To(float v) { v = rintf(v); v = v < -32768.f ? -32768.f : (v > 32767.f ? 32767.f : v); return static_cast<int16_t>(v); } };
template <> struct Caster<uint8_t>  { static __device__ __forceinline__ uint8_t  To(float v) { v = rintf(v); v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v); return static_cast<uint8_t>(v); } };
template <> struct Caster<int8_t>   { static __device__ __forceinline__ int8_t   To(float v) { v = rintf(v); v = v < -128.f ? -128.f : (v > 127.f ? 127.f : v); return static_cast<int8_t>(v); } };
template <> struct Caster<int32_t>  { static __device__ __forceinline__ int32_t  To(float v) { return static_cast<int32_t>(rintf(v)); } };
template <> struct Caster<uint32_t> { static __device__ __forceinline__ uint32_t To(float v) { v = rintf(v); return v < 0.f ? 0u : static_cast<uint32_t>(v); } };

// bf16 is carried as a tagged 16-bit type so it can share the store path.
struct Bf16 { uint16_t bits; };
template <> struct Caster<Bf16> {
  static __device__ __forceinline__ Bf16 To(float v) {
    uint32_t w = __float_as_uint(v);
    // round to nearest even; NaN stays NaN
    if ((w & 0x7FFFFFFFu) > 0x7F800000u) return Bf16{static_cast<uint16_t>((w >> 16) | 0x40)};
    w += 0x7FFFu + ((w >> 16) & 1u);
    return Bf16{static_cast<uint16_t>(w >> 16)};
  }
};

// Four consecutive elements as one aligned vector store when possible.
template <typename T>
__device__ __forceinline__ void Store4(T* __restrict__ dst, const T (&v)[4], int valid) {
  constexpr int kBytes = 4 * sizeof(T);
  const bool aligned = (reinterpret_cast<uintptr_t>(dst) % kBytes) == 0;
  if (valid == 4 && aligned) {
    if constexpr (kBytes == 4) {
      *reinterpret_cast<uint32_t*>(dst) = *reinterpret_cast<const uint32_t*>(v);
    } else if constexpr (kBytes == 8) {
      *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(v);
    } else if constexpr (kBytes == 16) {
      *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(v);
    } else {
      *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(v);
      *(reinterpret_cast<uint4*>(dst) + 1) = *(reinterpret_cast<const uint4*>(v) + 1);
    }
  } else {
    for (int j = 0; j < valid; ++j) dst[j] = v[j];
  }
}

// Two consecutive elements as one aligned store when possible.
template <typename T>
__device__ __forceinline__ void Store2(T* __restrict__ dst, const T (&v)[2], int valid) {
  constexpr int kBytes = 2 * sizeof(T);
  if (valid == 2 && (reinterpret_cast<uintptr_t>(dst) % kBytes) == 0) {
    if constexpr (kBytes == 2) {
      *reinterpret_cast<uint16_t*>(dst) = *reinterpret_cast<const uint16_t*>(v);
    } else if constexpr (kBytes == 4) {
      *reinterpret_cast<uint32_t*>(dst) = *reinterpret_cast<const uint32_t*>(v);
    } else if constexpr (kBytes == 8) {
      *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(v);
    } else {
      *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(v);
    }
  } else {
    for (int j = 0; j < valid; ++j) dst[j] = v[j];
  }
}

template <typename T, int C_OUT, bool kStaged>
__global__ __launch_bounds__(kThreads) void PreprocessKernel(
    const tca::PreprocImage* __restrict__ images, const int2* __restrict__ bounds, const int* __restrict__ coeffs,
    T* __restrict__ dst, tca::PreprocParams prm) {
  extern __shared__ __attribute__((aligned(16))) uint32_t hbuf[];  // [rows][C_OUT * kTileW/4 + 1] dwords, then the staged source

  const tca::PreprocImage img = images[blockIdx.z];
  const int ox0 = blockIdx.x * kTileW;
  const int oy0 = blockIdx.y * prm.tile_h;
  const int tile_w = min(kTileW, prm.out_w - ox0);
  const int tile_h = min(prm.tile_h, prm.out_h - oy0);

  const int2* __restrict__ xb = bounds + img.xb_off;
  const int2* __restrict__ yb = bounds + img.yb_off;
  const int* __restrict__ xk = coeffs + img.xk_off;
  const int* __restrict__ yk = coeffs + img.yk_off;

  // source rows needed by this tile: windows are monotonic in y
  const int row_first = yb[oy0].x;
  const int2 last_b = yb[oy0 + tile_h - 1];
  const int rows = last_b.x + last_b.y - row_first;
  if (rows > prm.max_rows || rows < 1) return;  // host sized the LDS buffer from max_rows; never overrun it

  // ---------------- stage: source rows -> LDS ----------------
  // Staged row r starts exactly at the tile's first source byte &src[row_first + r][x_first]: the row is
  // fetched as aligned dwords and shifted into place, so pass 1 addresses every row alike.
  constexpr int kQuads = kTileW / 4;
  constexpr int kHPitch = C_OUT * kQuads + 1;  // dwords per filtered row; odd, so lanes = rows hit distinct banks
  const int x_first = xb[ox0].x;
  uint32_t* const swords = hbuf + prm.max_rows * kHPitch;
  const int pitch_words = prm.src_pitch >> 2;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  if constexpr (kStaged) {
    const int2 xl = xb[ox0 + tile_w - 1];
    const int row_bytes = (xl.x + xl.y - x_first) * img.channels;
    const int words = (row_bytes + 3) >> 2;
    if (words > pitch_words) return;  // host sized the pitch from the widest tile; never overrun it
    const uint8_t* const img_end = img.src + static_cast<size_t>(img.height) * img.stride;
    auto load_word = [&](const uint8_t* wp) -> uint32_t {
      if (wp >= img.src && wp + 4 <= img_end) return *reinterpret_cast<const uint32_t*>(wp);
      uint32_t v = 0;  // the word straddles the image's first / last byte: only bytes of the image are touched
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (wp + k >= img.src && wp + k < img_end) v |= static_cast<uint32_t>(wp[k]) << (8 * k);
      return v;
    };
    for (int r = wave; r < rows; r += kThreads / 64) {
      const uint8_t* start = img.src + static_cast<size_t>(row_first + r) * img.stride + static_cast<size_t>(x_first) * img.channels;
      const uint32_t shift = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(start) & 3);
      const uint8_t* base = start - shift;
      for (int d = lane; d < words; d += 64) {
        const uint32_t lo = load_word(base + 4 * d);
        const uint32_t hi = shift ? load_word(base + 4 * d + 4) : 0u;
        swords[r * pitch_words + d] = static_cast<uint32_t>(((static_cast<uint64_t>(hi) << 32) | lo) >> (8 * shift));
      }
    }
    __syncthreads();
  }

  // ---------------- pass 1: horizontal ----------------
  if constexpr (kStaged) {
    // lane = source row, wave = group of 4 output columns: window and weights of a column are the same
    // for the whole wave (scalar loads), and lane r reads byte r * pitch + const - with an odd dword
    // pitch the rows sit on distinct LDS banks, reads and the packed b32 stores alike.
    const uint8_t* const sb = reinterpret_cast<const uint8_t*>(swords);
    for (int r0 = 0; r0 < rows; r0 += 64) {
      const int r = r0 + lane;
      const bool live = r < rows;
      const uint8_t* const lrow = sb + (live ? r : rows - 1) * prm.src_pitch;
      for (int q = wave; q < kQuads; q += kThreads / 64) {
        uint32_t packed[C_OUT];
#pragma unroll
        for (int c = 0; c < C_OUT; ++c) packed[c] = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int x = q * 4 + j;
          if (x < tile_w) {
            const int2 b = xb[ox0 + x];
            const int* __restrict__ k = xk + static_cast<size_t>(ox0 + x) * img.kx;
            int acc[C_OUT];
#pragma unroll
            for (int c = 0; c < C_OUT; ++c) acc[c] = 1 << (kPrecisionBits - 1);
            const uint8_t* p = lrow + (b.x - x_first) * img.channels;
            for (int t = 0; t < b.y; ++t) {
              uint32_t px[C_OUT];
              LoadPixel<C_OUT>(p, img.channels, img.bgr, px);
              const int w = k[t];
#pragma unroll
              for (int c = 0; c < C_OUT; ++c) acc[c] += static_cast<int>(px[c]) * w;
              p += img.channels;
            }
#pragma unroll
            for (int c = 0; c < C_OUT; ++c) packed[c] |= Clip8(acc[c]) << (8 * j);
          }
        }
        if (live) {
#pragma unroll
          for (int c = 0; c < C_OUT; ++c) hbuf[r * kHPitch + c * kQuads + q] = packed[c];
        }
      }
    }
  } else {
    // source too wide for LDS: rows x (64/4) work items reading the source from memory
    const int items1 = rows * kQuads;
    for (int item = threadIdx.x; item < items1; item += kThreads) {
      const int r = item / kQuads;
      const int q = item - r * kQuads;
      const uint8_t* __restrict__ srow = img.src + static_cast<size_t>(row_first + r) * img.stride;
      uint32_t packed[C_OUT];
#pragma unroll
      for (int c = 0; c < C_OUT; ++c) packed[c] = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int x = q * 4 + j;
        if (x < tile_w) {
          const int2 b = xb[ox0 + x];
          const int* __restrict__ k = xk + static_cast<size_t>(ox0 + x) * img.kx;
          int acc[C_OUT];
#pragma unroll
          for (int c = 0; c < C_OUT; ++c) acc[c] = 1 << (kPrecisionBits - 1);
          const uint8_t* __restrict__ p = srow + static_cast<size_t>(b.x) * img.channels;
          for (int t = 0; t < b.y; ++t) {
            uint32_t px[C_OUT];
            LoadPixel<C_OUT>(p, img.channels, img.bgr, px);
            const int w = k[t];
#pragma unroll
            for (int c = 0; c < C_OUT; ++c) acc[c] += static_cast<int>(px[c]) * w;
            p += img.channels;
          }
#pragma unroll
          for (int c = 0; c < C_OUT; ++c) packed[c] |= Clip8(acc[c]) << (8 * j);
        }
      }
#pragma unroll
      for (int c = 0; c < C_OUT; ++c) hbuf[r * kHPitch + c * kQuads + q] = packed[c];
    }
  }
  __syncthreads();

  // ---------------- pass 2: vertical + epilogue, tile_h x (64/2) items ----
  constexpr int kPairs = kTileW / 2;
  const uint16_t* const hpairs = reinterpret_cast<const uint16_t*>(hbuf);
  const int items2 = tile_h * kPairs;
  for (int item = threadIdx.x; item < items2; item += kThreads) {
    const int ty = item / kPairs;
    const int pq = item - ty * kPairs;
    const int x = pq * 2;
    if (x >= tile_w) continue;
    const int valid = min(2, tile_w - x);
    const int oy = oy0 + ty;
    const int2 b = yb[oy];
    const int* __restrict__ k = yk + static_cast<size_t>(oy) * img.ky;
    const int r0 = b.x - row_first;

    T out[C_OUT][2];
#pragma unroll
    for (int c = 0; c < C_OUT; ++c) {
      int acc0 = 1 << (kPrecisionBits - 1), acc1 = acc0;
      for (int t = 0; t < b.y; ++t) {
        const uint32_t w2 = hpairs[(r0 + t) * (2 * kHPitch) + c * kPairs + pq];
        const int w = k[t];
        acc0 += static_cast<int>(w2 & 0xFFu) * w;
        acc1 += static_cast<int>(w2 >> 8) * w;
      }
      const float s = prm.scale[c], o = prm.bias[c];
      out[c][0] = Caster<T>::To(fmaf(static_cast<float>(Clip8(acc0)), s, o));
      out[c][1] = Caster<T>::To(fmaf(static_cast<float>(Clip8(acc1)), s, o));
    }

    T* __restrict__ base = dst + static_cast<size_t>(blockIdx.z) * prm.image_stride;
    if (prm.nhwc) {
      T* __restrict__ p = base + (static_cast<size_t>(oy) * prm.out_w + ox0 + x) * C_OUT;
      if constexpr (C_OUT == 1) {
        Store2<T>(p, out[0], valid);
      } else {
        for (int j = 0; j < valid; ++j)
#pragma unroll
          for (int c = 0; c < C_OUT; ++c) p[j * C_OUT + c] = out[c][j];
      }
    } else {
      const size_t plane = static_cast<size_t>(prm.out_h) * prm.out_w;
#pragma unroll
      for (int c = 0; c < C_OUT; ++c) {
        T* __restrict__ p = base + c * plane + static_cast<size_t>(oy) * prm.out_w + ox0 + x;
        Store2<T>(p, out[c], valid);
      }
    }
  }
}

template <typename T>
hipError_t Launch(const tca::PreprocImage* images, const int2* bounds, const int* coeffs, void* dst,
                  const tca::PreprocParams& prm, int batch, hipStream_t stream) {
  dim3 grid((prm.out_w + kTileW - 1) / kTileW, (prm.out_h + prm.tile_h - 1) / prm.tile_h, batch);
  const size_t lds = static_cast<size_t>(prm.max_rows) * (prm.out_c * kTileW + 4 + prm.src_pitch);
#define TCA_PREPROCESS(C_, STAGED_)                                                                              \
  hipLaunchKernelGGL((PreprocessKernel<T, C_, STAGED_>), grid, dim3(kThreads), lds, stream, images, bounds, coeffs, \
                     static_cast<T*>(dst), prm)
  if (prm.out_c == 1) {
    if (prm.src_pitch > 0) TCA_PREPROCESS(1, true);
    else TCA_PREPROCESS(1, false);
  } else {
    if (prm.src_pitch > 0) TCA_PREPROCESS(3, true);
    else TCA_PREPROCESS(3, false);
  }
#undef TCA_PREPROCESS
  return hipGetLastError();
}

}  // namespace

extern "C" int tca_preprocess_launch(const void* images, const void* bounds, const void* coeffs, void* dst,
                                     const tca::PreprocParams* prm, int batch, int dtype, void* stream) {
  if (!images || !bounds || !coeffs || !dst || !prm || batch <= 0) return -1;
  if (prm->out_c != 1 && prm->out_c != 3) return -2;
  if (prm->tile_h < 1 || prm->tile_h > 16 || prm->max_rows < 1) return -3;
  if (prm->src_pitch < 0 || (prm->src_pitch & 3)) return -3;
  if (static_cast<size_t>(prm->max_rows) * (prm->out_c * kTileW + 4 + prm->src_pitch) > 64 * 1024) return -4;  // default dynamic LDS limit per block
  auto* im = static_cast<const tca::PreprocImage*>(images);
  auto* bd = static_cast<const int2*>(bounds);
  auto* cf = static_cast<const int*>(coeffs);
  auto st = static_cast<hipStream_t>(stream);
  hipError_t e;
  switch (dtype) {
    case tca::kF32: e = Launch<float>(im, bd, cf, dst, *prm, batch, st); break;
    case tca::kF16: e = Launch<__half>(im, bd, cf, dst, *prm, batch, st); break;
    case tca::kBF16: e = Launch<Bf16>(im, bd, cf, dst, *prm, batch, st); break;
    case tca::kI8: e = Launch<int8_t>(im, bd, cf, dst, *prm, batch, st); break;
    case tca::kU8: e = Launch<uint8_t>(im, bd, cf, dst, *prm, batch, st); break;
    case tca::kI16: e = Launch<int16_t>(im, bd, cf, dst, *prm, batch, st); break;
    case tca::kU16: e = Launch<uint16_t>(im, bd, cf, dst, *prm, batch, st); break;
    case tca::kI32: e = Launch<int32_t>(im, bd, cf, dst, *prm, batch, st); break;
    case tca::kU32: e = Launch<uint32_t>(im, bd, cf, dst, *prm, batch, st); break;
    case tca::kF64: e = Launch<double>(im, bd, cf, dst, *prm, batch, st); break;
    default: return -5;
  }
  return e == hipSuccess ? 0 : -1000 - static_cast<int>(e);
}
