// BYTES tensor (length-prefixed strings) pack / index / gather and BF16
// conversion kernels for gfx950.
//
// Wire form: for each element, u32 little-endian length then the payload
// (reference: `tritonclient/utils/__init__.py:219-276`, a per-element Python
// loop; `src/c++/library/common.cc:168-183`).
//
//  pack    CSR (payload + n+1 offsets) -> wire.  Output-stationary: every
//          thread produces 16 consecutive OUTPUT bytes (one dwordx4 store), so
//          stores are perfectly coalesced whatever the element sizes are; the
//          owning element is found by binary search over `offsets[i] + 4*i`.
//  scan    lengths -> exclusive prefix sums (+4 per element) when the caller
//          only has lengths.  Single pass over 64-wide wavefront shuffles.
//  index   wire -> (payload offset, length) per element.  The prefix chain is
//          inherently serial, so one workgroup streams the buffer through LDS
//          (coalesced 16-byte loads by all 256 threads) and a single lane walks
//          the chain out of LDS; the 100-cycle LDS hop replaces a ~2 us HBM hop
//          per element.  `status` reports count and the first malformed offset.
//          Streams of more than a few chunks use the chunk-parallel form (see
//          "parallel index" below): serial only across 4 KiB chunks.
//  gather  wire + index -> fixed pitch rows (zero padded) for device-side
//          consumers such as a tokenizer front end.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels.h"

namespace {

constexpr int kThreads = 256;

// ------------------------------------------------------------------- pack
__global__ __launch_bounds__(kThreads) void BytesPackKernel(const uint8_t* __restrict__ payload,
                                                            const uint64_t* __restrict__ offsets, uint64_t n,
                                                            uint8_t* __restrict__ out, uint64_t out_bytes) {
  const uint64_t chunks = (out_bytes + 15) / 16;
  for (uint64_t chunk = blockIdx.x * static_cast<uint64_t>(kThreads) + threadIdx.x; chunk < chunks;
       chunk += static_cast<uint64_t>(gridDim.x) * kThreads) {
    const uint64_t p0 = chunk * 16;
    // last element e with start(e) = offsets[e] + 4e <= p0
    uint64_t lo = 0, hi = n;  // invariant: start(lo) <= p0 < start(hi)  (start(n) = out_bytes)
    while (hi - lo > 1) {
      const uint64_t mid = (lo + hi) >> 1;
      if (offsets[mid] + 4 * mid <= p0) lo = mid; else hi = mid;
    }
    uint64_t e = lo;
    uint64_t start = offsets[e] + 4 * e;
    uint64_t next = offsets[e + 1] + 4 * (e + 1);
    uint32_t len = static_cast<uint32_t>(offsets[e + 1] - offsets[e]);
    alignas(16) uint8_t buf[16];
    const int valid = static_cast<int>(out_bytes - p0 < 16 ? out_bytes - p0 : 16);
    for (int j = 0; j < valid; ++j) {
      const uint64_t p = p0 + j;
      while (p >= next) {  // skips empty-payload elements too (their prefix still occupies 4 bytes)
        ++e;
        start = next;
        next = offsets[e + 1] + 4 * (e + 1);
        len = static_cast<uint32_t>(offsets[e + 1] - offsets[e]);
      }
      const uint64_t rel = p - start;
      buf[j] = rel < 4 ? static_cast<uint8_t>(len >> (8 * rel)) : payload[offsets[e] + rel - 4];
    }
    if (valid == 16) {
      *reinterpret_cast<uint4*>(out + p0) = *reinterpret_cast<const uint4*>(buf);
    } else {
      for (int j = 0; j < valid; ++j) out[p0 + j] = buf[j];
    }
  }
}

// ------------------------------------------------------------------- scan
// out[i] = sum_{j<i} (lens[j] + add_each), out[n] = total.  One workgroup:
// element counts of BYTES tensors are small compared with their byte volume.
__global__ __launch_bounds__(kThreads) void ExclusiveScanKernel(const uint32_t* __restrict__ lens, uint64_t n,
                                                                uint32_t add_each, uint64_t* __restrict__ out) {
  __shared__ uint64_t wave_sums[kThreads / 64];
  __shared__ uint64_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  for (uint64_t base = 0; base < n; base += kThreads) {
    const uint64_t i = base + threadIdx.x;
    uint64_t v = i < n ? static_cast<uint64_t>(lens[i]) + add_each : 0;
    uint64_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint64_t up = __shfl_up(incl, d, 64);
      if (lane >= d) incl += up;
    }
    if (lane == 63) wave_sums[wave] = incl;
    __syncthreads();
    uint64_t before = carry;
    for (int w = 0; w < wave; ++w) before += wave_sums[w];
    if (i < n) out[i] = before + incl - v;
    __syncthreads();
    if (threadIdx.x == kThreads - 1) carry = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) out[n] = carry;
}

// ------------------------------------------------------------------ index
constexpr uint32_t kIndexChunk = 48 * 1024;  // bytes of the stream resident in LDS at a time

__global__ __launch_bounds__(kThreads) void BytesIndexKernel(const uint8_t* __restrict__ s, uint64_t size,
                                                             uint64_t max_elems, uint64_t* __restrict__ offsets,
                                                             uint32_t* __restrict__ lens,
                                                             uint64_t* __restrict__ status) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[kIndexChunk + 16];
  __shared__ uint64_t cursor_s;   // absolute position of the next length prefix
  __shared__ uint64_t count_s;
  __shared__ int state_s;         // 0 running, 1 done, 2 malformed
  if (threadIdx.x == 0) { cursor_s = 0; count_s = 0; state_s = size == 0 ? 1 : 0; }
  __syncthreads();
  while (state_s == 0) {
    const uint64_t base = cursor_s & ~static_cast<uint64_t>(15);  // keep global loads 16-byte aligned
    const uint64_t avail = size - base < kIndexChunk ? size - base : kIndexChunk;
    // cooperative, coalesced stage: 256 threads x 16 bytes per iteration
    for (uint64_t off = static_cast<uint64_t>(threadIdx.x) * 16; off < avail; off += kThreads * 16) {
      if (off + 16 <= avail && ((reinterpret_cast<uintptr_t>(s + base + off) & 15) == 0)) {
        *reinterpret_cast<uint4*>(lds + off) = *reinterpret_cast<const uint4*>(s + base + off);
      } else {
        for (uint64_t j = off; j < avail && j < off + 16; ++j) lds[j] = s[base + j];
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      uint64_t cur = cursor_s, cnt = count_s;
      int state = 0;
      const uint64_t end = base + avail;
      while (true) {
        if (cnt >= max_elems || cur == size) { state = 1; break; }
        if (cur + 4 > size) { state = 2; break; }
        if (cur + 4 > end) break;  // prefix straddles the chunk: restage from cur
        // The walk is one dependent LDS read per element: fetch the (unaligned) prefix as
        // the two aligned dwords around it - they issue together, one LDS latency - instead
        // of four byte reads.
        const uint32_t rel = static_cast<uint32_t>(cur - base);
        const uint32_t* words = reinterpret_cast<const uint32_t*>(lds + (rel & ~3u));
        const uint32_t lo = words[0], hi = words[1];
        const uint32_t shift = (rel & 3u) * 8u;
        const uint32_t len = shift ? (lo >> shift) | (hi << (32u - shift)) : lo;
        if (len > size - (cur + 4)) { state = 2; break; }
        offsets[cnt] = cur + 4;
        lens[cnt] = len;
        ++cnt;
        cur += 4 + static_cast<uint64_t>(len);
        if (cur >= end && cur != size) break;  // next prefix is outside the staged window
      }
      cursor_s = cur;
      count_s = cnt;
      state_s = state;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    status[0] = count_s;
    status[1] = state_s == 2 ? 1 : 0;
    status[2] = cursor_s;
  }
}

// ---------------------------------------------------------- parallel index
// The prefix chain is serial only ACROSS chunks.  Inside a chunk of kMapChunk bytes the
// question "entering at byte e, where does the chain leave the chunk and how many
// elements does it pass" is answered for EVERY e at once by pointer jumping in LDS
// (each position first points at the header that would follow it; log2 rounds of
// "jump to my successor's successor" collapse every path).  A single lane then crosses
// the stream with ONE table lookup per chunk instead of one dependent load per
// element, and finally every chunk that the chain entered emits its elements on its own.
//
//   map    (one workgroup per chunk)  table[c][e] = packed {exit, count, flag}
//   chain  (one lane)                 entry byte + first element number of every chunk
//   emit   (one workgroup per chunk)  offsets[] / lens[] of the chunk's elements
constexpr uint32_t kMapChunk = 4096;
constexpr uint32_t kMapExited = 0x1FFF;          // 13-bit "next" field: the path has left the chunk
constexpr uint32_t kFlagOk = 0, kFlagEnd = 1, kFlagBad = 2;
constexpr uint32_t kNoEntry = 0xFFFFFFFFu;

// 64-bit node: [12:0] next header in the chunk (or kMapExited), [23:13] headers passed,
// [25:24] flag, [63:26] once exited: bytes past the chunk end where the chain continues
// (END / BAD: the position inside the chunk where it stopped).
__device__ inline uint64_t MapNode(uint32_t next, uint32_t count, uint32_t flag, uint64_t payload) {
  return static_cast<uint64_t>(next) | (static_cast<uint64_t>(count) << 13) | (static_cast<uint64_t>(flag) << 24) |
         (payload << 26);
}
__device__ inline uint32_t NodeNext(uint64_t t) { return static_cast<uint32_t>(t) & 0x1FFFu; }
__device__ inline uint32_t NodeCount(uint64_t t) { return (static_cast<uint32_t>(t) >> 13) & 0x7FFu; }
__device__ inline uint32_t NodeFlag(uint64_t t) { return (static_cast<uint32_t>(t) >> 24) & 3u; }
__device__ inline uint64_t NodePayload(uint64_t t) { return t >> 26; }

// Stages bytes [base, base + kMapChunk + 8) of the stream (zero past its end) into LDS.
__device__ inline void StageChunk(const uint8_t* __restrict__ s, uint64_t size, uint64_t base, uint8_t* lds) {
  const uint64_t avail = size - base < kMapChunk + 8 ? size - base : kMapChunk + 8;
  const bool aligned = (reinterpret_cast<uintptr_t>(s + base) & 15) == 0;
  for (uint32_t off = threadIdx.x * 16; off < kMapChunk + 16; off += kThreads * 16) {
    if (aligned && off + 16 <= avail) {
      *reinterpret_cast<uint4*>(lds + off) = *reinterpret_cast<const uint4*>(s + base + off);
    } else {
      for (uint32_t j = off; j < off + 16; ++j) lds[j] = j < avail ? s[base + j] : 0;
    }
  }
}

__device__ inline uint32_t LdsPrefix(const uint8_t* lds, uint32_t rel) {
  const uint32_t* words = reinterpret_cast<const uint32_t*>(lds + (rel & ~3u));
  const uint32_t lo = words[0], hi = words[1];
  const uint32_t shift = (rel & 3u) * 8u;
  return shift ? (lo >> shift) | (hi << (32u - shift)) : lo;
}

__global__ __launch_bounds__(kThreads) void BytesChunkMapKernel(const uint8_t* __restrict__ s, uint64_t size,
                                                                uint64_t* __restrict__ table,
                                                                uint32_t* __restrict__ chunk_entry) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[kMapChunk + 16];
  __shared__ uint64_t node[kMapChunk];
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kMapChunk;
  if (threadIdx.x == 0) chunk_entry[blockIdx.x] = kNoEntry;
  StageChunk(s, size, base, lds);
  __syncthreads();
  for (uint32_t e = threadIdx.x; e < kMapChunk; e += kThreads) {
    const uint64_t abs = base + e;
    uint64_t t;
    if (abs == size) {
      t = MapNode(kMapExited, 0, kFlagEnd, e);
    } else if (abs + 4 > size) {
      t = MapNode(kMapExited, 0, kFlagBad, e);
    } else {
      const uint32_t len = LdsPrefix(lds, e);
      if (len > size - (abs + 4)) {
        t = MapNode(kMapExited, 0, kFlagBad, e);
      } else {
        const uint64_t q = static_cast<uint64_t>(e) + 4 + len;
        t = q >= kMapChunk ? MapNode(kMapExited, 1, kFlagOk, q - kMapChunk) : MapNode(static_cast<uint32_t>(q), 1, kFlagOk, 0);
      }
    }
    node[e] = t;
  }
  __syncthreads();
  // Pointer jumping.  A node is one 64-bit LDS word, read and written whole, so rounds
  // need no double buffering: whatever mix of old and new successors a lane sees, the
  // invariant "count = headers from e up to, not including, next" holds.  At most
  // kMapChunk/4 headers lie on a path, so 11 doublings always finish; usually far fewer.
  for (int round = 0; round < 12; ++round) {
    int pending = 0;
    for (uint32_t e = threadIdx.x; e < kMapChunk; e += kThreads) {
      const uint64_t t = node[e];
      const uint32_t nx = NodeNext(t);
      if (nx != kMapExited) {
        const uint64_t u = node[nx];
        node[e] = MapNode(NodeNext(u), NodeCount(t) + NodeCount(u), NodeFlag(u), NodePayload(u));
        pending |= NodeNext(u) != kMapExited;
      }
    }
    if (!__syncthreads_or(pending)) break;
  }
  uint64_t* out = table + base;
  for (uint32_t e = threadIdx.x; e < kMapChunk; e += kThreads) out[e] = node[e];
}

__global__ void BytesChainKernel(const uint64_t* __restrict__ table, uint64_t size, uint64_t max_elems,
                                 uint32_t* __restrict__ chunk_entry, uint64_t* __restrict__ chunk_first,
                                 uint64_t* __restrict__ status) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint64_t pos = 0, count = 0;
  int bad = 0;
  while (count < max_elems && pos != size) {
    const uint64_t c = pos / kMapChunk;
    const uint32_t e = static_cast<uint32_t>(pos % kMapChunk);
    const uint64_t t = table[c * kMapChunk + e];
    chunk_entry[c] = e;
    chunk_first[c] = count;
    count += NodeCount(t);
    const uint32_t flag = NodeFlag(t);
    if (flag == kFlagOk) {
      pos = (c + 1) * kMapChunk + NodePayload(t);
    } else {
      pos = c * kMapChunk + NodePayload(t);
      bad = flag == kFlagBad && count < max_elems;  // a broken prefix past the requested elements is never looked at
      break;
    }
  }
  status[0] = count < max_elems ? count : max_elems;
  status[1] = bad;
  status[2] = pos;
}

__global__ __launch_bounds__(kThreads) void BytesChunkEmitKernel(const uint8_t* __restrict__ s, uint64_t size,
                                                                 uint64_t max_elems,
                                                                 const uint32_t* __restrict__ chunk_entry,
                                                                 const uint64_t* __restrict__ chunk_first,
                                                                 uint64_t* __restrict__ offsets,
                                                                 uint32_t* __restrict__ lens) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[kMapChunk + 16];
  const uint32_t entry = chunk_entry[blockIdx.x];
  if (entry == kNoEntry) return;  // uniform: the chain never starts an element in this chunk
  const uint64_t base = static_cast<uint64_t>(blockIdx.x) * kMapChunk;
  StageChunk(s, size, base, lds);
  __syncthreads();
  if (threadIdx.x != 0) return;
  uint64_t idx = chunk_first[blockIdx.x];
  uint64_t p = entry;
  while (p < kMapChunk && idx < max_elems) {
    const uint64_t abs = base + p;
    if (abs + 4 > size) break;  // clean end (abs == size) or a truncated prefix the chain kernel reported
    const uint32_t len = LdsPrefix(lds, static_cast<uint32_t>(p));
    if (len > size - (abs + 4)) break;
    offsets[idx] = abs + 4;
    lens[idx] = len;
    ++idx;
    p += 4 + static_cast<uint64_t>(len);
  }
}

// ----------------------------------------------------------------- gather
// One wavefront per element: lanes copy the payload with coalesced byte loads
// and zero-fill the rest of the row.
__global__ __launch_bounds__(kThreads) void BytesGatherKernel(const uint8_t* __restrict__ s,
                                                              const uint64_t* __restrict__ offsets,
                                                              const uint32_t* __restrict__ lens, uint64_t n,
                                                              uint8_t* __restrict__ out, uint32_t pitch) {
  const int lane = threadIdx.x & 63;
  const uint64_t waves = static_cast<uint64_t>(gridDim.x) * (kThreads / 64);
  for (uint64_t e = blockIdx.x * static_cast<uint64_t>(kThreads / 64) + (threadIdx.x >> 6); e < n; e += waves) {
    const uint8_t* __restrict__ src = s + offsets[e];
    const uint32_t len = lens[e] < pitch ? lens[e] : pitch;
    uint8_t* __restrict__ dst = out + e * pitch;
    for (uint32_t j = lane; j < pitch; j += 64) dst[j] = j < len ? src[j] : 0;
  }
}

// ------------------------------------------------------------------- bf16
__global__ __launch_bounds__(kThreads) void Fp32ToBf16Kernel(const float* __restrict__ src,
                                                             uint16_t* __restrict__ dst, uint64_t n, int rne) {
  const uint64_t quads = n / 4;
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kThreads;
  for (uint64_t q = blockIdx.x * static_cast<uint64_t>(kThreads) + threadIdx.x; q < quads; q += stride) {
    const uint4 w = reinterpret_cast<const uint4*>(src)[q];
    uint32_t v[4] = {w.x, w.y, w.z, w.w};
    if (rne) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if ((v[j] & 0x7FFFFFFFu) <= 0x7F800000u) v[j] += 0x7FFFu + ((v[j] >> 16) & 1u);
    }
    uint2 o;
    o.x = (v[0] >> 16) | (v[1] & 0xFFFF0000u);
    o.y = (v[2] >> 16) | (v[3] & 0xFFFF0000u);
    reinterpret_cast<uint2*>(dst)[q] = o;
  }
  // tail (n not a multiple of 4)
  for (uint64_t i = quads * 4 + blockIdx.x * static_cast<uint64_t>(kThreads) + threadIdx.x; i < n; i += stride) {
    uint32_t v = __float_as_uint(src[i]);
    if (rne && (v & 0x7FFFFFFFu) <= 0x7F800000u) v += 0x7FFFu + ((v >> 16) & 1u);
    dst[i] = static_cast<uint16_t>(v >> 16);
  }
}

__global__ __launch_bounds__(kThreads) void Bf16ToFp32Kernel(const uint16_t* __restrict__ src,
                                                             float* __restrict__ dst, uint64_t n) {
  const uint64_t quads = n / 4;
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kThreads;
  for (uint64_t q = blockIdx.x * static_cast<uint64_t>(kThreads) + threadIdx.x; q < quads; q += stride) {
    const uint2 w = reinterpret_cast<const uint2*>(src)[q];
    uint4 o;
    o.x = w.x << 16;
    o.y = w.x & 0xFFFF0000u;
    o.z = w.y << 16;
    o.w = w.y & 0xFFFF0000u;
    reinterpret_cast<uint4*>(dst)[q] = o;
  }
  for (uint64_t i = quads * 4 + blockIdx.x * static_cast<uint64_t>(kThreads) + threadIdx.x; i < n; i += stride)
    dst[i] = __uint_as_float(static_cast<uint32_t>(src[i]) << 16);
}

int GridFor(uint64_t work_items) {
  uint64_t blocks = (work_items + kThreads - 1) / kThreads;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;  // 256 CUs x 8 blocks, grid-stride beyond that
  return static_cast<int>(blocks);
}

int Status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -1000 - static_cast<int>(e);
}

}  // namespace

extern "C" {

int tca_bytes_pack_launch(const uint8_t* payload, const uint64_t* offsets, uint64_t n, uint8_t* out,
                          uint64_t out_bytes, void* stream) {
  if (n == 0 || out_bytes == 0) return 0;
  if (!offsets || !out) return -1;
  hipLaunchKernelGGL(BytesPackKernel, dim3(GridFor((out_bytes + 15) / 16)), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream), payload, offsets, n, out, out_bytes);
  return Status();
}

int tca_exclusive_scan_u64_launch(const uint32_t* lens, uint64_t n, uint32_t add_each, uint64_t* out, void* stream) {
  if (!out || (n && !lens)) return -1;
  hipLaunchKernelGGL(ExclusiveScanKernel, dim3(1), dim3(kThreads), 0, static_cast<hipStream_t>(stream), lens, n,
                     add_each, out);
  return Status();
}

int tca_bytes_index_launch(const uint8_t* stream_bytes, uint64_t size, uint64_t max_elems, uint64_t* offsets,
                           uint32_t* lens, uint64_t* status, void* stream) {
  if (!status || (size && !stream_bytes) || (max_elems && (!offsets || !lens))) return -1;
  hipLaunchKernelGGL(BytesIndexKernel, dim3(1), dim3(kThreads), 0, static_cast<hipStream_t>(stream), stream_bytes,
                     size, max_elems, offsets, lens, status);
  return Status();
}

// Workspace of the parallel index: the chunk table (8 bytes per stream byte, rounded up to
// whole chunks) followed by 4 + 8 bytes per chunk.
uint64_t tca_bytes_index_workspace_bytes(uint64_t size) {
  const uint64_t chunks = (size + kMapChunk - 1) / kMapChunk;
  return chunks * kMapChunk * 8 + chunks * 8 + ((chunks * 4 + 7) & ~7ull);
}

// Same contract as tca_bytes_index_launch, with the walk spread over one workgroup per
// 4 KiB of stream.  `workspace` (device memory, 8-byte aligned, at least
// tca_bytes_index_workspace_bytes(size) bytes) needs no initialisation.
int tca_bytes_index_parallel_launch(const uint8_t* stream_bytes, uint64_t size, uint64_t max_elems, uint64_t* offsets,
                                    uint32_t* lens, uint64_t* status, void* workspace, uint64_t workspace_bytes,
                                    void* stream) {
  if (!status || (size && !stream_bytes) || (max_elems && (!offsets || !lens))) return -1;
  const uint64_t chunks = (size + kMapChunk - 1) / kMapChunk;
  if (chunks == 0 || chunks > 0x7FFFFFFFull)
    return tca_bytes_index_launch(stream_bytes, size, max_elems, offsets, lens, status, stream);
  if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 7) || workspace_bytes < tca_bytes_index_workspace_bytes(size))
    return -2;
  auto* table = static_cast<uint64_t*>(workspace);
  uint64_t* chunk_first = table + chunks * kMapChunk;
  auto* chunk_entry = reinterpret_cast<uint32_t*>(chunk_first + chunks);
  auto hs = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(BytesChunkMapKernel, dim3(static_cast<unsigned>(chunks)), dim3(kThreads), 0, hs, stream_bytes, size,
                     table, chunk_entry);
  hipLaunchKernelGGL(BytesChainKernel, dim3(1), dim3(64), 0, hs, table, size, max_elems, chunk_entry, chunk_first,
                     status);
  hipLaunchKernelGGL(BytesChunkEmitKernel, dim3(static_cast<unsigned>(chunks)), dim3(kThreads), 0, hs, stream_bytes, size,
                     max_elems, chunk_entry, chunk_first, offsets, lens);
  return Status();
}

int tca_bytes_gather_launch(const uint8_t* stream_bytes, const uint64_t* offsets, const uint32_t* lens, uint64_t n,
                            uint8_t* out, uint32_t pitch, void* stream) {
  if (n == 0 || pitch == 0) return 0;
  if (!stream_bytes || !offsets || !lens || !out) return -1;
  hipLaunchKernelGGL(BytesGatherKernel, dim3(GridFor(n * 64)), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                     stream_bytes, offsets, lens, n, out, pitch);
  return Status();
}

int tca_fp32_to_bf16_launch(const float* src, uint16_t* dst, uint64_t n, int round_nearest, void* stream) {
  if (n == 0) return 0;
  if (!src || !dst) return -1;
  if ((reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(dst) & 7)) return -2;
  hipLaunchKernelGGL(Fp32ToBf16Kernel, dim3(GridFor((n + 3) / 4)), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream), src, dst, n, round_nearest);
  return Status();
}

int tca_bf16_to_fp32_launch(const uint16_t* src, float* dst, uint64_t n, void* stream) {
  if (n == 0) return 0;
  if (!src || !dst) return -1;
  if ((reinterpret_cast<uintptr_t>(src) & 7) || (reinterpret_cast<uintptr_t>(dst) & 15)) return -2;
  hipLaunchKernelGGL(Bf16ToFp32Kernel, dim3(GridFor((n + 3) / 4)), dim3(kThreads), 0,
                     static_cast<hipStream_t>(stream), src, dst, n);
  return Status();
}

}  // extern "C"
