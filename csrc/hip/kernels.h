// Host/device shared declarations for the CDNA4 kernels in this directory.
// All launchers have C linkage and take raw device pointers plus a hipStream_t
// (as void*), so they can be driven from C++ and from Python/ctypes alike.
#pragma once

#include <cstdint>

namespace tca {

// Element types understood by the kernels (values are part of the C ABI).
enum DType : int { kF32 = 0, kF16 = 1, kBF16 = 2, kI8 = 3, kU8 = 4, kI16 = 5, kU16 = 6, kI32 = 7, kU32 = 8, kF64 = 9 };

// One source image of a preprocessing batch (device-visible array).
struct PreprocImage {
  const uint8_t* src;  // HWC, 8 bit, device memory
  int height;
  int width;
  int stride;    // bytes between rows
  int channels;  // 1, 3 or 4
  int bgr;       // 1: B,G,R(,A) order
  int kx;        // taps per output column (row pitch of the x coefficient table)
  int ky;        // taps per output row
  int xb_off;    // index of this image's first x window in `bounds`
  int yb_off;    // index of its first y window
  int xk_off;    // index of its first x coefficient in `coeffs`
  int yk_off;    // index of its first y coefficient
  int pad_;
};

struct PreprocParams {
  int out_h;
  int out_w;
  int out_c;     // 1 or 3
  int nhwc;      // 0: NCHW planes, 1: NHWC interleaved
  int tile_h;    // output rows per block (1..16)
  int max_rows;  // max source rows any tile needs (sizes the LDS buffer)
  // bytes per staged source row in LDS (multiple of 4, >= widest tile's source bytes + 6), or 0: do not
  // stage the source (pass 1 then reads it from memory)
  int src_pitch;
  int pad_;
  long long image_stride;  // elements between consecutive batch entries in dst
  float scale[4];
  float bias[4];
};

}  // namespace tca

extern "C" {
// Fused convert+resize+normalise+layout+cast.  Returns 0 or a negative error.
int tca_preprocess_launch(const void* images, const void* bounds, const void* coeffs, void* dst,
                          const tca::PreprocParams* prm, int batch, int dtype, void* stream);

// BYTES tensors (length-prefixed wire form).
int tca_bytes_pack_launch(const uint8_t* payload, const uint64_t* offsets, uint64_t n, uint8_t* out,
                          uint64_t out_bytes, void* stream);
int tca_bytes_index_launch(const uint8_t* stream_bytes, uint64_t size, uint64_t max_elems, uint64_t* offsets,
                           uint32_t* lens, uint64_t* status, void* stream);
uint64_t tca_bytes_index_workspace_bytes(uint64_t size);
int tca_bytes_index_parallel_launch(const uint8_t* stream_bytes, uint64_t size, uint64_t max_elems, uint64_t* offsets,
                                    uint32_t* lens, uint64_t* status, void* workspace, uint64_t workspace_bytes,
                                    void* stream);
int tca_bytes_gather_launch(const uint8_t* stream_bytes, const uint64_t* offsets, const uint32_t* lens, uint64_t n,
                            uint8_t* out, uint32_t pitch, void* stream);
int tca_exclusive_scan_u64_launch(const uint32_t* lens, uint64_t n, uint32_t add_each, uint64_t* out,
                                  void* stream);

// BF16 <-> FP32 (truncate / round-to-nearest-even selectable).
int tca_fp32_to_bf16_launch(const float* src, uint16_t* dst, uint64_t n, int round_nearest, void* stream);
int tca_bf16_to_fp32_launch(const uint16_t* src, float* dst, uint64_t n, void* stream);

// Classification head helpers.
int tca_topk_launch(const void* logits, int dtype, int rows, int cols, int k, float* values, int* indices,
                    void* stream);
int tca_classify_strings_launch(const float* values, const int* indices, int rows, int k, const uint8_t* labels,
                                const uint32_t* label_offsets, int num_labels, uint8_t* out, uint64_t out_cap,
                                uint64_t* out_size, void* stream);

// NHWC fp16 convolution pieces.  1x1 convolution as an MFMA GEMM with bias /
// residual / ReLU in its epilogue (x: [batch,in_h,in_w,cin], w: [cout,cin],
// y: [batch,oh,ow,cout]; variant 0 = pick the block tile), and the same
// epilogue as an in-place pass over a library convolution's output.
int tca_conv1x1_launch(const void* x, const void* w, const void* bias, const void* residual, void* y, int batch,
                       int in_h, int in_w, int stride, int cin, int cout, int relu, int variant, void* stream);
int tca_conv_gemm_launch(const void* x, const void* w, const void* bias, const void* residual, void* y, int batch,
                         int in_h, int in_w, int ksize, int stride, int pad, int cin, int cout, int relu, int variant,
                         void* stream);
int tca_avgpool_launch(const void* y, void* feats, int batch, int pixels, int channels, void* stream);
int tca_linear_launch(const void* x, const void* w, const void* bias, void* out, int batch, int n_out, int k_total,
                      void* stream);
int tca_conv_pointwise_launch(const void* x, const void* w1, const void* bias1, const void* w2, const void* bias2,
                              const void* residual, void* y, int batch, int in_h, int in_w, int ksize, int stride, int pad,
                              int cin, int mid, int cout, int relu, const void* w3, const void* bias3, void* out3, int n3,
                              int tile_m, const void* xd, void* stream);
// Bottleneck tail with 64 middle channels organised by rows (bottleneck_rows.hip): 3x3 convolution, 1x1
// expansion to 256 channels (+ projection shortcut from xd, or residual), ReLU, and the next block's first
// 1x1 convolution (n3 = 64 or 128 outputs), no workgroup barrier behind the 3x3 part.  w2's first 64 K
// columns are stored in accumulator order (ops/conv.py: permute_mid_k).
int tca_bottleneck_rows_launch(const void* x, const void* w1, const void* bias1, const void* w2, const void* bias2,
                               const void* residual, void* y, int batch, int h, int w, int cout, const void* w3,
                               const void* bias3, void* out3, int n3, const void* xd, void* stream);
int tca_conv_gemm_splitk_launch(const void* x, const void* w, const void* bias, const void* residual, void* y, int batch,
                                int in_h, int in_w, int ksize, int stride, int pad, int cin, int cout, int relu,
                                int variant, int splits, void* partials, unsigned long long partial_bytes,
                                void* arrivals, int arrival_slots, void* stream);
// n (source, destination, size) device copies in one launch (64 per kernel).
int tca_batched_copy_launch(const void* const* srcs, void* const* dsts, const uint64_t* bytes, int n, void* stream);
int tca_bias_act_launch(void* y, const void* bias, const void* residual, long long rows, int channels, int relu,
                        void* stream);

// y[batch][oh][ow][channels] = relu(maxpool3x3/2/pad1(x[batch][in_h][in_w][channels]) + bias), fp16 NHWC, channels % 8 == 0.
int tca_bias_relu_maxpool_launch(const void* x, const void* bias, void* y, int batch, int in_h, int in_w, int channels,
                                 void* stream);

// Fused network stem (conv 7x7/2 pad 3 over 3 channels + bias + ReLU + max pool 3x3/2 pad 1).
// x: [batch][3][in_h][in_w] fp16 NCHW; wk: [64][192] fp16 (see conv_gemm.hip); y: [batch][PH][PW][64] fp16 NHWC.
int tca_stem_conv_pool_launch(const void* x, const void* wk, const void* bias, void* y, int batch, int in_h, int in_w,
                              void* stream);
}
