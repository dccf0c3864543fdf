// Image file readers for the examples and tools: PNG (on zlib), JPEG (Huffman-coded, sequential and progressive) and
// binary PNM.  The reference's C++ image clients lean on OpenCV's imread
// (`src/c++/examples/image_client.cc:85-105`); these cover the formats image classifiers are fed with
// and need nothing beyond zlib.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "tca/error.h"

namespace tca {

// 8-bit pixels, rows top to bottom, channels interleaved (HWC).
struct DecodedImage {
  int height = 0;
  int width = 0;
  int channels = 0;  // 1 (gray) or 3 (RGB)
  std::vector<uint8_t> pixels;
};

// PNG: gray, gray+alpha, RGB, RGBA and palette images of any legal bit depth,
// plain or Adam7-interlaced.  Alpha is dropped and 16-bit samples keep their
// high byte, as a conversion to 8-bit gray / RGB does.  Chunk CRCs and the
// zlib checksum are verified.
Error DecodePng(const uint8_t* data, size_t size, DecodedImage* out);

// JPEG: sequential (SOF0 / SOF1) and progressive (SOF2) Huffman-coded DCT, 8-bit, gray or YCbCr / RGB with any
// mix of 1x and 2x sampling factors, restart intervals, one or several scans.  Pixels equal libjpeg's default
// decode (slow integer IDCT, triangle-filter chroma upsampling).  Arithmetic-coded, 12-bit and CMYK files
// are refused with a message that says so.
Error DecodeJpeg(const uint8_t* data, size_t size, DecodedImage* out);

// Binary PGM (P5) / PPM (P6) with maxval 255.
Error DecodePnm(const uint8_t* data, size_t size, DecodedImage* out);

// Decodes an image file held in memory by its signature (PNG, JPEG, binary PNM).
Error DecodeImage(const uint8_t* data, size_t size, DecodedImage* out);

// Reads a file and decodes it by its signature.
Error ReadImageFile(const std::string& path, DecodedImage* out);

// Converts to `channels` (1 or 3): gray is replicated, RGB is reduced with the
// ITU-R 601 luma weights in 16-bit fixed point, (R*19595 + G*38470 + B*7471 + 0x8000) >> 16.
void ConvertChannels(const DecodedImage& in, int channels, DecodedImage* out);

}  // namespace tca
