// JPEG decoder for the image examples: Huffman-coded DCT, 8-bit - sequential (SOF0 baseline, SOF1 extended) and
// progressive (SOF2: spectral selection and successive approximation, coefficients kept until the last scan) -
// gray or three components, any mix of 1x / 2x sampling factors, interleaved or one scan per component,
// restart intervals, 8- and 16-bit quantisation tables.  Arithmetic-coded, lossless, 12-bit and
// four-component (CMYK) files are reported as unsupported.
//
// The reference's image clients read files with OpenCV's imread (`src/c++/examples/image_client.cc:85-105`),
// which decodes JPEG with libjpeg.  To produce the same pixels this decoder uses libjpeg's arithmetic: the
// 13-bit fixed-point "slow integer" inverse DCT (two passes, 2 extra bits kept between them), the triangle
// ("fancy") chroma upsampling filters for 2:1 horizontal, 2:1 in both directions and 2:1 vertical, sample
// replication for any other ratio, and the 16-bit fixed-point YCbCr -> RGB conversion.
#include <algorithm>
#include <cstring>

#include "tca/image_io.h"

namespace tca {
namespace {

constexpr int kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTable {
  bool present = false;
  uint16_t fast[512];     // 9-bit prefix -> (length << 8) | symbol, 0 when the code is longer
  int32_t maxcode[18];    // largest code of each length (-1: none), [17] = sentinel
  int32_t valptr[17];     // index of the first symbol of each length
  int32_t mincode[17];
  uint8_t symbols[256];

  Error Build(const uint8_t counts[16], const uint8_t* syms, int total) {
    std::memcpy(symbols, syms, static_cast<size_t>(total));
    std::memset(fast, 0, sizeof(fast));
    int code = 0, k = 0;
    for (int len = 1; len <= 16; ++len) {
      valptr[len] = k;
      mincode[len] = code;
      for (int i = 0; i < counts[len - 1]; ++i, ++k, ++code) {
        if (len <= 9) {
          const int shift = 9 - len;
          for (int fill = 0; fill < (1 << shift); ++fill)
            fast[(code << shift) | fill] = static_cast<uint16_t>((len << 8) | symbols[k]);
        }
      }
      if (code > (1 << len)) return Error("JPEG: a Huffman table assigns more codes than a length has");
      maxcode[len] = counts[len - 1] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7FFFFFFF;
    present = true;
    return Error::Success();
  }
};

// Entropy-coded bytes: 0xFF 0x00 is a data byte 0xFF; any other marker ends the segment (zero bits are fed
// from there on, the marker stays unread).
struct BitReader {
  const uint8_t* p = nullptr;
  const uint8_t* end = nullptr;
  uint64_t acc = 0;
  int bits = 0;
  bool at_marker = false;

  void Fill() {
    while (bits <= 56) {
      uint64_t byte = 0;
      if (!at_marker && p < end) {
        if (*p == 0xFF) {
          if (p + 1 < end && p[1] == 0x00) {
            byte = 0xFF;
            p += 2;
          } else {
            at_marker = true;
          }
        } else {
          byte = *p++;
        }
      } else {
        at_marker = true;
      }
      acc |= byte << (56 - bits);
      bits += 8;
    }
  }
  uint32_t Peek(int n) {
    if (bits < n) Fill();
    return static_cast<uint32_t>(acc >> (64 - n));
  }
  void Skip(int n) {
    acc <<= n;
    bits -= n;
  }
  int Receive(int n) {
    if (n == 0) return 0;
    const uint32_t v = Peek(n);
    Skip(n);
    return static_cast<int>(v);
  }
  void Reset() {
    acc = 0;
    bits = 0;
    at_marker = false;
  }
};

inline int Extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

inline int DecodeSymbol(BitReader* br, const HuffTable& t) {
  const uint32_t look = br->Peek(16);
  const uint16_t f = t.fast[look >> 7];
  if (f != 0) {
    br->Skip(f >> 8);
    return f & 0xFF;
  }
  for (int len = 10; len <= 16; ++len) {
    const int32_t code = static_cast<int32_t>(look >> (16 - len));
    if (code <= t.maxcode[len]) {
      br->Skip(len);
      const int32_t index = t.valptr[len] + code - t.mincode[len];
      return index >= 0 && index < 256 ? t.symbols[index] : -1;
    }
  }
  return -1;
}

struct Component {
  int id = 0, h = 1, v = 1, tq = 0;
  int td = 0, ta = 0;       // tables of the current scan
  int dw = 0, dh = 0;       // samples that belong to the image
  int pw = 0, ph = 0;       // plane size: whole MCUs
  int pred = 0;
  bool scanned = false;
  std::vector<uint8_t> plane;
  std::vector<int16_t> coefs;  // progressive files: every block's 64 coefficients (natural order) until the last scan
};

// libjpeg's jpeg_idct_islow: dequantised coefficients -> 8x8 samples (level shift and clamp included).
constexpr int kConstBits = 13, kPass1Bits = 2;
constexpr int32_t kF0_298 = 2446, kF0_390 = 3196, kF0_541 = 4433, kF0_765 = 6270, kF0_899 = 7373, kF1_175 = 9633,
                  kF1_501 = 12299, kF1_847 = 15137, kF1_961 = 16069, kF2_053 = 16819, kF2_562 = 20995, kF3_072 = 25172;

inline int32_t Descale(int32_t x, int n) { return (x + (1 << (n - 1))) >> n; }

// libjpeg's range-limit table behind a 10-bit mask: 0..127 -> +128, 128..511 -> 255, 512..895 -> 0, 896.. -> -896
inline uint8_t RangeLimit(int32_t x) {
  const int idx = x & 1023;
  if (idx < 128) return static_cast<uint8_t>(idx + 128);
  if (idx < 512) return 255;
  if (idx < 896) return 0;
  return static_cast<uint8_t>(idx - 896);
}

void IdctIslow(const int16_t* coef, const uint16_t* quant, uint8_t* dst, int stride) {
  int32_t ws[64];
  for (int c = 0; c < 8; ++c) {
    const int16_t* in = coef + c;
    const uint16_t* q = quant + c;
    int32_t* w = ws + c;
    if (in[8] == 0 && in[16] == 0 && in[24] == 0 && in[32] == 0 && in[40] == 0 && in[48] == 0 && in[56] == 0) {
      const int32_t dc = static_cast<int32_t>(in[0]) * q[0] * (1 << kPass1Bits);
      for (int r = 0; r < 8; ++r) w[8 * r] = dc;
      continue;
    }
    int32_t z2 = in[16] * q[16], z3 = in[48] * q[48];
    int32_t z1 = (z2 + z3) * kF0_541;
    int32_t tmp2 = z1 + z3 * (-kF1_847);
    int32_t tmp3 = z1 + z2 * kF0_765;
    z2 = in[0] * q[0];
    z3 = in[32] * q[32];
    int32_t tmp0 = (z2 + z3) * (1 << kConstBits);
    int32_t tmp1 = (z2 - z3) * (1 << kConstBits);
    const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[56] * q[56];
    tmp1 = in[40] * q[40];
    tmp2 = in[24] * q[24];
    tmp3 = in[8] * q[8];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    int32_t z4 = tmp1 + tmp3;
    const int32_t z5 = (z3 + z4) * kF1_175;
    tmp0 *= kF0_298;
    tmp1 *= kF2_053;
    tmp2 *= kF3_072;
    tmp3 *= kF1_501;
    z1 *= -kF0_899;
    z2 *= -kF2_562;
    z3 *= -kF1_961;
    z4 *= -kF0_390;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    const int n = kConstBits - kPass1Bits;
    w[0] = Descale(tmp10 + tmp3, n);
    w[56] = Descale(tmp10 - tmp3, n);
    w[8] = Descale(tmp11 + tmp2, n);
    w[48] = Descale(tmp11 - tmp2, n);
    w[16] = Descale(tmp12 + tmp1, n);
    w[40] = Descale(tmp12 - tmp1, n);
    w[24] = Descale(tmp13 + tmp0, n);
    w[32] = Descale(tmp13 - tmp0, n);
  }
  for (int r = 0; r < 8; ++r) {
    const int32_t* w = ws + 8 * r;
    uint8_t* out = dst + static_cast<size_t>(r) * stride;
    int32_t z2 = w[2], z3 = w[6];
    int32_t z1 = (z2 + z3) * kF0_541;
    int32_t tmp2 = z1 + z3 * (-kF1_847);
    int32_t tmp3 = z1 + z2 * kF0_765;
    int32_t tmp0 = (w[0] + w[4]) * (1 << kConstBits);
    int32_t tmp1 = (w[0] - w[4]) * (1 << kConstBits);
    const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = w[7];
    tmp1 = w[5];
    tmp2 = w[3];
    tmp3 = w[1];
    z1 = tmp0 + tmp3;
    z2 = tmp1 + tmp2;
    z3 = tmp0 + tmp2;
    int32_t z4 = tmp1 + tmp3;
    const int32_t z5 = (z3 + z4) * kF1_175;
    tmp0 *= kF0_298;
    tmp1 *= kF2_053;
    tmp2 *= kF3_072;
    tmp3 *= kF1_501;
    z1 *= -kF0_899;
    z2 *= -kF2_562;
    z3 *= -kF1_961;
    z4 *= -kF0_390;
    z3 += z5;
    z4 += z5;
    tmp0 += z1 + z3;
    tmp1 += z2 + z4;
    tmp2 += z2 + z3;
    tmp3 += z1 + z4;
    const int n = kConstBits + kPass1Bits + 3;
    out[0] = RangeLimit(Descale(tmp10 + tmp3, n));
    out[7] = RangeLimit(Descale(tmp10 - tmp3, n));
    out[1] = RangeLimit(Descale(tmp11 + tmp2, n));
    out[6] = RangeLimit(Descale(tmp11 - tmp2, n));
    out[2] = RangeLimit(Descale(tmp12 + tmp1, n));
    out[5] = RangeLimit(Descale(tmp12 - tmp1, n));
    out[3] = RangeLimit(Descale(tmp13 + tmp0, n));
    out[4] = RangeLimit(Descale(tmp13 - tmp0, n));
  }
}

struct Decoder {
  const uint8_t* data;
  size_t size;
  size_t pos = 2;
  int width = 0, height = 0, hmax = 1, vmax = 1;
  int restart_interval = 0;
  bool have_frame = false, adobe = false, progressive = false;
  int ss = 0, se = 63, ah = 0, al = 0;  // spectral selection / successive approximation of the current scan
  int eobrun = 0;                       // progressive AC scans: blocks left whose band is all zero
  int adobe_transform = -1;
  std::vector<Component> comps;
  uint16_t quant[4][64];  // natural order
  bool have_quant[4] = {false, false, false, false};
  HuffTable dc[4], ac[4];

  Error Dqt(const uint8_t* p, size_t n) {
    while (n > 0) {
      const int pq = p[0] >> 4, tq = p[0] & 15;
      const size_t need = 1 + (pq ? 128 : 64);
      if (tq > 3 || pq > 1 || n < need) return Error("JPEG: bad quantisation table segment");
      for (int i = 0; i < 64; ++i)
        quant[tq][kZigzag[i]] = pq ? static_cast<uint16_t>((p[1 + 2 * i] << 8) | p[2 + 2 * i]) : p[1 + i];
      have_quant[tq] = true;
      p += need;
      n -= need;
    }
    return Error::Success();
  }

  Error Dht(const uint8_t* p, size_t n) {
    while (n > 0) {
      if (n < 17) return Error("JPEG: bad Huffman table segment");
      const int tc = p[0] >> 4, th = p[0] & 15;
      int total = 0;
      for (int i = 0; i < 16; ++i) total += p[1 + i];
      if (tc > 1 || th > 3 || total > 256 || n < static_cast<size_t>(17 + total)) return Error("JPEG: bad Huffman table segment");
      const Error e = (tc == 0 ? dc[th] : ac[th]).Build(p + 1, p + 17, total);
      if (!e.IsOk()) return e;
      p += 17 + total;
      n -= static_cast<size_t>(17 + total);
    }
    return Error::Success();
  }

  Error Sof(const uint8_t* p, size_t n, bool is_progressive) {
    if (have_frame) return Error("JPEG: more than one frame");
    progressive = is_progressive;
    if (n < 6) return Error("JPEG: bad frame header");
    if (p[0] != 8) return Error("JPEG: only 8-bit samples are supported");
    height = (p[1] << 8) | p[2];
    width = (p[3] << 8) | p[4];
    const int nc = p[5];
    if (width == 0 || height == 0) return Error("JPEG: empty image (or a height defined by a later segment)");
    if (nc != 1 && nc != 3) return Error("JPEG: " + std::to_string(nc) + "-component images are not supported");
    if (n < static_cast<size_t>(6 + 3 * nc)) return Error("JPEG: bad frame header");
    if (static_cast<uint64_t>(width) * height > (1ull << 26)) return Error("JPEG: image too large (more than 64 M pixels)");
    comps.resize(static_cast<size_t>(nc));
    for (int i = 0; i < nc; ++i) {
      Component& c = comps[static_cast<size_t>(i)];
      c.id = p[6 + 3 * i];
      c.h = p[7 + 3 * i] >> 4;
      c.v = p[7 + 3 * i] & 15;
      c.tq = p[8 + 3 * i];
      if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return Error("JPEG: bad component description");
      if (nc == 1) c.h = c.v = 1;  // a single component is never subsampled, whatever the header says
      hmax = std::max(hmax, c.h);
      vmax = std::max(vmax, c.v);
    }
    const int mcux = (width + 8 * hmax - 1) / (8 * hmax), mcuy = (height + 8 * vmax - 1) / (8 * vmax);
    for (auto& c : comps) {
      if (hmax % c.h != 0 || vmax % c.v != 0) return Error("JPEG: fractional sampling ratios are not supported");
      c.dw = (width * c.h + hmax - 1) / hmax;
      c.dh = (height * c.v + vmax - 1) / vmax;
      c.pw = mcux * c.h * 8;
      c.ph = mcuy * c.v * 8;
      c.plane.assign(static_cast<size_t>(c.pw) * c.ph, 128);
      if (progressive) c.coefs.assign(static_cast<size_t>(c.pw) * c.ph, 0);
    }
    have_frame = true;
    return Error::Success();
  }

  Error DecodeBlock(BitReader* br, Component* c, int bx, int by) {
    int16_t coef[64] = {0};
    const HuffTable& d = dc[c->td];
    const HuffTable& a = ac[c->ta];
    int s = DecodeSymbol(br, d);
    if (s < 0 || s > 11) return Error("JPEG: bad DC code");
    c->pred += s ? Extend(br->Receive(s), s) : 0;
    coef[0] = static_cast<int16_t>(c->pred);
    for (int k = 1; k < 64;) {
      const int rs = DecodeSymbol(br, a);
      if (rs < 0) return Error("JPEG: bad AC code");
      const int r = rs >> 4;
      s = rs & 15;
      if (s == 0) {
        if (r != 15) break;  // end of block
        k += 16;
        continue;
      }
      k += r;
      if (k > 63) return Error("JPEG: coefficient index out of range");
      coef[kZigzag[k]] = static_cast<int16_t>(Extend(br->Receive(s), s));
      ++k;
    }
    IdctIslow(coef, quant[c->tq], &c->plane[static_cast<size_t>(by) * 8 * c->pw + static_cast<size_t>(bx) * 8], c->pw);
    return Error::Success();
  }

  // Progressive scans (libjpeg jdphuff.c): each adds bits to the stored coefficients of one band.
  Error ProgressiveBlock(BitReader* br, Component* c, int bx, int by) {
    int16_t* block = &c->coefs[(static_cast<size_t>(by) * (c->pw / 8) + bx) * 64];
    if (ss == 0) {
      if (ah == 0) {  // first DC scan: the difference, scaled by the point transform
        const int s = DecodeSymbol(br, dc[c->td]);
        if (s < 0 || s > 11) return Error("JPEG: bad DC code");
        c->pred += s ? Extend(br->Receive(s), s) : 0;
        block[0] = static_cast<int16_t>(c->pred * (1 << al));
      } else if (br->Receive(1)) {  // DC refinement: one more bit
        block[0] = static_cast<int16_t>(block[0] | (1 << al));
      }
      return Error::Success();
    }
    const HuffTable& a = ac[c->ta];
    if (ah == 0) {  // first AC scan of the band
      if (eobrun > 0) {
        --eobrun;
        return Error::Success();
      }
      for (int k = ss; k <= se; ++k) {
        const int rs = DecodeSymbol(br, a);
        if (rs < 0) return Error("JPEG: bad AC code");
        const int r = rs >> 4, sz = rs & 15;
        if (sz) {
          k += r;
          if (k > se) return Error("JPEG: coefficient index out of range");
          block[kZigzag[k]] = static_cast<int16_t>(Extend(br->Receive(sz), sz) * (1 << al));
        } else if (r == 15) {
          k += 15;
        } else {  // end of band for this block and eobrun - 1 more
          eobrun = (1 << r) + (r ? br->Receive(r) : 0) - 1;
          break;
        }
      }
      return Error::Success();
    }
    // AC refinement: correction bits for the coefficients that are nonzero already, new ones of magnitude 1 << al
    const int p1 = 1 << al, m1 = -(1 << al);
    auto correct = [&](int16_t* coef) {
      if (br->Receive(1) && (*coef & p1) == 0) *coef = static_cast<int16_t>(*coef >= 0 ? *coef + p1 : *coef + m1);
    };
    int k = ss;
    if (eobrun == 0) {
      for (; k <= se; ++k) {
        const int rs = DecodeSymbol(br, a);
        if (rs < 0) return Error("JPEG: bad AC code");
        int r = rs >> 4, value = 0;
        if (rs & 15) {
          if ((rs & 15) != 1) return Error("JPEG: bad refinement code");
          value = br->Receive(1) ? p1 : m1;
        } else if (r != 15) {
          eobrun = (1 << r) + (r ? br->Receive(r) : 0);
          break;  // the rest of the band below
        }
        do {  // over nonzero coefficients (a correction bit each) and r zero ones
          int16_t* coef = block + kZigzag[k];
          if (*coef != 0) {
            correct(coef);
          } else if (--r < 0) {
            break;
          }
          ++k;
        } while (k <= se);
        if (value) {
          if (k > se) return Error("JPEG: coefficient index out of range");
          block[kZigzag[k]] = static_cast<int16_t>(value);
        }
      }
    }
    if (eobrun > 0) {
      for (; k <= se; ++k) {
        int16_t* coef = block + kZigzag[k];
        if (*coef != 0) correct(coef);
      }
      --eobrun;
    }
    return Error::Success();
  }

  // After the scan header at `pos`: entropy-coded data up to the next marker that is not a restart marker.
  Error Scan(const uint8_t* p, size_t n) {
    if (!have_frame) return Error("JPEG: scan before the frame header");
    if (n < 1) return Error("JPEG: bad scan header");
    const int ns = p[0];
    if (ns < 1 || ns > static_cast<int>(comps.size()) || n < static_cast<size_t>(4 + 2 * ns)) return Error("JPEG: bad scan header");
    std::vector<Component*> in_scan;
    for (int i = 0; i < ns; ++i) {
      Component* found = nullptr;
      for (auto& c : comps)
        if (c.id == p[1 + 2 * i]) found = &c;
      if (found == nullptr) return Error("JPEG: scan names an unknown component");
      found->td = p[2 + 2 * i] >> 4;
      found->ta = p[2 + 2 * i] & 15;
      if (found->td > 3 || found->ta > 3) return Error("JPEG: bad table selector in a scan header");
      if (!have_quant[found->tq]) return Error("JPEG: component uses a quantisation table that was not defined");
      found->pred = 0;
      found->scanned = true;
      in_scan.push_back(found);
    }
    ss = p[1 + 2 * ns];
    se = p[2 + 2 * ns];
    ah = p[3 + 2 * ns] >> 4;
    al = p[3 + 2 * ns] & 15;
    if (!progressive) {
      if (ss != 0 || se != 63 || ah != 0 || al != 0)
        return Error("JPEG: spectral selection / successive approximation belong to progressive files");
    } else if (ss > se || se > 63 || al > 13 || ah > 13 || (ss == 0 && se != 0) || (ss != 0 && ns != 1)) {
      return Error("JPEG: bad progressive scan parameters");
    }
    for (const Component* c : in_scan) {  // the tables this kind of scan reads
      const bool need_dc = !progressive || (ss == 0 && ah == 0), need_ac = !progressive || ss != 0;
      if ((need_dc && !dc[c->td].present) || (need_ac && !ac[c->ta].present))
        return Error("JPEG: scan uses a Huffman table that was not defined");
    }
    eobrun = 0;
    BitReader br;
    br.p = data + pos;
    br.end = data + size;
    int mcus_x, mcus_y;
    if (ns == 1) {
      mcus_x = (in_scan[0]->dw + 7) / 8;
      mcus_y = (in_scan[0]->dh + 7) / 8;
    } else {
      mcus_x = (width + 8 * hmax - 1) / (8 * hmax);
      mcus_y = (height + 8 * vmax - 1) / (8 * vmax);
    }
    const long total = static_cast<long>(mcus_x) * mcus_y;
    int until_restart = restart_interval;
    int next_rst = 0;
    for (long m = 0; m < total; ++m) {
      if (restart_interval > 0 && until_restart == 0) {
        // byte-align, then the restart marker; a damaged stream is searched forward for it
        br.Reset();
        const uint8_t* q = br.p;
        while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) {
          if (q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF) return Error("JPEG: restart marker missing");
          ++q;
        }
        if (q + 1 >= br.end) return Error("JPEG: data ends inside a scan");
        next_rst = (next_rst + 1) & 7;
        br.p = q + 2;
        for (Component* c : in_scan) c->pred = 0;
        eobrun = 0;
        until_restart = restart_interval;
      }
      const int mx = static_cast<int>(m % mcus_x), my = static_cast<int>(m / mcus_x);
      if (ns == 1) {
        const Error e = progressive ? ProgressiveBlock(&br, in_scan[0], mx, my) : DecodeBlock(&br, in_scan[0], mx, my);
        if (!e.IsOk()) return e;
      } else {
        for (Component* c : in_scan)
          for (int vy = 0; vy < c->v; ++vy)
            for (int hx = 0; hx < c->h; ++hx) {
              const int bx = mx * c->h + hx, by = my * c->v + vy;
              const Error e = progressive ? ProgressiveBlock(&br, c, bx, by) : DecodeBlock(&br, c, bx, by);
              if (!e.IsOk()) return e;
            }
      }
      --until_restart;
    }
    // the reader may have buffered bytes past the end of the segment: find the next marker from where the
    // entropy-coded data started being consumed
    const uint8_t* q = br.p;
    // (bytes already in the accumulator are data bytes; the marker is at or after p)
    while (q + 1 < br.end && !(q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF && !(q[1] >= 0xD0 && q[1] <= 0xD7))) ++q;
    pos = static_cast<size_t>(q - data);
    return Error::Success();
  }

  // ---- chroma upsampling (libjpeg jdsample.c) ----
  static void H2V1Fancy(const Component& c, std::vector<uint8_t>* out, int* ow) {
    *ow = 2 * c.dw;
    out->resize(static_cast<size_t>(*ow) * c.dh);
    for (int r = 0; r < c.dh; ++r) {
      const uint8_t* in = &c.plane[static_cast<size_t>(r) * c.pw];
      uint8_t* o = &(*out)[static_cast<size_t>(r) * *ow];
      int v = in[0];
      *o++ = static_cast<uint8_t>(v);
      *o++ = static_cast<uint8_t>((v * 3 + in[1] + 2) >> 2);
      for (int x = 1; x < c.dw - 1; ++x) {
        v = in[x] * 3;
        *o++ = static_cast<uint8_t>((v + in[x - 1] + 1) >> 2);
        *o++ = static_cast<uint8_t>((v + in[x + 1] + 2) >> 2);
      }
      v = in[c.dw - 1];
      *o++ = static_cast<uint8_t>((v * 3 + in[c.dw - 2] + 1) >> 2);
      *o++ = static_cast<uint8_t>(v);
    }
  }

  static void H2V2Fancy(const Component& c, std::vector<uint8_t>* out, int* ow) {
    *ow = 2 * c.dw;
    out->resize(static_cast<size_t>(*ow) * c.dh * 2);
    for (int r = 0; r < c.dh; ++r)
      for (int v = 0; v < 2; ++v) {
        const int other = v == 0 ? std::max(r - 1, 0) : std::min(r + 1, c.dh - 1);
        const uint8_t* in0 = &c.plane[static_cast<size_t>(r) * c.pw];
        const uint8_t* in1 = &c.plane[static_cast<size_t>(other) * c.pw];
        uint8_t* o = &(*out)[static_cast<size_t>(2 * r + v) * *ow];
        int thiscol = in0[0] * 3 + in1[0], nextcol = in0[1] * 3 + in1[1], lastcol;
        *o++ = static_cast<uint8_t>((thiscol * 4 + 8) >> 4);
        *o++ = static_cast<uint8_t>((thiscol * 3 + nextcol + 7) >> 4);
        lastcol = thiscol;
        thiscol = nextcol;
        for (int x = 2; x < c.dw; ++x) {
          nextcol = in0[x] * 3 + in1[x];
          *o++ = static_cast<uint8_t>((thiscol * 3 + lastcol + 8) >> 4);
          *o++ = static_cast<uint8_t>((thiscol * 3 + nextcol + 7) >> 4);
          lastcol = thiscol;
          thiscol = nextcol;
        }
        *o++ = static_cast<uint8_t>((thiscol * 3 + lastcol + 8) >> 4);
        *o++ = static_cast<uint8_t>((thiscol * 4 + 7) >> 4);
      }
  }

  static void H1V2Fancy(const Component& c, std::vector<uint8_t>* out, int* ow) {
    *ow = c.dw;
    out->resize(static_cast<size_t>(*ow) * c.dh * 2);
    for (int r = 0; r < c.dh; ++r)
      for (int v = 0; v < 2; ++v) {
        const int other = v == 0 ? std::max(r - 1, 0) : std::min(r + 1, c.dh - 1);
        const int bias = v == 0 ? 1 : 2;
        const uint8_t* in0 = &c.plane[static_cast<size_t>(r) * c.pw];
        const uint8_t* in1 = &c.plane[static_cast<size_t>(other) * c.pw];
        uint8_t* o = &(*out)[static_cast<size_t>(2 * r + v) * *ow];
        for (int x = 0; x < c.dw; ++x) o[x] = static_cast<uint8_t>((in0[x] * 3 + in1[x] + bias) >> 2);
      }
  }

  static void Replicate(const Component& c, int hr, int vr, std::vector<uint8_t>* out, int* ow) {
    *ow = c.dw * hr;
    out->resize(static_cast<size_t>(*ow) * c.dh * vr);
    for (int r = 0; r < c.dh * vr; ++r) {
      const uint8_t* in = &c.plane[static_cast<size_t>(r / vr) * c.pw];
      uint8_t* o = &(*out)[static_cast<size_t>(r) * *ow];
      for (int x = 0; x < c.dw; ++x)
        for (int k = 0; k < hr; ++k) *o++ = in[x];
    }
  }

  Error Finish(DecodedImage* out) {
    for (const auto& c : comps)
      if (!c.scanned) return Error("JPEG: a component has no scan");
    if (progressive) {  // every scan is in: coefficients -> samples
      for (auto& c : comps) {
        const int bw = c.pw / 8, bh = c.ph / 8;
        for (int by = 0; by < bh; ++by)
          for (int bx = 0; bx < bw; ++bx)
            IdctIslow(&c.coefs[(static_cast<size_t>(by) * bw + bx) * 64], quant[c.tq],
                      &c.plane[static_cast<size_t>(by) * 8 * c.pw + static_cast<size_t>(bx) * 8], c.pw);
        std::vector<int16_t>().swap(c.coefs);
      }
    }
    out->width = width;
    out->height = height;
    out->channels = static_cast<int>(comps.size());
    out->pixels.resize(static_cast<size_t>(width) * height * comps.size());
    if (comps.size() == 1) {
      for (int y = 0; y < height; ++y)
        std::memcpy(&out->pixels[static_cast<size_t>(y) * width], &comps[0].plane[static_cast<size_t>(y) * comps[0].pw],
                    static_cast<size_t>(width));
      return Error::Success();
    }
    const uint8_t* planes[3];
    int strides[3];
    std::vector<uint8_t> up[3];
    for (int i = 0; i < 3; ++i) {
      const Component& c = comps[static_cast<size_t>(i)];
      const int hr = hmax / c.h, vr = vmax / c.v;
      if (hr == 1 && vr == 1) {
        planes[i] = c.plane.data();
        strides[i] = c.pw;
        continue;
      }
      int ow = 0;
      if (hr == 2 && vr == 1 && c.dw > 2) H2V1Fancy(c, &up[i], &ow);
      else if (hr == 2 && vr == 2 && c.dw > 2) H2V2Fancy(c, &up[i], &ow);
      else if (hr == 1 && vr == 2) H1V2Fancy(c, &up[i], &ow);
      else Replicate(c, hr, vr, &up[i], &ow);
      planes[i] = up[i].data();
      strides[i] = ow;
    }
    // Adobe transform 0, or components named R, G, B without an Adobe segment: the samples are RGB already
    const bool rgb = (adobe && adobe_transform == 0) ||
                     (!adobe && comps[0].id == 'R' && comps[1].id == 'G' && comps[2].id == 'B');
    for (int y = 0; y < height; ++y) {
      const uint8_t* p0 = planes[0] + static_cast<size_t>(y) * strides[0];
      const uint8_t* p1 = planes[1] + static_cast<size_t>(y) * strides[1];
      const uint8_t* p2 = planes[2] + static_cast<size_t>(y) * strides[2];
      uint8_t* o = &out->pixels[static_cast<size_t>(y) * width * 3];
      if (rgb) {
        for (int x = 0; x < width; ++x, o += 3) {
          o[0] = p0[x];
          o[1] = p1[x];
          o[2] = p2[x];
        }
        continue;
      }
      for (int x = 0; x < width; ++x, o += 3) {
        const int yy = p0[x], cb = p1[x] - 128, cr = p2[x] - 128;
        const int r = yy + ((91881 * cr + 32768) >> 16);
        const int g = yy + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
        const int b = yy + ((116130 * cb + 32768) >> 16);
        o[0] = static_cast<uint8_t>(std::min(std::max(r, 0), 255));
        o[1] = static_cast<uint8_t>(std::min(std::max(g, 0), 255));
        o[2] = static_cast<uint8_t>(std::min(std::max(b, 0), 255));
      }
    }
    return Error::Success();
  }

  Error Run(DecodedImage* out) {
    for (;;) {
      // next marker (fill bytes 0xFF may precede it)
      while (pos < size && data[pos] != 0xFF) ++pos;
      while (pos < size && data[pos] == 0xFF) ++pos;
      if (pos >= size) return Error("JPEG: data ends before the end-of-image marker");
      const uint8_t marker = data[pos++];
      if (marker == 0xD9) break;
      if (marker == 0x01 || (marker >= 0xD0 && marker <= 0xD7) || marker == 0x00) continue;
      if (pos + 2 > size) return Error("JPEG: truncated segment");
      const size_t len = (static_cast<size_t>(data[pos]) << 8) | data[pos + 1];
      if (len < 2 || pos + len > size) return Error("JPEG: truncated segment");
      const uint8_t* p = data + pos + 2;
      const size_t n = len - 2;
      pos += len;
      Error e = Error::Success();
      switch (marker) {
        case 0xC0:
        case 0xC1:
          e = Sof(p, n, false);
          break;
        case 0xC2:
          e = Sof(p, n, true);
          break;
        case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
          return Error("JPEG: lossless, hierarchical and arithmetic-coded files are not supported");
        case 0xC4:
          e = Dht(p, n);
          break;
        case 0xDB:
          e = Dqt(p, n);
          break;
        case 0xDD:
          if (n < 2) return Error("JPEG: bad restart interval segment");
          restart_interval = (p[0] << 8) | p[1];
          break;
        case 0xEE:
          if (n >= 12 && std::memcmp(p, "Adobe", 5) == 0) {
            adobe = true;
            adobe_transform = p[11];
          }
          break;
        case 0xDA:
          e = Scan(p, n);
          break;
        default:
          break;  // APPn, COM and anything else with a length: skipped
      }
      if (!e.IsOk()) return e;
    }
    if (!have_frame) return Error("JPEG: no frame header");
    return Finish(out);
  }
};

}  // namespace

Error DecodeJpeg(const uint8_t* data, size_t size, DecodedImage* out) {
  if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) return Error("JPEG: missing start-of-image marker");
  Decoder d;
  d.data = data;
  d.size = size;
  return d.Run(out);
}

}  // namespace tca
