// Image classification client (HTTP or gRPC, -i): reads the model's metadata +
// config to learn its input layout / type, preprocesses images, sends them in
// batches (sync or -a async) and prints the top `-c` classes returned by the
// classification extension.
//
// Two preprocessing paths (--preprocess):
//   host  2-tap bilinear resize on the CPU as cv::resize does, NONE / VGG /
//         INCEPTION scaling, NHWC or NCHW; the tensor travels in the request
//   hip   the decoded u8 images go to the GPU and ONE fused kernel (resize +
//         scale + layout + cast, same 2-tap windows in 22-bit fixed point)
//         writes the batch into a slot of a HIP shared-memory region the
//         server reads in place; with -a up to --ring-depth requests are in
//         flight while the next batch is preprocessed.  The kernels come from
//         libtca_hip.so, loaded at run time (--hip-lib or TCA_HIP_LIB).
//
// Images are PNG or binary PPM/PGM files (tca/image_io.h - OpenCV is not part of
// this toolchain), sequential JPEG files, or `--synthetic N` generated ones; `--decode-only` prints
// geometry and a pixel hash per file and exits (`--dump DIR` also writes the decoded pixels there as PNM).
//
// Behavioural reference: `src/c++/examples/image_client.cc:85-276,537-702,1030-1052`.
#include <dlfcn.h>

#include <cmath>
#include <condition_variable>
#include <deque>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>

#include "../hip/kernels.h"
#include "example_utils.h"
#include "tca/codec.h"
#include "tca/image_io.h"
#include "tca/model_info.h"

namespace {

enum class Scale { NONE, VGG, INCEPTION };

struct Image {
  int h = 0, w = 0, c = 0;
  std::vector<uint8_t> px;  // HWC
  std::string name;
};

struct ModelInfo {
  std::string input_name, output_name, input_datatype;
  int c = 0, h = 0, w = 0;
  bool nhwc = false;
  int max_batch_size = 0;
};

Image Synthetic(int index) {
  Image img;
  img.h = 240 + 16 * index;
  img.w = 320;
  img.c = 3;
  img.px.resize(static_cast<size_t>(img.h) * img.w * 3);
  uint32_t state = 12345u + 977u * static_cast<uint32_t>(index);
  for (auto& p : img.px) {
    state = state * 1664525u + 1013904223u;
    p = static_cast<uint8_t>(state >> 24);
  }
  img.name = "synthetic_" + std::to_string(index);
  return img;
}

// Channel conversion + bilinear resize to HxWxC float in [0,255].
std::vector<float> ResizeToFloat(const Image& img, int out_h, int out_w, int out_c) {
  std::vector<float> src(static_cast<size_t>(img.h) * img.w * out_c);
  for (int y = 0; y < img.h; ++y)
    for (int x = 0; x < img.w; ++x) {
      const uint8_t* p = &img.px[(static_cast<size_t>(y) * img.w + x) * img.c];
      float* d = &src[(static_cast<size_t>(y) * img.w + x) * out_c];
      if (out_c == 1) d[0] = img.c == 1 ? p[0] : 0.299f * p[0] + 0.587f * p[1] + 0.114f * p[2];
      else for (int c = 0; c < 3; ++c) d[c] = img.c == 1 ? p[0] : p[c];
    }
  std::vector<float> out(static_cast<size_t>(out_h) * out_w * out_c);
  const float sy = static_cast<float>(img.h) / out_h, sx = static_cast<float>(img.w) / out_w;
  for (int y = 0; y < out_h; ++y) {
    float fy = (y + 0.5f) * sy - 0.5f;
    int y0 = static_cast<int>(std::floor(fy));
    fy -= y0;
    const int y1 = std::min(std::max(y0 + 1, 0), img.h - 1);
    y0 = std::min(std::max(y0, 0), img.h - 1);
    for (int x = 0; x < out_w; ++x) {
      float fx = (x + 0.5f) * sx - 0.5f;
      int x0 = static_cast<int>(std::floor(fx));
      fx -= x0;
      const int x1 = std::min(std::max(x0 + 1, 0), img.w - 1);
      x0 = std::min(std::max(x0, 0), img.w - 1);
      for (int c = 0; c < out_c; ++c) {
        auto at = [&](int yy, int xx) { return src[(static_cast<size_t>(yy) * img.w + xx) * out_c + c]; };
        const float top = at(y0, x0) * (1 - fx) + at(y0, x1) * fx;
        const float bot = at(y1, x0) * (1 - fx) + at(y1, x1) * fx;
        out[(static_cast<size_t>(y) * out_w + x) * out_c + c] = top * (1 - fy) + bot * fy;
      }
    }
  }
  return out;
}

template <typename T>
void Emit(float v, std::vector<uint8_t>* out) {
  const T t = static_cast<T>(v);
  const uint8_t* p = reinterpret_cast<const uint8_t*>(&t);
  out->insert(out->end(), p, p + sizeof(T));
}

tca::Error Preprocess(const Image& img, const ModelInfo& m, Scale scale, std::vector<uint8_t>* out) {
  std::vector<float> hwc = ResizeToFloat(img, m.h, m.w, m.c);
  for (size_t i = 0; i < hwc.size(); ++i) {
    const int c = static_cast<int>(i % m.c);
    if (scale == Scale::INCEPTION) hwc[i] = hwc[i] / 127.5f - 1.0f;
    else if (scale == Scale::VGG) hwc[i] -= m.c == 1 ? 128.0f : (c == 0 ? 123.0f : c == 1 ? 117.0f : 104.0f);
  }
  std::vector<float> ordered;
  if (m.nhwc) {
    ordered.swap(hwc);
  } else {
    ordered.resize(hwc.size());
    const size_t plane = static_cast<size_t>(m.h) * m.w;
    for (size_t p = 0; p < plane; ++p)
      for (int c = 0; c < m.c; ++c) ordered[c * plane + p] = hwc[p * m.c + c];
  }
  out->clear();
  for (float v : ordered) {
    const std::string& dt = m.input_datatype;
    if (dt == "FP32") Emit<float>(v, out);
    else if (dt == "FP64") Emit<double>(v, out);
    else if (dt == "FP16") { const uint16_t h = tca::FloatToHalf(v); out->push_back(h & 0xFF); out->push_back(h >> 8); }
    else if (dt == "UINT8") Emit<uint8_t>(v, out);
    else if (dt == "INT8") Emit<int8_t>(v, out);
    else if (dt == "UINT16") Emit<uint16_t>(v, out);
    else if (dt == "INT16") Emit<int16_t>(v, out);
    else if (dt == "INT32") Emit<int32_t>(v, out);
    else return tca::Error("unsupported input datatype " + dt);
  }
  return tca::Error::Success();
}

tca::Error ParseModel(const std::string& metadata, const std::string& config, ModelInfo* m) {
  tca::ModelMetadataInfo md;
  tca::ModelConfigInfo cfg;  // either protocol's document (tca/model_info.h)
  TCA_RETURN_IF_ERROR(tca::ParseModelMetadata(metadata, &md));
  TCA_RETURN_IF_ERROR(tca::ParseModelConfig(config, &cfg));
  if (md.inputs.size() != 1) return tca::Error("expecting 1 input, got " + std::to_string(md.inputs.size()));
  if (md.outputs.size() != 1) return tca::Error("expecting 1 output, got " + std::to_string(md.outputs.size()));
  if (cfg.inputs.size() != 1) return tca::Error("expecting 1 input in the model configuration");
  const tca::TensorInfo& in = md.inputs[0];
  const tca::TensorInfo& out = md.outputs[0];
  if (out.datatype != "FP32") return tca::Error("expecting output datatype to be FP32, model output is " + out.datatype);
  m->max_batch_size = static_cast<int>(cfg.max_batch_size);
  const size_t b = m->max_batch_size > 0 ? 1 : 0;
  // output must be a vector (all other dims 1)
  size_t non_one = 0;
  for (size_t i = b; i < out.shape.size(); ++i) non_one += out.shape[i] > 1;
  if (non_one > 1) return tca::Error("expecting model output to be a vector");
  if (in.shape.size() != 3 + b) return tca::Error("expecting input to have " + std::to_string(3 + b) + " dimensions");
  const std::string& format = cfg.inputs[0].format;
  if (format != "FORMAT_NCHW" && format != "FORMAT_NHWC") return tca::Error("unexpected input format " + format + ", expecting FORMAT_NCHW or FORMAT_NHWC");
  m->nhwc = format == "FORMAT_NHWC";
  if (m->nhwc) { m->h = static_cast<int>(in.shape[b]); m->w = static_cast<int>(in.shape[b + 1]); m->c = static_cast<int>(in.shape[b + 2]); }
  else { m->c = static_cast<int>(in.shape[b]); m->h = static_cast<int>(in.shape[b + 1]); m->w = static_cast<int>(in.shape[b + 2]); }
  if (m->c != 1 && m->c != 3) return tca::Error("expecting 1 or 3 channels, got " + std::to_string(m->c));
  m->input_name = in.name;
  m->output_name = out.name;
  m->input_datatype = in.datatype;
  return tca::Error::Success();
}

void PrintResult(tc::InferResult* result, const ModelInfo& m, const std::vector<std::string>& names, size_t classes) {
  std::vector<std::string> strs;
  FAIL_IF_ERR(result->StringData(m.output_name, &strs), "unable to get classification strings");
  if (strs.size() != names.size() * classes) FAIL("expected " << names.size() * classes << " strings, got " << strs.size());
  for (size_t b = 0; b < names.size(); ++b) {
    std::cout << "Image '" << names[b] << "':" << std::endl;
    for (size_t c = 0; c < classes; ++c) {
      std::istringstream is(strs[b * classes + c]);
      std::string score, idx, label;
      std::getline(is, score, ':');
      std::getline(is, idx, ':');
      std::getline(is, label, ':');
      std::cout << "    " << score << " (" << idx << ") = " << label << std::endl;
    }
  }
}


// ---------------------------------------------------------------------------
// GPU path: libtca_hip.so through dlopen, so the example builds with the host
// compiler and only needs the library (and a GPU) when --preprocess hip is used
// ---------------------------------------------------------------------------
struct HipApi {
  void* lib = nullptr;
  int (*set_device)(int) = nullptr;
  int (*stream_create)(int, void**) = nullptr;
  int (*stream_sync)(void*) = nullptr;
  int (*stream_destroy)(void*) = nullptr;
  int (*host_alloc)(uint64_t, void**) = nullptr;
  int (*host_free)(void*) = nullptr;
  int (*dev_alloc)(int, uint64_t, void**) = nullptr;
  int (*dev_free)(void*) = nullptr;
  int (*copy_async)(void*, const void*, uint64_t, int, void*) = nullptr;
  int (*preprocess)(const void*, const void*, const void*, void*, const tca::PreprocParams*, int, int, void*) = nullptr;
  int (*shm_create)(const char*, uint64_t, int, void**) = nullptr;
  int (*shm_info)(void*, void**, uint8_t*, int*, uint64_t*, int*) = nullptr;
  int (*shm_destroy)(void*) = nullptr;
  const char* (*rt_error)() = nullptr;
  const char* (*shm_error)() = nullptr;

  template <typename F>
  void Bind(F* fn, const char* symbol) {
    *fn = reinterpret_cast<F>(dlsym(lib, symbol));
    if (*fn == nullptr) FAIL("libtca_hip.so has no symbol " << symbol);
  }

  void Load(const std::string& wanted, const char* argv0) {
    std::vector<std::string> candidates;
    if (!wanted.empty()) candidates.push_back(wanted);
    if (const char* env = std::getenv("TCA_HIP_LIB")) candidates.push_back(env);
    const std::string exe = argv0;
    const size_t slash = exe.rfind('/');
    if (slash != std::string::npos) candidates.push_back(exe.substr(0, slash) + "/../../triton_client_amd/_lib/libtca_hip.so");
    candidates.push_back("libtca_hip.so");
    std::string errors;
    for (const auto& path : candidates) {
      lib = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (lib != nullptr) break;
      errors += std::string("\n  ") + dlerror();
    }
    if (lib == nullptr) FAIL("unable to load libtca_hip.so (build it with `python -m triton_client_amd._build hip`):" << errors);
    Bind(&set_device, "tca_hip_set_device");
    Bind(&stream_create, "tca_hip_stream_create");
    Bind(&stream_sync, "tca_hip_stream_sync");
    Bind(&stream_destroy, "tca_hip_stream_destroy");
    Bind(&host_alloc, "tca_hip_host_alloc");
    Bind(&host_free, "tca_hip_host_free");
    Bind(&dev_alloc, "tca_hip_malloc");
    Bind(&dev_free, "tca_hip_free");
    Bind(&copy_async, "tca_hip_memcpy_async");
    Bind(&preprocess, "tca_preprocess_launch");
    Bind(&shm_create, "tca_hipshm_create");
    Bind(&shm_info, "tca_hipshm_info");
    Bind(&shm_destroy, "tca_hipshm_destroy");
    Bind(&rt_error, "tca_hip_rt_error");
    Bind(&shm_error, "tca_hip_last_error");
  }
};

#define HIP_CALL(API, EXPR, WHAT)                                                   \
  do {                                                                              \
    const int rc__ = (EXPR);                                                        \
    if (rc__ != 0) FAIL(WHAT << " failed (" << rc__ << "): " << (API).rt_error()); \
  } while (0)

// 2-tap windows with half-pixel centres (the geometry of ResizeToFloat) in the kernel's table form:
// bounds = (first source index, taps) per output index, coeffs = 22-bit fixed-point weights.
void LinearTable(int in_size, int out_size, std::vector<int32_t>* bounds, std::vector<int32_t>* coeffs) {
  const double scale = static_cast<double>(in_size) / out_size;
  const int32_t one = 1 << 22;
  for (int o = 0; o < out_size; ++o) {
    if (in_size == 1) {
      bounds->insert(bounds->end(), {0, 1});
      coeffs->insert(coeffs->end(), {one, 0});
      continue;
    }
    double f = (o + 0.5) * scale - 0.5;
    int s = static_cast<int>(std::floor(f));
    f -= s;
    if (s < 0) { s = 0; f = 0.0; }
    if (s >= in_size - 1) { s = in_size - 2; f = 1.0; }
    const int32_t w1 = static_cast<int32_t>(f * one + 0.5);
    bounds->insert(bounds->end(), {s, 2});
    coeffs->insert(coeffs->end(), {one - w1, w1});
  }
}

int DtypeCode(const std::string& dt) {
  static const std::map<std::string, int> codes = {{"FP32", tca::kF32}, {"FP16", tca::kF16}, {"BF16", tca::kBF16}, {"INT8", tca::kI8},
                                                   {"UINT8", tca::kU8}, {"INT16", tca::kI16}, {"UINT16", tca::kU16},
                                                   {"INT32", tca::kI32}, {"UINT32", tca::kU32}, {"FP64", tca::kF64}};
  const auto it = codes.find(dt);
  if (it == codes.end()) FAIL("unsupported input datatype " << dt << " for --preprocess hip");
  return it->second;
}

size_t DtypeBytes(const std::string& dt) {
  if (dt == "FP64") return 8;
  if (dt == "FP32" || dt == "INT32" || dt == "UINT32") return 4;
  if (dt == "FP16" || dt == "BF16" || dt == "INT16" || dt == "UINT16") return 2;
  return 1;
}

// Device-side input stage: staging buffers, coefficient tables of every image geometry in the run,
// and one HIP shared-memory region cut into `depth` batch slots.
class HipStage {
 public:
  HipStage(const HipApi& api, int device, const ModelInfo& m, Scale scale, size_t batch, size_t depth,
           const std::vector<Image>& images)
      : api_(api), device_(device), model_(m), batch_(batch), depth_(depth) {
    HIP_CALL(api_, api_.set_device(device), "hipSetDevice");
    HIP_CALL(api_, api_.stream_create(device, &stream_), "hipStreamCreate");
    // tables: one x/y pair per distinct (height, width)
    std::vector<int32_t> bounds, coeffs;
    size_t largest = 0;
    int tile_h = 16;
    for (const auto& img : images) {
      largest = std::max(largest, (img.px.size() + 15) & ~size_t(15));
      const auto key = std::make_pair(img.h, img.w);
      if (geometry_.count(key)) continue;
      Geometry g;
      g.xb_off = static_cast<int>(bounds.size() / 2);
      g.xk_off = static_cast<int>(coeffs.size());
      LinearTable(img.w, m.w, &bounds, &coeffs);
      g.yb_off = static_cast<int>(bounds.size() / 2);
      g.yk_off = static_cast<int>(coeffs.size());
      LinearTable(img.h, m.h, &bounds, &coeffs);
      geometry_[key] = g;
    }
    // tile plan: the tallest tile whose source rows fit the LDS budget for every geometry, then the
    // widest source span of a 64-column tile (sizes the staged rows)
    auto rows_of = [&](const Geometry& g, int th) {
      int worst = 1;
      for (int oy0 = 0; oy0 < m.h; oy0 += th) {
        const int last = std::min(oy0 + th, m.h) - 1;
        worst = std::max(worst, bounds[2 * (g.yb_off + last)] + bounds[2 * (g.yb_off + last) + 1] - bounds[2 * (g.yb_off + oy0)]);
      }
      return worst;
    };
    int max_rows = 1, span_px = 1;
    for (;; tile_h /= 2) {
      max_rows = 1;
      for (const auto& kv : geometry_) max_rows = std::max(max_rows, rows_of(kv.second, tile_h));
      if (static_cast<size_t>(max_rows) * (m.c * 64 + 4) <= 48 * 1024) break;
      if (tile_h == 1) FAIL("down-scaling factor too large for the preprocessing kernel");
    }
    for (const auto& kv : geometry_)
      for (int ox0 = 0; ox0 < m.w; ox0 += 64) {
        const int last = std::min(ox0 + 64, m.w) - 1;
        const int xb = kv.second.xb_off;
        span_px = std::max(span_px, bounds[2 * (xb + last)] + bounds[2 * (xb + last) + 1] - bounds[2 * (xb + ox0)]);
      }
    prm_.out_h = m.h;
    prm_.out_w = m.w;
    prm_.out_c = m.c;
    prm_.nhwc = m.nhwc ? 1 : 0;
    prm_.tile_h = tile_h;
    prm_.max_rows = max_rows;
    const int pitch_words = ((span_px * 3 + 3) / 4) | 1;
    prm_.src_pitch = static_cast<size_t>(max_rows) * (m.c * 64 + 4 + pitch_words * 4) <= 64 * 1024 ? pitch_words * 4 : 0;
    prm_.pad_ = 0;
    prm_.image_stride = static_cast<long long>(m.c) * m.h * m.w;
    for (int c = 0; c < 4; ++c) {
      prm_.scale[c] = scale == Scale::INCEPTION ? 1.0f / 127.5f : 1.0f;
      prm_.bias[c] = scale == Scale::INCEPTION ? -1.0f
                     : scale == Scale::VGG     ? (m.c == 1 ? -128.0f : c == 0 ? -123.0f : c == 1 ? -117.0f : c == 2 ? -104.0f : 0.0f)
                                               : 0.0f;
    }
    dtype_ = DtypeCode(m.input_datatype);
    slot_bytes_ = batch * static_cast<size_t>(m.c) * m.h * m.w * DtypeBytes(m.input_datatype);
    stage_bytes_ = largest * batch;
    HIP_CALL(api_, api_.host_alloc(stage_bytes_, &pinned_), "hipHostMalloc");
    HIP_CALL(api_, api_.dev_alloc(device, stage_bytes_, &dev_src_), "hipMalloc");
    HIP_CALL(api_, api_.dev_alloc(device, sizeof(tca::PreprocImage) * batch, &dev_desc_), "hipMalloc");
    HIP_CALL(api_, api_.dev_alloc(device, bounds.size() * 4, &dev_bounds_), "hipMalloc");
    HIP_CALL(api_, api_.dev_alloc(device, coeffs.size() * 4, &dev_coeffs_), "hipMalloc");
    HIP_CALL(api_, api_.copy_async(dev_bounds_, bounds.data(), bounds.size() * 4, 1, stream_), "table upload");
    HIP_CALL(api_, api_.copy_async(dev_coeffs_, coeffs.data(), coeffs.size() * 4, 1, stream_), "table upload");
    HIP_CALL(api_, api_.stream_sync(stream_), "hipStreamSynchronize");
    if (api_.shm_create(kRegion, slot_bytes_ * depth, device, &region_) != 0) FAIL("creating the input region: " << api_.shm_error());
    int dev = 0, is_host = 0;
    uint64_t size = 0;
    if (api_.shm_info(region_, &region_ptr_, raw_handle_, &dev, &size, &is_host) != 0) FAIL("region info: " << api_.shm_error());
  }

  ~HipStage() {
    if (region_) api_.shm_destroy(region_);
    for (void* p : {dev_src_, dev_desc_, dev_bounds_, dev_coeffs_})
      if (p) api_.dev_free(p);
    if (pinned_) api_.host_free(pinned_);
    if (stream_) api_.stream_destroy(stream_);
  }

  static constexpr const char* kRegion = "image_client_input";
  const uint8_t* raw_handle() const { return raw_handle_; }
  size_t region_bytes() const { return slot_bytes_ * depth_; }
  size_t slot_bytes() const { return slot_bytes_; }
  size_t depth() const { return depth_; }

  // Preprocesses `batch` into slot `slot` and waits for the kernel: the server reads from another process.
  void Fill(const std::vector<const Image*>& batch, size_t slot) {
    std::vector<tca::PreprocImage> desc(batch.size());
    size_t cursor = 0;
    for (size_t i = 0; i < batch.size(); ++i) {
      const Image& img = *batch[i];
      const Geometry& g = geometry_.at(std::make_pair(img.h, img.w));
      std::memcpy(static_cast<uint8_t*>(pinned_) + cursor, img.px.data(), img.px.size());
      tca::PreprocImage& d = desc[i];
      d.src = static_cast<const uint8_t*>(dev_src_) + cursor;
      d.height = img.h;
      d.width = img.w;
      d.stride = img.w * img.c;
      d.channels = img.c;
      d.bgr = 0;
      d.kx = d.ky = 2;
      d.xb_off = g.xb_off;
      d.yb_off = g.yb_off;
      d.xk_off = g.xk_off;
      d.yk_off = g.yk_off;
      d.pad_ = 0;
      cursor += (img.px.size() + 15) & ~size_t(15);
    }
    HIP_CALL(api_, api_.copy_async(dev_src_, pinned_, cursor, 1, stream_), "image upload");
    HIP_CALL(api_, api_.copy_async(dev_desc_, desc.data(), desc.size() * sizeof(tca::PreprocImage), 1, stream_), "descriptor upload");
    const int rc = api_.preprocess(dev_desc_, dev_bounds_, dev_coeffs_, static_cast<uint8_t*>(region_ptr_) + slot * slot_bytes_,
                                   &prm_, static_cast<int>(batch.size()), dtype_, stream_);
    if (rc != 0) FAIL("tca_preprocess_launch failed (" << rc << ")");
    HIP_CALL(api_, api_.stream_sync(stream_), "hipStreamSynchronize");
  }

 private:
  struct Geometry { int xb_off, yb_off, xk_off, yk_off; };
  const HipApi& api_;
  int device_;
  ModelInfo model_;
  size_t batch_, depth_;
  std::map<std::pair<int, int>, Geometry> geometry_;
  tca::PreprocParams prm_{};
  int dtype_ = 0;
  size_t slot_bytes_ = 0, stage_bytes_ = 0;
  void *stream_ = nullptr, *pinned_ = nullptr, *dev_src_ = nullptr, *dev_desc_ = nullptr, *dev_bounds_ = nullptr, *dev_coeffs_ = nullptr;
  void *region_ = nullptr, *region_ptr_ = nullptr;
  uint8_t raw_handle_[64] = {};
};

struct Options {
  std::string model, version, hip_lib, dump_dir;
  size_t batch = 1, classes = 1, ring_depth = 4;
  Scale scale = Scale::NONE;
  bool async = false, hip = false, decode_only = false;
  int synthetic = 0, device = 0;
  std::vector<std::string> files;
};

template <typename Client>
int Run(Client* client, const CommonFlags& flags, const Options& opt, std::vector<Image>& images, const char* argv0) {
  std::string metadata, config;
  FAIL_IF_ERR(client->ModelMetadata(&metadata, opt.model, opt.version, flags.headers), "failed to get model metadata");
  FAIL_IF_ERR(client->ModelConfig(&config, opt.model, opt.version, flags.headers), "failed to get model config");
  ModelInfo info;
  FAIL_IF_ERR(ParseModel(metadata, config, &info), "model is not an image classifier");
  const size_t batch = opt.batch, classes = opt.classes;
  if (info.max_batch_size == 0 && batch != 1) FAIL("this model does not support batching");
  if (info.max_batch_size > 0 && static_cast<int>(batch) > info.max_batch_size)
    FAIL("batch size " << batch << " exceeds the model's maximum " << info.max_batch_size);

  // the model's channel count decides gray / RGB, as the decode step of the reference does
  for (auto& img : images) {
    if (img.c == info.c) continue;
    tca::DecodedImage in, out;
    in.height = img.h; in.width = img.w; in.channels = img.c; in.pixels.swap(img.px);
    tca::ConvertChannels(in, info.c, &out);
    img.c = info.c;
    img.px.swap(out.pixels);
  }

  HipApi api;
  std::unique_ptr<HipStage> stage;
  std::vector<std::vector<uint8_t>> tensors;
  if (opt.hip) {
    api.Load(opt.hip_lib, argv0);
    stage.reset(new HipStage(api, opt.device, info, opt.scale, batch, opt.async ? opt.ring_depth : 1, images));
    FAIL_IF_ERR(client->RegisterCudaSharedMemory(HipStage::kRegion, stage->raw_handle(), opt.device, stage->region_bytes()),
                "registering the input region");
  } else {
    tensors.resize(images.size());
    for (size_t i = 0; i < images.size(); ++i) FAIL_IF_ERR(Preprocess(images[i], info, opt.scale, &tensors[i]), "preprocessing");
  }

  std::vector<int64_t> shape;
  if (info.max_batch_size > 0) shape.push_back(static_cast<int64_t>(batch));
  if (info.nhwc) { shape.push_back(info.h); shape.push_back(info.w); shape.push_back(info.c); }
  else { shape.push_back(info.c); shape.push_back(info.h); shape.push_back(info.w); }
  // host path: one InferInput reused for every batch (Reset + AppendRaw per image);
  // hip path: one InferInput per ring slot, pointing at the slot for good
  const size_t slots = stage ? stage->depth() : 1;
  std::vector<std::unique_ptr<tc::InferInput>> inputs;
  for (size_t s = 0; s < slots; ++s) {
    inputs.push_back(NewInput(info.input_name, shape, info.input_datatype));
    if (stage) FAIL_IF_ERR(inputs[s]->SetSharedMemory(HipStage::kRegion, stage->slot_bytes(), s * stage->slot_bytes()), "input region");
  }
  auto output = NewOutput(info.output_name, classes);
  tc::InferOptions options(opt.model);
  options.model_version_ = opt.version;
  options.client_timeout_ = flags.client_timeout_us;

  std::mutex mu;
  std::condition_variable cv;
  size_t pending = 0;
  std::deque<size_t> free_slots;
  for (size_t s = 0; s < slots; ++s) free_slots.push_back(s);
  std::map<size_t, std::pair<std::unique_ptr<tc::InferResult>, std::vector<std::string>>> async_results;
  // the last batch wraps around to the first images to stay full
  size_t idx = 0, sent = 0;
  bool last = false;
  while (!last) {
    size_t slot = 0;
    if (stage) {  // a slot is refilled only after the response of the request that read it
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return !free_slots.empty(); });
      slot = free_slots.front();
      free_slots.pop_front();
    }
    tc::InferInput* input = inputs[slot].get();
    if (!stage) FAIL_IF_ERR(input->Reset(), "resetting input");
    std::vector<std::string> names;
    std::vector<const Image*> members;
    for (size_t b = 0; b < batch; ++b) {
      if (stage) members.push_back(&images[idx]);
      else FAIL_IF_ERR(input->AppendRaw(tensors[idx]), "appending image");
      names.push_back(images[idx].name);
      idx = (idx + 1) % images.size();
      if (idx == 0) last = true;
    }
    if (stage) stage->Fill(members, slot);
    const size_t seq = ++sent;
    options.request_id_ = std::to_string(seq);
    if (opt.async) {
      {
        std::lock_guard<std::mutex> lk(mu);
        ++pending;
      }
      // the request references `tensors` / the region, which outlive the call
      FAIL_IF_ERR(client->AsyncInfer(
                      [&, names, seq, slot](tc::InferResult* r) {
                        std::lock_guard<std::mutex> lk(mu);
                        async_results[seq] = std::make_pair(std::unique_ptr<tc::InferResult>(r), names);
                        --pending;
                        if (stage) free_slots.push_back(slot);
                        cv.notify_all();
                      },
                      options, {input}, {output.get()}, flags.headers),
                  "failed to send asynchronous infer request");
      if (!stage) {  // the shared InferInput is rewritten for the next batch: wait for this one
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return pending == 0; });
      }
    } else {
      tc::InferResult* result = nullptr;
      FAIL_IF_ERR(client->Infer(&result, options, {input}, {output.get()}, flags.headers),
                  "failed to send synchronous infer request");
      std::unique_ptr<tc::InferResult> guard(result);
      std::cout << "Request " << options.request_id_ << ", batch size " << batch << std::endl;
      PrintResult(result, info, names, classes);
      if (stage) {
        std::lock_guard<std::mutex> lk(mu);
        free_slots.push_back(slot);
      }
    }
  }
  {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return pending == 0; });
  }
  for (auto& r : async_results) {
    FAIL_IF_ERR(r.second.first->RequestStatus(), "asynchronous inference failed");
    std::cout << "Request " << r.first << ", batch size " << batch << std::endl;
    PrintResult(r.second.first.get(), info, r.second.second, classes);
  }
  if (stage) FAIL_IF_ERR(client->UnregisterCudaSharedMemory(HipStage::kRegion), "unregistering the input region");
  std::cout << "PASS : Image Client" << std::endl;
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  CommonFlags flags = ParseCommonFlags(argc, argv);
  Options opt;
  std::string protocol = "http";
  for (size_t i = 0; i < flags.rest.size(); ++i) {
    const std::string& a = flags.rest[i];
    auto value = [&]() -> std::string { return i + 1 < flags.rest.size() ? flags.rest[++i] : ""; };
    if (a == "-m") opt.model = value();
    else if (a == "-x") opt.version = value();
    else if (a == "-i") protocol = value();
    else if (a == "-b") opt.batch = std::strtoul(value().c_str(), nullptr, 10);
    else if (a == "-c") opt.classes = std::strtoul(value().c_str(), nullptr, 10);
    else if (a == "-a") opt.async = true;
    else if (a == "-g") opt.device = std::atoi(value().c_str());
    else if (a == "--synthetic") opt.synthetic = std::atoi(value().c_str());
    else if (a == "--decode-only") opt.decode_only = true;
    else if (a == "--dump") opt.dump_dir = value();
    else if (a == "--hip-lib") opt.hip_lib = value();
    else if (a == "--ring-depth") opt.ring_depth = std::max<size_t>(1, std::strtoul(value().c_str(), nullptr, 10));
    else if (a == "--preprocess") {
      const std::string p = value();
      if (p != "host" && p != "hip") FAIL("--preprocess takes host or hip");
      opt.hip = p == "hip";
    } else if (a == "-s") {
      const std::string s = value();
      opt.scale = s == "INCEPTION" ? Scale::INCEPTION : s == "VGG" ? Scale::VGG : Scale::NONE;
    } else opt.files.push_back(a);
  }
  for (auto& ch : protocol) ch = static_cast<char>(std::tolower(ch));
  if (protocol != "http" && protocol != "grpc") FAIL("-i takes http or grpc");

  std::vector<Image> images;
  for (const auto& f : opt.files) {
    tca::DecodedImage decoded;
    FAIL_IF_ERR(tca::ReadImageFile(f, &decoded), "unable to read '" + f + "'");
    Image img;
    img.h = decoded.height;
    img.w = decoded.width;
    img.c = decoded.channels;
    img.px.swap(decoded.pixels);
    img.name = f;
    images.push_back(std::move(img));
  }
  if (opt.decode_only) {
    for (const auto& img : images) {
      uint64_t hash = 1469598103934665603ull;  // FNV-1a over the pixels
      for (uint8_t p : img.px) hash = (hash ^ p) * 1099511628211ull;
      std::cout << img.name << " " << img.h << " " << img.w << " " << img.c << " " << hash << std::endl;
      if (!opt.dump_dir.empty()) {  // the decoded pixels as binary PGM / PPM, named after the file
        const size_t slash = img.name.find_last_of('/');
        const std::string base = slash == std::string::npos ? img.name : img.name.substr(slash + 1);
        std::ofstream f(opt.dump_dir + "/" + base + (img.c == 1 ? ".pgm" : ".ppm"), std::ios::binary);
        f << (img.c == 1 ? "P5" : "P6") << "\n" << img.w << " " << img.h << "\n255\n";
        f.write(reinterpret_cast<const char*>(img.px.data()), static_cast<std::streamsize>(img.px.size()));
        if (!f) FAIL("unable to write into '" << opt.dump_dir << "'");
      }
    }
    return 0;
  }
  for (int i = 0; i < opt.synthetic; ++i) images.push_back(Synthetic(i));
  if (images.empty()) FAIL("no images given (pass .png / .jpg / .ppm files or --synthetic N)");
  if (opt.model.empty()) FAIL("-m <model name> is required");
  if (protocol == "grpc") {
    auto client = MakeGrpcClient(flags);
    return Run(client.get(), flags, opt, images, argv[0]);
  }
  auto client = MakeClient(flags);
  return Run(client.get(), flags, opt, images, argv[0]);
}
