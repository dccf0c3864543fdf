// Image classification client (HTTP): reads the model's metadata + config to
// learn its input layout / type, preprocesses images on the CPU (2-tap bilinear
// resize as cv::resize does, NONE / VGG / INCEPTION scaling, NHWC or NCHW),
// sends them in batches (sync or -a async) and prints the top `-c` classes
// returned by the classification extension.
//
// Behavioural reference: `src/c++/examples/image_client.cc:85-276,537-702,1030-1052`.
// OpenCV is not available in this toolchain, so images are binary PPM/PGM
// files (P6/P5) or `--synthetic N` generated ones.  For GPU-side preprocessing
// see the Python `image_client.py --gpu-preprocess` (HIP kernel + hipshm).
#include <cmath>
#include <condition_variable>
#include <fstream>
#include <mutex>
#include <sstream>

#include "example_utils.h"
#include "tca/codec.h"

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

bool ReadPnm(const std::string& path, Image* img) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::string magic;
  f >> magic;
  if (magic != "P6" && magic != "P5") return false;
  auto next_int = [&]() {
    std::string tok;
    while (f >> tok) {
      if (tok[0] == '#') { std::string rest; std::getline(f, rest); continue; }
      return std::atoi(tok.c_str());
    }
    return 0;
  };
  img->w = next_int();
  img->h = next_int();
  const int maxval = next_int();
  f.get();
  if (img->w <= 0 || img->h <= 0 || maxval != 255) return false;
  img->c = magic == "P6" ? 3 : 1;
  img->px.resize(static_cast<size_t>(img->w) * img->h * img->c);
  f.read(reinterpret_cast<char*>(img->px.data()), static_cast<std::streamsize>(img->px.size()));
  img->name = path;
  return static_cast<size_t>(f.gcount()) == img->px.size();
}

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
  tca::json::Value md, cfg;
  TCA_RETURN_IF_ERROR(tca::json::Parse(metadata, &md));
  TCA_RETURN_IF_ERROR(tca::json::Parse(config, &cfg));
  if (md["inputs"].Size() != 1) return tca::Error("expecting 1 input, got " + std::to_string(md["inputs"].Size()));
  if (md["outputs"].Size() != 1) return tca::Error("expecting 1 output, got " + std::to_string(md["outputs"].Size()));
  if (cfg["input"].Size() != 1) return tca::Error("expecting 1 input in the model configuration");
  const auto& in = md["inputs"][0];
  const auto& out = md["outputs"][0];
  if (out.GetString("datatype") != "FP32") return tca::Error("expecting output datatype to be FP32, model output is " + out.GetString("datatype"));
  m->max_batch_size = static_cast<int>(cfg.GetInt("max_batch_size"));
  // output must be a vector (all other dims 1)
  size_t non_one = 0;
  const auto& oshape = out.Find("shape")->Items();
  for (size_t i = m->max_batch_size > 0 ? 1 : 0; i < oshape.size(); ++i) non_one += oshape[i].AsInt() > 1;
  if (non_one > 1) return tca::Error("expecting model output to be a vector");
  const auto& ishape = in.Find("shape")->Items();
  const size_t expected_dims = 3 + (m->max_batch_size > 0 ? 1 : 0);
  if (ishape.size() != expected_dims) return tca::Error("expecting input to have " + std::to_string(expected_dims) + " dimensions");
  const std::string format = cfg["input"][0].GetString("format", "FORMAT_NONE");
  if (format != "FORMAT_NCHW" && format != "FORMAT_NHWC") return tca::Error("unexpected input format " + format + ", expecting FORMAT_NCHW or FORMAT_NHWC");
  m->nhwc = format == "FORMAT_NHWC";
  const size_t b = m->max_batch_size > 0 ? 1 : 0;
  if (m->nhwc) { m->h = ishape[b].AsInt(); m->w = ishape[b + 1].AsInt(); m->c = ishape[b + 2].AsInt(); }
  else { m->c = ishape[b].AsInt(); m->h = ishape[b + 1].AsInt(); m->w = ishape[b + 2].AsInt(); }
  if (m->c != 1 && m->c != 3) return tca::Error("expecting 1 or 3 channels, got " + std::to_string(m->c));
  m->input_name = in.GetString("name");
  m->output_name = out.GetString("name");
  m->input_datatype = in.GetString("datatype");
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

}  // namespace

int main(int argc, char** argv) {
  CommonFlags flags = ParseCommonFlags(argc, argv);
  std::string model, version;
  size_t batch = 1, classes = 1;
  Scale scale = Scale::NONE;
  bool async = false;
  int synthetic = 0;
  std::vector<std::string> files;
  for (size_t i = 0; i < flags.rest.size(); ++i) {
    const std::string& a = flags.rest[i];
    auto value = [&]() -> std::string { return i + 1 < flags.rest.size() ? flags.rest[++i] : ""; };
    if (a == "-m") model = value();
    else if (a == "-x") version = value();
    else if (a == "-b") batch = std::strtoul(value().c_str(), nullptr, 10);
    else if (a == "-c") classes = std::strtoul(value().c_str(), nullptr, 10);
    else if (a == "-a") async = true;
    else if (a == "--synthetic") synthetic = std::atoi(value().c_str());
    else if (a == "-s") {
      const std::string s = value();
      scale = s == "INCEPTION" ? Scale::INCEPTION : s == "VGG" ? Scale::VGG : Scale::NONE;
    } else files.push_back(a);
  }
  if (model.empty()) FAIL("-m <model name> is required");
  auto client = MakeClient(flags);
  std::string metadata, config;
  FAIL_IF_ERR(client->ModelMetadata(&metadata, model, version, flags.headers), "failed to get model metadata");
  FAIL_IF_ERR(client->ModelConfig(&config, model, version, flags.headers), "failed to get model config");
  ModelInfo info;
  FAIL_IF_ERR(ParseModel(metadata, config, &info), "model is not an image classifier");
  if (info.max_batch_size == 0 && batch != 1) FAIL("this model does not support batching");
  if (info.max_batch_size > 0 && static_cast<int>(batch) > info.max_batch_size)
    FAIL("batch size " << batch << " exceeds the model's maximum " << info.max_batch_size);

  std::vector<Image> images;
  for (const auto& f : files) {
    Image img;
    if (!ReadPnm(f, &img)) FAIL("unable to read '" << f << "' (binary PPM/PGM expected)");
    images.push_back(std::move(img));
  }
  for (int i = 0; i < synthetic; ++i) images.push_back(Synthetic(i));
  if (images.empty()) FAIL("no images given (pass .ppm files or --synthetic N)");
  std::vector<std::vector<uint8_t>> tensors(images.size());
  for (size_t i = 0; i < images.size(); ++i) FAIL_IF_ERR(Preprocess(images[i], info, scale, &tensors[i]), "preprocessing");

  std::vector<int64_t> shape;
  if (info.max_batch_size > 0) shape.push_back(static_cast<int64_t>(batch));
  if (info.nhwc) { shape.push_back(info.h); shape.push_back(info.w); shape.push_back(info.c); }
  else { shape.push_back(info.c); shape.push_back(info.h); shape.push_back(info.w); }
  auto input = NewInput(info.input_name, shape, info.input_datatype);
  auto output = NewOutput(info.output_name, classes);
  tc::InferOptions options(model);
  options.model_version_ = version;
  options.client_timeout_ = flags.client_timeout_us;

  std::mutex mu;
  std::condition_variable cv;
  size_t pending = 0;
  std::vector<std::pair<std::unique_ptr<tc::InferResult>, std::vector<std::string>>> async_results;
  // one InferInput is reused for every batch: Reset() + AppendRaw per image; the
  // last batch wraps around to the first images to stay full
  size_t idx = 0;
  bool last = false;
  while (!last) {
    FAIL_IF_ERR(input->Reset(), "resetting input");
    std::vector<std::string> names;
    for (size_t b = 0; b < batch; ++b) {
      FAIL_IF_ERR(input->AppendRaw(tensors[idx]), "appending image");
      names.push_back(images[idx].name);
      idx = (idx + 1) % images.size();
      if (idx == 0) last = true;
    }
    options.request_id_ = std::to_string(idx);
    if (async) {
      {
        std::lock_guard<std::mutex> lk(mu);
        ++pending;
      }
      // the request references `tensors`, which outlive the call
      FAIL_IF_ERR(client->AsyncInfer(
                      [&, names](tc::InferResult* r) {
                        std::lock_guard<std::mutex> lk(mu);
                        async_results.emplace_back(std::unique_ptr<tc::InferResult>(r), names);
                        --pending;
                        cv.notify_one();
                      },
                      options, {input.get()}, {output.get()}, flags.headers),
                  "failed to send asynchronous infer request");
      // the shared InferInput is rewritten for the next batch: wait for this one to be on the wire
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return pending == 0; });
    } else {
      tc::InferResult* result = nullptr;
      FAIL_IF_ERR(client->Infer(&result, options, {input.get()}, {output.get()}, flags.headers),
                  "failed to send synchronous infer request");
      std::unique_ptr<tc::InferResult> guard(result);
      std::cout << "Request " << options.request_id_ << ", batch size " << batch << std::endl;
      PrintResult(result, info, names, classes);
    }
  }
  for (auto& r : async_results) {
    FAIL_IF_ERR(r.first->RequestStatus(), "asynchronous inference failed");
    PrintResult(r.first.get(), info, r.second, classes);
  }
  std::cout << "PASS : Image Client" << std::endl;
  return 0;
}
