// Ensemble image client: the raw encoded image file is sent as one BYTES element
// per image; decoding and preprocessing happen on the server inside the
// `preprocess_inception_ensemble` model.  (Reference:
// `src/c++/examples/ensemble_image_client.cc`.)
#include <fstream>
#include <sstream>

#include "example_utils.h"

static std::string SyntheticPpm(int index) {
  const int h = 60 + 4 * index, w = 80;
  std::string out = "P6\n" + std::to_string(w) + " " + std::to_string(h) + "\n255\n";
  uint32_t state = 99u + 31u * static_cast<uint32_t>(index);
  for (int i = 0; i < h * w * 3; ++i) {
    state = state * 1664525u + 1013904223u;
    out.push_back(static_cast<char>(state >> 24));
  }
  return out;
}

int main(int argc, char** argv) {
  CommonFlags flags = ParseCommonFlags(argc, argv);
  size_t classes = 1;
  int synthetic = 0;
  std::vector<std::string> files;
  for (size_t i = 0; i < flags.rest.size(); ++i) {
    const std::string& a = flags.rest[i];
    if (a == "-c" && i + 1 < flags.rest.size()) classes = std::strtoul(flags.rest[++i].c_str(), nullptr, 10);
    else if (a == "--synthetic" && i + 1 < flags.rest.size()) synthetic = std::atoi(flags.rest[++i].c_str());
    else files.push_back(a);
  }
  std::vector<std::string> blobs, names;
  for (const auto& f : files) {
    std::ifstream in(f, std::ios::binary);
    if (!in) FAIL("unable to open " << f);
    std::ostringstream ss;
    ss << in.rdbuf();
    blobs.push_back(ss.str());
    names.push_back(f);
  }
  for (int i = 0; i < synthetic; ++i) {
    blobs.push_back(SyntheticPpm(i));
    names.push_back("synthetic_" + std::to_string(i));
  }
  if (blobs.empty()) FAIL("no images given");
  auto client = MakeClient(flags);
  auto input = NewInput("INPUT", {static_cast<int64_t>(blobs.size()), 1}, "BYTES");
  FAIL_IF_ERR(input->AppendFromString(blobs), "setting image bytes");
  auto output = NewOutput("OUTPUT", classes);
  tc::InferResult* result = nullptr;
  FAIL_IF_ERR(client->Infer(&result, tc::InferOptions("preprocess_inception_ensemble"), {input.get()}, {output.get()},
                            flags.headers),
              "failed sending synchronous infer request");
  std::unique_ptr<tc::InferResult> guard(result);
  std::vector<std::string> strs;
  FAIL_IF_ERR(result->StringData("OUTPUT", &strs), "classification results");
  if (strs.size() != blobs.size() * classes) FAIL("unexpected number of classification strings " << strs.size());
  for (size_t b = 0; b < blobs.size(); ++b) {
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
  std::cout << "PASS : Ensemble Image Client" << std::endl;
  return 0;
}
