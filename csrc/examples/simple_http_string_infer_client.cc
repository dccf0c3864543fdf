// BYTES tensors over HTTP on `simple_string` (decimal strings in, sum/difference
// strings out) and `simple_identity` (payloads with embedded NULs).
// (Reference: `src/c++/examples/simple_http_string_infer_client.cc`.)
#include "example_utils.h"

int main(int argc, char** argv) {
  CommonFlags flags = ParseCommonFlags(argc, argv);
  auto client = MakeClient(flags);
  std::vector<std::string> a, b;
  for (int i = 0; i < 16; ++i) {
    a.push_back(std::to_string(i));
    b.push_back("1");
  }
  auto in0 = NewInput("INPUT0", {1, 16}, "BYTES");
  auto in1 = NewInput("INPUT1", {1, 16}, "BYTES");
  FAIL_IF_ERR(in0->AppendFromString(a), "setting INPUT0");
  FAIL_IF_ERR(in1->AppendFromString(b), "setting INPUT1");
  tc::InferResult* result = nullptr;
  FAIL_IF_ERR(client->Infer(&result, tc::InferOptions("simple_string"), {in0.get(), in1.get()}, {}, flags.headers),
              "unable to run simple_string");
  std::vector<std::string> sum, diff;
  FAIL_IF_ERR(result->StringData("OUTPUT0", &sum), "OUTPUT0");
  FAIL_IF_ERR(result->StringData("OUTPUT1", &diff), "OUTPUT1");
  for (int i = 0; i < 16; ++i) {
    if (sum[i] != std::to_string(i + 1)) FAIL("incorrect sum " << sum[i]);
    if (diff[i] != std::to_string(i - 1)) FAIL("incorrect difference " << diff[i]);
  }
  delete result;

  std::vector<std::string> payload = {std::string("a\0b", 3), "", std::string("\0\0", 2), "xyz"};
  auto idin = NewInput("INPUT0", {1, 4}, "BYTES");
  FAIL_IF_ERR(idin->AppendFromString(payload), "setting identity input");
  FAIL_IF_ERR(client->Infer(&result, tc::InferOptions("simple_identity"), {idin.get()}), "unable to run simple_identity");
  std::vector<std::string> echoed;
  FAIL_IF_ERR(result->StringData("OUTPUT0", &echoed), "identity output");
  if (echoed != payload) FAIL("identity output differs from its input");
  delete result;
  std::cout << "PASS : String Infer" << std::endl;
  return 0;
}
