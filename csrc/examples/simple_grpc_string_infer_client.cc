// BYTES tensors over gRPC: decimal strings through `simple_string`, a value
// with an embedded NUL through `simple_identity`.  (Reference:
// `src/c++/examples/simple_grpc_string_infer_client.cc`.)
#include "example_utils.h"

int main(int argc, char** argv) {
  CommonFlags flags = ParseCommonFlags(argc, argv);
  auto client = MakeGrpcClient(flags);
  std::vector<std::string> s0, s1;
  for (int i = 0; i < 16; ++i) {
    s0.push_back(std::to_string(i));
    s1.push_back("1");
  }
  auto in0 = NewInput("INPUT0", {1, 16}, "BYTES");
  auto in1 = NewInput("INPUT1", {1, 16}, "BYTES");
  FAIL_IF_ERR(in0->AppendFromString(s0), "INPUT0 data");
  FAIL_IF_ERR(in1->AppendFromString(s1), "INPUT1 data");
  tc::InferResult* result = nullptr;
  FAIL_IF_ERR(client->Infer(&result, tc::InferOptions("simple_string"), {in0.get(), in1.get()}), "unable to run model");
  auto holder = Own(result);
  std::vector<std::string> sum, diff;
  FAIL_IF_ERR(result->StringData("OUTPUT0", &sum), "OUTPUT0");
  FAIL_IF_ERR(result->StringData("OUTPUT1", &diff), "OUTPUT1");
  if (sum.size() != 16 || diff.size() != 16) FAIL("unexpected element count");
  for (int i = 0; i < 16; ++i) {
    std::cout << s0[i] << " + 1 = " << sum[i] << ", " << s0[i] << " - 1 = " << diff[i] << std::endl;
    if (std::stoi(sum[i]) != i + 1 || std::stoi(diff[i]) != i - 1) FAIL("incorrect result at " << i);
  }
  auto id = NewInput("INPUT0", {1, 1}, "BYTES");
  const std::string with_nul("hello\0world", 11);
  FAIL_IF_ERR(id->AppendFromString({with_nul}), "identity data");
  FAIL_IF_ERR(client->Infer(&result, tc::InferOptions("simple_identity"), {id.get()}), "identity");
  holder.reset(result);
  std::vector<std::string> echoed;
  FAIL_IF_ERR(result->StringData("OUTPUT0", &echoed), "identity output");
  if (echoed.size() != 1 || echoed[0] != with_nul) FAIL("embedded NUL was not preserved");
  std::cout << "PASS : String Infer" << std::endl;
  return 0;
}
