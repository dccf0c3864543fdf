// Synchronous inference over gRPC on the `simple` add/sub model: default
// outputs, explicitly requested outputs, request compression, custom headers,
// client timeout, statistics and the unknown-model error.  (Reference:
// `src/c++/examples/simple_grpc_infer_client.cc`.)
#include "example_utils.h"

int main(int argc, char** argv) {
  CommonFlags flags = ParseCommonFlags(argc, argv);
  tc::GrpcCompression compression = tc::GrpcCompression::NONE;
  for (size_t i = 0; i + 1 < flags.rest.size(); ++i) {
    if (flags.rest[i] == "-C") {
      if (flags.rest[i + 1] == "gzip") compression = tc::GrpcCompression::GZIP;
      else if (flags.rest[i + 1] == "deflate") compression = tc::GrpcCompression::DEFLATE;
      else if (flags.rest[i + 1] != "none") FAIL("unsupported compression algorithm " << flags.rest[i + 1]);
    }
  }
  auto client = MakeGrpcClient(flags);
  std::vector<int32_t> a(16), b(16, 1);
  for (int i = 0; i < 16; ++i) a[i] = i;
  auto in0 = NewInput("INPUT0", {1, 16}, "INT32");
  auto in1 = NewInput("INPUT1", {1, 16}, "INT32");
  FAIL_IF_ERR(in0->AppendRaw(reinterpret_cast<const uint8_t*>(a.data()), 64), "INPUT0 data");
  FAIL_IF_ERR(in1->AppendRaw(reinterpret_cast<const uint8_t*>(b.data()), 64), "INPUT1 data");
  auto out0 = NewOutput("OUTPUT0");
  auto out1 = NewOutput("OUTPUT1");
  tc::InferOptions options("simple");
  options.request_id_ = "1";
  options.client_timeout_ = flags.client_timeout_us;

  tc::InferResult* result = nullptr;
  FAIL_IF_ERR(client->Infer(&result, options, {in0.get(), in1.get()}, {out0.get(), out1.get()}, flags.headers,
                            compression),
              "unable to run model");
  auto holder = Own(result);
  ValidateSimple(result, a, b);
  std::string id;
  result->Id(&id);
  if (id != "1") FAIL("request id was not echoed");
  for (int i = 0; i < 16; ++i) std::cout << a[i] << " + " << b[i] << " = " << a[i] + b[i] << std::endl;

  // no outputs requested: the server returns all of them
  FAIL_IF_ERR(client->Infer(&result, options, {in0.get(), in1.get()}), "default outputs");
  holder.reset(result);
  ValidateSimple(result, a, b);

  tc::InferOptions unknown("wrong_model_name");
  const tca::Error err = client->Infer(&result, unknown, {in0.get(), in1.get()});
  if (err.IsOk() || err.Message().find("Request for unknown model") == std::string::npos)
    FAIL("expected an unknown-model error, got: " << err.Message());

  tc::InferStat stat;
  FAIL_IF_ERR(client->ClientInferStat(&stat), "client statistics");
  if (stat.completed_request_count != 2) FAIL("expected 2 completed requests, got " << stat.completed_request_count);
  std::string server_stat;
  FAIL_IF_ERR(client->ModelInferenceStatistics(&server_stat, "simple"), "server statistics");
  std::cout << "PASS : Infer" << std::endl;
  return 0;
}
