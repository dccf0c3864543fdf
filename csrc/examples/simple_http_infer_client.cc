// Synchronous inference on `simple` over HTTP: binary and JSON tensors, custom
// headers / query parameters, request+response compression, client statistics
// and the unknown-model error.  (Reference example:
// `src/c++/examples/simple_http_infer_client.cc`.)
#include "example_utils.h"

int main(int argc, char** argv) {
  CommonFlags flags = ParseCommonFlags(argc, argv);
  auto client = MakeClient(flags);
  std::vector<int32_t> a(16), b(16, 1);
  for (int i = 0; i < 16; ++i) a[i] = i;
  auto in0 = NewInput("INPUT0", {1, 16}, "INT32");
  auto in1 = NewInput("INPUT1", {1, 16}, "INT32");
  FAIL_IF_ERR(in0->AppendRaw(reinterpret_cast<uint8_t*>(a.data()), 64), "setting INPUT0");
  FAIL_IF_ERR(in1->AppendRaw(reinterpret_cast<uint8_t*>(b.data()), 64), "setting INPUT1");
  auto out0 = NewOutput("OUTPUT0");
  auto out1 = NewOutput("OUTPUT1");
  tc::InferOptions options("simple");
  options.client_timeout_ = flags.client_timeout_us;
  options.request_id_ = "1";
  std::vector<tc::InferInput*> inputs = {in0.get(), in1.get()};
  std::vector<const tc::InferRequestedOutput*> outputs = {out0.get(), out1.get()};
  using CT = tc::InferenceServerHttpClient::CompressionType;

  tc::InferResult* result = nullptr;
  FAIL_IF_ERR(client->Infer(&result, options, inputs, outputs, flags.headers, {{"test_1", "1"}, {"test_2", "2"}}),
              "unable to run model");
  ValidateSimple(result, a, b);
  std::string name, id;
  FAIL_IF_ERR(result->ModelName(&name), "model name");
  FAIL_IF_ERR(result->Id(&id), "request id");
  if (name != "simple" || id != "1") FAIL("unexpected model name / id: " << name << " / " << id);
  if (flags.verbose) std::cout << result->DebugString() << std::endl;
  delete result;

  // one input as JSON, one output as JSON
  in0->SetBinaryData(false);
  out1->SetBinaryData(false);
  FAIL_IF_ERR(client->Infer(&result, options, inputs, outputs, flags.headers), "JSON tensor request");
  ValidateSimple(result, a, b);
  delete result;
  in0->SetBinaryData(true);
  out1->SetBinaryData(true);

  // no outputs requested: the server returns all of them
  FAIL_IF_ERR(client->Infer(&result, options, inputs, {}, flags.headers), "request without outputs");
  ValidateSimple(result, a, b);
  delete result;

  for (CT alg : {CT::GZIP, CT::DEFLATE}) {
    FAIL_IF_ERR(client->Infer(&result, options, inputs, outputs, flags.headers, {}, alg, alg), "compressed request");
    ValidateSimple(result, a, b);
    delete result;
  }

  tc::InferStat stat;
  FAIL_IF_ERR(client->ClientInferStat(&stat), "client statistics");
  if (stat.completed_request_count != 5) FAIL("expected 5 completed requests, got " << stat.completed_request_count);
  std::string server_stat;
  FAIL_IF_ERR(client->ModelInferenceStatistics(&server_stat, "simple"), "server statistics");
  if (server_stat.find("\"simple\"") == std::string::npos) FAIL("statistics do not mention the model");

  tc::InferOptions bad("wrong_model_name");
  tca::Error e = client->Infer(&result, bad, inputs, outputs);
  if (e.IsOk() || e.Message().find("Request for unknown model") == std::string::npos)
    FAIL("expected an unknown model error, got: " << (e.IsOk() ? "OK" : e.Message()));
  std::cout << "PASS : Infer" << std::endl;
  return 0;
}
