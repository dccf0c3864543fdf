// Health, metadata, configuration and statistics over gRPC; answers are JSON
// text with the .proto field names.  (Reference:
// `src/c++/examples/simple_grpc_health_metadata.cc`.)
#include "example_utils.h"
#include "tca/json.h"

int main(int argc, char** argv) {
  CommonFlags flags = ParseCommonFlags(argc, argv);
  auto client = MakeGrpcClient(flags);
  bool ok = false;
  FAIL_IF_ERR(client->IsServerLive(&ok, flags.headers), "server liveness");
  if (!ok) FAIL("server is not live");
  FAIL_IF_ERR(client->IsServerReady(&ok), "server readiness");
  if (!ok) FAIL("server is not ready");
  FAIL_IF_ERR(client->IsModelReady(&ok, "simple"), "model readiness");
  if (!ok) FAIL("model is not ready");
  std::string text;
  tca::json::Value v;
  FAIL_IF_ERR(client->ServerMetadata(&text), "server metadata");
  FAIL_IF_ERR(tca::json::Parse(text.data(), text.size(), &v), "server metadata JSON");
  if (v.GetString("name") != "triton") FAIL("unexpected server name " << v.GetString("name"));
  std::cout << text << std::endl;
  FAIL_IF_ERR(client->ModelMetadata(&text, "simple"), "model metadata");
  FAIL_IF_ERR(tca::json::Parse(text.data(), text.size(), &v), "model metadata JSON");
  if (v.GetString("name") != "simple" || v["inputs"].Size() != 2) FAIL("unexpected model metadata " << text);
  std::cout << text << std::endl;
  FAIL_IF_ERR(client->ModelConfig(&text, "simple"), "model config");
  FAIL_IF_ERR(tca::json::Parse(text.data(), text.size(), &v), "model config JSON");
  if (v["config"].GetString("name") != "simple") FAIL("unexpected model config " << text);
  const tca::Error err = client->ModelMetadata(&text, "wrong_model_name");
  if (err.IsOk() || err.Message().find("Request for unknown model") == std::string::npos)
    FAIL("expected an unknown-model error, got: " << err.Message());
  FAIL_IF_ERR(client->ModelInferenceStatistics(&text), "statistics");
  std::cout << "PASS : Health Metadata" << std::endl;
  return 0;
}
