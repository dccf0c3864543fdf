// Health, server / model metadata and configuration over HTTP.  (Reference:
// `src/c++/examples/simple_http_health_metadata.cc`.)
#include "example_utils.h"

int main(int argc, char** argv) {
  CommonFlags flags = ParseCommonFlags(argc, argv);
  auto client = MakeClient(flags);
  bool live = false, ready = false, model_ready = false;
  FAIL_IF_ERR(client->IsServerLive(&live, flags.headers), "unable to get server liveness");
  FAIL_IF_ERR(client->IsServerReady(&ready, flags.headers), "unable to get server readiness");
  FAIL_IF_ERR(client->IsModelReady(&model_ready, "simple", "", flags.headers), "unable to get model readiness");
  if (!live || !ready || !model_ready) FAIL("server or model not ready");

  std::string text;
  tca::json::Value doc;
  FAIL_IF_ERR(client->ServerMetadata(&text, flags.headers), "unable to get server metadata");
  FAIL_IF_ERR(tca::json::Parse(text, &doc), "parsing server metadata");
  if (doc.GetString("name") != "triton") FAIL("unexpected server name " << doc.GetString("name"));
  if (flags.verbose) std::cout << text << std::endl;

  FAIL_IF_ERR(client->ModelMetadata(&text, "simple", "", flags.headers), "unable to get model metadata");
  FAIL_IF_ERR(tca::json::Parse(text, &doc), "parsing model metadata");
  if (doc.GetString("name") != "simple") FAIL("unexpected model name");
  if (doc["inputs"].Size() != 2 || doc["outputs"].Size() != 2) FAIL("unexpected tensor counts in metadata");

  FAIL_IF_ERR(client->ModelConfig(&text, "simple", "", flags.headers), "unable to get model config");
  FAIL_IF_ERR(tca::json::Parse(text, &doc), "parsing model config");
  if (doc.GetInt("max_batch_size") != 8) FAIL("unexpected max_batch_size");

  tca::Error e = client->ModelMetadata(&text, "wrong_model_name");
  if (e.IsOk() || e.Message().find("Request for unknown model") == std::string::npos)
    FAIL("expected an unknown model error");
  std::cout << "PASS : Health Metadata" << std::endl;
  return 0;
}
