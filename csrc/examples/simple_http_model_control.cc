// Model repository control over HTTP: index, unload, load, config override and
// the failure text for an unknown model.  (Reference:
// `src/c++/examples/simple_http_model_control.cc`.)
#include "example_utils.h"

int main(int argc, char** argv) {
  CommonFlags flags = ParseCommonFlags(argc, argv);
  auto client = MakeClient(flags);
  std::string index;
  FAIL_IF_ERR(client->ModelRepositoryIndex(&index, flags.headers), "unable to get repository index");
  tca::json::Value doc;
  FAIL_IF_ERR(tca::json::Parse(index, &doc), "parsing repository index");
  if (doc.Size() < 8) FAIL("expected at least 8 models in the repository index, got " << doc.Size());
  bool ready = false;
  FAIL_IF_ERR(client->UnloadModel("simple", flags.headers), "unable to unload model");
  FAIL_IF_ERR(client->IsModelReady(&ready, "simple"), "readiness after unload");
  if (ready) FAIL("model still ready after unload");
  FAIL_IF_ERR(client->LoadModel("simple", flags.headers), "unable to load model");
  FAIL_IF_ERR(client->IsModelReady(&ready, "simple"), "readiness after load");
  if (!ready) FAIL("model not ready after load");
  FAIL_IF_ERR(client->LoadModel("simple", flags.headers, {}, R"({"max_batch_size": 16})"), "config override");
  std::string cfg;
  FAIL_IF_ERR(client->ModelConfig(&cfg, "simple"), "config after override");
  FAIL_IF_ERR(tca::json::Parse(cfg, &doc), "parsing config");
  if (doc.GetInt("max_batch_size") != 16) FAIL("config override did not take effect");
  FAIL_IF_ERR(client->LoadModel("simple", flags.headers), "restoring model");
  tca::Error e = client->LoadModel("wrong_model_name");
  if (e.IsOk() || e.Message().find("failed to load") == std::string::npos) FAIL("expected a load failure");
  std::cout << "PASS : Model Control" << std::endl;
  return 0;
}
