// Model repository control over gRPC: index, unload / load, config override,
// the error for an unknown model.  (Reference:
// `src/c++/examples/simple_grpc_model_control.cc`.)
#include "example_utils.h"
#include "tca/json.h"

int main(int argc, char** argv) {
  CommonFlags flags = ParseCommonFlags(argc, argv);
  auto client = MakeGrpcClient(flags);
  std::string text;
  tca::json::Value v;
  FAIL_IF_ERR(client->ModelRepositoryIndex(&text), "repository index");
  FAIL_IF_ERR(tca::json::Parse(text.data(), text.size(), &v), "index JSON");
  if (v["models"].Size() < 8) FAIL("expected at least 8 models in the repository index, got " << v["models"].Size());
  bool ready = true;
  FAIL_IF_ERR(client->UnloadModel("simple"), "unload");
  FAIL_IF_ERR(client->IsModelReady(&ready, "simple"), "readiness");
  if (ready) FAIL("model still ready after unload");
  FAIL_IF_ERR(client->LoadModel("simple"), "load");
  FAIL_IF_ERR(client->IsModelReady(&ready, "simple"), "readiness");
  if (!ready) FAIL("model not ready after load");
  FAIL_IF_ERR(client->LoadModel("simple", flags.headers, R"({"backend":"python","max_batch_size":16})"),
              "load with config override");
  FAIL_IF_ERR(client->ModelConfig(&text, "simple"), "model config");
  FAIL_IF_ERR(tca::json::Parse(text.data(), text.size(), &v), "config JSON");
  if (v["config"].GetInt("max_batch_size") != 16) FAIL("config override was not applied: " << text);
  FAIL_IF_ERR(client->LoadModel("simple"), "reload");
  const tca::Error err = client->LoadModel("wrong_model_name");
  if (err.IsOk() || err.Message().find("failed to load") == std::string::npos)
    FAIL("expected a load failure, got: " << err.Message());
  std::cout << "PASS : Model Control" << std::endl;
  return 0;
}
