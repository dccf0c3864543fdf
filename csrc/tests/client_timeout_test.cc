// Client timeout behaviour:
//   client_timeout_test -u host:port [-i http|grpc] -t <usec> [-d <delay ms>] [-p]
// Inference on `custom_identity_int32` while the server is configured with an execution delay
// must fail with "Deadline Exceeded" when the timeout is shorter than the delay and succeed
// when it is longer - synchronously, asynchronously and (gRPC) over the bidirectional stream,
// whose timeout bounds the whole stream.  With -p (gRPC) every control call is made under the
// deadline as well: with a deadline no server can meet each of them has to report
// "Deadline Exceeded", and without one each has to succeed; the calls that did not are counted.
// Intent: reference `src/c++/tests/client_timeout_test.cc:59-151,211-341,361-506`.
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iostream>
#include <mutex>
#include <vector>

#include "tca/grpc_client.h"
#include "tca/http_client.h"

namespace tc = tca::client;

#define FAIL_IF_ERR(X, MSG)                                                  \
  do {                                                                       \
    tca::Error err = (X);                                                    \
    if (!err.IsOk()) {                                                       \
      std::cerr << "error: " << (MSG) << ": " << err.Message() << std::endl; \
      return 1;                                                              \
    }                                                                        \
  } while (0)

namespace {

const char* const kModel = "custom_identity_int32";

bool IsDeadline(const tca::Error& e) { return !e.IsOk() && e.Message().find("Deadline Exceeded") != std::string::npos; }

// One verdict per mode: did the outcome match the expectation?
int Judge(const char* mode, const tca::Error& status, bool expect_timeout) {
  if (expect_timeout ? IsDeadline(status) : status.IsOk()) return 0;
  std::cerr << mode << ": expected " << (expect_timeout ? "Deadline Exceeded" : "success") << ", got "
            << (status.IsOk() ? "OK" : status.Message()) << std::endl;
  return 1;
}

struct Waiter {
  std::mutex mu;
  std::condition_variable cv;
  bool done = false;
  tca::Error status;
  void Set(const tca::Error& e) {
    std::lock_guard<std::mutex> lk(mu);
    if (done) return;
    status = e;
    done = true;
    cv.notify_all();
  }
  tca::Error Get() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return done; });
    return status;
  }
};

std::string DelayConfig(int delay_ms) {
  return "{\"parameters\": {\"execute_delay_ms\": {\"string_value\": \"" + std::to_string(delay_ms) + "\"}}}";
}

int RunHttp(const std::string& url, uint64_t timeout_us, int delay_ms) {
  std::unique_ptr<tc::InferenceServerHttpClient> client;
  FAIL_IF_ERR(tc::InferenceServerHttpClient::Create(&client, url), "unable to create client");
  FAIL_IF_ERR(client->LoadModel(kModel, {}, {}, DelayConfig(delay_ms)), "configuring server side delay");
  int32_t data[16];
  for (int i = 0; i < 16; ++i) data[i] = i;
  tc::InferInput* in = nullptr;
  FAIL_IF_ERR(tc::InferInput::Create(&in, "INPUT0", {1, 16}, "INT32"), "creating input");
  std::unique_ptr<tc::InferInput> guard(in);
  in->AppendRaw(reinterpret_cast<const uint8_t*>(data), sizeof(data));
  tc::InferOptions opt(kModel);
  opt.client_timeout_ = timeout_us;
  const bool expect_timeout = timeout_us != 0 && timeout_us < static_cast<uint64_t>(delay_ms) * 1000;
  int failures = 0;
  tc::InferResult* res = nullptr;
  failures += Judge("http sync", client->Infer(&res, opt, {in}), expect_timeout);
  delete res;
  Waiter w;
  FAIL_IF_ERR(client->AsyncInfer(
                  [&](tc::InferResult* r) {
                    const tca::Error s = r->RequestStatus();
                    delete r;
                    w.Set(s);
                  },
                  opt, {in}),
              "submitting async request");
  failures += Judge("http async", w.Get(), expect_timeout);
  // let a timed-out request finish on the server before restoring the model
  client.reset();
  FAIL_IF_ERR(tc::InferenceServerHttpClient::Create(&client, url), "unable to re-create client");
  FAIL_IF_ERR(client->LoadModel(kModel), "restoring model");
  if (failures) return 1;
  std::cout << (expect_timeout ? "PASS: timeouts reported" : "PASS: requests completed") << std::endl;
  return 0;
}

int RunGrpc(const std::string& url, uint64_t timeout_us, int delay_ms) {
  std::unique_ptr<tc::InferenceServerGrpcClient> client;
  FAIL_IF_ERR(tc::InferenceServerGrpcClient::Create(&client, url), "unable to create client");
  FAIL_IF_ERR(client->LoadModel(kModel, {}, DelayConfig(delay_ms)), "configuring server side delay");
  int32_t data[16];
  for (int i = 0; i < 16; ++i) data[i] = i;
  tc::InferInput* in = nullptr;
  FAIL_IF_ERR(tc::InferInput::Create(&in, "INPUT0", {1, 16}, "INT32"), "creating input");
  std::unique_ptr<tc::InferInput> guard(in);
  in->AppendRaw(reinterpret_cast<const uint8_t*>(data), sizeof(data));
  tc::InferOptions opt(kModel);
  opt.client_timeout_ = timeout_us;
  const bool expect_timeout = timeout_us != 0 && timeout_us < static_cast<uint64_t>(delay_ms) * 1000;
  int failures = 0;
  tc::InferResult* res = nullptr;
  failures += Judge("grpc sync", client->Infer(&res, opt, {in}), expect_timeout);
  delete res;
  {
    Waiter w;
    FAIL_IF_ERR(client->AsyncInfer(
                    [&](tc::InferResult* r) {
                      const tca::Error s = r->RequestStatus();
                      delete r;
                      w.Set(s);
                    },
                    opt, {in}),
                "submitting async request");
    failures += Judge("grpc async", w.Get(), expect_timeout);
  }
  {
    // the stream's own deadline (whole stream, microseconds); the request carries none
    Waiter w;
    FAIL_IF_ERR(client->StartStream(
                    [&](tc::InferResult* r) {
                      const tca::Error s = r->RequestStatus();
                      delete r;
                      w.Set(s);
                    },
                    true, static_cast<uint32_t>(timeout_us)),
                "starting the stream");
    FAIL_IF_ERR(client->AsyncStreamInfer(tc::InferOptions(kModel), {in}), "sending on the stream");
    failures += Judge("grpc stream", w.Get(), expect_timeout);
    client->StopStream();
  }
  client.reset();
  FAIL_IF_ERR(tc::InferenceServerGrpcClient::Create(&client, url), "unable to re-create client");
  FAIL_IF_ERR(client->LoadModel(kModel), "restoring model");
  if (failures) return 1;
  std::cout << (expect_timeout ? "PASS: timeouts reported" : "PASS: requests completed") << std::endl;
  return 0;
}

// Every control call of the gRPC client under `timeout_us` (reference `:211-341`).
int RunGrpcControlApis(const std::string& url, uint64_t timeout_us) {
  std::unique_ptr<tc::InferenceServerGrpcClient> client;
  FAIL_IF_ERR(tc::InferenceServerGrpcClient::Create(&client, url), "unable to create client");
  const tc::Headers h;
  const uint8_t handle[64] = {0};
  auto calls = [&](uint64_t t) {
    bool flag = false;
    std::string text;
    return std::vector<std::pair<const char*, tca::Error>>{
        {"IsServerLive", client->IsServerLive(&flag, h, t)},
        {"IsServerReady", client->IsServerReady(&flag, h, t)},
        {"IsModelReady", client->IsModelReady(&flag, kModel, "", h, t)},
        {"ServerMetadata", client->ServerMetadata(&text, h, t)},
        {"ModelMetadata", client->ModelMetadata(&text, kModel, "", h, t)},
        {"ModelConfig", client->ModelConfig(&text, kModel, "", h, t)},
        {"ModelRepositoryIndex", client->ModelRepositoryIndex(&text, h, t)},
        {"LoadModel", client->LoadModel(kModel, h, "", {}, t)},
        {"ModelInferenceStatistics", client->ModelInferenceStatistics(&text, kModel, "", h, t)},
        {"UpdateTraceSettings", client->UpdateTraceSettings(&text, kModel, {{"trace_rate", {"1"}}}, h, t)},
        {"GetTraceSettings", client->GetTraceSettings(&text, kModel, h, t)},
        {"UpdateLogSettings", client->UpdateLogSettings(&text, "{\"log_info\": true}", h, t)},
        {"GetLogSettings", client->GetLogSettings(&text, h, t)},
        {"SystemSharedMemoryStatus", client->SystemSharedMemoryStatus(&text, "", h, t)},
        {"UnregisterSystemSharedMemory", client->UnregisterSystemSharedMemory("", h, t)},
        {"CudaSharedMemoryStatus", client->CudaSharedMemoryStatus(&text, "", h, t)},
        {"UnregisterCudaSharedMemory", client->UnregisterCudaSharedMemory("", h, t)},
        // a region that does not exist: the answer is an error either way, a deadline error only under the deadline
        {"RegisterSystemSharedMemory", client->RegisterSystemSharedMemory("timeout_test_region", "/tca_no_such_key", 64, 0, h, t)},
        {"RegisterCudaSharedMemory", client->RegisterCudaSharedMemory("timeout_test_region", handle, 0, 64, h, t)},
        {"UnloadModel+LoadModel", [&] {
           tca::Error e = client->UnloadModel(kModel, h, t);
           client->LoadModel(kModel);  // whatever happened, the model is back for the calls that follow
           return e;
         }()},
    };
  };
  int failures = 0, total = 0;
  for (const auto& kv : calls(timeout_us)) {
    ++total;
    if (!IsDeadline(kv.second)) {
      std::cerr << kv.first << ": expected Deadline Exceeded, got " << (kv.second.IsOk() ? "OK" : kv.second.Message())
                << std::endl;
      ++failures;
    }
  }
  for (const auto& kv : calls(0)) {
    const bool registers = std::strncmp(kv.first, "Register", 8) == 0;
    // without a deadline: success, or (the two registrations of things that do not exist) the server's own error
    if (registers ? (kv.second.IsOk() || IsDeadline(kv.second)) : !kv.second.IsOk()) {
      std::cerr << kv.first << " without a deadline: " << (kv.second.IsOk() ? "unexpected OK" : kv.second.Message()) << std::endl;
      ++failures;
    }
  }
  std::cout << total << " control calls under a " << timeout_us << " us deadline, " << failures << " unexpected outcomes"
            << std::endl;
  if (failures) return 1;
  std::cout << "PASS: control calls honour the deadline" << std::endl;
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  std::string url, protocol = "http";
  uint64_t timeout_us = 0;
  int delay_ms = 300;
  bool control_apis = false;
  for (int i = 1; i < argc; ++i) {
    if (!std::strcmp(argv[i], "-u") && i + 1 < argc) url = argv[++i];
    else if (!std::strcmp(argv[i], "-i") && i + 1 < argc) protocol = argv[++i];
    else if (!std::strcmp(argv[i], "-t") && i + 1 < argc) timeout_us = std::strtoull(argv[++i], nullptr, 10);
    else if (!std::strcmp(argv[i], "-d") && i + 1 < argc) delay_ms = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "-p")) control_apis = true;
  }
  for (auto& ch : protocol) ch = static_cast<char>(std::tolower(static_cast<unsigned char>(ch)));
  if (protocol != "http" && protocol != "grpc") {
    std::cerr << "unknown protocol '" << protocol << "' (http | grpc)" << std::endl;
    return 2;
  }
  if (url.empty()) url = protocol == "grpc" ? "localhost:8001" : "localhost:8000";
  if (control_apis) {
    if (protocol != "grpc") {
      std::cerr << "-p tests the control calls of the gRPC client: use -i grpc" << std::endl;
      return 2;
    }
    return RunGrpcControlApis(url, timeout_us ? timeout_us : 1);
  }
  return protocol == "grpc" ? RunGrpc(url, timeout_us, delay_ms) : RunHttp(url, timeout_us, delay_ms);
}
