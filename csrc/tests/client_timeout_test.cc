// Client timeout behaviour (`client_timeout_test -u host:port -t <usec>`):
// sync and async inference on `custom_identity_int32` while the server is
// configured with an execution delay must fail with "Deadline Exceeded" when
// the timeout is shorter than the delay and succeed when it is longer.
// Intent: reference `src/c++/tests/client_timeout_test.cc:59-506`.
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <mutex>

#include "tca/http_client.h"

namespace tc = tca::client;

#define FAIL_IF_ERR(X, MSG)                                        \
  do {                                                             \
    tca::Error err = (X);                                          \
    if (!err.IsOk()) {                                             \
      std::cerr << "error: " << (MSG) << ": " << err.Message() << std::endl; \
      return 1;                                                    \
    }                                                              \
  } while (0)

int main(int argc, char** argv) {
  std::string url = "localhost:8000";
  uint64_t timeout_us = 0;
  int delay_ms = 300;
  for (int i = 1; i < argc; ++i) {
    if (!std::strcmp(argv[i], "-u") && i + 1 < argc) url = argv[++i];
    else if (!std::strcmp(argv[i], "-t") && i + 1 < argc) timeout_us = std::strtoull(argv[++i], nullptr, 10);
    else if (!std::strcmp(argv[i], "-d") && i + 1 < argc) delay_ms = std::atoi(argv[++i]);
  }
  std::unique_ptr<tc::InferenceServerHttpClient> client;
  FAIL_IF_ERR(tc::InferenceServerHttpClient::Create(&client, url), "unable to create client");
  const std::string model = "custom_identity_int32";
  FAIL_IF_ERR(client->LoadModel(model, {}, {}, "{\"parameters\": {\"execute_delay_ms\": {\"string_value\": \"" +
                                                   std::to_string(delay_ms) + "\"}}}"),
              "configuring server side delay");
  int32_t data[16];
  for (int i = 0; i < 16; ++i) data[i] = i;
  tc::InferInput* in = nullptr;
  FAIL_IF_ERR(tc::InferInput::Create(&in, "INPUT0", {1, 16}, "INT32"), "creating input");
  std::unique_ptr<tc::InferInput> guard(in);
  in->AppendRaw(reinterpret_cast<const uint8_t*>(data), sizeof(data));
  tc::InferOptions opt(model);
  opt.client_timeout_ = timeout_us;
  const bool expect_timeout = timeout_us != 0 && timeout_us < static_cast<uint64_t>(delay_ms) * 1000;
  int failures = 0;

  // synchronous
  tc::InferResult* res = nullptr;
  tca::Error e = client->Infer(&res, opt, {in});
  if (expect_timeout) {
    if (e.IsOk() || e.Message().find("Deadline Exceeded") == std::string::npos) {
      std::cerr << "sync: expected Deadline Exceeded, got " << (e.IsOk() ? "OK" : e.Message()) << std::endl;
      ++failures;
    }
  } else if (!e.IsOk()) {
    std::cerr << "sync: unexpected failure " << e.Message() << std::endl;
    ++failures;
  }
  delete res;

  // asynchronous
  std::mutex mu;
  std::condition_variable cv;
  bool done = false;
  tca::Error async_status;
  FAIL_IF_ERR(client->AsyncInfer(
                  [&](tc::InferResult* r) {
                    std::lock_guard<std::mutex> lk(mu);
                    async_status = r->RequestStatus();
                    delete r;
                    done = true;
                    cv.notify_one();
                  },
                  opt, {in}),
              "submitting async request");
  {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return done; });
  }
  if (expect_timeout) {
    if (async_status.IsOk() || async_status.Message().find("Deadline Exceeded") == std::string::npos) {
      std::cerr << "async: expected Deadline Exceeded, got " << (async_status.IsOk() ? "OK" : async_status.Message()) << std::endl;
      ++failures;
    }
  } else if (!async_status.IsOk()) {
    std::cerr << "async: unexpected failure " << async_status.Message() << std::endl;
    ++failures;
  }
  // let a timed-out request finish on the server before restoring the model
  client.reset();
  FAIL_IF_ERR(tc::InferenceServerHttpClient::Create(&client, url), "unable to re-create client");
  FAIL_IF_ERR(client->LoadModel(model), "restoring model");
  if (failures) return 1;
  std::cout << (expect_timeout ? "PASS: timeouts reported" : "PASS: requests completed") << std::endl;
  return 0;
}
