// Asynchronous inference on `simple` over HTTP: several requests in flight on
// the client's epoll reactor, results delivered to callbacks.  (Reference:
// `src/c++/examples/simple_http_async_infer_client.cc`.)
#include <condition_variable>
#include <mutex>

#include "example_utils.h"

int main(int argc, char** argv) {
  CommonFlags flags = ParseCommonFlags(argc, argv);
  auto client = MakeClient(flags);
  std::vector<int32_t> a(16), b(16, 1);
  for (int i = 0; i < 16; ++i) a[i] = i;
  auto in0 = NewInput("INPUT0", {1, 16}, "INT32");
  auto in1 = NewInput("INPUT1", {1, 16}, "INT32");
  in0->AppendRaw(reinterpret_cast<uint8_t*>(a.data()), 64);
  in1->AppendRaw(reinterpret_cast<uint8_t*>(b.data()), 64);
  auto out0 = NewOutput("OUTPUT0");
  auto out1 = NewOutput("OUTPUT1");
  tc::InferOptions options("simple");
  options.client_timeout_ = flags.client_timeout_us;
  std::vector<tc::InferInput*> inputs = {in0.get(), in1.get()};
  std::vector<const tc::InferRequestedOutput*> outputs = {out0.get(), out1.get()};

  constexpr int kRequests = 16;
  std::mutex mu;
  std::condition_variable cv;
  std::vector<std::unique_ptr<tc::InferResult>> done;
  for (int i = 0; i < kRequests; ++i) {
    options.request_id_ = std::to_string(i);
    FAIL_IF_ERR(client->AsyncInfer(
                    [&](tc::InferResult* r) {
                      std::lock_guard<std::mutex> lk(mu);
                      done.emplace_back(r);
                      cv.notify_one();
                    },
                    options, inputs, outputs, flags.headers),
                "unable to submit request");
  }
  {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return done.size() == kRequests; });
  }
  for (auto& r : done) {
    FAIL_IF_ERR(r->RequestStatus(), "inference failed");
    ValidateSimple(r.get(), a, b);
  }
  tc::InferStat stat;
  FAIL_IF_ERR(client->ClientInferStat(&stat), "client statistics");
  if (stat.completed_request_count != kRequests) FAIL("statistics count mismatch");
  std::cout << "PASS : Async Infer" << std::endl;
  return 0;
}
