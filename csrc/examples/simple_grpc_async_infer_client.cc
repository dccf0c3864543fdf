// Asynchronous inference over gRPC: several requests in flight, completion
// callbacks on the channel's reactor thread.  (Reference:
// `src/c++/examples/simple_grpc_async_infer_client.cc`.)
#include <condition_variable>
#include <mutex>

#include "example_utils.h"
#include "tca/sync.h"

int main(int argc, char** argv) {
  CommonFlags flags = ParseCommonFlags(argc, argv);
  auto client = MakeGrpcClient(flags);
  std::vector<int32_t> a(16), b(16, 1);
  for (int i = 0; i < 16; ++i) a[i] = i;
  auto in0 = NewInput("INPUT0", {1, 16}, "INT32");
  auto in1 = NewInput("INPUT1", {1, 16}, "INT32");
  in0->AppendRaw(reinterpret_cast<const uint8_t*>(a.data()), 64);
  in1->AppendRaw(reinterpret_cast<const uint8_t*>(b.data()), 64);
  tc::InferOptions options("simple");
  options.client_timeout_ = flags.client_timeout_us;

  constexpr int kRequests = 8;
  std::mutex mu;
  std::condition_variable cv;
  std::vector<std::unique_ptr<tc::InferResult>> results;
  for (int i = 0; i < kRequests; ++i) {
    options.request_id_ = std::to_string(i);
    FAIL_IF_ERR(client->AsyncInfer(
                    [&](tc::InferResult* r) {
                      std::lock_guard<std::mutex> lk(mu);
                      results.emplace_back(r);
                      cv.notify_all();
                    },
                    options, {in0.get(), in1.get()}, {}, flags.headers),
                "unable to send request");
  }
  {
    std::unique_lock<std::mutex> lk(mu);
    if (!tca::WaitFor(cv, lk, std::chrono::seconds(30), [&] { return results.size() == kRequests; }))
      FAIL("timed out waiting for the responses");
  }
  for (auto& r : results) {
    FAIL_IF_ERR(r->RequestStatus(), "inference failed");
    ValidateSimple(r.get(), a, b);
  }
  tc::InferStat stat;
  client->ClientInferStat(&stat);
  if (stat.completed_request_count != kRequests) FAIL("statistics count " << stat.completed_request_count);
  std::cout << "PASS : Async Infer" << std::endl;
  return 0;
}
