// Concurrency stress (`stress_test -u http_host:port -g grpc_host:port -n threads -r rounds`):
// N threads hammer the server at once through every client mode - sync and
// async HTTP on per-thread clients, sync / async gRPC on per-thread clients
// that SHARE one cached channel, and control RPCs on one gRPC client shared by
// all threads (the calls documented as thread-safe).  Every answer is checked.
// Meant to be run under -fsanitize=thread and -fsanitize=address,undefined
// (`make SAN=...`); the reference has no such test and documents its clients'
// thread-safety only in comments (`src/c++/library/http_client.h:89-94`,
// `grpc_client.h:85-89`).
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <mutex>
#include <thread>
#include <vector>

#include "tca/grpc_client.h"
#include "tca/http_client.h"

namespace tc = tca::client;

namespace {

std::atomic<int> g_failures{0};
std::mutex g_print;

void Fail(const std::string& what) {
  g_failures.fetch_add(1);
  std::lock_guard<std::mutex> lk(g_print);
  std::cerr << "FAIL: " << what << std::endl;
}

struct Request {
  std::vector<int32_t> a, b;
  std::unique_ptr<tc::InferInput> in0, in1;
  std::vector<tc::InferInput*> inputs;
  explicit Request(int seed) : a(16), b(16) {
    for (int i = 0; i < 16; ++i) {
      a[i] = seed + i;
      b[i] = 2 * seed - i;
    }
    tc::InferInput* p = nullptr;
    tc::InferInput::Create(&p, "INPUT0", {1, 16}, "INT32");
    in0.reset(p);
    tc::InferInput::Create(&p, "INPUT1", {1, 16}, "INT32");
    in1.reset(p);
    in0->AppendRaw(reinterpret_cast<const uint8_t*>(a.data()), 64);
    in1->AppendRaw(reinterpret_cast<const uint8_t*>(b.data()), 64);
    inputs = {in0.get(), in1.get()};
  }
  bool Check(tc::InferResult* result, const char* mode) const {
    std::unique_ptr<tc::InferResult> guard(result);
    if (result == nullptr) {
      Fail(std::string(mode) + ": null result");
      return false;
    }
    if (!result->RequestStatus().IsOk()) {
      Fail(std::string(mode) + ": " + result->RequestStatus().Message());
      return false;
    }
    const uint8_t* sum = nullptr;
    const uint8_t* diff = nullptr;
    size_t n0 = 0, n1 = 0;
    if (!result->RawData("OUTPUT0", &sum, &n0).IsOk() || !result->RawData("OUTPUT1", &diff, &n1).IsOk() || n0 != 64 ||
        n1 != 64) {
      Fail(std::string(mode) + ": missing outputs");
      return false;
    }
    for (int i = 0; i < 16; ++i) {
      int32_t s, d;
      std::memcpy(&s, sum + 4 * i, 4);
      std::memcpy(&d, diff + 4 * i, 4);
      if (s != a[i] + b[i] || d != a[i] - b[i]) {
        Fail(std::string(mode) + ": wrong values");
        return false;
      }
    }
    return true;
  }
};

// waits until `pending` asynchronous answers of one thread have arrived
struct Latch {
  std::mutex mu;
  std::condition_variable cv;
  int pending = 0;
  void Add() {
    std::lock_guard<std::mutex> lk(mu);
    ++pending;
  }
  void Done() {
    std::lock_guard<std::mutex> lk(mu);
    if (--pending == 0) cv.notify_all();
  }
  void Wait() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [this] { return pending == 0; });
  }
};

}  // namespace

int main(int argc, char** argv) {
  std::string http_url, grpc_url;
  int threads = 8, rounds = 20;
  for (int i = 1; i < argc; ++i) {
    if (!std::strcmp(argv[i], "-u") && i + 1 < argc) http_url = argv[++i];
    else if (!std::strcmp(argv[i], "-g") && i + 1 < argc) grpc_url = argv[++i];
    else if (!std::strcmp(argv[i], "-n") && i + 1 < argc) threads = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "-r") && i + 1 < argc) rounds = std::atoi(argv[++i]);
  }
  if (http_url.empty() || grpc_url.empty()) {
    std::cerr << "usage: stress_test -u http_host:port -g grpc_host:port [-n threads] [-r rounds]" << std::endl;
    return 2;
  }
  std::unique_ptr<tc::InferenceServerGrpcClient> shared;
  if (!tc::InferenceServerGrpcClient::Create(&shared, grpc_url).IsOk()) {
    std::cerr << "unable to create the shared gRPC client" << std::endl;
    return 1;
  }
  std::atomic<long> completed{0};
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) {
    pool.emplace_back([&, t] {
      std::unique_ptr<tc::InferenceServerHttpClient> http;
      std::unique_ptr<tc::InferenceServerGrpcClient> grpc;
      if (!tc::InferenceServerHttpClient::Create(&http, http_url).IsOk() ||
          !tc::InferenceServerGrpcClient::Create(&grpc, grpc_url).IsOk()) {  // same url: shares the cached channel
        Fail("client creation");
        return;
      }
      tc::InferOptions options("simple");
      Latch latch;
      std::vector<std::unique_ptr<Request>> keep;  // async requests reference their buffers until completion
      for (int r = 0; r < rounds; ++r) {
        const int seed = t * 1000 + r;
        {
          Request req(seed);
          tc::InferResult* result = nullptr;
          const tca::Error e = http->Infer(&result, options, req.inputs);
          if (!e.IsOk()) Fail("http sync: " + e.Message());
          else if (req.Check(result, "http sync")) completed.fetch_add(1);
        }
        {
          Request req(seed + 1);
          tc::InferResult* result = nullptr;
          const tca::Error e = grpc->Infer(&result, options, req.inputs);
          if (!e.IsOk()) Fail("grpc sync: " + e.Message());
          else if (req.Check(result, "grpc sync")) completed.fetch_add(1);
        }
        for (int k = 0; k < 2; ++k) {
          keep.emplace_back(new Request(seed + 2 + k));
          Request* req = keep.back().get();
          latch.Add();
          const tca::Error e = k == 0 ? http->AsyncInfer(
                                            [&, req](tc::InferResult* result) {
                                              if (req->Check(result, "http async")) completed.fetch_add(1);
                                              latch.Done();
                                            },
                                            options, req->inputs)
                                      : grpc->AsyncInfer(
                                            [&, req](tc::InferResult* result) {
                                              if (req->Check(result, "grpc async")) completed.fetch_add(1);
                                              latch.Done();
                                            },
                                            options, req->inputs);
          if (!e.IsOk()) {
            Fail("async submit: " + e.Message());
            latch.Done();
          }
        }
        bool live = false;
        if (!shared->IsServerLive(&live).IsOk() || !live) Fail("shared client: IsServerLive");
        std::string metadata;
        if (!shared->ModelMetadata(&metadata, "simple").IsOk() || metadata.find("INPUT0") == std::string::npos)
          Fail("shared client: ModelMetadata");
      }
      latch.Wait();
      tc::InferStat stat;
      http->ClientInferStat(&stat);
      if (stat.completed_request_count != static_cast<size_t>(2 * rounds)) Fail("http client statistics are off");
    });
  }
  for (auto& th : pool) th.join();
  const long expected = static_cast<long>(threads) * rounds * 4;
  if (completed.load() != expected) Fail("completed " + std::to_string(completed.load()) + " of " + std::to_string(expected));
  if (g_failures.load() != 0) return 1;
  std::cout << "PASS : stress (" << threads << " threads x " << rounds << " rounds, " << expected << " inferences)" << std::endl;
  return 0;
}
