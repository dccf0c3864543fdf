// The native gRPC serving front under load, in one process, for the race / memory-error
// detectors: `grpc_front_stress [-n threads] [-r rounds]`.
//
// A GrpcServerFront (epoll reactors: HTTP/2, HPACK, gRPC framing) is started on a free port
// with a stand-in backend that behaves like the GPU executor behind the real server: the
// reactor's callback only queues the request, and completer threads answer later from
// outside the reactor - Reply / Send / Finish race with the reactors' own work on the same
// connections, which is the hot path of the serving benchmark.  Client threads drive it
// with the in-tree gRPC client: synchronous and asynchronous inference (add / subtract on
// INT32[1,16], checked), control calls through a shared client, bidirectional streams with
// several responses per request, deadlines that expire while the backend still holds the
// call, and clients that go away with calls in flight.  A hostile peer runs beside them: raw
// connections that send random bytes, and valid HTTP/2 prefaces followed by frames with random
// types, flags, lengths, stream ids and payloads (HPACK blocks included) - the front may answer
// with GOAWAY or close, it may not crash, leak or stop serving the others.  The reference has no server and
// configures no sanitizers; this is the serving-side counterpart of stress_test.cc.
#include <arpa/inet.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <iostream>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

#include "tca/grpc_client.h"
#include "tca/grpc_server.h"
#include "tca/json.h"
#include "tca/pbjson.h"
#include "tca/pbwire.h"
#include "tca/sync.h"

namespace tc = tca::client;
namespace pb = tca::pb;

namespace {

tc::RequestParameter IntParam(int64_t v) {
  tc::RequestParameter p;
  p.type = tc::RequestParameter::Type::INT;
  p.int_value = v;
  return p;
}

std::atomic<int> g_failures{0};
void Fail(const std::string& what) {
  ++g_failures;
  std::cerr << "FAIL: " << what << std::endl;
}

// ---- stand-in backend ------------------------------------------------------------------
struct Job {
  uint64_t call;
  std::string method, request;
  bool stream;
};

class Backend {
 public:
  explicit Backend(int completers) {
    for (int i = 0; i < completers; ++i) threads_.emplace_back([this, i] { Run(i); });
  }
  ~Backend() { Stop(); }
  void Attach(tca::server::GrpcServerFront* front) { front_ = front; }
  void Submit(Job&& job) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      queue_.push_back(std::move(job));
    }
    cv_.notify_one();
  }
  void Stop() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (stop_) return;
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
  }

 private:
  void Run(int seed) {
    std::mt19937 rng(1234u + seed);
    while (true) {
      Job job;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || !queue_.empty(); });
        if (queue_.empty()) return;
        job = std::move(queue_.front());
        queue_.pop_front();
      }
      // some answers come at once, some after the reactor has gone on to other work
      const unsigned d = rng() % 8;
      if (d >= 5) std::this_thread::sleep_for(std::chrono::microseconds(50 * d));
      Answer(job);
    }
  }

  void Answer(const Job& job) {
    if (job.method == "ModelInfer" || job.method == "ModelStreamInfer") {
      std::string model, id, error, body;
      int delay_ms = 0, repeat = 1;
      if (!Infer(job.request, &model, &id, &body, &error, &delay_ms, &repeat)) {
        if (job.stream) {
          std::string wrapped;
          pb::Writer w(&wrapped);
          w.String(1, error);
          front_->Send(job.call, std::move(wrapped));
        } else {
          front_->Finish(job.call, tc::kGrpcInvalidArgument, error);
        }
        return;
      }
      if (delay_ms) std::this_thread::sleep_for(std::chrono::milliseconds(delay_ms));
      if (job.stream) {
        for (int i = 0; i < repeat; ++i) {
          std::string wrapped;
          pb::Writer w(&wrapped);
          w.Message(2, body);  // ModelStreamInferResponse.infer_response
          front_->Send(job.call, std::move(wrapped));
        }
      } else {
        front_->Reply(job.call, std::move(body));
      }
      return;
    }
    int req_msg = -1, resp_msg = -1;
    if (!pb::FindMethod(job.method, &req_msg, &resp_msg)) {
      front_->Finish(job.call, tc::kGrpcUnimplemented, "Method not found: " + job.method);
      return;
    }
    tca::json::Value v = tca::json::Value::Object();
    if (job.method == "ServerLive") v["live"] = true;
    else if (job.method == "ServerReady" || job.method == "ModelReady") v["ready"] = true;
    else if (job.method == "ServerMetadata") {
      v["name"] = "triton";
      v["version"] = "front-stress";
    } else if (job.method == "ModelMetadata") {
      v["name"] = "simple";
      v["platform"] = "stand-in";
      tca::json::Value in = tca::json::Value::Array();
      for (const char* n : {"INPUT0", "INPUT1"}) {
        tca::json::Value t = tca::json::Value::Object();
        t["name"] = n;
        t["datatype"] = "INT32";
        in.Append(std::move(t));
      }
      v["inputs"] = std::move(in);
    }
    std::string wire;
    const tca::Error e = pb::FromJson(resp_msg, v, &wire);
    if (!e.IsOk()) front_->Finish(job.call, tc::kGrpcInternal, e.Message());
    else front_->Reply(job.call, std::move(wire));
  }

  // ModelInferRequest (simple: OUTPUT0 = INPUT0 + INPUT1, OUTPUT1 = INPUT0 - INPUT1) -> ModelInferResponse.
  // Request parameters understood: "delay_ms" (int64) and "repeat" (int64, streams).
  static bool Infer(const std::string& request, std::string* model, std::string* id, std::string* body, std::string* error,
                    int* delay_ms, int* repeat) {
    pb::Reader r(request);
    std::vector<std::string> raws;
    uint32_t f, wt;
    while (r.Next(&f, &wt)) {
      if (f == 1 && wt == pb::kLen) r.String(model);
      else if (f == 3 && wt == pb::kLen) r.String(id);
      else if (f == 4 && wt == pb::kLen) {  // map<string, InferParameter> entry
        const uint8_t* d;
        size_t n;
        if (!r.Len(&d, &n)) break;
        pb::Reader e(d, n);
        std::string key;
        int64_t value = 0;
        uint32_t ef, ewt;
        while (e.Next(&ef, &ewt)) {
          if (ef == 1 && ewt == pb::kLen) e.String(&key);
          else if (ef == 2 && ewt == pb::kLen) {
            const uint8_t* pd;
            size_t pn;
            if (!e.Len(&pd, &pn)) break;
            pb::Reader p(pd, pn);
            uint32_t pf, pwt;
            while (p.Next(&pf, &pwt)) {
              uint64_t u;
              if (pf == 2 && pwt == pb::kVarint && p.Varint(&u)) value = static_cast<int64_t>(u);
              else if (!p.Skip(pwt)) break;
            }
          } else if (!e.Skip(ewt)) break;
        }
        if (key == "delay_ms") *delay_ms = static_cast<int>(value);
        if (key == "repeat") *repeat = static_cast<int>(value);
      } else if (f == 7 && wt == pb::kLen) {
        std::string raw;
        r.String(&raw);
        raws.push_back(std::move(raw));
      } else if (!r.Skip(wt)) break;
    }
    if (r.Failed()) {
      *error = "malformed ModelInferRequest";
      return false;
    }
    if (*model != "simple") {
      *error = "Request for unknown model: '" + *model + "' is not found";
      return false;
    }
    if (raws.size() != 2 || raws[0].size() != 64 || raws[1].size() != 64) {
      *error = "simple expects two INT32[1,16] tensors as raw contents";
      return false;
    }
    int32_t a[16], b[16], sum[16], diff[16];
    std::memcpy(a, raws[0].data(), 64);
    std::memcpy(b, raws[1].data(), 64);
    for (int i = 0; i < 16; ++i) {
      sum[i] = a[i] + b[i];
      diff[i] = a[i] - b[i];
    }
    pb::Writer w(body);
    w.String(1, *model);
    w.String(2, "1");
    if (!id->empty()) w.String(3, *id);
    for (const char* name : {"OUTPUT0", "OUTPUT1"}) {
      std::string t;
      pb::Writer tw(&t);
      tw.String(1, name);
      tw.String(2, "INT32");
      tw.UInt(3, 1);
      tw.UInt(3, 16);
      w.Message(5, t);
    }
    w.Bytes(6, sum, 64);
    w.Bytes(6, diff, 64);
    return true;
  }

  tca::server::GrpcServerFront* front_ = nullptr;
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<Job> queue_;
  bool stop_ = false;
  std::vector<std::thread> threads_;
};

// ---- clients -----------------------------------------------------------------------------
struct Tensors {
  int32_t a[16], b[16];
  std::unique_ptr<tc::InferInput> in0, in1;
  std::vector<tc::InferInput*> inputs;
  explicit Tensors(int seed) {
    for (int i = 0; i < 16; ++i) {
      a[i] = seed * 31 + i;
      b[i] = seed - 3 * i;
    }
    tc::InferInput* p = nullptr;
    tc::InferInput::Create(&p, "INPUT0", {1, 16}, "INT32");
    in0.reset(p);
    tc::InferInput::Create(&p, "INPUT1", {1, 16}, "INT32");
    in1.reset(p);
    in0->AppendRaw(reinterpret_cast<const uint8_t*>(a), 64);
    in1->AppendRaw(reinterpret_cast<const uint8_t*>(b), 64);
    inputs = {in0.get(), in1.get()};
  }
  bool Check(tc::InferResult* r, const char* mode) const {
    if (!r->RequestStatus().IsOk()) {
      Fail(std::string(mode) + ": " + r->RequestStatus().Message());
      return false;
    }
    const uint8_t *s = nullptr, *d = nullptr;
    size_t ns = 0, nd = 0;
    if (!r->RawData("OUTPUT0", &s, &ns).IsOk() || !r->RawData("OUTPUT1", &d, &nd).IsOk() || ns != 64 || nd != 64) {
      Fail(std::string(mode) + ": outputs missing");
      return false;
    }
    for (int i = 0; i < 16; ++i) {
      int32_t sv, dv;
      std::memcpy(&sv, s + 4 * i, 4);
      std::memcpy(&dv, d + 4 * i, 4);
      if (sv != a[i] + b[i] || dv != a[i] - b[i]) {
        Fail(std::string(mode) + ": wrong values");
        return false;
      }
    }
    return true;
  }
};

struct Latch {
  std::mutex mu;
  std::condition_variable cv;
  int left;
  explicit Latch(int n) : left(n) {}
  void Down() {
    std::lock_guard<std::mutex> lk(mu);
    if (--left <= 0) cv.notify_all();
  }
  bool Wait(int seconds) {
    std::unique_lock<std::mutex> lk(mu);
    return tca::WaitFor(cv, lk, std::chrono::seconds(seconds), [&] { return left <= 0; });
  }
};

void ClientThread(int t, int rounds, const std::string& url, tc::InferenceServerGrpcClient* shared) {
  Tensors req(t + 1);
  for (int round = 0; round < rounds; ++round) {
    std::unique_ptr<tc::InferenceServerGrpcClient> c;
    // odd threads: a channel of their own, opened and dropped every round
    if (!tc::InferenceServerGrpcClient::Create(&c, url, false, false, tc::SslOptions(), tc::KeepAliveOptions(), t % 2 == 0).IsOk()) {
      Fail("creating a client");
      return;
    }
    tc::InferOptions opt("simple");
    opt.request_id_ = "t" + std::to_string(t) + "r" + std::to_string(round);
    for (int k = 0; k < 4; ++k) {
      tc::InferResult* res = nullptr;
      const tca::Error e = c->Infer(&res, opt, req.inputs);
      if (!e.IsOk()) Fail("sync infer: " + e.Message());
      else req.Check(res, "sync");
      delete res;
    }
    {
      const int n = 8;
      Latch latch(n);
      for (int k = 0; k < n; ++k) {
        const tca::Error e = c->AsyncInfer(
            [&](tc::InferResult* r) {
              req.Check(r, "async");
              delete r;
              latch.Down();
            },
            opt, req.inputs);
        if (!e.IsOk()) {
          Fail("async submit: " + e.Message());
          latch.Down();
        }
      }
      if (!latch.Wait(60)) Fail("async requests did not complete");
    }
    {
      // a stream with three responses per request
      const int requests = 4, per = 3;
      Latch latch(requests * per);
      if (!c->StartStream([&](tc::InferResult* r) {
             req.Check(r, "stream");
             delete r;
             latch.Down();
           }).IsOk()) {
        Fail("StartStream");
      } else {
        tc::InferOptions sopt("simple");
        sopt.request_parameters["repeat"] = IntParam(3);
        for (int k = 0; k < requests; ++k)
          if (!c->AsyncStreamInfer(sopt, req.inputs).IsOk()) Fail("AsyncStreamInfer");
        if (!latch.Wait(60)) Fail("stream responses did not arrive");
        c->StopStream();
      }
    }
    {
      // a deadline that runs out while the backend holds the call; the answer arrives for a call that is gone
      tc::InferOptions slow("simple");
      slow.client_timeout_ = 2000;
      slow.request_parameters["delay_ms"] = IntParam(20);
      tc::InferResult* res = nullptr;
      const tca::Error e = c->Infer(&res, slow, req.inputs);
      if (e.IsOk() || e.Message().find("Deadline Exceeded") == std::string::npos)
        Fail("expected Deadline Exceeded, got " + (e.IsOk() ? std::string("OK") : e.Message()));
      delete res;
    }
    {
      // an unknown model is an error answer, not a broken call
      tc::InferResult* res = nullptr;
      const tca::Error e = c->Infer(&res, tc::InferOptions("no_such_model"), req.inputs);
      if (e.IsOk() || e.Message().find("unknown model") == std::string::npos) Fail("unknown model: " + e.Message());
      delete res;
    }
    bool live = false;
    std::string text;
    if (!shared->IsServerLive(&live).IsOk() || !live) Fail("shared client: IsServerLive");
    if (!shared->ModelMetadata(&text, "simple").IsOk() || text.find("INPUT0") == std::string::npos)
      Fail("shared client: ModelMetadata");
    if (t % 3 == 0) {
      // leave with calls in flight: the client (and, for odd threads, its connection) goes away under them
      tc::InferOptions held("simple");
      held.request_parameters["delay_ms"] = IntParam(10);
      for (int k = 0; k < 3; ++k) c->AsyncInfer([](tc::InferResult* r) { delete r; }, held, req.inputs);
    }
  }
}

// One hostile connection: `preface` decides whether the bytes start like HTTP/2.
void HostileConnection(int port, std::mt19937_64* rng, bool preface) {
  const int fd = socket(AF_INET, SOCK_STREAM, 0);
  if (fd < 0) return;
  sockaddr_in addr{};
  addr.sin_family = AF_INET;
  addr.sin_port = htons(static_cast<uint16_t>(port));
  addr.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
  timeval tv{0, 200000};
  setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
  setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof(tv));
  if (connect(fd, reinterpret_cast<sockaddr*>(&addr), sizeof(addr)) != 0) {
    close(fd);
    return;
  }
  std::string out;
  if (preface) {
    out = "PRI * HTTP/2.0\r\n\r\nSM\r\n\r\n";
    out.append("\0\0\0\4\0\0\0\0\0", 9);  // empty SETTINGS
    const int frames = 1 + static_cast<int>((*rng)() % 12);
    for (int f = 0; f < frames; ++f) {
      const uint32_t declared = (*rng)() % 5 == 0 ? static_cast<uint32_t>((*rng)() % (1u << 24)) : static_cast<uint32_t>((*rng)() % 64);
      const uint32_t actual = std::min<uint32_t>(declared, static_cast<uint32_t>((*rng)() % 96));
      const uint8_t type = static_cast<uint8_t>((*rng)() % 12);
      const uint8_t flags = static_cast<uint8_t>((*rng)());
      const uint32_t stream = (*rng)() % 4 == 0 ? static_cast<uint32_t>((*rng)()) : static_cast<uint32_t>((*rng)() % 8);
      const char head[9] = {static_cast<char>(declared >> 16), static_cast<char>(declared >> 8), static_cast<char>(declared),
                            static_cast<char>(type), static_cast<char>(flags), static_cast<char>(stream >> 24),
                            static_cast<char>(stream >> 16), static_cast<char>(stream >> 8), static_cast<char>(stream)};
      out.append(head, 9);
      for (uint32_t k = 0; k < actual; ++k) out.push_back(static_cast<char>((*rng)()));
    }
  } else {
    const int n = 1 + static_cast<int>((*rng)() % 600);
    for (int k = 0; k < n; ++k) out.push_back(static_cast<char>((*rng)()));
  }
  (void)!send(fd, out.data(), out.size(), MSG_NOSIGNAL);
  char buf[4096];
  for (int k = 0; k < 3 && recv(fd, buf, sizeof(buf), 0) > 0; ++k) {
  }
  close(fd);
}

void HostilePeer(int port, int connections, uint64_t seed) {
  std::mt19937_64 rng(seed);
  for (int c = 0; c < connections; ++c) HostileConnection(port, &rng, c % 4 != 0);
}

}  // namespace

int main(int argc, char** argv) {
  int threads = 6, rounds = 10;
  for (int i = 1; i < argc; ++i) {
    if (!std::strcmp(argv[i], "-n") && i + 1 < argc) threads = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "-r") && i + 1 < argc) rounds = std::atoi(argv[++i]);
  }
  Backend backend(3);
  tca::server::GrpcServerFront::Handlers handlers;
  handlers.on_unary = [&](uint64_t call, const std::string& method, std::string&& request) {
    backend.Submit(Job{call, method, std::move(request), false});
  };
  handlers.on_stream_message = [&](uint64_t call, const std::string& method, std::string&& request) {
    backend.Submit(Job{call, method, std::move(request), true});
  };
  std::atomic<int> cancels{0};
  tca::server::GrpcServerFront* front_ptr = nullptr;
  handlers.on_stream_end = [&](uint64_t call) {
    // answers of the stream's requests may still be queued in the backend: close it a little later, from another thread
    std::thread([&, call] {
      std::this_thread::sleep_for(std::chrono::milliseconds(100));
      if (front_ptr != nullptr) front_ptr->Finish(call, 0);
    }).detach();
  };
  handlers.on_cancel = [&](uint64_t) { ++cancels; };
  tca::server::GrpcServerFront front("127.0.0.1", 0, 2, {"ModelStreamInfer"}, handlers);
  front_ptr = &front;
  backend.Attach(&front);
  const tca::Error started = front.Start();
  if (!started.IsOk()) {
    std::cerr << "unable to start the front: " << started.Message() << std::endl;
    return 1;
  }
  const std::string url = "127.0.0.1:" + std::to_string(front.port());
  {
    std::unique_ptr<tc::InferenceServerGrpcClient> shared;
    if (!tc::InferenceServerGrpcClient::Create(&shared, url).IsOk()) {
      std::cerr << "unable to create the shared client" << std::endl;
      return 1;
    }
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) pool.emplace_back(ClientThread, t, rounds, url, shared.get());
    pool.emplace_back(HostilePeer, front.port(), 40 * rounds, 12345);
    for (auto& th : pool) th.join();
    // the front still serves after all that
    bool live = false;
    if (!shared->IsServerLive(&live).IsOk()) Fail("the front stopped answering after the hostile connections");
  }
  // late stream closers and held calls drain before the front goes away
  std::this_thread::sleep_for(std::chrono::milliseconds(300));
  backend.Stop();
  const uint64_t calls = front.CallsStarted(), conns = front.ConnectionsAccepted();
  front.Stop();
  front_ptr = nullptr;
  std::cout << conns << " connections, " << calls << " calls, " << cancels.load() << " cancelled" << std::endl;
  if (g_failures.load()) {
    std::cerr << g_failures.load() << " failure(s)" << std::endl;
    return 1;
  }
  std::cout << "PASS : front stress" << std::endl;
  return 0;
}
