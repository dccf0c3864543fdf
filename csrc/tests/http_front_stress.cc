// The native HTTP/1.1 serving front under load, in one process, for the race / memory-error
// detectors: `http_front_stress [-n threads] [-r rounds]`.
//
// An HttpServerFront (epoll reactors: request parsing, keep-alive, response writing) is
// started on a free port with a stand-in backend that behaves like the GPU executor behind
// the real server: the reactor's callback only queues the request, completer threads
// answer later through Respond() from outside the reactor.  Client threads drive it with
// the in-tree HTTP client (synchronous and asynchronous inference with binary tensors,
// checked; health and metadata calls through a shared client; deadlines that expire while
// the backend still holds the request; clients that go away with requests in flight) and
// with raw sockets (pipelined requests, chunked bodies, malformed input, connections
// dropped mid-request).  Serving-side counterpart of stress_test.cc for REST.
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

#include "tca/http_client.h"
#include "tca/http_server.h"
#include "tca/json.h"
#include "tca/sync.h"

namespace tc = tca::client;
using tca::server::HttpServerFront;

namespace {

std::atomic<int> g_failures{0};
void Fail(const std::string& what) {
  ++g_failures;
  std::cerr << "FAIL: " << what << std::endl;
}

struct Job {
  uint64_t call;
  HttpServerFront::Request request;
};

class Backend {
 public:
  explicit Backend(int completers) {
    for (int i = 0; i < completers; ++i) threads_.emplace_back([this, i] { Run(i); });
  }
  ~Backend() { Stop(); }
  void Attach(HttpServerFront* front) { front_ = front; }
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
    std::mt19937 rng(99u + seed);
    while (true) {
      Job job;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || !queue_.empty(); });
        if (queue_.empty()) return;
        job = std::move(queue_.front());
        queue_.pop_front();
      }
      const unsigned d = rng() % 8;
      if (d >= 5) std::this_thread::sleep_for(std::chrono::microseconds(50 * d));
      Answer(job);
    }
  }

  void Error(uint64_t call, int status, const std::string& message) {
    tca::json::Value v = tca::json::Value::Object();
    v["error"] = message;
    front_->Respond(call, status, {}, {v.Dump()});
  }

  void Answer(const Job& job) {
    const HttpServerFront::Request& r = job.request;
    if (r.target == "/v2/health/live" || r.target == "/v2/health/ready") {
      front_->Respond(job.call, 200, {}, {});
      return;
    }
    if (r.target == "/v2/models/simple") {
      front_->Respond(job.call, 200, {}, {"{\"name\":\"simple\",\"inputs\":[{\"name\":\"INPUT0\"},{\"name\":\"INPUT1\"}]}"});
      return;
    }
    if (r.target != "/v2/models/simple/infer") {
      Error(job.call, r.target.find("/infer") != std::string::npos ? 400 : 404,
            r.target.find("/infer") != std::string::npos ? "Request for unknown model: not found" : "no handler");
      return;
    }
    size_t json_len = r.body.size();
    if (const std::string* hl = r.Header("inference-header-content-length")) json_len = std::strtoull(hl->c_str(), nullptr, 10);
    tca::json::Value doc;
    if (json_len > r.body.size() || !tca::json::Parse(r.body.data(), json_len, &doc).IsOk() || doc.Find("inputs") == nullptr) {
      Error(job.call, 400, "failed to parse the request JSON buffer");
      return;
    }
    int32_t a[16], b[16];
    size_t cursor = json_len;
    int seen = 0;
    for (const auto& in : doc["inputs"].Items()) {
      int32_t* dst = in.GetString("name") == "INPUT0" ? a : b;
      const tca::json::Value* params = in.Find("parameters");
      if (params != nullptr && params->Find("binary_data_size") != nullptr) {
        const size_t n = static_cast<size_t>(params->GetInt("binary_data_size"));
        if (n != 64 || cursor + n > r.body.size()) {
          Error(job.call, 400, "unexpected binary data size");
          return;
        }
        std::memcpy(dst, r.body.data() + cursor, 64);
        cursor += n;
      } else if (in.Find("data") != nullptr && in.Find("data")->Size() == 16) {
        for (int i = 0; i < 16; ++i) dst[i] = static_cast<int32_t>(in.Find("data")->Items()[i].AsInt());
      } else {
        Error(job.call, 400, "input without data");
        return;
      }
      ++seen;
    }
    if (seen != 2) {
      Error(job.call, 400, "simple expects two inputs");
      return;
    }
    int delay_ms = 0;
    if (const tca::json::Value* p = doc.Find("parameters")) delay_ms = static_cast<int>(p->GetInt("delay_ms", 0));
    if (delay_ms) std::this_thread::sleep_for(std::chrono::milliseconds(delay_ms));
    int32_t sum[16], diff[16];
    for (int i = 0; i < 16; ++i) {
      sum[i] = a[i] + b[i];
      diff[i] = a[i] - b[i];
    }
    std::string head = "{\"model_name\":\"simple\",\"model_version\":\"1\"";
    if (!doc.GetString("id").empty()) head += ",\"id\":\"" + doc.GetString("id") + "\"";
    head += ",\"outputs\":[{\"name\":\"OUTPUT0\",\"datatype\":\"INT32\",\"shape\":[1,16],\"parameters\":{\"binary_data_size\":64}},"
            "{\"name\":\"OUTPUT1\",\"datatype\":\"INT32\",\"shape\":[1,16],\"parameters\":{\"binary_data_size\":64}}]}";
    HttpServerFront::HeaderList headers = {{"Inference-Header-Content-Length", std::to_string(head.size())},
                                           {"Content-Type", "application/octet-stream"}};
    std::vector<std::string> parts;
    parts.push_back(std::move(head));
    parts.emplace_back(reinterpret_cast<const char*>(sum), 64);
    parts.emplace_back(reinterpret_cast<const char*>(diff), 64);
    front_->Respond(job.call, 200, std::move(headers), std::move(parts));
  }

  HttpServerFront* front_ = nullptr;
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<Job> queue_;
  bool stop_ = false;
  std::vector<std::thread> threads_;
};

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

// One exchange over a raw socket; returns what came back until the peer went quiet or closed.
std::string Raw(int port, const std::string& payload, bool drop_early = false) {
  const int fd = socket(AF_INET, SOCK_STREAM, 0);
  sockaddr_in addr{};
  addr.sin_family = AF_INET;
  addr.sin_port = htons(static_cast<uint16_t>(port));
  addr.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
  std::string out;
  if (connect(fd, reinterpret_cast<sockaddr*>(&addr), sizeof(addr)) == 0) {
    size_t sent = 0;
    while (sent < payload.size()) {
      const ssize_t n = send(fd, payload.data() + sent, payload.size() - sent, MSG_NOSIGNAL);
      if (n <= 0) break;
      sent += static_cast<size_t>(n);
    }
    if (!drop_early) {
      timeval tv{0, 300000};
      setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
      char buf[8192];
      for (;;) {
        const ssize_t n = recv(fd, buf, sizeof(buf), 0);
        if (n <= 0) break;
        out.append(buf, static_cast<size_t>(n));
      }
    }
  }
  close(fd);
  return out;
}

size_t Count(const std::string& hay, const std::string& needle) {
  size_t n = 0;
  for (size_t pos = hay.find(needle); pos != std::string::npos; pos = hay.find(needle, pos + needle.size())) ++n;
  return n;
}

// A hostile peer: well-formed requests (Content-Length, chunked, pipelined, Expect) with a few bytes flipped,
// overwritten, inserted or removed, and plain random bytes.  Any answer or a closed connection is fine; a crash,
// a leak or an access outside a buffer is what the detectors are watching for.
void HostilePeer(int port, int connections, uint64_t seed) {
  std::mt19937_64 rng(seed);
  const std::string json = "{\"inputs\":[{\"name\":\"INPUT0\",\"datatype\":\"INT32\",\"shape\":[1,16],\"data\":[1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16]},"
                           "{\"name\":\"INPUT1\",\"datatype\":\"INT32\",\"shape\":[1,16],\"data\":[1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1]}]}";
  const std::string seeds[4] = {
      "POST /v2/models/simple/infer HTTP/1.1\r\nHost: x\r\nContent-Length: " + std::to_string(json.size()) + "\r\n\r\n" + json,
      "POST /v2/models/simple/infer HTTP/1.1\r\nHost: x\r\nTransfer-Encoding: chunked\r\n\r\n20\r\n" + json.substr(0, 32) + "\r\n" +
          "a\r\n" + json.substr(32, 10) + "\r\n0\r\n\r\n",
      "GET /v2/health/live HTTP/1.1\r\nHost: x\r\n\r\nGET /v2/models/simple HTTP/1.1\r\nHost: x\r\nConnection: close\r\n\r\n",
      "POST /v2/models/simple/infer HTTP/1.1\r\nHost: x\r\nExpect: 100-continue\r\nInference-Header-Content-Length: 40\r\nContent-Length: 99999999\r\n\r\n" + json};
  for (int c = 0; c < connections; ++c) {
    std::string m;
    if (c % 5 == 4) {
      const int n = 1 + static_cast<int>(rng() % 500);
      for (int k = 0; k < n; ++k) m.push_back(static_cast<char>(rng()));
    } else {
      m = seeds[rng() % 4];
      const int edits = 1 + static_cast<int>(rng() % 4);
      for (int k = 0; k < edits && !m.empty(); ++k) {
        const size_t at = rng() % std::min<size_t>(m.size(), 160);  // mostly the head: that is where the parser works
        switch (rng() % 4) {
          case 0: m[at] = static_cast<char>(m[at] ^ (1 << (rng() % 8))); break;
          case 1: m[at] = static_cast<char>(rng()); break;
          case 2: m.insert(at, 1, static_cast<char>(rng())); break;
          default: m.erase(at, 1 + rng() % 3); break;
        }
      }
    }
    Raw(port, m, /*drop_early=*/c % 3 == 0);
  }
}

// The HTTP client is a one-thread object (like the reference's): the shared one is used under a lock.
std::mutex g_shared_mu;

void ClientThread(int t, int rounds, const std::string& url, int port, tc::InferenceServerHttpClient* shared) {
  Tensors req(t + 1);
  const std::string json = "{\"inputs\":[{\"name\":\"INPUT0\",\"datatype\":\"INT32\",\"shape\":[1,16],\"data\":[1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16]},"
                           "{\"name\":\"INPUT1\",\"datatype\":\"INT32\",\"shape\":[1,16],\"data\":[1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1]}]}";
  for (int round = 0; round < rounds; ++round) {
    std::unique_ptr<tc::InferenceServerHttpClient> c;
    if (!tc::InferenceServerHttpClient::Create(&c, url, false).IsOk()) {
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
      // a deadline that runs out while the backend holds the request; the answer arrives for a connection that is gone
      tc::InferOptions slow("simple");
      slow.client_timeout_ = 2000;
      slow.request_parameters["delay_ms"].type = tc::RequestParameter::Type::INT;
      slow.request_parameters["delay_ms"].int_value = 20;
      tc::InferResult* res = nullptr;
      const tca::Error e = c->Infer(&res, slow, req.inputs);
      if (e.IsOk()) Fail("expected a deadline error");
      delete res;
    }
    {
      tc::InferResult* res = nullptr;
      const tca::Error e = c->Infer(&res, tc::InferOptions("no_such_model"), req.inputs);
      if (e.IsOk() || e.Message().find("unknown model") == std::string::npos) Fail("unknown model: " + e.Message());
      delete res;
    }
    {
      std::lock_guard<std::mutex> lk(g_shared_mu);
      bool live = false;
      std::string text;
      if (!shared->IsServerLive(&live).IsOk() || !live) Fail("shared client: IsServerLive");
      if (!shared->ModelMetadata(&text, "simple").IsOk() || text.find("INPUT0") == std::string::npos)
        Fail("shared client: ModelMetadata");
    }
    // raw sockets: pipelined pair, chunked body, garbage, a connection dropped with the request half sent
    const std::string post = "POST /v2/models/simple/infer HTTP/1.1\r\nHost: x\r\nContent-Length: " + std::to_string(json.size()) + "\r\n\r\n" + json;
    std::string out = Raw(port, post + "GET /v2/health/live HTTP/1.1\r\nHost: x\r\n\r\n");
    if (Count(out, "HTTP/1.1 200 OK") != 2) Fail("pipelined pair: " + out.substr(0, 200));
    char sizes[2][32];
    const size_t half = json.size() / 2;
    std::snprintf(sizes[0], sizeof(sizes[0]), "%zx", half);
    std::snprintf(sizes[1], sizeof(sizes[1]), "%zx", json.size() - half);
    out = Raw(port, "POST /v2/models/simple/infer HTTP/1.1\r\nHost: x\r\nTransfer-Encoding: chunked\r\n\r\n" + std::string(sizes[0]) + "\r\n" +
                        json.substr(0, half) + "\r\n" + sizes[1] + "\r\n" + json.substr(half) + "\r\n0\r\n\r\n");
    if (Count(out, "HTTP/1.1 200 OK") != 1 || out.find("OUTPUT1") == std::string::npos) Fail("chunked body: " + out.substr(0, 200));
    out = Raw(port, "NONSENSE\r\n\r\n");
    if (out.compare(0, 12, "HTTP/1.1 400") != 0) Fail("garbage request line: " + out.substr(0, 100));
    Raw(port, post.substr(0, post.size() - 20), /*drop_early=*/true);
    Raw(port, post, /*drop_early=*/true);  // complete request, client gone before the answer
    if (t % 3 == 0) {
      tc::InferOptions held("simple");
      held.request_parameters["delay_ms"].type = tc::RequestParameter::Type::INT;
      held.request_parameters["delay_ms"].int_value = 10;
      for (int k = 0; k < 3; ++k) c->AsyncInfer([](tc::InferResult* r) { delete r; }, held, req.inputs);
    }
  }
}

}  // namespace

int main(int argc, char** argv) {
  int threads = 6, rounds = 10;
  for (int i = 1; i < argc; ++i) {
    if (!std::strcmp(argv[i], "-n") && i + 1 < argc) threads = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "-r") && i + 1 < argc) rounds = std::atoi(argv[++i]);
  }
  Backend backend(3);
  std::atomic<int> cancels{0};
  HttpServerFront::Handlers handlers;
  handlers.on_request = [&](uint64_t call, HttpServerFront::Request&& request) { backend.Submit(Job{call, std::move(request)}); };
  handlers.on_cancel = [&](uint64_t) { ++cancels; };
  HttpServerFront front("127.0.0.1", 0, 2, handlers);
  backend.Attach(&front);
  const tca::Error started = front.Start();
  if (!started.IsOk()) {
    std::cerr << "unable to start the front: " << started.Message() << std::endl;
    return 1;
  }
  const std::string url = "127.0.0.1:" + std::to_string(front.port());
  {
    std::unique_ptr<tc::InferenceServerHttpClient> shared;
    if (!tc::InferenceServerHttpClient::Create(&shared, url, false).IsOk()) {
      std::cerr << "unable to create the shared client" << std::endl;
      return 1;
    }
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) pool.emplace_back(ClientThread, t, rounds, url, front.port(), shared.get());
    pool.emplace_back(HostilePeer, front.port(), 30 * rounds, 4242);
    for (auto& t : pool) t.join();
    bool live = false;
    if (!shared->IsServerLive(&live).IsOk() || !live) Fail("the front stopped answering after the hostile connections");
  }
  std::this_thread::sleep_for(std::chrono::milliseconds(100));  // answers of abandoned requests drain
  backend.Stop();
  front.Stop();
  std::cout << "connections " << front.ConnectionsAccepted() << ", requests " << front.RequestsStarted() << ", cancelled "
            << cancels.load() << ", failures " << g_failures.load() << std::endl;
  if (g_failures.load() != 0) return 1;
  std::cout << "PASS : http front stress" << std::endl;
  return 0;
}
