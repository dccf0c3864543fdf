// Native HTTP/1.1 serving front: accepts keep-alive connections on epoll
// reactor threads, parses requests (Content-Length and chunked bodies,
// `Expect: 100-continue`) and hands each complete request to the application;
// the response may be sent from any thread at any later time, so a request
// waiting for the GPU never occupies a thread.
//
// It is the REST twin of the gRPC front in tca/grpc_server.h: same reactor
// layout (one SO_REUSEPORT listener per reactor), same asynchronous
// completion, optional TLS.  The contract it has to satisfy is the one the
// HTTP clients speak (`src/c++/library/http_client.cc:2098-2246`: POST with an
// `Inference-Header-Content-Length` header in front of binary tensors, no
// `Expect:` round trip, multi-megabyte bodies).
#pragma once

#include <atomic>
#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "tca/error.h"
#include "tca/tls.h"

namespace tca {
namespace server {

class HttpServerFront {
 public:
  using HeaderList = std::vector<std::pair<std::string, std::string>>;

  struct Request {
    std::string method;   // "GET", "POST", ...
    std::string target;   // request target as sent: path[?query]
    HeaderList headers;   // names lower-cased, in arrival order
    std::string body;     // de-chunked
    bool keep_alive = true;
    const std::string* Header(const std::string& lower_name) const {
      for (const auto& h : headers)
        if (h.first == lower_name) return &h.second;
      return nullptr;
    }
  };

  struct Handlers {
    // A request is complete.  Answer with Respond(), from any thread, any time.
    std::function<void(uint64_t call, Request&& request)> on_request;
    // The connection went away before Respond().
    std::function<void(uint64_t call)> on_cancel;
  };

  struct Limits {
    size_t max_header_bytes = 1 << 20;
    size_t max_body_bytes = size_t(2) << 30;
  };

  HttpServerFront(std::string host, int port, int threads, Handlers handlers);
  ~HttpServerFront();

  // Serve HTTPS instead of cleartext; call before Start().  `context` must be a server context.
  void SetTls(std::shared_ptr<net::TlsContext> context) { tls_ = std::move(context); }
  void SetLimits(const Limits& limits) { limits_ = limits; }

  Error Start();
  void Stop();
  int port() const { return port_; }

  // Thread-safe; unknown / already answered calls are ignored.  `Content-Length`, and
  // `Connection: close` where due, are added here; `parts` are sent back to back as the body.
  void Respond(uint64_t call, int status, HeaderList&& headers, std::vector<std::string>&& parts);

  uint64_t ConnectionsAccepted() const { return accepted_.load(); }
  uint64_t RequestsStarted() const { return requests_.load(); }

 private:
  struct Reactor;
  std::string host_;
  int port_;
  int nthreads_;
  Handlers handlers_;
  Limits limits_;
  std::shared_ptr<net::TlsContext> tls_;
  std::vector<std::unique_ptr<Reactor>> reactors_;
  std::atomic<bool> running_{false};
  std::atomic<uint64_t> accepted_{0};
  std::atomic<uint64_t> requests_{0};
};

}  // namespace server
}  // namespace tca
