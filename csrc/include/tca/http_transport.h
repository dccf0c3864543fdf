// Self-contained HTTP/1.1 transport on POSIX sockets (no libcurl in this
// toolchain; the reference drives curl easy/multi handles,
// `src/c++/library/http_client.cc:2098-2534`).
//
// One non-blocking connection state machine (`HttpConn`) serves both modes:
//   * synchronous calls pump it with poll() on the caller's thread;
//   * asynchronous calls are multiplexed by `HttpReactor`, one epoll thread
//     that owns a pool of keep-alive connections (any number in flight).
// Request bodies are gather lists sent with writev (JSON header first, then the
// caller's tensor buffers, never concatenated); responses are read into one
// contiguous string, de-chunking when needed.
#pragma once

#include <atomic>
#include <cstdint>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "tca/common.h"
#include "tca/error.h"
#include "tca/tls.h"

namespace tca {
namespace client {

struct HttpRequestSpec {
  std::string method = "GET";
  std::string target = "/";
  Headers headers;
  // body pieces in order; `owned` keeps storage for pieces this request created itself
  std::vector<std::pair<const uint8_t*, size_t>> body;
  std::deque<std::string> owned;
  uint64_t timeout_us = 0;  // whole exchange, 0 = none

  void AddOwned(std::string piece) {
    owned.emplace_back(std::move(piece));
    body.emplace_back(reinterpret_cast<const uint8_t*>(owned.back().data()), owned.back().size());
  }
  size_t BodySize() const {
    size_t n = 0;
    for (const auto& p : body) n += p.second;
    return n;
  }
};

struct HttpResponseData {
  int status = 0;
  Headers headers;  // names lower-cased
  std::string body;
  const std::string* Header(const std::string& lower_name) const {
    auto it = headers.find(lower_name);
    return it == headers.end() ? nullptr : &it->second;
  }
};

struct HttpEndpoint {
  std::string host;
  int port = 80;
  std::string base_path;  // "" or "/prefix"
  bool https = false;     // the url carried the https:// scheme
  std::shared_ptr<net::TlsContext> tls;  // set for https endpoints before the first connection
};

// Splits "[http://|https://]host[:port][/base]"; the port defaults to 80 / 443.
Error ParseHttpUrl(const std::string& url, HttpEndpoint* out);

class HttpConn {
 public:
  enum class State { IDLE, CONNECTING, TLS_HANDSHAKE, SENDING, READ_HEAD, READ_BODY, READ_CHUNKED, READ_TO_CLOSE, DONE, FAILED };

  explicit HttpConn(const HttpEndpoint& ep) : ep_(ep) {}
  ~HttpConn() { Close(); }
  HttpConn(const HttpConn&) = delete;
  HttpConn& operator=(const HttpConn&) = delete;

  // Begin an exchange.  `spec`, `resp` and `timers` must stay valid until the
  // state becomes DONE or FAILED.
  Error Begin(const HttpRequestSpec* spec, HttpResponseData* resp, RequestTimers* timers);
  // Advance after the fd became readable/writable (or unconditionally).
  void Pump();
  // Blocking helper: pump with poll() until finished or the deadline passes.
  Error RunToCompletion();

  State state() const { return state_; }
  bool Finished() const { return state_ == State::DONE || state_ == State::FAILED; }
  const Error& error() const { return error_; }
  int fd() const { return fd_; }
  // bumps every time a new socket is opened (fd numbers get reused by the kernel)
  uint64_t generation() const { return generation_; }
  // direction to wait for; a TLS session may need the opposite of what the HTTP state suggests
  bool WantsWrite() const {
    if (state_ == State::TLS_HANDSHAKE) return tls_ && tls_->WantsWrite();
    if (tls_ && tls_blocked_) return tls_->WantsWrite();
    return state_ == State::CONNECTING || state_ == State::SENDING;
  }
  bool Reusable() const { return fd_ >= 0 && keep_alive_ && state_ == State::DONE; }
  uint64_t deadline_ns() const { return deadline_ns_; }
  void Fail(const Error& e);
  void Close();

 private:
  Error OpenSocket();
  void BuildHead();
  bool SendSome();
  bool RecvSome();   // false when nothing more can be done right now
  ssize_t RecvRaw(void* buf, size_t len);  // recv() semantics (errno) over the socket or the TLS session
  bool Reconnect();  // stale keep-alive connection: open a fresh socket and start over
  bool ParseHead();
  bool ParseChunks();
  void Finish();

  HttpEndpoint ep_;
  int fd_ = -1;
  std::unique_ptr<net::TlsSession> tls_;
  bool tls_blocked_ = false;  // the last TLS call returned kWouldBlock
  uint64_t generation_ = 0;
  State state_ = State::IDLE;
  Error error_;
  bool keep_alive_ = true;
  bool reused_ = false;
  bool retried_ = false;

  const HttpRequestSpec* spec_ = nullptr;
  HttpResponseData* resp_ = nullptr;
  RequestTimers* timers_ = nullptr;
  uint64_t deadline_ns_ = 0;

  std::string head_;
  size_t piece_ = 0;      // index into {head_, body...}
  size_t piece_off_ = 0;
  std::string rbuf_;
  size_t body_expected_ = 0;
  bool recv_started_ = false;
  // chunked decoding
  size_t chunk_left_ = 0;
  int chunk_phase_ = 0;  // 0 size line, 1 data, 2 CRLF after data, 3 trailers
};

// Epoll-driven executor of asynchronous exchanges.
class HttpReactor {
 public:
  struct Op {
    HttpRequestSpec spec;
    HttpResponseData resp;
    RequestTimers timers;
    std::function<void(const Error&, Op*)> done;  // invoked on the reactor thread
  };

  explicit HttpReactor(const HttpEndpoint& ep);
  ~HttpReactor();
  void Submit(std::unique_ptr<Op> op);
  size_t InFlight() const { return inflight_.load(); }

 private:
  void Loop();
  struct Active;
  HttpEndpoint ep_;
  int epfd_ = -1;
  int wake_fd_ = -1;
  std::thread thread_;
  std::atomic<bool> stop_{false};
  std::atomic<size_t> inflight_{0};
  std::mutex mu_;
  std::deque<std::unique_ptr<Op>> incoming_;
};

}  // namespace client
}  // namespace tca
