// Native gRPC serving front: accepts HTTP/2 connections, reassembles gRPC
// messages and hands them to the application; responses may be sent from any
// thread at any later time (the GPU executor completes requests
// asynchronously, so no thread ever blocks on a request).
//
// This is the wire-facing half of the in-repo v2 test server: the model
// executor stays in Python (PyTorch-ROCm), the protocol work (HTTP/2 framing,
// HPACK, flow control, gRPC message framing) runs here on epoll reactor
// threads.  The reference has no server; its clients are exercised against
// this one (`src/python/library/tritonclient/grpc/_client.py:298-1790` is the
// API surface it has to satisfy).
#pragma once

#include <atomic>
#include <cstdint>
#include <functional>
#include <memory>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "tca/error.h"
#include "tca/tls.h"

namespace tca {
namespace server {

class GrpcServerFront {
 public:
  struct Handlers {
    // A unary RPC's request is complete.  `method` is the last path component
    // ("ModelInfer").  Answer with Reply()/Finish(), from any thread, any time.
    std::function<void(uint64_t call, const std::string& method, std::string&& request)> on_unary;
    // One message of a client-streaming RPC.
    std::function<void(uint64_t call, const std::string& method, std::string&& request)> on_stream_message;
    // The client half-closed a streaming RPC.
    std::function<void(uint64_t call)> on_stream_end;
    // The call went away (reset by the client or connection lost) before Finish().
    std::function<void(uint64_t call)> on_cancel;
  };

  GrpcServerFront(std::string host, int port, int threads, std::set<std::string> streaming_methods,
                  Handlers handlers);
  ~GrpcServerFront();

  // Serve TLS (ALPN "h2") instead of cleartext HTTP/2; call before Start().
  // `context` must be a server context (TlsOptions::server).
  void SetTls(std::shared_ptr<net::TlsContext> context) { tls_ = std::move(context); }

  Error Start();
  void Stop();
  int port() const { return port_; }

  // Thread-safe.  Unknown / finished calls are ignored.
  void Send(uint64_t call, std::string&& message);  // one response message
  void Finish(uint64_t call, int grpc_status, const std::string& grpc_message = std::string());
  void Reply(uint64_t call, std::string&& message);  // Send + Finish(OK) in one step (unary success)

  // Counters for tests / introspection.
  uint64_t ConnectionsAccepted() const { return accepted_.load(); }
  uint64_t CallsStarted() const { return calls_.load(); }

 private:
  struct Reactor;
  std::string host_;
  int port_;
  int nthreads_;
  std::set<std::string> streaming_;
  Handlers handlers_;
  std::shared_ptr<net::TlsContext> tls_;
  std::vector<std::unique_ptr<Reactor>> reactors_;
  std::atomic<bool> running_{false};
  std::atomic<uint64_t> accepted_{0};
  std::atomic<uint64_t> calls_{0};
};

}  // namespace server
}  // namespace tca
