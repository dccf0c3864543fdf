// gRPC-over-HTTP/2 channel for the native client: one TCP connection, one
// reactor thread (epoll) that owns the HTTP/2 state machine, any number of
// concurrent calls (unary and bidirectional streaming) multiplexed on it.
//
// The reference rides on grpc++ (`src/c++/library/grpc_client.cc:80-152`:
// channel + stub cache); this toolchain has no grpc++, so the transport is
// implemented here on tca/h2.h.  Calls are started from any thread; events are
// delivered on the reactor thread.
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

#include "tca/common.h"
#include "tca/error.h"
#include "tca/h2.h"
#include "tca/tls.h"

namespace tca {
namespace client {

// gRPC status codes used by this library.
enum GrpcCode : int {
  kGrpcOk = 0, kGrpcCancelled = 1, kGrpcUnknown = 2, kGrpcInvalidArgument = 3, kGrpcDeadlineExceeded = 4,
  kGrpcNotFound = 5, kGrpcResourceExhausted = 8, kGrpcUnimplemented = 12, kGrpcInternal = 13, kGrpcUnavailable = 14,
};

struct KeepAliveOptions {
  // Parity with the reference's struct (`grpc_client.h:77-97`).  An HTTP/2 PING
  // is sent after `keepalive_time_ms` without traffic; the connection is
  // dropped when the ack does not arrive within `keepalive_timeout_ms`.
  int keepalive_time_ms = INT32_MAX;
  int keepalive_timeout_ms = 20000;
  bool keepalive_permit_without_calls = false;
  int http2_max_pings_without_data = 2;
};

// Per-channel settings beyond keep-alive: what code written against grpc++ passes as
// grpc::ChannelArguments (reference `grpc_client.h:146-150`, `grpc_client.cc:475-498`).  Fields can be
// set directly or through the gRPC argument names ("grpc.max_send_message_length", ...).
struct ChannelOptions {
  KeepAliveOptions keepalive;
  // largest request / response message in bytes (the reference's default is INT32_MAX for both)
  int64_t max_send_message_bytes = INT32_MAX;
  int64_t max_receive_message_bytes = INT32_MAX;
  // goes in front of the library's own user-agent token
  std::string primary_user_agent;
  // HTTP/2 receive windows and frame size advertised to the server; 0 = the library's defaults
  uint32_t http2_stream_window_bytes = 0;
  uint32_t http2_connection_window_bytes = 0;
  uint32_t http2_max_frame_bytes = 0;

  // By gRPC argument name.  Unknown names are an error (the reference passes them on unchecked, "at the
  // user's own risk"; here a typo is reported instead of silently doing nothing).
  Error SetInt(const std::string& grpc_arg, int64_t value);
  Error SetString(const std::string& grpc_arg, const std::string& value);
  // Everything that distinguishes two channels to the same URL (the channel cache shares only equal ones).
  std::string CacheKey() const;
};

class GrpcChannel {
 public:
  struct CallEvents {
    // One complete response message (without the 5-byte gRPC prefix).
    std::function<void(std::string&& message)> on_message;
    // Exactly once per call: gRPC status code and message.
    std::function<void(int status, const std::string& message)> on_done;
  };

  // `tls` non-null: the connection is TLS with ALPN "h2" (the context must offer it).
  static Error Create(std::shared_ptr<GrpcChannel>* channel, const std::string& url, bool verbose,
                      const KeepAliveOptions& keepalive = KeepAliveOptions(),
                      std::shared_ptr<net::TlsContext> tls = nullptr);
  static Error Create(std::shared_ptr<GrpcChannel>* channel, const std::string& url, bool verbose,
                      const ChannelOptions& options, std::shared_ptr<net::TlsContext> tls = nullptr);
  const ChannelOptions& Options() const { return options_; }
  ~GrpcChannel();

  // Starts a call on "/inference.GRPCInferenceService/<method>".  Never fails
  // synchronously: connection problems surface through on_done.
  uint64_t StartCall(const std::string& method, const Headers& metadata, uint64_t timeout_us, CallEvents events);
  // Sends one request message; `last` half-closes the call.  `compressed` marks
  // a payload the caller compressed with the algorithm it announced in the
  // call's "grpc-encoding" metadata.
  void SendMessage(uint64_t call, std::string&& serialized, bool last, bool compressed = false);
  void CloseSend(uint64_t call);
  void Cancel(uint64_t call);

  // Bytes accepted by SendMessage that have not reached the socket yet.
  size_t PendingBytes() const { return pending_bytes_.load(); }
  const std::string& Url() const { return url_; }

 private:
  GrpcChannel(std::string url, std::string host, int port, bool verbose, const ChannelOptions& options,
              std::shared_ptr<net::TlsContext> tls);
  struct Call {
    uint32_t stream = 0;
    CallEvents events;
    std::string method;
    Headers metadata;
    uint64_t deadline_ns = 0;
    uint64_t timeout_us = 0;
    std::string rbuf;  // partial gRPC message
    bool got_status = false;
    int http_status = 0;
    int grpc_status = -1;
    std::string grpc_message;
    bool send_closed = false;
    std::deque<std::pair<std::string, bool>> unsent;  // queued before the stream opened
  };
  void Loop();
  void Post(std::function<void()> fn);
  bool EnsureConnected(std::string* why);
  void Disconnect(const std::string& why);
  void OpenCallStream(uint64_t id, Call* call);
  void Finish(uint64_t id, int status, std::string message);  // by value: callers pass fields of the call being erased
  void OnHeaders(uint32_t stream, h2::HeaderList&& headers, bool end_stream);
  void OnData(uint32_t stream, const uint8_t* data, size_t len, bool end_stream);
  void WriteSocket(size_t budget = static_cast<size_t>(-1));
  bool write_blocked_ = false;  // the last send() found the socket full: EPOLLOUT will tell
  bool ReadSocket();

  const std::string url_;
  const std::string host_;
  const int port_;
  const bool verbose_;
  const ChannelOptions options_;
  const KeepAliveOptions keepalive_;

  int fd_ = -1;
  std::shared_ptr<net::TlsContext> tls_context_;
  std::unique_ptr<net::TlsSession> tls_;
  int epfd_ = -1;
  int wake_fd_ = -1;
  std::unique_ptr<h2::Connection> conn_;
  std::map<uint64_t, Call> calls_;
  std::map<uint32_t, uint64_t> by_stream_;
  uint64_t last_activity_ns_ = 0;
  uint64_t ping_sent_ns_ = 0;

  std::thread thread_;
  std::atomic<bool> stop_{false};
  std::atomic<uint64_t> next_call_{1};
  std::atomic<size_t> pending_bytes_{0};
  std::mutex mu_;
  std::deque<std::function<void()>> commands_;
};

}  // namespace client
}  // namespace tca
