// gRPC channel on the in-tree HTTP/2 engine (see tca/grpc_channel.h).
#include "tca/grpc_channel.h"

#include "tca/grpc_compress.h"

#include <arpa/inet.h>
#include <fcntl.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/epoll.h>
#include <sys/eventfd.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cctype>
#include <cerrno>
#include <chrono>
#include <cstring>
#include <iostream>
#include <vector>

namespace tca {
namespace client {
namespace {

uint64_t NowNs() {
  return static_cast<uint64_t>(
      std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count());
}

// grpc-message is percent-encoded (gRPC over HTTP/2 spec, "Status-Message").
std::string PercentDecode(const std::string& s) {
  std::string out;
  out.reserve(s.size());
  for (size_t i = 0; i < s.size(); ++i) {
    if (s[i] == '%' && i + 2 < s.size() && std::isxdigit(static_cast<unsigned char>(s[i + 1])) &&
        std::isxdigit(static_cast<unsigned char>(s[i + 2]))) {
      out.push_back(static_cast<char>(std::stoi(s.substr(i + 1, 2), nullptr, 16)));
      i += 2;
    } else {
      out.push_back(s[i]);
    }
  }
  return out;
}

std::string GrpcTimeout(uint64_t timeout_us) {
  // at most 8 digits + unit
  if (timeout_us < 100000000ull) return std::to_string(timeout_us) + "u";
  const uint64_t ms = timeout_us / 1000;
  if (ms < 100000000ull) return std::to_string(ms) + "m";
  return std::to_string(ms / 1000) + "S";
}

int HttpToGrpc(int http_status) {
  switch (http_status) {
    case 400: return kGrpcInternal;
    case 401: return 16;  // UNAUTHENTICATED
    case 403: return 7;   // PERMISSION_DENIED
    case 404: return kGrpcUnimplemented;
    case 429: case 502: case 503: case 504: return kGrpcUnavailable;
    default: return kGrpcUnknown;
  }
}

}  // namespace

namespace {
// what h2::Connection advertises unless told otherwise
constexpr uint32_t kDefaultStreamWindow = 8u << 20, kDefaultConnectionWindow = 64u << 20, kDefaultMaxFrame = 1u << 20;
}  // namespace

Error ChannelOptions::SetInt(const std::string& arg, int64_t value) {
  if (arg == "grpc.max_send_message_length") max_send_message_bytes = value < 0 ? INT64_MAX : value;
  else if (arg == "grpc.max_receive_message_length") max_receive_message_bytes = value < 0 ? INT64_MAX : value;
  else if (arg == "grpc.keepalive_time_ms") keepalive.keepalive_time_ms = static_cast<int>(value);
  else if (arg == "grpc.keepalive_timeout_ms") keepalive.keepalive_timeout_ms = static_cast<int>(value);
  else if (arg == "grpc.keepalive_permit_without_calls") keepalive.keepalive_permit_without_calls = value != 0;
  else if (arg == "grpc.http2.max_pings_without_data") keepalive.http2_max_pings_without_data = static_cast<int>(value);
  else if (arg == "grpc.http2.lookahead_bytes" || arg == "grpc.http2.stream_window_bytes")
    http2_stream_window_bytes = static_cast<uint32_t>(value);
  else if (arg == "grpc.http2.connection_window_bytes") http2_connection_window_bytes = static_cast<uint32_t>(value);
  else if (arg == "grpc.http2.max_frame_size") http2_max_frame_bytes = static_cast<uint32_t>(value);
  else if (arg == "triton_client_channel_idx") (void)value;  // the reference's trick against grpc's own channel sharing: nothing to do
  else return Error("unsupported integer channel argument '" + arg + "'");
  return Error::Success();
}

Error ChannelOptions::SetString(const std::string& arg, const std::string& value) {
  if (arg == "grpc.primary_user_agent") primary_user_agent = value;
  else return Error("unsupported string channel argument '" + arg + "'");
  return Error::Success();
}

std::string ChannelOptions::CacheKey() const {
  return std::to_string(keepalive.keepalive_time_ms) + "," + std::to_string(keepalive.keepalive_timeout_ms) + "," +
         std::to_string(keepalive.keepalive_permit_without_calls) + "," +
         std::to_string(keepalive.http2_max_pings_without_data) + "," + std::to_string(max_send_message_bytes) + "," +
         std::to_string(max_receive_message_bytes) + "," + primary_user_agent + "," +
         std::to_string(http2_stream_window_bytes) + "," + std::to_string(http2_connection_window_bytes) + "," +
         std::to_string(http2_max_frame_bytes);
}

Error GrpcChannel::Create(std::shared_ptr<GrpcChannel>* channel, const std::string& url, bool verbose,
                          const KeepAliveOptions& keepalive, std::shared_ptr<net::TlsContext> tls) {
  ChannelOptions options;
  options.keepalive = keepalive;
  return Create(channel, url, verbose, options, std::move(tls));
}

Error GrpcChannel::Create(std::shared_ptr<GrpcChannel>* channel, const std::string& url, bool verbose,
                          const ChannelOptions& options, std::shared_ptr<net::TlsContext> tls) {
  if (options.http2_max_frame_bytes != 0 && (options.http2_max_frame_bytes < 16384 || options.http2_max_frame_bytes > 16777215))
    return Error("http2_max_frame_bytes must be within [16384, 16777215]");
  if ((options.http2_stream_window_bytes | options.http2_connection_window_bytes) > 0x7FFFFFFFu)
    return Error("HTTP/2 windows are at most 2^31 - 1 bytes");
  std::string u = url;
  const size_t scheme = u.find("://");
  if (scheme != std::string::npos) u = u.substr(scheme + 3);
  const size_t colon = u.rfind(':');
  std::string host = colon == std::string::npos ? u : u.substr(0, colon);
  int port = 8001;
  if (colon != std::string::npos) {
    try {
      port = std::stoi(u.substr(colon + 1));
    } catch (...) {
      return Error("invalid port in gRPC url '" + url + "'");
    }
  }
  if (host.empty()) return Error("invalid gRPC url '" + url + "'");
  channel->reset(new GrpcChannel(url, host, port, verbose, options, std::move(tls)));
  return Error::Success();
}

GrpcChannel::GrpcChannel(std::string url, std::string host, int port, bool verbose, const ChannelOptions& options,
                         std::shared_ptr<net::TlsContext> tls)
    : url_(std::move(url)), host_(std::move(host)), port_(port), verbose_(verbose), options_(options),
      keepalive_(options.keepalive),
      tls_context_(std::move(tls)) {
  epfd_ = epoll_create1(EPOLL_CLOEXEC);
  wake_fd_ = eventfd(0, EFD_NONBLOCK | EFD_CLOEXEC);
  epoll_event ev{};
  ev.events = EPOLLIN;
  ev.data.fd = wake_fd_;
  epoll_ctl(epfd_, EPOLL_CTL_ADD, wake_fd_, &ev);
  thread_ = std::thread([this] { Loop(); });
}

GrpcChannel::~GrpcChannel() {
  stop_.store(true);
  const uint64_t one = 1;
  if (write(wake_fd_, &one, sizeof(one)) < 0) { /* the loop also polls stop_ on its timeout */ }
  if (thread_.joinable()) thread_.join();
  tls_.reset();
  if (fd_ >= 0) close(fd_);
  close(wake_fd_);
  close(epfd_);
}

void GrpcChannel::Post(std::function<void()> fn) {
  {
    std::lock_guard<std::mutex> lk(mu_);
    commands_.push_back(std::move(fn));
  }
  const uint64_t one = 1;
  if (write(wake_fd_, &one, sizeof(one)) < 0) { /* counter saturated: the loop is already awake */ }
}

bool GrpcChannel::EnsureConnected(std::string* why) {
  if (fd_ >= 0 && conn_ && !conn_->GoawayReceived() && !conn_->GoawaySent()) return true;
  if (fd_ >= 0) {
    // draining connection: calls in flight keep it; new calls need a fresh one,
    // which this single-connection channel only opens once the old one is idle
    if (!calls_.empty() && conn_ && conn_->OpenStreams() > 0) {
      *why = "connection is shutting down (GOAWAY)";
      return false;
    }
    Disconnect("reconnect");
  }
  addrinfo hints{};
  hints.ai_family = AF_UNSPEC;
  hints.ai_socktype = SOCK_STREAM;
  addrinfo* res = nullptr;
  const int gai = getaddrinfo(host_.c_str(), std::to_string(port_).c_str(), &hints, &res);
  if (gai != 0) {
    *why = "failed to resolve '" + host_ + "': " + gai_strerror(gai);
    return false;
  }
  int fd = -1;
  std::string last = "no address";
  for (addrinfo* ai = res; ai != nullptr; ai = ai->ai_next) {
    fd = socket(ai->ai_family, ai->ai_socktype | SOCK_NONBLOCK | SOCK_CLOEXEC, ai->ai_protocol);
    if (fd < 0) continue;
    int rc = connect(fd, ai->ai_addr, ai->ai_addrlen);
    if (rc < 0 && errno == EINPROGRESS) {
      pollfd pfd{fd, POLLOUT, 0};
      rc = poll(&pfd, 1, 5000);
      int err = 0;
      socklen_t len = sizeof(err);
      if (rc == 1 && getsockopt(fd, SOL_SOCKET, SO_ERROR, &err, &len) == 0 && err == 0) {
        rc = 0;
      } else {
        last = rc == 0 ? "connect timed out" : std::strerror(err ? err : errno);
        rc = -1;
      }
    } else if (rc < 0) {
      last = std::strerror(errno);
    }
    if (rc == 0) break;
    close(fd);
    fd = -1;
  }
  freeaddrinfo(res);
  if (fd < 0) {
    *why = "failed to connect to " + host_ + ":" + std::to_string(port_) + ": " + last;
    return false;
  }
  const int one = 1;
  setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
  const int buf = 4 << 20;
  setsockopt(fd, SOL_SOCKET, SO_SNDBUF, &buf, sizeof(buf));
  setsockopt(fd, SOL_SOCKET, SO_RCVBUF, &buf, sizeof(buf));
  if (tls_context_) {
    // the connect above already waits on this thread; the handshake does the same
    std::unique_ptr<net::TlsSession> session(new net::TlsSession(tls_context_, host_, fd));
    Error e = net::HandshakeBlocking(session.get(), fd, 5000);
    if (e.IsOk() && session->Alpn() != "h2") e = Error("the server did not agree to HTTP/2 (ALPN '" + session->Alpn() + "')");
    if (!e.IsOk()) {
      session.reset();
      close(fd);
      *why = "failed to connect to " + host_ + ":" + std::to_string(port_) + ": " + e.Message();
      return false;
    }
    tls_ = std::move(session);
  }
  fd_ = fd;
  h2::Connection::Callbacks cb;
  cb.on_headers = [this](uint32_t s, h2::HeaderList&& h, bool end) { OnHeaders(s, std::move(h), end); };
  cb.on_data = [this](uint32_t s, const uint8_t* d, size_t n, bool end) { OnData(s, d, n, end); };
  cb.on_reset = [this](uint32_t s, uint32_t code) {
    auto it = by_stream_.find(s);
    if (it == by_stream_.end()) return;
    const uint64_t id = it->second;
    Finish(id, code == h2::kRefusedStream ? kGrpcUnavailable : (code == h2::kCancel ? kGrpcCancelled : kGrpcInternal),
           "stream reset by the server (HTTP/2 error " + std::to_string(code) + ")");
  };
  cb.on_goaway = [this](uint32_t last_stream, uint32_t) {
    // streams the server never saw are safe to report as UNAVAILABLE
    std::vector<uint64_t> dead;
    for (const auto& kv : by_stream_)
      if (kv.first > last_stream) dead.push_back(kv.second);
    for (uint64_t id : dead) Finish(id, kGrpcUnavailable, "server is going away");
  };
  conn_.reset(new h2::Connection(h2::Connection::Role::CLIENT, std::move(cb)));
  if (options_.http2_stream_window_bytes || options_.http2_connection_window_bytes || options_.http2_max_frame_bytes)
    conn_->SetLocalWindow(options_.http2_stream_window_bytes ? options_.http2_stream_window_bytes : kDefaultStreamWindow,
                          options_.http2_connection_window_bytes ? options_.http2_connection_window_bytes
                                                                 : kDefaultConnectionWindow,
                          options_.http2_max_frame_bytes ? options_.http2_max_frame_bytes : kDefaultMaxFrame);
  conn_->Start();
  epoll_event ev{};
  ev.events = EPOLLIN | EPOLLOUT | EPOLLET | EPOLLRDHUP;
  ev.data.fd = fd_;
  epoll_ctl(epfd_, EPOLL_CTL_ADD, fd_, &ev);
  last_activity_ns_ = NowNs();
  ping_sent_ns_ = 0;
  if (verbose_) std::cerr << "grpc: connected to " << host_ << ":" << port_ << std::endl;
  return true;
}

void GrpcChannel::Disconnect(const std::string& why) {
  tls_.reset();
  if (fd_ >= 0) {
    epoll_ctl(epfd_, EPOLL_CTL_DEL, fd_, nullptr);
    close(fd_);
    fd_ = -1;
  }
  conn_.reset();
  std::vector<uint64_t> ids;
  for (const auto& kv : calls_)
    if (kv.second.stream != 0) ids.push_back(kv.first);
  for (uint64_t id : ids) Finish(id, kGrpcUnavailable, why);
  by_stream_.clear();
}

void GrpcChannel::Finish(uint64_t id, int status, std::string message) {
  auto it = calls_.find(id);
  if (it == calls_.end()) return;
  Call call = std::move(it->second);
  calls_.erase(it);
  for (const auto& m : call.unsent) pending_bytes_.fetch_sub(m.first.size());
  if (call.stream) by_stream_.erase(call.stream);
  if (call.events.on_done) call.events.on_done(status, message);
}

uint64_t GrpcChannel::StartCall(const std::string& method, const Headers& metadata, uint64_t timeout_us,
                                CallEvents events) {
  const uint64_t id = next_call_.fetch_add(1);
  auto shared = std::make_shared<CallEvents>(std::move(events));
  Post([this, id, method, metadata, timeout_us, shared] {
    Call& call = calls_[id];
    call.events = std::move(*shared);
    call.method = method;
    call.metadata = metadata;
    call.timeout_us = timeout_us;
    if (timeout_us) call.deadline_ns = NowNs() + timeout_us * 1000;
    OpenCallStream(id, &call);
  });
  return id;
}

void GrpcChannel::OpenCallStream(uint64_t id, Call* call) {
  std::string why;
  if (!EnsureConnected(&why)) {
    Finish(id, kGrpcUnavailable, why);
    return;
  }
  h2::HeaderList h;
  h.push_back({":method", "POST"});
  h.push_back({":scheme", tls_context_ ? "https" : "http"});
  h.push_back({":path", "/inference.GRPCInferenceService/" + call->method});
  h.push_back({":authority", host_ + ":" + std::to_string(port_)});
  h.push_back({"te", "trailers"});
  h.push_back({"content-type", "application/grpc"});
  h.push_back({"user-agent", options_.primary_user_agent.empty() ? std::string("tca-grpc-cpp/1.0")
                                                                 : options_.primary_user_agent + " tca-grpc-cpp/1.0"});
  h.push_back({"grpc-accept-encoding", "identity,deflate,gzip"});
  if (call->timeout_us) h.push_back({"grpc-timeout", GrpcTimeout(call->timeout_us)});
  for (const auto& kv : call->metadata) {
    std::string name = kv.first;
    for (auto& c : name) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
    h.push_back({name, kv.second});
  }
  call->stream = conn_->OpenStream(h, false);
  if (call->stream == 0) {
    Finish(id, kGrpcUnavailable, "could not open an HTTP/2 stream");
    return;
  }
  by_stream_[call->stream] = id;
}

void GrpcChannel::SendMessage(uint64_t call_id, std::string&& serialized, bool last, bool compressed) {
  // The 5-byte message prefix travels as a DATA frame of its own in front of the message: a message may be
  // split over frames any way, and this way a request of megabytes is not copied to put 5 bytes in front.
  const uint32_t n = static_cast<uint32_t>(serialized.size());
  const std::string prefix = {static_cast<char>(compressed ? 1 : 0), static_cast<char>(n >> 24), static_cast<char>(n >> 16),
                              static_cast<char>(n >> 8), static_cast<char>(n)};
  pending_bytes_.fetch_add(serialized.size() + 5);
  auto shared = std::make_shared<std::string>(std::move(serialized));
  Post([this, call_id, shared, last, prefix] {
    pending_bytes_.fetch_sub(shared->size() + 5);
    auto it = calls_.find(call_id);
    if (it == calls_.end() || it->second.send_closed || !conn_) return;
    if (last) it->second.send_closed = true;
    const uint32_t stream = it->second.stream;
    Error e = conn_->SendData(stream, std::string(prefix), false);
    if (e.IsOk()) e = conn_->SendData(stream, std::move(*shared), last);
    if (!e.IsOk()) Finish(call_id, kGrpcInternal, e.Message());
  });
}

void GrpcChannel::CloseSend(uint64_t call_id) {
  Post([this, call_id] {
    auto it = calls_.find(call_id);
    if (it == calls_.end() || it->second.send_closed || !conn_) return;
    it->second.send_closed = true;
    conn_->SendData(it->second.stream, std::string(), true);
  });
}

void GrpcChannel::Cancel(uint64_t call_id) {
  Post([this, call_id] {
    auto it = calls_.find(call_id);
    if (it == calls_.end()) return;
    if (conn_ && it->second.stream) conn_->ResetStream(it->second.stream, h2::kCancel);
    Finish(call_id, kGrpcCancelled, "Locally cancelled by application!");
  });
}

void GrpcChannel::OnHeaders(uint32_t stream, h2::HeaderList&& headers, bool end_stream) {
  auto it = by_stream_.find(stream);
  if (it == by_stream_.end()) return;
  const uint64_t id = it->second;
  Call& call = calls_[id];
  for (const auto& h : headers) {
    if (verbose_) std::cerr << "grpc: stream " << stream << " header " << h.name << ": " << h.value << std::endl;
    if (h.name == ":status") call.http_status = std::atoi(h.value.c_str());
    else if (h.name == "grpc-status") { call.grpc_status = std::atoi(h.value.c_str()); call.got_status = true; }
    else if (h.name == "grpc-message") call.grpc_message = PercentDecode(h.value);
    else if (h.name == "grpc-encoding" && h.value != "identity" && h.value != "gzip" && h.value != "deflate") {
      conn_->ResetStream(stream, h2::kCancel);
      Finish(id, kGrpcUnimplemented, "response compression '" + h.value + "' is not supported by this client");
      return;
    }
  }
  if (end_stream) {
    if (call.got_status) {
      Finish(id, call.grpc_status, call.grpc_message);
    } else if (call.http_status != 200 && call.http_status != 0) {
      Finish(id, HttpToGrpc(call.http_status), "unexpected HTTP status " + std::to_string(call.http_status));
    } else {
      Finish(id, kGrpcUnknown, "stream closed without grpc-status");
    }
  }
}

void GrpcChannel::OnData(uint32_t stream, const uint8_t* data, size_t len, bool end_stream) {
  auto it = by_stream_.find(stream);
  if (it == by_stream_.end()) return;
  const uint64_t id = it->second;
  Call& call = calls_[id];
  call.rbuf.append(reinterpret_cast<const char*>(data), len);
  size_t off = 0;
  while (call.rbuf.size() - off >= 5) {
    const uint8_t* p = reinterpret_cast<const uint8_t*>(call.rbuf.data()) + off;
    const size_t n = (static_cast<size_t>(p[1]) << 24) | (static_cast<size_t>(p[2]) << 16) |
                     (static_cast<size_t>(p[3]) << 8) | p[4];
    if (static_cast<int64_t>(n) > options_.max_receive_message_bytes) {
      conn_->ResetStream(stream, h2::kCancel);
      Finish(id, kGrpcResourceExhausted, "Received message larger than max (" + std::to_string(n) + " vs. " +
                                             std::to_string(options_.max_receive_message_bytes) + ")");
      return;
    }
    if (call.rbuf.size() - off - 5 < n) break;
    std::string message(call.rbuf, off + 5, n);
    off += 5 + n;
    if (p[0] != 0) {
      std::string plain;
      if (!grpcz::Decompress(message.data(), message.size(), &plain).IsOk()) {
        conn_->ResetStream(stream, h2::kCancel);
        Finish(id, kGrpcInternal, "corrupt compressed response message");
        return;
      }
      message.swap(plain);
    }
    if (call.events.on_message) {
      // the handler may cancel or finish the call: look it up again afterwards
      auto handler = call.events.on_message;
      handler(std::move(message));
      if (calls_.find(id) == calls_.end()) return;
    }
  }
  Call& again = calls_[id];
  again.rbuf.erase(0, off);
  if (end_stream) Finish(id, kGrpcUnknown, "stream closed without trailers");
}

void GrpcChannel::WriteSocket(size_t budget) {
  // `budget`: with tens of megabytes queued and a peer that keeps reading, this loop would not return for
  // many milliseconds - and the responses waiting in the socket (whose callbacks issue the next requests)
  // would wait with it.  The event loop comes straight back here after a look at the other direction.
  size_t written = 0;
  write_blocked_ = false;
  while (conn_ && conn_->WantsWrite() && fd_ >= 0 && written < budget) {
    const std::string& out = conn_->Output();
    const size_t off = conn_->OutputOffset();
    ssize_t n;
    if (tls_) {
      n = tls_->Write(out.data() + off, out.size() - off);
      if (n == net::TlsSession::kWouldBlock) {  // the edge-triggered loop calls again on the next event
        write_blocked_ = true;
        return;
      }
      if (n < 0) {
        const std::string why = tls_->LastError();  // Disconnect() destroys the session that owns the text
        Disconnect(why);
        return;
      }
    } else {
      n = send(fd_, out.data() + off, out.size() - off, MSG_NOSIGNAL);
    }
    if (n > 0) {
      conn_->ConsumeOutput(static_cast<size_t>(n));
      written += static_cast<size_t>(n);
      last_activity_ns_ = NowNs();
    } else if (n < 0 && (errno == EAGAIN || errno == EWOULDBLOCK)) {
      write_blocked_ = true;
      return;  // EPOLLOUT (edge) fires when the socket drains
    } else if (n < 0 && errno == EINTR) {
      continue;
    } else {
      Disconnect(std::string("send failed: ") + std::strerror(errno));
      return;
    }
  }
}

bool GrpcChannel::ReadSocket() {
  static thread_local std::vector<uint8_t> buf(256 * 1024);
  while (fd_ >= 0) {
    ssize_t n;
    if (tls_) {
      n = tls_->Read(buf.data(), buf.size());
      if (n == net::TlsSession::kWouldBlock) return true;
      if (n < 0) {
        const std::string why = tls_->LastError();
        Disconnect(why);
        return false;
      }
    } else {
      n = recv(fd_, buf.data(), buf.size(), 0);
    }
    if (n > 0) {
      last_activity_ns_ = NowNs();
      const Error e = conn_->Receive(buf.data(), static_cast<size_t>(n));
      if (!e.IsOk()) {
        WriteSocket();
        Disconnect(e.Message());
        return false;
      }
      if (!conn_) return false;  // a callback tore the connection down
    } else if (n == 0) {
      Disconnect("Socket closed");
      return false;
    } else if (errno == EAGAIN || errno == EWOULDBLOCK) {
      return true;
    } else if (errno != EINTR) {
      Disconnect(std::string("recv failed: ") + std::strerror(errno));
      return false;
    }
  }
  return false;
}

void GrpcChannel::Loop() {
  epoll_event events[8];
  while (!stop_.load()) {
    // next wake-up: earliest call deadline / keep-alive tick
    int timeout_ms = 1000;
    const uint64_t now = NowNs();
    for (const auto& kv : calls_) {
      if (kv.second.deadline_ns == 0) continue;
      const int64_t left = static_cast<int64_t>(kv.second.deadline_ns) - static_cast<int64_t>(now);
      const int ms = left <= 0 ? 0 : static_cast<int>(left / 1000000) + 1;
      if (ms < timeout_ms) timeout_ms = ms;
    }
    if (conn_ && conn_->WantsWrite() && !write_blocked_) timeout_ms = 0;  // more to send: only a look at the socket
    const int n = epoll_wait(epfd_, events, 8, timeout_ms);
    for (int i = 0; i < n; ++i) {
      if (events[i].data.fd == wake_fd_) {
        uint64_t v;
        while (read(wake_fd_, &v, sizeof(v)) > 0) {}
      } else if (events[i].data.fd == fd_) {
        if (events[i].events & (EPOLLIN | EPOLLHUP | EPOLLRDHUP | EPOLLERR)) ReadSocket();
      }
    }
    std::deque<std::function<void()>> cmds;
    {
      std::lock_guard<std::mutex> lk(mu_);
      cmds.swap(commands_);
    }
    for (auto& fn : cmds) fn();
    // deadlines
    const uint64_t t = NowNs();
    std::vector<uint64_t> expired;
    for (const auto& kv : calls_)
      if (kv.second.deadline_ns && kv.second.deadline_ns <= t) expired.push_back(kv.first);
    for (uint64_t id : expired) {
      auto it = calls_.find(id);
      if (it == calls_.end()) continue;
      if (conn_ && it->second.stream) conn_->ResetStream(it->second.stream, h2::kCancel);
      Finish(id, kGrpcDeadlineExceeded, "Deadline Exceeded");
    }
    // keep-alive
    if (conn_ && keepalive_.keepalive_time_ms < INT32_MAX &&
        (keepalive_.keepalive_permit_without_calls || !calls_.empty())) {
      const uint64_t idle_ns = t - last_activity_ns_;
      if (ping_sent_ns_ == 0 && idle_ns > static_cast<uint64_t>(keepalive_.keepalive_time_ms) * 1000000ull) {
        conn_->SendPing();
        ping_sent_ns_ = t;
      } else if (ping_sent_ns_ != 0) {
        if (conn_->PingsAcked() >= conn_->PingsSent()) ping_sent_ns_ = 0;
        else if (t - ping_sent_ns_ > static_cast<uint64_t>(keepalive_.keepalive_timeout_ms) * 1000000ull)
          Disconnect("keepalive watchdog timeout");
      }
    }
    WriteSocket(4u << 20);
  }
  // shutting down: fail whatever is left
  std::vector<uint64_t> ids;
  for (const auto& kv : calls_) ids.push_back(kv.first);
  for (uint64_t id : ids) Finish(id, kGrpcCancelled, "channel closed");
}

}  // namespace client
}  // namespace tca
