// Native gRPC serving front (see tca/grpc_server.h).
#include "tca/grpc_server.h"

#include <arpa/inet.h>
#include <fcntl.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/epoll.h>
#include <sys/eventfd.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cerrno>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>

#include "tca/grpc_compress.h"
#include "tca/h2.h"

namespace tca {
namespace server {
namespace {

constexpr char kServicePrefix[] = "/inference.GRPCInferenceService/";

// call id layout: [reactor:8][connection:24][stream:32]
inline uint64_t MakeCall(uint32_t reactor, uint32_t conn, uint32_t stream) {
  return (static_cast<uint64_t>(reactor & 0xFF) << 56) | (static_cast<uint64_t>(conn & 0xFFFFFF) << 32) | stream;
}

std::string PercentEncode(const std::string& s) {
  static const char* hex = "0123456789ABCDEF";
  std::string out;
  for (unsigned char c : s) {
    if (c >= 0x20 && c <= 0x7E && c != '%') {
      out.push_back(static_cast<char>(c));
    } else {
      out.push_back('%');
      out.push_back(hex[c >> 4]);
      out.push_back(hex[c & 15]);
    }
  }
  return out;
}

std::string Framed(std::string&& message) {
  std::string out;
  out.reserve(message.size() + 5);
  out.push_back(0);
  const uint32_t n = static_cast<uint32_t>(message.size());
  const char len[4] = {static_cast<char>(n >> 24), static_cast<char>(n >> 16), static_cast<char>(n >> 8),
                       static_cast<char>(n)};
  out.append(len, 4);
  out.append(message);
  return out;
}

}  // namespace

struct GrpcServerFront::Reactor {
  struct StreamState {
    std::string method;
    bool streaming = false;
    bool valid = true;        // path and content-type accepted
    std::string rbuf;
    bool headers_sent = false;
    bool finished = false;
    bool dispatched = false;  // unary request handed to the application
    bool compressed = false;  // grpc-encoding: gzip | deflate
  };
  struct Conn {
    int fd = -1;
    uint32_t id = 0;
    std::unique_ptr<h2::Connection> h2;
    std::map<uint32_t, StreamState> streams;
    bool want_write = false;  // EPOLLOUT armed
    bool dead = false;
    std::unique_ptr<net::TlsSession> tls;  // null = cleartext
    bool handshaking = false;
  };
  struct Command {
    enum Kind { SEND, FINISH, REPLY } kind;
    uint64_t call;
    std::string payload;
    int status = 0;
  };

  GrpcServerFront* owner = nullptr;
  uint32_t index = 0;
  int listen_fd = -1;
  int epfd = -1;
  int wake_fd = -1;
  std::thread thread;
  std::mutex mu;
  std::deque<Command> commands;
  std::map<uint32_t, std::unique_ptr<Conn>> conns;
  uint32_t next_conn = 1;

  void Post(Command&& c) {
    {
      std::lock_guard<std::mutex> lk(mu);
      commands.push_back(std::move(c));
    }
    const uint64_t one = 1;
    if (write(wake_fd, &one, sizeof(one)) < 0) { /* already signalled */ }
  }

  void Accept() {
    for (;;) {
      const int fd = accept4(listen_fd, nullptr, nullptr, SOCK_NONBLOCK | SOCK_CLOEXEC);
      if (fd < 0) return;
      const int one = 1;
      setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
      const int buf = 4 << 20;
      setsockopt(fd, SOL_SOCKET, SO_SNDBUF, &buf, sizeof(buf));
      setsockopt(fd, SOL_SOCKET, SO_RCVBUF, &buf, sizeof(buf));
      std::unique_ptr<Conn> c(new Conn());
      c->fd = fd;
      c->id = next_conn++ & 0xFFFFFF;
      if (c->id == 0) c->id = next_conn++ & 0xFFFFFF;
      Conn* raw = c.get();
      h2::Connection::Callbacks cb;
      cb.on_headers = [this, raw](uint32_t s, h2::HeaderList&& h, bool end) { OnHeaders(raw, s, std::move(h), end); };
      cb.on_data = [this, raw](uint32_t s, const uint8_t* d, size_t n, bool end) { OnData(raw, s, d, n, end); };
      cb.on_reset = [this, raw](uint32_t s, uint32_t) { Cancelled(raw, s); };
      c->h2.reset(new h2::Connection(h2::Connection::Role::SERVER, std::move(cb)));
      c->h2->Start();
      epoll_event ev{};
      ev.events = EPOLLIN | EPOLLRDHUP;
      ev.data.u64 = (1ull << 32) | c->id;
      epoll_ctl(epfd, EPOLL_CTL_ADD, fd, &ev);
      if (owner->tls_) {
        c->tls.reset(new net::TlsSession(owner->tls_, std::string(), fd));
        c->handshaking = true;
      }
      conns[c->id] = std::move(c);
      owner->accepted_.fetch_add(1);
      if (raw->handshaking) Handshake(raw);
      else Write(raw);
    }
  }

  void Cancelled(Conn* c, uint32_t stream) {
    auto it = c->streams.find(stream);
    if (it == c->streams.end()) return;
    const bool notify = !it->second.finished && (it->second.dispatched || it->second.streaming) && it->second.valid;
    c->streams.erase(it);
    if (notify && owner->handlers_.on_cancel) owner->handlers_.on_cancel(MakeCall(index, c->id, stream));
  }

  void RejectStream(Conn* c, uint32_t stream, int status, const std::string& message) {
    c->h2->SendHeaders(stream, {{":status", "200"}, {"content-type", "application/grpc"},
                                {"grpc-status", std::to_string(status)}, {"grpc-message", PercentEncode(message)}},
                       true);
    c->h2->ResetStream(stream, h2::kNoError);
    c->streams.erase(stream);
  }

  void OnHeaders(Conn* c, uint32_t stream, h2::HeaderList&& headers, bool end_stream) {
    if (c->streams.count(stream)) {  // request trailers: nothing to take from them
      if (end_stream) OnData(c, stream, nullptr, 0, true);
      return;
    }
    StreamState st;
    std::string path, content_type, encoding;
    for (const auto& h : headers) {
      if (h.name == ":path") path = h.value;
      else if (h.name == "content-type") content_type = h.value;
      else if (h.name == "grpc-encoding") encoding = h.value;
    }
    c->streams[stream] = StreamState();
    StreamState& s = c->streams[stream];
    if (content_type.compare(0, 16, "application/grpc") != 0) {
      c->h2->SendHeaders(stream, {{":status", "415"}}, true);
      c->h2->ResetStream(stream, h2::kNoError);
      c->streams.erase(stream);
      return;
    }
    if (path.compare(0, sizeof(kServicePrefix) - 1, kServicePrefix) != 0) {
      RejectStream(c, stream, 12, "unknown service " + path);
      return;
    }
    if (!encoding.empty() && encoding != "identity" && encoding != "gzip" && encoding != "deflate") {
      RejectStream(c, stream, 12, "request compression '" + encoding + "' is not supported");
      return;
    }
    s.compressed = encoding == "gzip" || encoding == "deflate";
    s.method = path.substr(sizeof(kServicePrefix) - 1);
    s.streaming = owner->streaming_.count(s.method) != 0;
    owner->calls_.fetch_add(1);
    if (end_stream) OnData(c, stream, nullptr, 0, true);
  }

  void OnData(Conn* c, uint32_t stream, const uint8_t* data, size_t len, bool end_stream) {
    auto it = c->streams.find(stream);
    if (it == c->streams.end()) return;
    StreamState* s = &it->second;
    if (len) {
      if (s->rbuf.size() < 5 && s->rbuf.size() + len >= 5) {
        // the length prefix is complete: size the buffer for the whole message once instead of
        // growing (and copying) it through the doublings of a megabyte-sized tensor
        uint8_t prefix[5];
        const size_t have = s->rbuf.size();
        for (size_t i = 0; i < 5; ++i) prefix[i] = i < have ? static_cast<uint8_t>(s->rbuf[i]) : data[i - have];
        const size_t n = (static_cast<size_t>(prefix[1]) << 24) | (static_cast<size_t>(prefix[2]) << 16) |
                         (static_cast<size_t>(prefix[3]) << 8) | prefix[4];
        if (n <= (256u << 20)) s->rbuf.reserve(5 + n);
      }
      s->rbuf.append(reinterpret_cast<const char*>(data), len);
    }
    const uint64_t call = MakeCall(index, c->id, stream);
    size_t off = 0;
    while (s->rbuf.size() - off >= 5) {
      const uint8_t* p = reinterpret_cast<const uint8_t*>(s->rbuf.data()) + off;
      const bool packed = p[0] != 0;
      if (packed && !s->compressed) {
        RejectStream(c, stream, 13, "compressed message without a grpc-encoding header");
        return;
      }
      const size_t n = (static_cast<size_t>(p[1]) << 24) | (static_cast<size_t>(p[2]) << 16) |
                       (static_cast<size_t>(p[3]) << 8) | p[4];
      if (s->rbuf.size() - off - 5 < n) break;
      if (s->streaming) {
        std::string message(s->rbuf, off + 5, n);
        off += 5 + n;
        if (packed) {
          std::string plain;
          if (!grpcz::Decompress(message.data(), message.size(), &plain).IsOk()) {
            RejectStream(c, stream, 13, "corrupt compressed message");
            return;
          }
          message.swap(plain);
        }
        const std::string method = s->method;
        if (owner->handlers_.on_stream_message) owner->handlers_.on_stream_message(call, method, std::move(message));
        it = c->streams.find(stream);  // the handler may have finished the call
        if (it == c->streams.end()) return;
        s = &it->second;
      } else {
        if (!end_stream || s->rbuf.size() - off - 5 != n) break;  // unary: wait for the whole request
        std::string message;
        if (off == 0) {
          // take the receive buffer itself and drop the prefix in place: no second
          // allocation (page faults included) and copy of a large request
          message = std::move(s->rbuf);
          message.erase(0, 5);
        } else {
          message.assign(s->rbuf, off + 5, n);
        }
        s->rbuf.clear();
        off = 0;
        if (packed) {
          std::string plain;
          if (!grpcz::Decompress(message.data(), message.size(), &plain).IsOk()) {
            RejectStream(c, stream, 13, "corrupt compressed message");
            return;
          }
          message.swap(plain);
        }
        s->dispatched = true;
        const std::string method = s->method;
        if (owner->handlers_.on_unary) owner->handlers_.on_unary(call, method, std::move(message));
        return;
      }
    }
    if (off) s->rbuf.erase(0, off);
    if (end_stream) {
      if (s->streaming) {
        if (owner->handlers_.on_stream_end) owner->handlers_.on_stream_end(call);
      } else if (!s->dispatched) {
        if (s->rbuf.empty()) {  // a unary call without a message body: treat as an empty message
          s->dispatched = true;
          const std::string method = s->method;
          if (owner->handlers_.on_unary) owner->handlers_.on_unary(call, method, std::string());
        } else {
          RejectStream(c, stream, 13, "truncated gRPC message");
        }
      }
    }
  }

  void Apply(Command& cmd) {
    const uint32_t conn_id = static_cast<uint32_t>((cmd.call >> 32) & 0xFFFFFF);
    const uint32_t stream = static_cast<uint32_t>(cmd.call & 0xFFFFFFFF);
    auto cit = conns.find(conn_id);
    if (cit == conns.end()) return;
    Conn* c = cit->second.get();
    auto sit = c->streams.find(stream);
    if (sit == c->streams.end() || sit->second.finished) return;
    StreamState& s = sit->second;
    if (cmd.kind == Command::SEND || cmd.kind == Command::REPLY) {
      if (!s.headers_sent) {
        c->h2->SendHeaders(stream, {{":status", "200"}, {"content-type", "application/grpc"},
                                    {"grpc-accept-encoding", "identity"}}, false);
        s.headers_sent = true;
      }
      c->h2->SendData(stream, Framed(std::move(cmd.payload)), false);
    }
    if (cmd.kind == Command::FINISH || cmd.kind == Command::REPLY) {
      h2::HeaderList trailers;
      if (!s.headers_sent) {
        trailers.push_back({":status", "200"});
        trailers.push_back({"content-type", "application/grpc"});
      }
      trailers.push_back({"grpc-status", std::to_string(cmd.status)});
      if (cmd.kind == Command::FINISH && !cmd.payload.empty())
        trailers.push_back({"grpc-message", PercentEncode(cmd.payload)});
      c->h2->SendHeaders(stream, trailers, true);
      s.finished = true;
      // no RST_STREAM here: the response may still be queued behind flow control;
      // a client that has not half-closed yet does so when it sees the trailers
      c->streams.erase(sit);
    }
    Write(c);
  }

  void ArmWrite(Conn* c, bool on) {
    if (c->want_write == on) return;
    epoll_event ev{};
    ev.events = EPOLLIN | EPOLLRDHUP;
    if (on) ev.events |= EPOLLOUT;
    ev.data.u64 = (1ull << 32) | c->id;
    epoll_ctl(epfd, EPOLL_CTL_MOD, c->fd, &ev);
    c->want_write = on;
  }

  // false while the TLS handshake is still in progress (or failed: c->dead)
  bool Handshake(Conn* c) {
    const int rc = c->tls->Handshake();
    if (rc == net::TlsSession::kWouldBlock) {
      ArmWrite(c, c->tls->WantsWrite());
      return false;
    }
    if (rc != 1 || c->tls->Alpn() != "h2") {
      c->dead = true;
      return false;
    }
    c->handshaking = false;
    ArmWrite(c, false);
    Write(c);  // the server SETTINGS queued at accept time
    return !c->dead;
  }

  void Write(Conn* c) {
    if (c->handshaking) return;
    while (c->h2->WantsWrite()) {
      const std::string& out = c->h2->Output();
      const size_t off = c->h2->OutputOffset();
      ssize_t n;
      if (c->tls) {
        n = c->tls->Write(out.data() + off, out.size() - off);
        if (n == net::TlsSession::kWouldBlock) {
          // waiting for the socket to drain, or (renegotiation) for the peer: the read path calls Write again
          ArmWrite(c, c->tls->WantsWrite());
          return;
        }
        if (n < 0) {
          c->dead = true;
          return;
        }
      } else {
        n = send(c->fd, out.data() + off, out.size() - off, MSG_NOSIGNAL);
      }
      if (n > 0) {
        c->h2->ConsumeOutput(static_cast<size_t>(n));
      } else if (n < 0 && (errno == EAGAIN || errno == EWOULDBLOCK)) {
        ArmWrite(c, true);
        return;
      } else if (n < 0 && errno == EINTR) {
        continue;
      } else {
        c->dead = true;
        return;
      }
    }
    ArmWrite(c, false);
  }

  void Read(Conn* c) {
    static thread_local std::vector<uint8_t> buf(512 * 1024);
    if (c->handshaking && !Handshake(c)) return;
    // a TLS read returns one record (<= 16 KiB) at a time: allow more rounds per wake-up
    const int max_rounds = c->tls ? 64 : 16;
    for (int rounds = 0; rounds < max_rounds && !c->dead; ++rounds) {
      ssize_t n;
      if (c->tls) {
        n = c->tls->Read(buf.data(), buf.size());
        if (n == net::TlsSession::kWouldBlock) break;
        if (n < 0) {
          c->dead = true;
          break;
        }
      } else {
        n = recv(c->fd, buf.data(), buf.size(), 0);
      }
      if (n > 0) {
        const Error e = c->h2->Receive(buf.data(), static_cast<size_t>(n));
        if (!e.IsOk()) {
          Write(c);  // flush the GOAWAY
          c->dead = true;
          return;
        }
        if (!c->tls && static_cast<size_t>(n) < buf.size()) break;
      } else if (n == 0) {
        c->dead = true;
      } else if (errno == EAGAIN || errno == EWOULDBLOCK) {
        break;
      } else if (errno != EINTR) {
        c->dead = true;
      }
    }
    if (!c->dead) Write(c);
  }

  void Close(Conn* c) {
    std::vector<uint32_t> open;
    for (const auto& kv : c->streams) open.push_back(kv.first);
    for (uint32_t s : open) Cancelled(c, s);
    epoll_ctl(epfd, EPOLL_CTL_DEL, c->fd, nullptr);
    c->tls.reset();
    close(c->fd);
    conns.erase(c->id);
  }

  void Loop() {
    epoll_event events[64];
    while (owner->running_.load()) {
      const int n = epoll_wait(epfd, events, 64, 500);
      for (int i = 0; i < n; ++i) {
        const uint64_t tag = events[i].data.u64;
        if (tag == 0) {
          Accept();
        } else if (tag == 1) {
          uint64_t v;
          while (read(wake_fd, &v, sizeof(v)) > 0) {}
        } else {
          auto it = conns.find(static_cast<uint32_t>(tag & 0xFFFFFF));
          if (it == conns.end()) continue;
          Conn* c = it->second.get();
          if (events[i].events & (EPOLLIN | EPOLLHUP | EPOLLRDHUP | EPOLLERR)) Read(c);
          if (!c->dead && (events[i].events & EPOLLOUT)) {
            if (c->handshaking) Handshake(c);
            else Write(c);
          }
          if (c->dead) Close(c);
        }
      }
      std::deque<Command> cmds;
      {
        std::lock_guard<std::mutex> lk(mu);
        cmds.swap(commands);
      }
      for (auto& cmd : cmds) Apply(cmd);
      std::vector<Conn*> dead;
      for (auto& kv : conns)
        if (kv.second->dead) dead.push_back(kv.second.get());
      for (Conn* c : dead) Close(c);
    }
    std::vector<Conn*> all;
    for (auto& kv : conns) all.push_back(kv.second.get());
    for (Conn* c : all) {
      c->h2->Goaway(h2::kNoError);
      Write(c);
      Close(c);
    }
  }
};

GrpcServerFront::GrpcServerFront(std::string host, int port, int threads, std::set<std::string> streaming_methods,
                                 Handlers handlers)
    : host_(std::move(host)), port_(port), nthreads_(threads < 1 ? 1 : (threads > 64 ? 64 : threads)),
      streaming_(std::move(streaming_methods)), handlers_(std::move(handlers)) {}

GrpcServerFront::~GrpcServerFront() { Stop(); }

Error GrpcServerFront::Start() {
  if (running_.load()) return Error("already started");
  sockaddr_in addr{};
  addr.sin_family = AF_INET;
  if (host_.empty() || host_ == "0.0.0.0") addr.sin_addr.s_addr = htonl(INADDR_ANY);
  else if (host_ == "localhost") addr.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
  else if (inet_pton(AF_INET, host_.c_str(), &addr.sin_addr) != 1) return Error("cannot parse listen address '" + host_ + "'");
  // one listening socket per reactor (SO_REUSEPORT): the kernel spreads new
  // connections over the reactors and no accept lock is needed
  for (int i = 0; i < nthreads_; ++i) {
    const int fd = socket(AF_INET, SOCK_STREAM | SOCK_NONBLOCK | SOCK_CLOEXEC, 0);
    if (fd < 0) return Error(std::string("socket: ") + std::strerror(errno));
    const int one = 1;
    setsockopt(fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    setsockopt(fd, SOL_SOCKET, SO_REUSEPORT, &one, sizeof(one));
    addr.sin_port = htons(static_cast<uint16_t>(port_));
    if (bind(fd, reinterpret_cast<sockaddr*>(&addr), sizeof(addr)) < 0 || listen(fd, 1024) < 0) {
      const std::string why = std::strerror(errno);
      close(fd);
      for (auto& r : reactors_) close(r->listen_fd);
      reactors_.clear();
      return Error("cannot listen on " + host_ + ":" + std::to_string(port_) + ": " + why);
    }
    if (port_ == 0) {  // ephemeral port picked by the first bind; the others join it
      sockaddr_in bound{};
      socklen_t len = sizeof(bound);
      getsockname(fd, reinterpret_cast<sockaddr*>(&bound), &len);
      port_ = ntohs(bound.sin_port);
    }
    std::unique_ptr<Reactor> r(new Reactor());
    r->owner = this;
    r->index = static_cast<uint32_t>(i);
    r->listen_fd = fd;
    r->epfd = epoll_create1(EPOLL_CLOEXEC);
    r->wake_fd = eventfd(0, EFD_NONBLOCK | EFD_CLOEXEC);
    epoll_event ev{};
    ev.events = EPOLLIN;
    ev.data.u64 = 0;
    epoll_ctl(r->epfd, EPOLL_CTL_ADD, fd, &ev);
    ev.data.u64 = 1;
    epoll_ctl(r->epfd, EPOLL_CTL_ADD, r->wake_fd, &ev);
    reactors_.push_back(std::move(r));
  }
  running_.store(true);
  for (auto& r : reactors_) {
    Reactor* raw = r.get();
    r->thread = std::thread([raw] { raw->Loop(); });
  }
  return Error::Success();
}

void GrpcServerFront::Stop() {
  if (!running_.exchange(false)) return;
  for (auto& r : reactors_) {
    const uint64_t one = 1;
    if (write(r->wake_fd, &one, sizeof(one)) < 0) { /* the loop times out by itself */ }
  }
  for (auto& r : reactors_) {
    if (r->thread.joinable()) r->thread.join();
    close(r->listen_fd);
    close(r->wake_fd);
    close(r->epfd);
  }
  reactors_.clear();
}

void GrpcServerFront::Send(uint64_t call, std::string&& message) {
  const size_t idx = static_cast<size_t>(call >> 56);
  if (!running_.load() || idx >= reactors_.size()) return;
  Reactor::Command c;
  c.kind = Reactor::Command::SEND;
  c.call = call;
  c.payload = std::move(message);
  reactors_[idx]->Post(std::move(c));
}

void GrpcServerFront::Finish(uint64_t call, int grpc_status, const std::string& grpc_message) {
  const size_t idx = static_cast<size_t>(call >> 56);
  if (!running_.load() || idx >= reactors_.size()) return;
  Reactor::Command c;
  c.kind = Reactor::Command::FINISH;
  c.call = call;
  c.status = grpc_status;
  c.payload = grpc_message;
  reactors_[idx]->Post(std::move(c));
}

void GrpcServerFront::Reply(uint64_t call, std::string&& message) {
  const size_t idx = static_cast<size_t>(call >> 56);
  if (!running_.load() || idx >= reactors_.size()) return;
  Reactor::Command c;
  c.kind = Reactor::Command::REPLY;
  c.call = call;
  c.payload = std::move(message);
  reactors_[idx]->Post(std::move(c));
}

}  // namespace server
}  // namespace tca
