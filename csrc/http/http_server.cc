// Native HTTP/1.1 serving front (see tca/http_server.h).
#include "tca/http_server.h"

#include <arpa/inet.h>
#include <fcntl.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/epoll.h>
#include <sys/eventfd.h>
#include <sys/socket.h>
#include <sys/uio.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>

namespace tca {
namespace server {
namespace {

// call id layout: [reactor:8][connection:24][request sequence on the connection:32]
inline uint64_t MakeCall(uint32_t reactor, uint32_t conn, uint32_t seq) {
  return (static_cast<uint64_t>(reactor & 0xFF) << 56) | (static_cast<uint64_t>(conn & 0xFFFFFF) << 32) | seq;
}

const char* Reason(int status) {
  switch (status) {
    case 100: return "Continue";
    case 200: return "OK";
    case 400: return "Bad Request";
    case 404: return "Not Found";
    case 405: return "Method Not Allowed";
    case 408: return "Request Timeout";
    case 411: return "Length Required";
    case 413: return "Payload Too Large";
    case 431: return "Request Header Fields Too Large";
    case 500: return "Internal Server Error";
    case 501: return "Not Implemented";
    case 503: return "Service Unavailable";
    default: return status < 400 ? "OK" : "Error";
  }
}

std::string Lower(std::string s) {
  for (auto& c : s) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
  return s;
}

std::string Trim(const std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && (s[a] == ' ' || s[a] == '\t')) ++a;
  while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r')) --b;
  return s.substr(a, b - a);
}

}  // namespace

struct HttpServerFront::Reactor {
  enum class Phase { HEAD, BODY, CHUNK_SIZE, CHUNK_DATA, CHUNK_END, TRAILERS, DISPATCHED };
  struct Conn {
    int fd = -1;
    uint32_t id = 0;
    uint32_t seq = 0;  // sequence number of the request being read / answered
    Phase phase = Phase::HEAD;
    std::string rbuf;  // bytes received and not consumed yet
    Request req;
    size_t body_left = 0;  // BODY: bytes still missing; CHUNK_DATA: bytes left of the current chunk
    size_t body_filled = 0;  // BODY: bytes of the (pre-sized) body received so far
    bool continued = false;
    std::deque<std::string> out;  // response bytes not sent yet
    size_t out_off = 0;           // bytes of out.front() already sent
    bool close_after_write = false;
    bool rejected = false;  // answered here with an error: the connection closes once that is written
    bool want_write = false;
    bool dead = false;
    std::unique_ptr<net::TlsSession> tls;
    bool handshaking = false;
  };
  struct Command {
    uint64_t call;
    int status;
    HeaderList headers;
    std::vector<std::string> parts;
  };

  HttpServerFront* owner = nullptr;
  uint32_t index = 0;
  int listen_fd = -1, epfd = -1, wake_fd = -1;
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
      epoll_event ev{};
      ev.events = EPOLLIN | EPOLLRDHUP;
      ev.data.u64 = (1ull << 32) | c->id;
      epoll_ctl(epfd, EPOLL_CTL_ADD, fd, &ev);
      Conn* raw = c.get();
      if (owner->tls_) {
        c->tls.reset(new net::TlsSession(owner->tls_, std::string(), fd));
        c->handshaking = true;
      }
      conns[c->id] = std::move(c);
      owner->accepted_.fetch_add(1);
      if (raw->handshaking) Handshake(raw);
    }
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

  bool Handshake(Conn* c) {
    const int rc = c->tls->Handshake();
    if (rc == net::TlsSession::kWouldBlock) {
      ArmWrite(c, c->tls->WantsWrite());
      return false;
    }
    if (rc != 1) {
      c->dead = true;
      return false;
    }
    c->handshaking = false;
    ArmWrite(c, false);
    return true;
  }

  // Queues a response built here (errors found while parsing).
  void Reject(Conn* c, int status, const std::string& message) {
    const std::string body = "{\"error\":\"" + message + "\"}";
    std::string head = "HTTP/1.1 " + std::to_string(status) + " " + Reason(status) +
                       "\r\nContent-Type: application/json\r\nContent-Length: " + std::to_string(body.size()) +
                       "\r\nConnection: close\r\n\r\n";
    c->out.push_back(head + body);
    c->close_after_write = true;
    c->rejected = true;
    c->phase = Phase::DISPATCHED;  // nothing more is parsed on this connection
    c->rbuf.clear();
    Write(c);
  }

  void Dispatch(Conn* c) {
    c->phase = Phase::DISPATCHED;
    owner->requests_.fetch_add(1);
    const uint64_t call = MakeCall(index, c->id, c->seq);
    Request req = std::move(c->req);
    c->req = Request();
    if (!req.keep_alive) c->close_after_write = true;
    if (owner->handlers_.on_request) owner->handlers_.on_request(call, std::move(req));
  }

  // Consumes what it can from c->rbuf.
  static constexpr size_t kBodyFirstAllocation = size_t(4) << 20;
  // Makes room for `need` more body bytes behind the ones received (never past the declared length): the
  // buffer doubles, so a body of n bytes is copied at most once more in total while it grows.
  static void GrowBody(Conn* c, size_t need) {
    std::string& body = c->req.body;
    if (body.size() - c->body_filled >= need) return;
    const size_t total = c->body_filled + c->body_left;
    body.resize(std::min(total, std::max(body.size() * 2, c->body_filled + need)));
  }

  void Parse(Conn* c) {
    for (;;) {
      switch (c->phase) {
        case Phase::DISPATCHED:
          return;
        case Phase::HEAD: {
          // tolerate blank lines between requests (RFC 9112 2.2)
          size_t skip = 0;
          while (skip + 1 < c->rbuf.size() && c->rbuf[skip] == '\r' && c->rbuf[skip + 1] == '\n') skip += 2;
          if (skip) c->rbuf.erase(0, skip);
          const size_t end = c->rbuf.find("\r\n\r\n");
          if (end == std::string::npos) {
            if (c->rbuf.size() > owner->limits_.max_header_bytes) Reject(c, 431, "request header section too large");
            return;
          }
          if (!ParseHead(c, end)) return;
          c->rbuf.erase(0, end + 4);
          break;
        }
        case Phase::BODY: {
          // the body was sized from Content-Length when the head was parsed
          const size_t take = std::min(c->body_left, c->rbuf.size());
          if (take) {
            GrowBody(c, take);
            std::memcpy(&c->req.body[c->body_filled], c->rbuf.data(), take);
            c->rbuf.erase(0, take);
            c->body_filled += take;
            c->body_left -= take;
          }
          if (c->body_left) return;
          Dispatch(c);
          return;
        }
        case Phase::CHUNK_SIZE: {
          const size_t eol = c->rbuf.find("\r\n");
          if (eol == std::string::npos) {
            if (c->rbuf.size() > 4096) Reject(c, 400, "malformed chunk size");
            return;
          }
          char* stop = nullptr;
          const unsigned long long n = std::strtoull(c->rbuf.c_str(), &stop, 16);
          if (stop == c->rbuf.c_str() || n > owner->limits_.max_body_bytes - std::min(owner->limits_.max_body_bytes, c->req.body.size())) {
            Reject(c, stop == c->rbuf.c_str() ? 400 : 413, stop == c->rbuf.c_str() ? "malformed chunk size" : "request body too large");
            return;
          }
          c->rbuf.erase(0, eol + 2);
          c->body_left = static_cast<size_t>(n);
          c->phase = n ? Phase::CHUNK_DATA : Phase::TRAILERS;
          break;
        }
        case Phase::CHUNK_DATA: {
          const size_t take = std::min(c->body_left, c->rbuf.size());
          c->req.body.append(c->rbuf, 0, take);
          c->rbuf.erase(0, take);
          c->body_left -= take;
          if (c->body_left) return;
          c->phase = Phase::CHUNK_END;
          break;
        }
        case Phase::CHUNK_END: {
          if (c->rbuf.size() < 2) return;
          if (c->rbuf[0] != '\r' || c->rbuf[1] != '\n') {
            Reject(c, 400, "malformed chunk");
            return;
          }
          c->rbuf.erase(0, 2);
          c->phase = Phase::CHUNK_SIZE;
          break;
        }
        case Phase::TRAILERS: {
          // zero or more trailer lines, then an empty line
          const size_t eol = c->rbuf.find("\r\n");
          if (eol == std::string::npos) {
            if (c->rbuf.size() > owner->limits_.max_header_bytes) Reject(c, 431, "request trailer section too large");
            return;
          }
          const bool blank = eol == 0;
          c->rbuf.erase(0, eol + 2);
          if (!blank) break;
          Dispatch(c);
          return;
        }
      }
    }
  }

  // Request line + header fields in c->rbuf[0, end); false when the request was rejected.
  bool ParseHead(Conn* c, size_t end) {
    const std::string& b = c->rbuf;
    size_t line_end = b.find("\r\n");
    const std::string first = b.substr(0, std::min(line_end, end));
    const size_t sp1 = first.find(' ');
    const size_t sp2 = sp1 == std::string::npos ? std::string::npos : first.find(' ', sp1 + 1);
    if (sp2 == std::string::npos || first.compare(sp2 + 1, 5, "HTTP/") != 0) {
      Reject(c, 400, "malformed request line");
      return false;
    }
    Request& r = c->req;
    r = Request();
    r.method = first.substr(0, sp1);
    r.target = first.substr(sp1 + 1, sp2 - sp1 - 1);
    const std::string version = first.substr(sp2 + 1);
    r.keep_alive = version != "HTTP/1.0";
    bool chunked = false, has_length = false, expect_continue = false;
    unsigned long long length = 0;
    for (size_t pos = std::min(line_end, end) + 2; pos < end;) {
      size_t eol = b.find("\r\n", pos);
      if (eol == std::string::npos || eol > end) eol = end;
      const size_t colon = b.find(':', pos);
      if (colon == std::string::npos || colon >= eol) {
        Reject(c, 400, "malformed header field");
        return false;
      }
      std::string name = Lower(b.substr(pos, colon - pos));
      std::string value = Trim(b.substr(colon + 1, eol - colon - 1));
      if (name == "content-length") {
        char* stop = nullptr;
        length = std::strtoull(value.c_str(), &stop, 10);
        if (value.empty() || *stop != '\0') {
          Reject(c, 400, "malformed Content-Length");
          return false;
        }
        has_length = true;
      } else if (name == "transfer-encoding") {
        chunked = Lower(value).find("chunked") != std::string::npos;
      } else if (name == "connection") {
        const std::string v = Lower(value);
        if (v.find("close") != std::string::npos) r.keep_alive = false;
        else if (v.find("keep-alive") != std::string::npos) r.keep_alive = true;
      } else if (name == "expect") {
        expect_continue = Lower(value) == "100-continue";
      }
      r.headers.emplace_back(std::move(name), std::move(value));
      pos = eol + 2;
    }
    if (chunked) {
      c->phase = Phase::CHUNK_SIZE;
    } else if (has_length && length > 0) {
      if (length > owner->limits_.max_body_bytes) {
        Reject(c, 413, "request body too large");
        return false;
      }
      c->body_left = static_cast<size_t>(length);
      c->body_filled = 0;
      // received in place: from the head's tail by memcpy, then straight from the socket.  Bodies up to 4 MiB are
      // sized once; larger ones grow as their bytes arrive (GrowBody), so a peer that only CLAIMS gigabytes costs nothing.
      r.body.resize(std::min(c->body_left, kBodyFirstAllocation));
      c->phase = Phase::BODY;
    } else {
      c->phase = Phase::BODY;
      c->body_left = 0;
      c->body_filled = 0;
    }
    // the client waits for the go-ahead only when it has not sent the body along
    const size_t have = b.size() - (end + 4);
    if (expect_continue && ((chunked && have == 0) || (!chunked && have < c->body_left))) {
      c->out.push_back("HTTP/1.1 100 Continue\r\n\r\n");
      Write(c);
    }
    return !c->dead;
  }

  void Apply(Command& cmd) {
    const uint32_t conn_id = static_cast<uint32_t>((cmd.call >> 32) & 0xFFFFFF);
    const uint32_t seq = static_cast<uint32_t>(cmd.call & 0xFFFFFFFF);
    auto it = conns.find(conn_id);
    if (it == conns.end()) return;
    Conn* c = it->second.get();
    if (c->phase != Phase::DISPATCHED || c->rejected || c->seq != seq || c->dead) return;
    size_t total = 0;
    for (const auto& p : cmd.parts) total += p.size();
    std::string head;
    head.reserve(256);
    head += "HTTP/1.1 ";
    head += std::to_string(cmd.status);
    head += ' ';
    head += Reason(cmd.status);
    head += "\r\n";
    bool has_type = false;
    for (const auto& h : cmd.headers) {
      if (Lower(h.first) == "content-type") has_type = true;
      head += h.first;
      head += ": ";
      head += h.second;
      head += "\r\n";
    }
    if (!has_type) head += "Content-Type: application/json\r\n";
    head += "Content-Length: ";
    head += std::to_string(total);
    head += "\r\n";
    if (c->close_after_write) head += "Connection: close\r\n";
    head += "\r\n";
    if (total <= 65536) {
      // one send() and one segment train: header and a small body written separately cost two
      // system calls and wake the client twice
      for (auto& p : cmd.parts) head += p;
      c->out.push_back(std::move(head));
    } else {
      c->out.push_back(std::move(head));
      for (auto& p : cmd.parts)
        if (!p.empty()) c->out.push_back(std::move(p));  // large tensors: no extra copy
    }
    ++c->seq;
    c->phase = Phase::HEAD;
    Write(c);
    if (!c->dead && !c->close_after_write && !c->rbuf.empty()) Parse(c);  // a pipelined request
  }

  void Write(Conn* c) {
    if (c->handshaking) return;
    while (!c->out.empty()) {
      ssize_t n;
      if (c->tls) {
        const std::string& s = c->out.front();
        n = c->tls->Write(s.data() + c->out_off, s.size() - c->out_off);
        if (n == net::TlsSession::kWouldBlock) {
          ArmWrite(c, c->tls->WantsWrite());
          return;
        }
        if (n < 0) {
          c->dead = true;
          return;
        }
      } else {
        iovec iov[8];
        int cnt = 0;
        for (auto it = c->out.begin(); it != c->out.end() && cnt < 8; ++it, ++cnt) {
          const size_t off = cnt == 0 ? c->out_off : 0;
          iov[cnt].iov_base = const_cast<char*>(it->data()) + off;
          iov[cnt].iov_len = it->size() - off;
        }
        msghdr msg{};
        msg.msg_iov = iov;
        msg.msg_iovlen = static_cast<size_t>(cnt);
        n = sendmsg(c->fd, &msg, MSG_NOSIGNAL);
      }
      if (n > 0) {
        size_t left = static_cast<size_t>(n);
        while (left && !c->out.empty()) {
          const size_t avail = c->out.front().size() - c->out_off;
          if (left >= avail) {
            left -= avail;
            c->out.pop_front();
            c->out_off = 0;
          } else {
            c->out_off += left;
            left = 0;
          }
        }
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
    if (c->close_after_write && (c->phase != Phase::DISPATCHED || c->rejected)) c->dead = true;
  }

  void Read(Conn* c) {
    static thread_local std::vector<char> buf(512 * 1024);
    if (c->handshaking && !Handshake(c)) return;
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
      } else if (c->phase == Phase::BODY && c->rbuf.empty() && c->body_left >= 65536) {
        // a large body: receive straight into the request's buffer, no bounce through `buf`
        GrowBody(c, 1);
        n = recv(c->fd, &c->req.body[c->body_filled], std::min(c->body_left, c->req.body.size() - c->body_filled), 0);
        if (n > 0) {
          c->body_filled += static_cast<size_t>(n);
          c->body_left -= static_cast<size_t>(n);
          if (c->body_left == 0) {
            Dispatch(c);
            break;
          }
          continue;
        }
      } else {
        n = recv(c->fd, buf.data(), buf.size(), 0);
      }
      if (n > 0) {
        if (c->rbuf.size() + static_cast<size_t>(n) > owner->limits_.max_header_bytes + (64u << 20) && c->phase == Phase::DISPATCHED) {
          c->dead = true;  // a client flooding a busy connection
          break;
        }
        c->rbuf.append(buf.data(), static_cast<size_t>(n));
        Parse(c);
        if (!c->tls && static_cast<size_t>(n) < buf.size()) break;
      } else if (n == 0) {
        c->dead = true;
      } else if (errno == EAGAIN || errno == EWOULDBLOCK) {
        break;
      } else if (errno != EINTR) {
        c->dead = true;
      }
    }
  }

  void Close(Conn* c) {
    if (c->phase == Phase::DISPATCHED && !c->rejected && owner->handlers_.on_cancel)
      owner->handlers_.on_cancel(MakeCall(index, c->id, c->seq));
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
    for (Conn* c : all) Close(c);
  }
};

HttpServerFront::HttpServerFront(std::string host, int port, int threads, Handlers handlers)
    : host_(std::move(host)), port_(port), nthreads_(threads < 1 ? 1 : (threads > 64 ? 64 : threads)),
      handlers_(std::move(handlers)) {}

HttpServerFront::~HttpServerFront() { Stop(); }

Error HttpServerFront::Start() {
  if (running_.load()) return Error("already started");
  sockaddr_in addr{};
  addr.sin_family = AF_INET;
  if (host_.empty() || host_ == "0.0.0.0") addr.sin_addr.s_addr = htonl(INADDR_ANY);
  else if (host_ == "localhost") addr.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
  else if (inet_pton(AF_INET, host_.c_str(), &addr.sin_addr) != 1) return Error("cannot parse listen address '" + host_ + "'");
  // one listening socket per reactor (SO_REUSEPORT): the kernel spreads new connections over them
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
      for (auto& r : reactors_) {
        close(r->listen_fd);
        close(r->wake_fd);
        close(r->epfd);
      }
      reactors_.clear();
      return Error("cannot listen on " + host_ + ":" + std::to_string(port_) + ": " + why);
    }
    if (port_ == 0) {
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

void HttpServerFront::Stop() {
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

void HttpServerFront::Respond(uint64_t call, int status, HeaderList&& headers, std::vector<std::string>&& parts) {
  const size_t idx = static_cast<size_t>(call >> 56);
  if (!running_.load() || idx >= reactors_.size()) return;
  Reactor::Command c;
  c.call = call;
  c.status = status;
  c.headers = std::move(headers);
  c.parts = std::move(parts);
  reactors_[idx]->Post(std::move(c));
}

}  // namespace server
}  // namespace tca
