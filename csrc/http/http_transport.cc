#include "tca/http_transport.h"

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cstring>
#include <unordered_map>

#include <arpa/inet.h>
#include <fcntl.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/epoll.h>
#include <sys/eventfd.h>
#include <sys/socket.h>
#include <sys/uio.h>
#include <unistd.h>

namespace tca {
namespace client {

namespace {

uint64_t NowNs() {
  return static_cast<uint64_t>(
      std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count());
}

std::string Lower(std::string s) {
  for (auto& c : s) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
  return s;
}

std::string Trim(const std::string& s) {
  size_t b = 0, e = s.size();
  while (b < e && (s[b] == ' ' || s[b] == '\t')) ++b;
  while (e > b && (s[e - 1] == ' ' || s[e - 1] == '\t' || s[e - 1] == '\r')) --e;
  return s.substr(b, e - b);
}

}  // namespace

Error ParseHttpUrl(const std::string& url, HttpEndpoint* out) {
  std::string rest = url;
  out->https = false;
  if (rest.rfind("https://", 0) == 0) {
    out->https = true;
    rest.erase(0, 8);
  } else if (rest.rfind("http://", 0) == 0) {
    rest.erase(0, 7);
  }
  std::string hostport = rest, base;
  const size_t slash = rest.find('/');
  if (slash != std::string::npos) {
    hostport = rest.substr(0, slash);
    base = rest.substr(slash);
    while (!base.empty() && base.back() == '/') base.pop_back();
  }
  std::string host = hostport, port = out->https ? "443" : "80";
  if (!hostport.empty() && hostport[0] == '[') {
    const size_t close = hostport.find(']');
    if (close == std::string::npos) return Error("malformed IPv6 literal in url '" + url + "'");
    host = hostport.substr(1, close - 1);
    if (close + 1 < hostport.size() && hostport[close + 1] == ':') port = hostport.substr(close + 2);
  } else {
    const size_t colon = hostport.rfind(':');
    if (colon != std::string::npos) {
      host = hostport.substr(0, colon);
      port = hostport.substr(colon + 1);
    }
  }
  if (host.empty()) return Error("url '" + url + "' has no host");
  out->host = host;
  out->port = std::atoi(port.c_str());
  if (out->port <= 0 || out->port > 65535) return Error("invalid port in url '" + url + "'");
  out->base_path = base;
  return Error::Success();
}

//==============================================================================
void HttpConn::Close() {
  tls_.reset();  // before the socket: the session may still write its close_notify
  tls_blocked_ = false;
  if (fd_ >= 0) {
    ::close(fd_);
    fd_ = -1;
  }
}

void HttpConn::Fail(const Error& e) {
  error_ = e;
  state_ = State::FAILED;
  Close();
}

Error HttpConn::OpenSocket() {
  struct addrinfo hints;
  std::memset(&hints, 0, sizeof(hints));
  hints.ai_family = AF_UNSPEC;
  hints.ai_socktype = SOCK_STREAM;
  struct addrinfo* res = nullptr;
  const std::string port = std::to_string(ep_.port);
  int rc = getaddrinfo(ep_.host.c_str(), port.c_str(), &hints, &res);
  if (rc != 0 || res == nullptr)
    return Error("unable to resolve '" + ep_.host + "': " + gai_strerror(rc), 503);
  Error last("unable to connect to " + ep_.host + ":" + port, 503);
  for (struct addrinfo* ai = res; ai != nullptr; ai = ai->ai_next) {
    int fd = ::socket(ai->ai_family, ai->ai_socktype | SOCK_NONBLOCK | SOCK_CLOEXEC, ai->ai_protocol);
    if (fd < 0) continue;
    int one = 1;
    setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
    int buf = 4 << 20;  // large tensors: fewer wakeups (the reference sizes curl's buffers to 16 MiB)
    setsockopt(fd, SOL_SOCKET, SO_SNDBUF, &buf, sizeof(buf));
    setsockopt(fd, SOL_SOCKET, SO_RCVBUF, &buf, sizeof(buf));
    rc = ::connect(fd, ai->ai_addr, ai->ai_addrlen);
    if (rc == 0 || errno == EINPROGRESS) {
      fd_ = fd;
      ++generation_;
      freeaddrinfo(res);
      return Error::Success();
    }
    last = Error("unable to connect to " + ep_.host + ":" + port + ": " + std::strerror(errno), 503);
    ::close(fd);
  }
  freeaddrinfo(res);
  return last;
}

void HttpConn::BuildHead() {
  head_.clear();
  head_.reserve(256);
  head_ += spec_->method;
  head_ += ' ';
  head_ += ep_.base_path;
  head_ += spec_->target;
  head_ += " HTTP/1.1\r\nHost: ";
  head_ += ep_.host;
  if (ep_.port != (ep_.https ? 443 : 80)) {
    head_ += ':';
    head_ += std::to_string(ep_.port);
  }
  head_ += "\r\n";
  bool has_conn = false;
  for (const auto& h : spec_->headers) {
    head_ += h.first;
    head_ += ": ";
    head_ += h.second;
    head_ += "\r\n";
    if (Lower(h.first) == "connection") has_conn = true;
  }
  const size_t body = spec_->BodySize();
  if (spec_->method != "GET" || body) {
    head_ += "Content-Length: ";
    head_ += std::to_string(body);
    head_ += "\r\n";
  }
  if (!has_conn) head_ += "Connection: keep-alive\r\n";
  head_ += "\r\n";
}

Error HttpConn::Begin(const HttpRequestSpec* spec, HttpResponseData* resp, RequestTimers* timers) {
  spec_ = spec;
  resp_ = resp;
  timers_ = timers;
  error_ = Error();
  resp_->status = 0;
  resp_->headers.clear();
  resp_->body.clear();
  rbuf_.clear();
  piece_ = piece_off_ = 0;
  body_expected_ = 0;
  recv_started_ = false;
  chunk_left_ = 0;
  chunk_phase_ = 0;
  retried_ = false;
  keep_alive_ = true;
  deadline_ns_ = spec->timeout_us ? NowNs() + spec->timeout_us * 1000 : 0;
  BuildHead();
  reused_ = fd_ >= 0;
  if (fd_ < 0) {
    Error e = OpenSocket();
    if (!e.IsOk()) {
      Fail(e);
      return e;
    }
    state_ = State::CONNECTING;
  } else {
    state_ = State::SENDING;
    if (timers_) timers_->CaptureTimestamp(RequestTimers::Kind::SEND_START);
  }
  return Error::Success();
}

bool HttpConn::Reconnect() {
  retried_ = true;
  Close();
  Error e = OpenSocket();
  if (!e.IsOk()) {
    Fail(e);
    return false;
  }
  reused_ = false;
  piece_ = piece_off_ = 0;
  state_ = State::CONNECTING;
  return true;
}

bool HttpConn::SendSome() {
  const size_t total_pieces = 1 + spec_->body.size();
  if (tls_) {
    // TLS records are built by the session: one piece at a time, no gather
    tls_blocked_ = false;
    while (piece_ < total_pieces) {
      const uint8_t* base = piece_ == 0 ? reinterpret_cast<const uint8_t*>(head_.data()) : spec_->body[piece_ - 1].first;
      const size_t len = piece_ == 0 ? head_.size() : spec_->body[piece_ - 1].second;
      if (piece_off_ >= len) {
        ++piece_;
        piece_off_ = 0;
        continue;
      }
      const ssize_t n = tls_->Write(base + piece_off_, len - piece_off_);
      if (n == net::TlsSession::kWouldBlock) {
        tls_blocked_ = true;
        return false;
      }
      if (n < 0) {
        if (reused_ && !retried_) {
          Reconnect();
          return false;
        }
        Fail(Error(tls_->LastError(), 503));
        return false;
      }
      piece_off_ += static_cast<size_t>(n);
    }
    if (timers_) timers_->CaptureTimestamp(RequestTimers::Kind::SEND_END);
    state_ = State::READ_HEAD;
    return true;
  }
  // gather up to 16 pieces per writev: {head_, body pieces...}
  while (piece_ < total_pieces) {
    struct iovec iov[16];
    int n = 0;
    size_t idx = piece_, off = piece_off_;
    while (idx < total_pieces && n < 16) {
      const uint8_t* base = idx == 0 ? reinterpret_cast<const uint8_t*>(head_.data()) : spec_->body[idx - 1].first;
      const size_t len = idx == 0 ? head_.size() : spec_->body[idx - 1].second;
      if (len > off) {
        iov[n].iov_base = const_cast<uint8_t*>(base + off);
        iov[n].iov_len = len - off;
        ++n;
      }
      ++idx;
      off = 0;
    }
    if (n == 0) {
      piece_ = total_pieces;
      break;
    }
    struct msghdr msg;
    std::memset(&msg, 0, sizeof(msg));
    msg.msg_iov = iov;
    msg.msg_iovlen = n;
    ssize_t sent = ::sendmsg(fd_, &msg, MSG_NOSIGNAL);
    if (sent < 0) {
      if (errno == EAGAIN || errno == EWOULDBLOCK) return false;
      if (errno == EINTR) continue;
      // stale keep-alive connection: reconnect once and resend from scratch
      if (reused_ && !retried_ && (errno == EPIPE || errno == ECONNRESET)) {
        Reconnect();
        return false;
      }
      Fail(Error(std::string("send failed: ") + std::strerror(errno), 503));
      return false;
    }
    size_t left = static_cast<size_t>(sent);
    while (left > 0 && piece_ < total_pieces) {
      const size_t len = piece_ == 0 ? head_.size() : spec_->body[piece_ - 1].second;
      const size_t avail = len - piece_off_;
      if (left >= avail) {
        left -= avail;
        ++piece_;
        piece_off_ = 0;
      } else {
        piece_off_ += left;
        left = 0;
      }
    }
    // skip empty pieces
    while (piece_ < total_pieces && (piece_ == 0 ? head_.size() : spec_->body[piece_ - 1].second) == piece_off_) {
      ++piece_;
      piece_off_ = 0;
    }
  }
  if (timers_) timers_->CaptureTimestamp(RequestTimers::Kind::SEND_END);
  state_ = State::READ_HEAD;
  return true;
}

ssize_t HttpConn::RecvRaw(void* buf, size_t len) {
  if (!tls_) return ::recv(fd_, buf, len, 0);
  tls_blocked_ = false;
  const ssize_t n = tls_->Read(buf, len);
  if (n >= 0) return n;
  if (n == net::TlsSession::kWouldBlock) {
    tls_blocked_ = true;
    errno = EAGAIN;
  } else {
    errno = EPROTO;
  }
  return -1;
}

bool HttpConn::RecvSome() {
  char buf[65536];
  // large bodies: read straight into the destination once its size is known
  if (state_ == State::READ_BODY && resp_->body.capacity() >= body_expected_ && body_expected_ > resp_->body.size()) {
    const size_t have = resp_->body.size();
    const size_t want = body_expected_ - have;
    resp_->body.resize(body_expected_);
    ssize_t n = RecvRaw(&resp_->body[have], want);
    if (n > 0) {
      resp_->body.resize(have + static_cast<size_t>(n));
      return true;
    }
    resp_->body.resize(have);
    if (n == 0) { Fail(Error("connection closed by peer mid-body", 503)); return false; }
    if (errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR) return false;
    Fail(Error(tls_ && errno == EPROTO ? tls_->LastError() : std::string("recv failed: ") + std::strerror(errno), 503));
    return false;
  }
  ssize_t n = RecvRaw(buf, sizeof(buf));
  if (n > 0) {
    if (!recv_started_) {
      recv_started_ = true;
      if (timers_) timers_->CaptureTimestamp(RequestTimers::Kind::RECV_START);
    }
    rbuf_.append(buf, static_cast<size_t>(n));
    return true;
  }
  if (n == 0) {
    if (state_ == State::READ_TO_CLOSE) {
      resp_->body.append(rbuf_);
      rbuf_.clear();
      keep_alive_ = false;
      Finish();
      return false;
    }
    if (state_ == State::READ_HEAD && reused_ && !retried_ && rbuf_.empty()) {
      // server dropped the idle connection before reading our request
      Reconnect();
      return false;
    }
    Fail(Error("connection closed by peer", 503));
    return false;
  }
  if (errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR) return false;
  if (state_ == State::READ_HEAD && reused_ && !retried_ && rbuf_.empty() && errno == ECONNRESET) {
    Reconnect();
    return false;
  }
  Fail(Error(tls_ && errno == EPROTO ? tls_->LastError() : std::string("recv failed: ") + std::strerror(errno), 503));
  return false;
}

bool HttpConn::ParseHead() {
  const size_t end = rbuf_.find("\r\n\r\n");
  if (end == std::string::npos) {
    if (rbuf_.size() > (1u << 20)) Fail(Error("response header section too large", 502));
    return false;
  }
  size_t pos = 0;
  size_t eol = rbuf_.find("\r\n");
  const std::string status_line = rbuf_.substr(0, eol);
  if (status_line.size() < 12 || status_line.compare(0, 5, "HTTP/") != 0) {
    Fail(Error("malformed HTTP status line", 502));
    return false;
  }
  resp_->status = std::atoi(status_line.c_str() + 9);
  pos = eol + 2;
  resp_->headers.clear();
  while (pos < end) {
    eol = rbuf_.find("\r\n", pos);
    if (eol == std::string::npos || eol > end) eol = end;
    const std::string line = rbuf_.substr(pos, eol - pos);
    const size_t colon = line.find(':');
    if (colon != std::string::npos) resp_->headers[Lower(Trim(line.substr(0, colon)))] = Trim(line.substr(colon + 1));
    pos = eol + 2;
  }
  rbuf_.erase(0, end + 4);
  if (resp_->status == 100) return true;  // interim response: parse the next head
  const std::string* conn = resp_->Header("connection");
  keep_alive_ = !(conn && Lower(*conn) == "close");
  const std::string* te = resp_->Header("transfer-encoding");
  if (spec_->method == "HEAD" || resp_->status == 204 || resp_->status == 304) {
    Finish();
  } else if (te && Lower(*te).find("chunked") != std::string::npos) {
    state_ = State::READ_CHUNKED;
    chunk_phase_ = 0;
  } else if (const std::string* cl = resp_->Header("content-length")) {
    body_expected_ = static_cast<size_t>(std::strtoull(cl->c_str(), nullptr, 10));
    resp_->body.reserve(body_expected_);
    resp_->body.assign(rbuf_, 0, std::min(rbuf_.size(), body_expected_));
    rbuf_.erase(0, std::min(rbuf_.size(), body_expected_));
    state_ = State::READ_BODY;
    if (resp_->body.size() >= body_expected_) Finish();
  } else {
    state_ = State::READ_TO_CLOSE;
  }
  return true;
}

bool HttpConn::ParseChunks() {
  while (true) {
    if (chunk_phase_ == 0) {
      const size_t eol = rbuf_.find("\r\n");
      if (eol == std::string::npos) return false;
      chunk_left_ = static_cast<size_t>(std::strtoull(rbuf_.c_str(), nullptr, 16));
      rbuf_.erase(0, eol + 2);
      chunk_phase_ = chunk_left_ == 0 ? 3 : 1;
    } else if (chunk_phase_ == 1) {
      if (rbuf_.empty()) return false;
      const size_t take = std::min(chunk_left_, rbuf_.size());
      resp_->body.append(rbuf_, 0, take);
      rbuf_.erase(0, take);
      chunk_left_ -= take;
      if (chunk_left_ == 0) chunk_phase_ = 2;
    } else if (chunk_phase_ == 2) {
      if (rbuf_.size() < 2) return false;
      rbuf_.erase(0, 2);
      chunk_phase_ = 0;
    } else {  // trailers until an empty line
      const size_t eol = rbuf_.find("\r\n");
      if (eol == std::string::npos) return false;
      const bool empty = eol == 0;
      rbuf_.erase(0, eol + 2);
      if (empty) {
        Finish();
        return true;
      }
    }
  }
}

void HttpConn::Finish() {
  if (timers_) {
    if (!recv_started_) timers_->CaptureTimestamp(RequestTimers::Kind::RECV_START);
    timers_->CaptureTimestamp(RequestTimers::Kind::RECV_END);
  }
  state_ = State::DONE;
  if (!keep_alive_) Close();
}

void HttpConn::Pump() {
  while (!Finished()) {
    if (deadline_ns_ && NowNs() > deadline_ns_) {
      Fail(Error("Deadline Exceeded", 499));
      return;
    }
    switch (state_) {
      case State::CONNECTING: {
        struct pollfd p = {fd_, POLLOUT, 0};
        if (::poll(&p, 1, 0) <= 0) return;
        int err = 0;
        socklen_t len = sizeof(err);
        getsockopt(fd_, SOL_SOCKET, SO_ERROR, &err, &len);
        if (err != 0) {
          Fail(Error("unable to connect to " + ep_.host + ":" + std::to_string(ep_.port) + ": " + std::strerror(err), 503));
          return;
        }
        if (ep_.https) {
          if (!ep_.tls) {
            Fail(Error("https endpoint without a TLS context", 400));
            return;
          }
          tls_.reset(new net::TlsSession(ep_.tls, ep_.host, fd_));
          state_ = State::TLS_HANDSHAKE;
          break;
        }
        state_ = State::SENDING;
        if (timers_) timers_->CaptureTimestamp(RequestTimers::Kind::SEND_START);
        break;
      }
      case State::TLS_HANDSHAKE: {
        const int rc = tls_->Handshake();
        if (rc == net::TlsSession::kWouldBlock) return;
        if (rc != 1) {
          Fail(Error(tls_->LastError(), 503));
          return;
        }
        state_ = State::SENDING;
        if (timers_) timers_->CaptureTimestamp(RequestTimers::Kind::SEND_START);
        break;
      }
      case State::SENDING:
        if (!SendSome()) return;
        break;
      case State::READ_HEAD:
        if (!ParseHead()) {
          if (Finished()) return;
          if (!RecvSome()) return;
        }
        break;
      case State::READ_BODY:
        if (resp_->body.size() >= body_expected_) { Finish(); break; }
        if (!rbuf_.empty()) {
          const size_t take = std::min(rbuf_.size(), body_expected_ - resp_->body.size());
          resp_->body.append(rbuf_, 0, take);
          rbuf_.erase(0, take);
          break;
        }
        if (!RecvSome()) return;
        break;
      case State::READ_CHUNKED:
        if (!ParseChunks()) {
          if (Finished()) return;
          if (!RecvSome()) return;
        }
        break;
      case State::READ_TO_CLOSE:
        if (!RecvSome()) return;
        break;
      default:
        return;
    }
  }
}

Error HttpConn::RunToCompletion() {
  while (!Finished()) {
    Pump();
    if (Finished()) break;
    int timeout_ms = -1;
    if (deadline_ns_) {
      const uint64_t now = NowNs();
      if (now >= deadline_ns_) {
        Fail(Error("Deadline Exceeded", 499));
        break;
      }
      timeout_ms = static_cast<int>((deadline_ns_ - now) / 1000000) + 1;
    }
    struct pollfd p = {fd_, static_cast<short>(WantsWrite() ? POLLOUT : POLLIN), 0};
    int rc = ::poll(&p, 1, timeout_ms);
    if (rc < 0 && errno != EINTR) {
      Fail(Error(std::string("poll failed: ") + std::strerror(errno), 503));
      break;
    }
  }
  return state_ == State::DONE ? Error::Success() : error_;
}

//==============================================================================
struct HttpReactor::Active {
  std::unique_ptr<HttpConn> conn;
  std::unique_ptr<Op> op;
  int registered_fd = -1;
  uint64_t registered_gen = 0;
  bool registered_write = false;
};

HttpReactor::HttpReactor(const HttpEndpoint& ep) : ep_(ep) {
  epfd_ = epoll_create1(EPOLL_CLOEXEC);
  wake_fd_ = eventfd(0, EFD_NONBLOCK | EFD_CLOEXEC);
  struct epoll_event ev;
  std::memset(&ev, 0, sizeof(ev));
  ev.events = EPOLLIN;
  ev.data.u64 = 0;  // id 0 = wake-up
  epoll_ctl(epfd_, EPOLL_CTL_ADD, wake_fd_, &ev);
  thread_ = std::thread(&HttpReactor::Loop, this);
}

HttpReactor::~HttpReactor() {
  stop_.store(true);
  uint64_t one = 1;
  ssize_t w = ::write(wake_fd_, &one, sizeof(one));
  (void)w;
  if (thread_.joinable()) thread_.join();
  ::close(wake_fd_);
  ::close(epfd_);
}

void HttpReactor::Submit(std::unique_ptr<Op> op) {
  inflight_.fetch_add(1);
  {
    std::lock_guard<std::mutex> lk(mu_);
    incoming_.push_back(std::move(op));
  }
  uint64_t one = 1;
  ssize_t w = ::write(wake_fd_, &one, sizeof(one));
  (void)w;
}

void HttpReactor::Loop() {
  std::unordered_map<uint64_t, Active> active;
  std::vector<std::unique_ptr<HttpConn>> idle;
  uint64_t next_id = 1;

  auto arm = [&](uint64_t id, Active& a) {
    const int fd = a.conn->fd();
    const bool want_write = a.conn->WantsWrite();
    struct epoll_event ev;
    std::memset(&ev, 0, sizeof(ev));
    ev.events = (want_write ? EPOLLOUT : EPOLLIN) | EPOLLRDHUP;
    ev.data.u64 = id;
    if (a.registered_fd != fd || a.registered_gen != a.conn->generation()) {
      // the old fd (if any) was closed by the connection: epoll forgot it already
      if (fd >= 0) epoll_ctl(epfd_, EPOLL_CTL_ADD, fd, &ev);
      a.registered_fd = fd;
      a.registered_gen = a.conn->generation();
    } else if (a.registered_write != want_write && fd >= 0) {
      epoll_ctl(epfd_, EPOLL_CTL_MOD, fd, &ev);
    }
    a.registered_write = want_write;
  };

  auto finish = [&](uint64_t id) {
    auto it = active.find(id);
    Active a = std::move(it->second);
    active.erase(it);
    if (a.registered_fd >= 0 && a.conn->fd() == a.registered_fd && a.registered_gen == a.conn->generation())
      epoll_ctl(epfd_, EPOLL_CTL_DEL, a.registered_fd, nullptr);
    const Error err = a.conn->state() == HttpConn::State::DONE ? Error::Success() : a.conn->error();
    if (a.conn->Reusable()) idle.push_back(std::move(a.conn));
    Op* raw = a.op.get();
    if (raw->done) raw->done(err, raw);
    inflight_.fetch_sub(1);
  };

  std::vector<struct epoll_event> events(256);
  while (true) {
    // adopt new operations
    std::deque<std::unique_ptr<Op>> fresh;
    {
      std::lock_guard<std::mutex> lk(mu_);
      fresh.swap(incoming_);
    }
    for (auto& op : fresh) {
      Active a;
      if (!idle.empty()) {
        a.conn = std::move(idle.back());
        idle.pop_back();
      } else {
        a.conn.reset(new HttpConn(ep_));
      }
      a.op = std::move(op);
      a.op->timers.CaptureTimestamp(RequestTimers::Kind::REQUEST_START);
      const uint64_t id = next_id++;
      a.conn->Begin(&a.op->spec, &a.op->resp, &a.op->timers);
      auto& slot = active[id] = std::move(a);
      if (!slot.conn->Finished()) slot.conn->Pump();
      if (slot.conn->Finished()) finish(id); else arm(id, slot);
    }
    if (stop_.load() && active.empty()) {
      std::lock_guard<std::mutex> lk(mu_);
      if (incoming_.empty()) break;
      continue;
    }
    // next deadline
    int timeout_ms = stop_.load() ? 10 : 1000;
    const uint64_t now = NowNs();
    for (auto& kv : active) {
      const uint64_t d = kv.second.conn->deadline_ns();
      if (d) {
        const int ms = d <= now ? 0 : static_cast<int>((d - now) / 1000000) + 1;
        timeout_ms = std::min(timeout_ms, ms);
      }
    }
    int n = epoll_wait(epfd_, events.data(), static_cast<int>(events.size()), timeout_ms);
    if (n < 0 && errno != EINTR) break;
    std::vector<uint64_t> ready;
    for (int i = 0; i < n; ++i) {
      if (events[i].data.u64 == 0) {
        uint64_t v;
        while (::read(wake_fd_, &v, sizeof(v)) > 0) {}
      } else {
        ready.push_back(events[i].data.u64);
      }
    }
    // deadlines are enforced inside Pump(); visit expired ones too
    const uint64_t now2 = NowNs();
    for (auto& kv : active)
      if (kv.second.conn->deadline_ns() && kv.second.conn->deadline_ns() <= now2) ready.push_back(kv.first);
    for (uint64_t id : ready) {
      auto it = active.find(id);
      if (it == active.end()) continue;
      it->second.conn->Pump();
      if (it->second.conn->Finished()) finish(id); else arm(id, it->second);
    }
  }
}

}  // namespace client
}  // namespace tca
