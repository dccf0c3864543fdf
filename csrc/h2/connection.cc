// HTTP/2 connection state machine (see tca/h2.h).
#include <algorithm>
#include <cstring>

#include "tca/h2.h"

namespace tca {
namespace h2 {
namespace {

constexpr char kClientMagic[] = "PRI * HTTP/2.0\r\n\r\nSM\r\n\r\n";
constexpr size_t kClientMagicLen = 24;

enum FrameType : uint8_t {
  kData = 0, kHeaders = 1, kPriority = 2, kRstStream = 3, kSettings = 4, kPushPromise = 5, kPing = 6, kGoaway = 7,
  kWindowUpdate = 8, kContinuation = 9,
};
enum Flags : uint8_t { kEndStream = 0x1, kAck = 0x1, kEndHeaders = 0x4, kPadded = 0x8, kPriorityFlag = 0x20 };
enum Setting : uint16_t {
  kHeaderTableSize = 1, kEnablePush = 2, kMaxConcurrentStreams = 3, kInitialWindowSize = 4, kMaxFrameSize = 5,
  kMaxHeaderListSize = 6,
};

inline uint32_t Load32(const uint8_t* p) {
  return (static_cast<uint32_t>(p[0]) << 24) | (static_cast<uint32_t>(p[1]) << 16) | (static_cast<uint32_t>(p[2]) << 8) |
         p[3];
}
inline void Put32(std::string* out, uint32_t v) {
  const char b[4] = {static_cast<char>(v >> 24), static_cast<char>(v >> 16), static_cast<char>(v >> 8),
                     static_cast<char>(v)};
  out->append(b, 4);
}
inline void Put16(std::string* out, uint16_t v) {
  const char b[2] = {static_cast<char>(v >> 8), static_cast<char>(v)};
  out->append(b, 2);
}

}  // namespace

Connection::Connection(Role role, Callbacks callbacks) : role_(role), cb_(std::move(callbacks)) {
  next_stream_id_ = role == Role::CLIENT ? 1 : 2;
  preface_pending_ = role == Role::SERVER;
}

void Connection::SetLocalWindow(uint32_t stream_window, uint32_t connection_window, uint32_t max_frame_size) {
  local_stream_window_ = stream_window;
  local_conn_window_ = connection_window;
  local_max_frame_ = max_frame_size;
}

void Connection::WriteFrameHeader(size_t len, uint8_t type, uint8_t flags, uint32_t stream) {
  const char h[9] = {static_cast<char>(len >> 16), static_cast<char>(len >> 8), static_cast<char>(len),
                     static_cast<char>(type), static_cast<char>(flags), static_cast<char>((stream >> 24) & 0x7F),
                     static_cast<char>(stream >> 16), static_cast<char>(stream >> 8), static_cast<char>(stream)};
  out_.append(h, 9);
}

void Connection::WriteSettings() {
  WriteFrameHeader(role_ == Role::CLIENT ? 24 : 18, kSettings, 0, 0);
  if (role_ == Role::CLIENT) {
    Put16(&out_, kEnablePush);
    Put32(&out_, 0);
  }
  Put16(&out_, kInitialWindowSize);
  Put32(&out_, local_stream_window_);
  Put16(&out_, kMaxFrameSize);
  Put32(&out_, local_max_frame_);
  Put16(&out_, kMaxHeaderListSize);
  Put32(&out_, 1u << 20);
}

void Connection::WriteWindowUpdate(uint32_t stream, uint32_t increment) {
  WriteFrameHeader(4, kWindowUpdate, 0, stream);
  Put32(&out_, increment & 0x7FFFFFFF);
}

void Connection::Start() {
  if (role_ == Role::CLIENT) out_.append(kClientMagic, kClientMagicLen);
  WriteSettings();
  if (local_conn_window_ > 65535) WriteWindowUpdate(0, local_conn_window_ - 65535);
}

void Connection::ConsumeOutput(size_t n) {
  out_off_ += n;
  if (out_off_ >= out_.size()) {
    out_.clear();
    out_off_ = 0;
  } else if (out_off_ > (1u << 20) && out_off_ >= out_.size() / 2) {
    // keep the buffer from growing without bound behind a slow peer - but only once the sent part is the
    // larger one: dropping it moves the rest, and with tens of megabytes queued (many large messages on
    // one connection) a move per megabyte sent costs many times the bytes sent
    out_.erase(0, out_off_);
    out_off_ = 0;
  }
}

Connection::Stream* Connection::Find(uint32_t id) {
  auto it = streams_.find(id);
  return it == streams_.end() ? nullptr : &it->second;
}

void Connection::MaybeErase(uint32_t id) {
  auto it = streams_.find(id);
  if (it != streams_.end() && it->second.local_closed && it->second.remote_closed && it->second.pending.empty())
    streams_.erase(it);
}

Error Connection::ConnectionError(uint32_t code, const std::string& what) {
  Goaway(code, what);
  return Error("HTTP/2 connection error: " + what);
}

void Connection::Goaway(uint32_t error_code, const std::string& debug) {
  if (goaway_sent_) return;
  goaway_sent_ = true;
  WriteFrameHeader(8 + debug.size(), kGoaway, 0, 0);
  Put32(&out_, last_peer_stream_);
  Put32(&out_, error_code);
  out_.append(debug);
}

void Connection::SendPing() {
  ++pings_sent_;
  WriteFrameHeader(8, kPing, 0, 0);
  const char payload[8] = {'t', 'c', 'a', 'p', 'i', 'n', 'g', static_cast<char>(pings_sent_)};
  out_.append(payload, 8);
}

void Connection::WriteHeaderBlock(uint32_t stream, const HeaderList& headers, bool end_stream) {
  std::string block;
  HpackEncoder::Encode(headers, &block);
  size_t off = 0;
  bool first = true;
  do {
    const size_t n = std::min<size_t>(block.size() - off, peer_max_frame_);
    const bool last = off + n == block.size();
    uint8_t flags = last ? kEndHeaders : 0;
    if (first && end_stream) flags |= kEndStream;
    WriteFrameHeader(n, first ? kHeaders : kContinuation, flags, stream);
    out_.append(block, off, n);
    off += n;
    first = false;
  } while (off < block.size());
}

uint32_t Connection::OpenStream(const HeaderList& headers, bool end_stream) {
  if (role_ != Role::CLIENT || goaway_sent_ || goaway_received_ || next_stream_id_ > 0x7FFFFFFD) return 0;
  const uint32_t id = next_stream_id_;
  next_stream_id_ += 2;
  Stream& s = streams_[id];
  s.send_window = peer_initial_window_;
  s.local_closed = end_stream;
  WriteHeaderBlock(id, headers, end_stream);
  return id;
}

Error Connection::SendHeaders(uint32_t stream, const HeaderList& headers, bool end_stream) {
  Stream* s = Find(stream);
  if (s == nullptr || s->local_closed) return Error("HTTP/2: stream is not open for sending");
  if (!s->pending.empty()) {
    // trailers must follow the queued DATA: keep them as a marker entry
    Pending p;
    HpackEncoder::Encode(headers, &p.data);
    p.end_stream = end_stream;
    p.off = static_cast<size_t>(-1);  // marks a header block
    s->pending.push_back(std::move(p));
    return Error::Success();
  }
  WriteHeaderBlock(stream, headers, end_stream);
  if (end_stream) {
    s->local_closed = true;
    MaybeErase(stream);
  }
  return Error::Success();
}

Error Connection::SendData(uint32_t stream, std::string&& data, bool end_stream) {
  Stream* s = Find(stream);
  if (s == nullptr || s->local_closed) return Error("HTTP/2: stream is not open for sending");
  Pending p;
  queued_bytes_ += data.size();
  p.data = std::move(data);
  p.end_stream = end_stream;
  s->pending.push_back(std::move(p));
  FlushStream(stream, s);
  MaybeErase(stream);
  return Error::Success();
}

void Connection::FlushStream(uint32_t id, Stream* s) {
  while (!s->pending.empty()) {
    Pending& p = s->pending.front();
    if (p.off == static_cast<size_t>(-1)) {  // queued header block (trailers)
      size_t off = 0;
      bool first = true;
      do {
        const size_t n = std::min<size_t>(p.data.size() - off, peer_max_frame_);
        const bool last = off + n == p.data.size();
        uint8_t flags = last ? kEndHeaders : 0;
        if (first && p.end_stream) flags |= kEndStream;
        WriteFrameHeader(n, first ? kHeaders : kContinuation, flags, id);
        out_.append(p.data, off, n);
        off += n;
        first = false;
      } while (off < p.data.size());
      if (p.end_stream) s->local_closed = true;
      s->pending.pop_front();
      continue;
    }
    const size_t left = p.data.size() - p.off;
    if (left == 0) {
      if (p.end_stream) {
        WriteFrameHeader(0, kData, kEndStream, id);
        s->local_closed = true;
      }
      s->pending.pop_front();
      continue;
    }
    const int64_t window = std::min(conn_send_window_, s->send_window);
    if (window <= 0) return;
    const size_t n = std::min<size_t>({left, static_cast<size_t>(window), peer_max_frame_});
    const bool done = n == left;
    WriteFrameHeader(n, kData, (done && p.end_stream) ? kEndStream : 0, id);
    out_.append(p.data, p.off, n);
    p.off += n;
    conn_send_window_ -= static_cast<int64_t>(n);
    s->send_window -= static_cast<int64_t>(n);
    queued_bytes_ -= n;
    if (done) {
      if (p.end_stream) s->local_closed = true;
      s->pending.pop_front();
    }
  }
}

void Connection::Flush() {
  std::vector<uint32_t> ids;
  for (auto& kv : streams_)
    if (!kv.second.pending.empty()) ids.push_back(kv.first);
  for (uint32_t id : ids) {
    Stream* s = Find(id);
    if (s == nullptr) continue;
    FlushStream(id, s);
    MaybeErase(id);
    if (conn_send_window_ <= 0) break;
  }
}

void Connection::ResetStream(uint32_t stream, uint32_t error_code) {
  auto it = streams_.find(stream);
  if (it == streams_.end()) return;
  for (const auto& p : it->second.pending)
    if (p.off != static_cast<size_t>(-1)) queued_bytes_ -= p.data.size() - p.off;
  streams_.erase(it);
  WriteFrameHeader(4, kRstStream, 0, stream);
  Put32(&out_, error_code);
}

Error Connection::Receive(const uint8_t* data, size_t len) {
  if (preface_pending_) {
    while (len && preface_seen_ < kClientMagicLen) {
      if (static_cast<char>(*data) != kClientMagic[preface_seen_]) return ConnectionError(kProtocolError, "bad client preface");
      ++data;
      --len;
      ++preface_seen_;
    }
    if (preface_seen_ < kClientMagicLen) return Error::Success();
    preface_pending_ = false;
  }
  const uint8_t* p = data;
  size_t n = len;
  if (!in_.empty()) {
    in_.append(reinterpret_cast<const char*>(data), len);
    p = reinterpret_cast<const uint8_t*>(in_.data());
    n = in_.size();
  }
  size_t off = 0;
  while (n - off >= 9) {
    const uint8_t* h = p + off;
    const size_t flen = (static_cast<size_t>(h[0]) << 16) | (static_cast<size_t>(h[1]) << 8) | h[2];
    if (flen > local_max_frame_) return ConnectionError(kFrameSizeError, "frame larger than SETTINGS_MAX_FRAME_SIZE");
    if (n - off < 9 + flen) break;
    const uint8_t type = h[3], flags = h[4];
    const uint32_t stream = Load32(h + 5) & 0x7FFFFFFF;
    if (hb_stream_ != 0 && (type != kContinuation || stream != hb_stream_))
      return ConnectionError(kProtocolError, "expected CONTINUATION");
    const Error e = HandleFrame(type, flags, stream, h + 9, flen);
    if (!e.IsOk()) return e;
    off += 9 + flen;
  }
  if (!in_.empty()) {
    in_.erase(0, off);
  } else if (off < n) {
    in_.assign(reinterpret_cast<const char*>(p + off), n - off);
  }
  return Error::Success();
}

Error Connection::HandleSettings(uint8_t flags, const uint8_t* p, size_t len) {
  if (flags & kAck) return len == 0 ? Error::Success() : ConnectionError(kFrameSizeError, "SETTINGS ack with payload");
  if (len % 6 != 0) return ConnectionError(kFrameSizeError, "bad SETTINGS length");
  for (size_t i = 0; i < len; i += 6) {
    const uint16_t id = static_cast<uint16_t>((p[i] << 8) | p[i + 1]);
    const uint32_t value = Load32(p + i + 2);
    switch (id) {
      case kInitialWindowSize: {
        if (value > 0x7FFFFFFF) return ConnectionError(kFlowControlError, "initial window too large");
        const int64_t delta = static_cast<int64_t>(value) - static_cast<int64_t>(peer_initial_window_);
        peer_initial_window_ = value;
        for (auto& kv : streams_) kv.second.send_window += delta;
        break;
      }
      case kMaxFrameSize:
        if (value < 16384 || value > 16777215) return ConnectionError(kProtocolError, "bad SETTINGS_MAX_FRAME_SIZE");
        peer_max_frame_ = value;
        break;
      case kMaxConcurrentStreams:
        peer_max_streams_ = value;
        break;
      default:
        break;  // header table size: our encoder keeps no dynamic table; the rest is advisory
    }
  }
  WriteFrameHeader(0, kSettings, kAck, 0);
  Flush();
  return Error::Success();
}

Error Connection::HandleHeaderBlockDone(uint32_t stream, bool end_stream) {
  HeaderList headers;
  const Error e = hpack_.Decode(reinterpret_cast<const uint8_t*>(hb_buf_.data()), hb_buf_.size(), &headers);
  hb_buf_.clear();
  hb_stream_ = 0;
  if (!e.IsOk()) return ConnectionError(kCompressionError, e.Message());
  Stream* s = Find(stream);
  if (s == nullptr) {
    if (role_ == Role::SERVER && (stream & 1) && stream > last_peer_stream_) {
      if (goaway_sent_) return Error::Success();  // refused: the client retries elsewhere
      last_peer_stream_ = stream;
      s = &streams_[stream];
      s->send_window = peer_initial_window_;
    } else {
      return Error::Success();  // headers for a stream we already dropped (e.g. after RST_STREAM)
    }
  }
  s->headers_seen = true;
  if (end_stream) s->remote_closed = true;
  if (cb_.on_headers) cb_.on_headers(stream, std::move(headers), end_stream);
  MaybeErase(stream);
  return Error::Success();
}

Error Connection::HandleFrame(uint8_t type, uint8_t flags, uint32_t stream, const uint8_t* p, size_t len) {
  switch (type) {
    case kData: {
      if (stream == 0) return ConnectionError(kProtocolError, "DATA on stream 0");
      size_t pad = 0;
      const size_t flow_len = len;
      if (flags & kPadded) {
        if (len < 1) return ConnectionError(kFrameSizeError, "bad padded DATA");
        pad = p[0];
        ++p;
        --len;
        if (pad > len) return ConnectionError(kProtocolError, "DATA padding exceeds the frame");
        len -= pad;
      }
      // flow control accounting covers the whole frame, delivered or not
      conn_recv_consumed_ += static_cast<int64_t>(flow_len);
      if (conn_recv_consumed_ >= static_cast<int64_t>(local_conn_window_ / 2)) {
        WriteWindowUpdate(0, static_cast<uint32_t>(conn_recv_consumed_));
        conn_recv_consumed_ = 0;
      }
      Stream* s = Find(stream);
      if (s == nullptr || s->remote_closed) return Error::Success();  // late frame of a stream we closed
      const bool end = (flags & kEndStream) != 0;
      if (end) {
        s->remote_closed = true;
      } else {
        s->recv_consumed += static_cast<int64_t>(flow_len);
        if (s->recv_consumed >= static_cast<int64_t>(local_stream_window_ / 2)) {
          WriteWindowUpdate(stream, static_cast<uint32_t>(s->recv_consumed));
          s->recv_consumed = 0;
        }
      }
      if (cb_.on_data) cb_.on_data(stream, p, len, end);
      MaybeErase(stream);
      return Error::Success();
    }
    case kHeaders: {
      if (stream == 0) return ConnectionError(kProtocolError, "HEADERS on stream 0");
      size_t pad = 0;
      if (flags & kPadded) {
        if (len < 1) return ConnectionError(kFrameSizeError, "bad padded HEADERS");
        pad = p[0];
        ++p;
        --len;
      }
      if (flags & kPriorityFlag) {
        if (len < 5) return ConnectionError(kFrameSizeError, "bad HEADERS priority");
        p += 5;
        len -= 5;
      }
      if (pad > len) return ConnectionError(kProtocolError, "HEADERS padding exceeds the frame");
      len -= pad;
      hb_buf_.assign(reinterpret_cast<const char*>(p), len);
      hb_end_stream_ = (flags & kEndStream) != 0;
      if (flags & kEndHeaders) return HandleHeaderBlockDone(stream, hb_end_stream_);
      hb_stream_ = stream;
      return Error::Success();
    }
    case kContinuation: {
      if (hb_stream_ == 0 || stream != hb_stream_) return ConnectionError(kProtocolError, "unexpected CONTINUATION");
      hb_buf_.append(reinterpret_cast<const char*>(p), len);
      if (hb_buf_.size() > (4u << 20)) return ConnectionError(kEnhanceYourCalm, "header block too large");
      if (flags & kEndHeaders) return HandleHeaderBlockDone(stream, hb_end_stream_);
      return Error::Success();
    }
    case kPriority:
      return len == 5 ? Error::Success() : ConnectionError(kFrameSizeError, "bad PRIORITY length");
    case kRstStream: {
      if (len != 4 || stream == 0) return ConnectionError(kFrameSizeError, "bad RST_STREAM");
      auto it = streams_.find(stream);
      if (it != streams_.end()) {
        for (const auto& q : it->second.pending)
          if (q.off != static_cast<size_t>(-1)) queued_bytes_ -= q.data.size() - q.off;
        streams_.erase(it);
        if (cb_.on_reset) cb_.on_reset(stream, Load32(p));
      }
      return Error::Success();
    }
    case kSettings:
      if (stream != 0) return ConnectionError(kProtocolError, "SETTINGS on a stream");
      return HandleSettings(flags, p, len);
    case kPushPromise:
      return ConnectionError(kProtocolError, "PUSH_PROMISE is disabled");
    case kPing:
      if (len != 8 || stream != 0) return ConnectionError(kFrameSizeError, "bad PING");
      if (!(flags & kAck)) {
        WriteFrameHeader(8, kPing, kAck, 0);
        out_.append(reinterpret_cast<const char*>(p), 8);
      } else {
        ++pings_acked_;
      }
      return Error::Success();
    case kGoaway: {
      if (len < 8 || stream != 0) return ConnectionError(kFrameSizeError, "bad GOAWAY");
      goaway_received_ = true;
      if (cb_.on_goaway) cb_.on_goaway(Load32(p) & 0x7FFFFFFF, Load32(p + 4));
      return Error::Success();
    }
    case kWindowUpdate: {
      if (len != 4) return ConnectionError(kFrameSizeError, "bad WINDOW_UPDATE");
      const uint32_t inc = Load32(p) & 0x7FFFFFFF;
      if (stream == 0) {
        if (inc == 0) return ConnectionError(kProtocolError, "zero WINDOW_UPDATE");
        conn_send_window_ += inc;
        if (conn_send_window_ > 0x7FFFFFFFll) return ConnectionError(kFlowControlError, "connection window overflow");
        Flush();
      } else if (Stream* s = Find(stream)) {
        s->send_window += inc;
        if (s->send_window > 0x7FFFFFFFll) {
          ResetStream(stream, kFlowControlError);
          if (cb_.on_reset) cb_.on_reset(stream, kFlowControlError);
        } else {
          FlushStream(stream, s);
          MaybeErase(stream);
        }
      }
      return Error::Success();
    }
    default:
      return Error::Success();  // unknown frame types are ignored (RFC 9113 section 4.1)
  }
}

}  // namespace h2
}  // namespace tca
