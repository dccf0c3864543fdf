// HTTP/2 (RFC 9113) + HPACK (RFC 7541) protocol engine, written for this
// library: the image ships neither grpc++ nor an HTTP/2 development package,
// and both the native gRPC client and the native serving front need one.
//
// `Connection` is a sans-IO state machine: the owner feeds it the bytes read
// from a socket (`Receive`) and writes out whatever accumulates in `Output()`.
// It serves both roles (client: `OpenStream`; server: streams appear through
// `on_headers`).  Flow control is handled inside: queued DATA is released as
// the peer's windows allow, and the local windows are replenished as data is
// delivered.  Not thread-safe; one reactor thread owns a connection.
#pragma once

#include <cstdint>
#include <deque>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "tca/error.h"

namespace tca {
namespace h2 {

struct HeaderField {
  std::string name;
  std::string value;
};
using HeaderList = std::vector<HeaderField>;

// ---------------------------------------------------------------------------
// HPACK
class HpackDecoder {
 public:
  // Decodes one complete header block, appending to `out`.
  Error Decode(const uint8_t* data, size_t len, HeaderList* out);
  // Upper bound the peer may set through a table size update (our
  // SETTINGS_HEADER_TABLE_SIZE).
  void SetMaxTableSize(size_t bytes) { settings_limit_ = bytes; }

 private:
  Error Lookup(uint64_t index, HeaderField* out) const;
  void Insert(HeaderField field);
  void Evict();
  std::deque<HeaderField> table_;  // newest first
  size_t table_bytes_ = 0;
  size_t table_limit_ = 4096;
  size_t settings_limit_ = 4096;
};

// Stateless encoder: exact static-table matches become indexed fields,
// everything else a literal WITHOUT indexing, never Huffman coded - so the
// peer's decoder needs no dynamic state from us and encoding is a few copies.
class HpackEncoder {
 public:
  static void Encode(const HeaderList& headers, std::string* out);
};

// Huffman string decoding (exposed for tests).
Error HuffmanDecode(const uint8_t* data, size_t len, std::string* out);
// Encoder counterpart, used by tests to cross-check the code table.
void HuffmanEncode(const std::string& in, std::string* out);

// ---------------------------------------------------------------------------
enum ErrorCode : uint32_t {
  kNoError = 0, kProtocolError = 1, kInternalError = 2, kFlowControlError = 3, kSettingsTimeout = 4,
  kStreamClosed = 5, kFrameSizeError = 6, kRefusedStream = 7, kCancel = 8, kCompressionError = 9,
  kConnectError = 10, kEnhanceYourCalm = 11,
};

class Connection {
 public:
  enum class Role { CLIENT, SERVER };

  struct Callbacks {
    // A complete header block (request/response headers or trailers).
    std::function<void(uint32_t stream, HeaderList&& headers, bool end_stream)> on_headers;
    // DATA payload; the pointer is valid during the call only.
    std::function<void(uint32_t stream, const uint8_t* data, size_t len, bool end_stream)> on_data;
    // The peer reset the stream.
    std::function<void(uint32_t stream, uint32_t error_code)> on_reset;
    // The peer is going away; streams above `last_stream` were not processed.
    std::function<void(uint32_t last_stream, uint32_t error_code)> on_goaway;
  };

  Connection(Role role, Callbacks callbacks);

  // Queue the connection preface (client magic + SETTINGS + window update).
  void Start();

  // Feed bytes from the transport.  An error is a connection error: the owner
  // should flush Output() (it holds the GOAWAY) and close.
  Error Receive(const uint8_t* data, size_t len);

  // Bytes to write to the transport.
  const std::string& Output() const { return out_; }
  size_t OutputOffset() const { return out_off_; }
  void ConsumeOutput(size_t n);
  bool WantsWrite() const { return out_off_ < out_.size(); }

  // Client: start a request.  Returns the stream id (0 when ids ran out or the
  // connection is closing).
  uint32_t OpenStream(const HeaderList& headers, bool end_stream);
  // Server: response headers or trailers.
  Error SendHeaders(uint32_t stream, const HeaderList& headers, bool end_stream);
  // Queue payload; it is framed and released as flow control allows.
  Error SendData(uint32_t stream, std::string&& data, bool end_stream);
  void ResetStream(uint32_t stream, uint32_t error_code);
  void Goaway(uint32_t error_code, const std::string& debug = std::string());
  // Keep-alive probe; PingsAcked() counts the acknowledgements received.
  void SendPing();
  uint64_t PingsSent() const { return pings_sent_; }
  uint64_t PingsAcked() const { return pings_acked_; }

  // Bytes accepted by SendData that flow control has not released yet.
  size_t QueuedBytes() const { return queued_bytes_; }
  size_t OpenStreams() const { return streams_.size(); }
  bool GoawayReceived() const { return goaway_received_; }
  bool GoawaySent() const { return goaway_sent_; }
  uint32_t PeerMaxConcurrentStreams() const { return peer_max_streams_; }

  // Receive-side windows advertised to the peer (set before Start()).
  void SetLocalWindow(uint32_t stream_window, uint32_t connection_window, uint32_t max_frame_size);

 private:
  struct Pending {
    std::string data;
    size_t off = 0;
    bool end_stream = false;
  };
  struct Stream {
    int64_t send_window = 0;
    int64_t recv_consumed = 0;  // bytes delivered since the last WINDOW_UPDATE
    bool local_closed = false;
    bool remote_closed = false;
    bool headers_seen = false;
    std::deque<Pending> pending;
  };

  void WriteFrameHeader(size_t len, uint8_t type, uint8_t flags, uint32_t stream);
  void WriteSettings();
  void WriteWindowUpdate(uint32_t stream, uint32_t increment);
  void WriteHeaderBlock(uint32_t stream, const HeaderList& headers, bool end_stream);
  Error HandleFrame(uint8_t type, uint8_t flags, uint32_t stream, const uint8_t* p, size_t len);
  Error HandleSettings(uint8_t flags, const uint8_t* p, size_t len);
  Error HandleHeaderBlockDone(uint32_t stream, bool end_stream);
  Error ConnectionError(uint32_t code, const std::string& what);
  void Flush();            // release queued DATA within the windows
  void FlushStream(uint32_t id, Stream* s);
  void MaybeErase(uint32_t id);
  Stream* Find(uint32_t id);

  Role role_;
  Callbacks cb_;
  std::string in_;       // unparsed input (partial frame)
  std::string out_;
  size_t out_off_ = 0;
  bool preface_pending_ = false;  // server: still expecting the client magic
  size_t preface_seen_ = 0;

  std::map<uint32_t, Stream> streams_;
  uint32_t next_stream_id_ = 1;
  uint32_t last_peer_stream_ = 0;
  size_t queued_bytes_ = 0;

  // peer settings
  uint32_t peer_initial_window_ = 65535;
  uint32_t peer_max_frame_ = 16384;
  uint32_t peer_max_streams_ = 0x7FFFFFFF;
  int64_t conn_send_window_ = 65535;
  // local settings
  uint32_t local_stream_window_ = 8u << 20;
  uint32_t local_conn_window_ = 64u << 20;
  uint32_t local_max_frame_ = 1u << 20;
  int64_t conn_recv_consumed_ = 0;

  // header block assembly (HEADERS + CONTINUATION)
  uint32_t hb_stream_ = 0;
  bool hb_end_stream_ = false;
  std::string hb_buf_;
  HpackDecoder hpack_;

  bool goaway_sent_ = false;
  bool goaway_received_ = false;
  uint64_t pings_sent_ = 0;
  uint64_t pings_acked_ = 0;
};

}  // namespace h2
}  // namespace tca
