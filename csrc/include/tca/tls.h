// TLS for the native clients (and the test server's native front) on OpenSSL.
//
// Capability parity: the reference's C++ HTTP client hands `HttpSslOptions`
// to libcurl (`src/c++/library/http_client.h:45-78`, `http_client.cc:2150-2180`)
// and its gRPC client builds `grpc::SslCredentials` from `SslOptions`
// (`src/c++/library/grpc_client.h:43-60`, `grpc_client.cc:104-130`).  Here both
// transports are in-tree, so TLS is one small non-blocking session object that
// sits between the socket and the HTTP/1.1 or HTTP/2 state machine.
//
// OpenSSL types stay out of this header (opaque pointers) so including it does
// not pull <openssl/ssl.h> into user code.
#pragma once

#include <sys/types.h>

#include <memory>
#include <string>
#include <vector>

#include "tca/error.h"

namespace tca {
namespace net {

struct TlsOptions {
  bool server = false;       // accept instead of connect
  std::string ca_file;       // PEM bundle to verify the peer with; empty = the system default locations
  std::string cert_file;     // own certificate (chain); PEM unless cert_der
  std::string key_file;      // own private key; PEM unless key_der
  bool cert_der = false;
  bool key_der = false;
  bool verify_peer = true;   // client: check the server chain; server: REQUIRE a client certificate
  bool verify_host = true;   // client: the certificate must name the host connected to
  std::vector<std::string> alpn;  // protocols offered (client) / accepted (server), e.g. {"h2"}
};

// Shareable between all connections of one client / server (wraps SSL_CTX).
class TlsContext {
 public:
  static Error Create(const TlsOptions& options, std::shared_ptr<TlsContext>* out);
  ~TlsContext();
  TlsContext(const TlsContext&) = delete;
  TlsContext& operator=(const TlsContext&) = delete;
  const TlsOptions& options() const { return options_; }
  void* native() const { return ctx_; }

 private:
  TlsContext() = default;
  TlsOptions options_;
  void* ctx_ = nullptr;
  std::string alpn_wire_;  // length-prefixed protocol list in the form OpenSSL takes
  friend class TlsSession;
};

// One TLS connection over a NON-BLOCKING socket the caller owns.  Every call
// returns immediately; kWouldBlock means "call again when the socket is
// readable (WantsWrite() false) or writable (WantsWrite() true)".
class TlsSession {
 public:
  enum : int { kWouldBlock = -1, kFailed = -2 };

  // `host` is used for SNI and certificate name checks (client side).
  TlsSession(std::shared_ptr<TlsContext> context, const std::string& host, int fd);
  ~TlsSession();
  TlsSession(const TlsSession&) = delete;
  TlsSession& operator=(const TlsSession&) = delete;

  int Handshake();                           // 1 = complete
  ssize_t Read(void* buf, size_t len);       // > 0 bytes, 0 = closed by the peer
  ssize_t Write(const void* buf, size_t len);  // > 0 bytes accepted
  bool WantsWrite() const { return wants_write_; }
  bool HandshakeDone() const { return handshake_done_; }
  const std::string& LastError() const { return last_error_; }
  std::string Alpn() const;                  // protocol the peers agreed on ("" = none)

 private:
  int Classify(int rc, const char* what);
  std::shared_ptr<TlsContext> context_;
  void* ssl_ = nullptr;
  bool wants_write_ = false;
  bool handshake_done_ = false;
  std::string last_error_;
};

// Blocking convenience for code that connects synchronously: drives
// Handshake() with poll() until it completes, fails or `timeout_ms` passes.
Error HandshakeBlocking(TlsSession* session, int fd, int timeout_ms);

}  // namespace net
}  // namespace tca
