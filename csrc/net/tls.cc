#include "tca/tls.h"

#include <openssl/err.h>
#include <openssl/ssl.h>
#include <openssl/x509v3.h>
#include <poll.h>

#include <cerrno>
#include <chrono>
#include <cstring>

namespace tca {
namespace net {

namespace {

std::string ErrorQueue() {
  std::string out;
  unsigned long code;
  char buf[256];
  while ((code = ERR_get_error()) != 0) {
    ERR_error_string_n(code, buf, sizeof(buf));
    if (!out.empty()) out += "; ";
    out += buf;
  }
  return out;
}

bool IsIpLiteral(const std::string& host) {
  if (host.find(':') != std::string::npos) return true;  // IPv6
  if (host.empty()) return false;
  for (char c : host)
    if (!((c >= '0' && c <= '9') || c == '.')) return false;
  return true;
}

// server side ALPN: first protocol of OUR list the client also offers
int SelectAlpn(SSL*, const unsigned char** out, unsigned char* outlen, const unsigned char* in, unsigned int inlen,
               void* arg) {
  const std::string* ours = static_cast<const std::string*>(arg);
  size_t i = 0;
  while (i < ours->size()) {
    const size_t n = static_cast<unsigned char>((*ours)[i]);
    unsigned int j = 0;
    while (j < inlen) {
      const unsigned int m = in[j];
      if (m == n && j + 1 + m <= inlen && std::memcmp(in + j + 1, ours->data() + i + 1, n) == 0) {
        *out = in + j + 1;
        *outlen = static_cast<unsigned char>(m);
        return SSL_TLSEXT_ERR_OK;
      }
      j += 1 + m;
    }
    i += 1 + n;
  }
  return SSL_TLSEXT_ERR_ALERT_FATAL;
}

}  // namespace

Error TlsContext::Create(const TlsOptions& options, std::shared_ptr<TlsContext>* out) {
  std::shared_ptr<TlsContext> self(new TlsContext());
  self->options_ = options;
  ERR_clear_error();
  SSL_CTX* ctx = SSL_CTX_new(options.server ? TLS_server_method() : TLS_client_method());
  if (ctx == nullptr) return Error("TLS: cannot create a context: " + ErrorQueue());
  self->ctx_ = ctx;
  SSL_CTX_set_min_proto_version(ctx, TLS1_2_VERSION);
  SSL_CTX_set_mode(ctx, SSL_MODE_ENABLE_PARTIAL_WRITE | SSL_MODE_ACCEPT_MOVING_WRITE_BUFFER | SSL_MODE_RELEASE_BUFFERS);
  if (!options.ca_file.empty()) {
    if (SSL_CTX_load_verify_locations(ctx, options.ca_file.c_str(), nullptr) != 1)
      return Error("TLS: cannot load CA certificates from '" + options.ca_file + "': " + ErrorQueue());
  } else if (!options.server) {
    SSL_CTX_set_default_verify_paths(ctx);
  }
  if (!options.cert_file.empty()) {
    const int rc = options.cert_der
                       ? SSL_CTX_use_certificate_file(ctx, options.cert_file.c_str(), SSL_FILETYPE_ASN1)
                       : SSL_CTX_use_certificate_chain_file(ctx, options.cert_file.c_str());
    if (rc != 1) return Error("TLS: cannot load the certificate '" + options.cert_file + "': " + ErrorQueue());
  }
  if (!options.key_file.empty()) {
    if (SSL_CTX_use_PrivateKey_file(ctx, options.key_file.c_str(),
                                    options.key_der ? SSL_FILETYPE_ASN1 : SSL_FILETYPE_PEM) != 1)
      return Error("TLS: cannot load the private key '" + options.key_file + "': " + ErrorQueue());
    if (SSL_CTX_check_private_key(ctx) != 1)
      return Error("TLS: the private key does not match the certificate: " + ErrorQueue());
  }
  if (options.server && (options.cert_file.empty() || options.key_file.empty()))
    return Error("TLS: a server needs a certificate and a private key");
  if (options.server) {
    SSL_CTX_set_verify(ctx, options.verify_peer ? (SSL_VERIFY_PEER | SSL_VERIFY_FAIL_IF_NO_PEER_CERT) : SSL_VERIFY_NONE,
                       nullptr);
  } else {
    SSL_CTX_set_verify(ctx, options.verify_peer ? SSL_VERIFY_PEER : SSL_VERIFY_NONE, nullptr);
  }
  for (const auto& p : options.alpn) {
    if (p.empty() || p.size() > 255) return Error("TLS: invalid ALPN protocol name");
    self->alpn_wire_.push_back(static_cast<char>(p.size()));
    self->alpn_wire_ += p;
  }
  if (!self->alpn_wire_.empty()) {
    if (options.server) {
      SSL_CTX_set_alpn_select_cb(ctx, SelectAlpn, &self->alpn_wire_);
    } else if (SSL_CTX_set_alpn_protos(ctx, reinterpret_cast<const unsigned char*>(self->alpn_wire_.data()),
                                       static_cast<unsigned int>(self->alpn_wire_.size())) != 0) {
      return Error("TLS: cannot set the ALPN protocol list");
    }
  }
  *out = std::move(self);
  return Error::Success();
}

TlsContext::~TlsContext() {
  if (ctx_ != nullptr) SSL_CTX_free(static_cast<SSL_CTX*>(ctx_));
}

TlsSession::TlsSession(std::shared_ptr<TlsContext> context, const std::string& host, int fd)
    : context_(std::move(context)) {
  SSL* ssl = SSL_new(static_cast<SSL_CTX*>(context_->ctx_));
  ssl_ = ssl;
  if (ssl == nullptr) {
    last_error_ = "TLS: cannot create a session: " + ErrorQueue();
    return;
  }
  SSL_set_fd(ssl, fd);
  const TlsOptions& opt = context_->options_;
  if (opt.server) {
    SSL_set_accept_state(ssl);
  } else {
    SSL_set_connect_state(ssl);
    if (!host.empty() && !IsIpLiteral(host)) SSL_set_tlsext_host_name(ssl, host.c_str());
    if (opt.verify_peer && opt.verify_host && !host.empty()) {
      X509_VERIFY_PARAM* param = SSL_get0_param(ssl);
      if (IsIpLiteral(host)) X509_VERIFY_PARAM_set1_ip_asc(param, host.c_str());
      else X509_VERIFY_PARAM_set1_host(param, host.c_str(), host.size());
    }
  }
}

TlsSession::~TlsSession() {
  if (ssl_ != nullptr) {
    SSL* ssl = static_cast<SSL*>(ssl_);
    if (handshake_done_) SSL_shutdown(ssl);  // one-way close_notify, best effort on a non-blocking socket
    SSL_free(ssl);
  }
}

int TlsSession::Classify(int rc, const char* what) {
  SSL* ssl = static_cast<SSL*>(ssl_);
  const int err = SSL_get_error(ssl, rc);
  switch (err) {
    case SSL_ERROR_WANT_READ:
      wants_write_ = false;
      return kWouldBlock;
    case SSL_ERROR_WANT_WRITE:
      wants_write_ = true;
      return kWouldBlock;
    case SSL_ERROR_ZERO_RETURN:
      return 0;
    case SSL_ERROR_SYSCALL: {
      const std::string queued = ErrorQueue();
      if (queued.empty() && (rc == 0 || errno == 0)) return 0;  // EOF without close_notify
      last_error_ = std::string("TLS ") + what + " failed: " + (queued.empty() ? std::strerror(errno) : queued);
      return kFailed;
    }
    default: {
      std::string detail = ErrorQueue();
      const long verify = SSL_get_verify_result(ssl);
      if (verify != X509_V_OK) detail += std::string(" (certificate verification: ") + X509_verify_cert_error_string(verify) + ")";
      last_error_ = std::string("TLS ") + what + " failed: " + detail;
      return kFailed;
    }
  }
}

int TlsSession::Handshake() {
  if (ssl_ == nullptr) return kFailed;
  if (handshake_done_) return 1;
  ERR_clear_error();
  const int rc = SSL_do_handshake(static_cast<SSL*>(ssl_));
  if (rc == 1) {
    handshake_done_ = true;
    return 1;
  }
  const int what = Classify(rc, "handshake");
  if (what == 0) {
    last_error_ = "TLS handshake failed: connection closed by the peer";
    return kFailed;
  }
  return what;
}

ssize_t TlsSession::Read(void* buf, size_t len) {
  if (ssl_ == nullptr) return kFailed;
  ERR_clear_error();
  const int want = len > 0x40000000u ? 0x40000000 : static_cast<int>(len);
  const int rc = SSL_read(static_cast<SSL*>(ssl_), buf, want);
  if (rc > 0) return rc;
  return Classify(rc, "read");
}

ssize_t TlsSession::Write(const void* buf, size_t len) {
  if (ssl_ == nullptr) return kFailed;
  if (len == 0) return 0;
  ERR_clear_error();
  const int want = len > 0x40000000u ? 0x40000000 : static_cast<int>(len);
  const int rc = SSL_write(static_cast<SSL*>(ssl_), buf, want);
  if (rc > 0) return rc;
  const int what = Classify(rc, "write");
  if (what == 0) {
    last_error_ = "TLS write failed: connection closed by the peer";
    return kFailed;
  }
  return what;
}

std::string TlsSession::Alpn() const {
  if (ssl_ == nullptr) return std::string();
  const unsigned char* data = nullptr;
  unsigned int len = 0;
  SSL_get0_alpn_selected(static_cast<SSL*>(ssl_), &data, &len);
  return data ? std::string(reinterpret_cast<const char*>(data), len) : std::string();
}

Error HandshakeBlocking(TlsSession* session, int fd, int timeout_ms) {
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms);
  while (true) {
    const int rc = session->Handshake();
    if (rc == 1) return Error::Success();
    if (rc == TlsSession::kFailed) return Error(session->LastError());
    const auto left = std::chrono::duration_cast<std::chrono::milliseconds>(deadline - std::chrono::steady_clock::now()).count();
    if (left <= 0) return Error("TLS handshake timed out");
    pollfd p{fd, static_cast<short>(session->WantsWrite() ? POLLOUT : POLLIN), 0};
    const int n = poll(&p, 1, static_cast<int>(left));
    if (n < 0 && errno != EINTR) return Error(std::string("poll failed during the TLS handshake: ") + std::strerror(errno));
  }
}

}  // namespace net
}  // namespace tca
