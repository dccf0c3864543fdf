// Shared helpers of the C++ examples: flag parsing (-v, -u URL, -H 'K:V', -t
// client timeout in microseconds, the TLS flags of the reference's
// simple_http_infer_client / simple_grpc_infer_client) and error handling.
//   HTTP:  -u https://host:port  --verify-peer 0|1  --verify-host 0|2  --ca-certs FILE  --cert-file FILE  --key-file FILE
//   gRPC:  --ssl  --root-certificates FILE  --private-key FILE  --certificate-chain FILE
#pragma once

#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "tca/grpc_client.h"
#include "tca/http_client.h"

namespace tc = tca::client;

#define FAIL_IF_ERR(X, MSG)                                                       \
  do {                                                                            \
    const tca::Error err__ = (X);                                                 \
    if (!err__.IsOk()) {                                                          \
      std::cerr << "error: " << (MSG) << ": " << err__.Message() << std::endl;    \
      std::exit(1);                                                               \
    }                                                                             \
  } while (0)

#define FAIL(MSG)                                  \
  do {                                             \
    std::cerr << "error: " << MSG << std::endl;    \
    std::exit(1);                                  \
  } while (0)

struct CommonFlags {
  bool verbose = false;
  std::string url = "localhost:8000";
  tc::Headers headers;
  uint64_t client_timeout_us = 0;
  tc::HttpSslOptions http_ssl;
  bool grpc_ssl = false;
  tc::SslOptions grpc_ssl_options;
  std::vector<std::string> rest;  // unparsed arguments, in order
};

inline CommonFlags ParseCommonFlags(int argc, char** argv) {
  CommonFlags f;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "-v") {
      f.verbose = true;
    } else if (a == "-u" && i + 1 < argc) {
      f.url = argv[++i];
    } else if (a == "-t" && i + 1 < argc) {
      f.client_timeout_us = std::strtoull(argv[++i], nullptr, 10);
    } else if (a == "-H" && i + 1 < argc) {
      const std::string h = argv[++i];
      const size_t colon = h.find(':');
      if (colon != std::string::npos) f.headers[h.substr(0, colon)] = h.substr(colon + 1);
    } else if (a == "--verify-peer" && i + 1 < argc) {
      f.http_ssl.verify_peer = std::atol(argv[++i]);
    } else if (a == "--verify-host" && i + 1 < argc) {
      f.http_ssl.verify_host = std::atol(argv[++i]);
    } else if (a == "--ca-certs" && i + 1 < argc) {
      f.http_ssl.ca_info = argv[++i];
    } else if (a == "--cert-file" && i + 1 < argc) {
      f.http_ssl.cert = argv[++i];
    } else if (a == "--key-file" && i + 1 < argc) {
      f.http_ssl.key = argv[++i];
    } else if (a == "--ssl") {
      f.grpc_ssl = true;
    } else if (a == "--root-certificates" && i + 1 < argc) {
      f.grpc_ssl_options.root_certificates = argv[++i];
    } else if (a == "--private-key" && i + 1 < argc) {
      f.grpc_ssl_options.private_key = argv[++i];
    } else if (a == "--certificate-chain" && i + 1 < argc) {
      f.grpc_ssl_options.certificate_chain = argv[++i];
    } else {
      f.rest.push_back(a);
    }
  }
  return f;
}

inline std::unique_ptr<tc::InferenceServerHttpClient> MakeClient(const CommonFlags& f) {
  std::unique_ptr<tc::InferenceServerHttpClient> client;
  FAIL_IF_ERR(tc::InferenceServerHttpClient::Create(&client, f.url, f.verbose, f.http_ssl), "unable to create http client");
  return client;
}

inline std::unique_ptr<tc::InferenceServerGrpcClient> MakeGrpcClient(
    const CommonFlags& f, const tc::KeepAliveOptions& keepalive = tc::KeepAliveOptions(), bool cached = true) {
  std::unique_ptr<tc::InferenceServerGrpcClient> client;
  // -u defaults to the HTTP port: the gRPC examples move it to 8001
  const std::string url = f.url == "localhost:8000" ? "localhost:8001" : f.url;
  FAIL_IF_ERR(tc::InferenceServerGrpcClient::Create(&client, url, f.verbose, f.grpc_ssl, f.grpc_ssl_options, keepalive, cached),
              "unable to create grpc client");
  return client;
}

template <typename T>
std::unique_ptr<T> Own(T* p) {
  return std::unique_ptr<T>(p);
}

inline std::unique_ptr<tc::InferInput> NewInput(const std::string& name, const std::vector<int64_t>& shape,
                                                const std::string& datatype) {
  tc::InferInput* p = nullptr;
  FAIL_IF_ERR(tc::InferInput::Create(&p, name, shape, datatype), "unable to create input " + name);
  return Own(p);
}

inline std::unique_ptr<tc::InferRequestedOutput> NewOutput(const std::string& name, size_t class_count = 0) {
  tc::InferRequestedOutput* p = nullptr;
  FAIL_IF_ERR(tc::InferRequestedOutput::Create(&p, name, class_count), "unable to create output " + name);
  return Own(p);
}

// Checks OUTPUT0 = a + b and OUTPUT1 = a - b for the `simple` model.
inline void ValidateSimple(tc::InferResult* result, const std::vector<int32_t>& a, const std::vector<int32_t>& b) {
  const uint8_t* buf = nullptr;
  size_t n = 0;
  FAIL_IF_ERR(result->RawData("OUTPUT0", &buf, &n), "unable to get OUTPUT0");
  if (n != a.size() * 4) FAIL("unexpected OUTPUT0 size " << n);
  const uint8_t* o0 = buf;
  FAIL_IF_ERR(result->RawData("OUTPUT1", &buf, &n), "unable to get OUTPUT1");
  if (n != a.size() * 4) FAIL("unexpected OUTPUT1 size " << n);
  const uint8_t* o1 = buf;
  for (size_t i = 0; i < a.size(); ++i) {
    if (tca::LoadAs<int32_t>(o0, i) != a[i] + b[i]) FAIL("incorrect sum at " << i);
    if (tca::LoadAs<int32_t>(o1, i) != a[i] - b[i]) FAIL("incorrect difference at " << i);
  }
}
