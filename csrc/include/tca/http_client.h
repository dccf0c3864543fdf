// C++ HTTP/REST client for the KServe-v2 / Triton protocol.
//
// Capability parity: reference `src/c++/library/http_client.h:105-649` /
// `http_client.cc` - every control endpoint, Infer / AsyncInfer / InferMulti /
// AsyncInferMulti, request + response compression, static
// GenerateRequestBody / ParseResponseBody, per-request timers feeding
// InferStat.  On top: UpdateLogSettings / GetLogSettings (absent from the
// reference's C++ client) and IsServerReady really queries v2/health/ready.
//
// Thread-safety: like the reference, one client is meant for one thread;
// AsyncInfer callbacks run on the client's reactor thread.
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "tca/common.h"
#include "tca/http_transport.h"
#include "tca/json.h"

namespace tca {
namespace client {

// TLS settings, used when the server url starts with "https://" (reference
// `src/c++/library/http_client.h:45-78`; there the values go to libcurl, here to
// the in-tree TLS session, `tca/tls.h`).
struct HttpSslOptions {
  enum CERTTYPE { CERT_PEM = 0, CERT_DER = 1 };
  enum KEYTYPE { KEY_PEM = 0, KEY_DER = 1 };
  long verify_peer = 1;  // 0 = accept any certificate
  long verify_host = 2;  // 0 = do not compare the certificate's names with the host (curl's 0 / 2 convention)
  std::string ca_info;   // PEM bundle of trusted CAs; empty = system default
  CERTTYPE cert_type = CERT_PEM;
  std::string cert;      // client certificate file
  KEYTYPE key_type = KEY_PEM;
  std::string key;       // client private key file
};

class InferResultHttp;

class InferenceServerHttpClient : public InferenceServerClient {
 public:
  enum class CompressionType { NONE, DEFLATE, GZIP };

  ~InferenceServerHttpClient() override;

  static Error Create(std::unique_ptr<InferenceServerHttpClient>* client, const std::string& server_url,
                      bool verbose = false, const HttpSslOptions& ssl_options = HttpSslOptions());

  Error IsServerLive(bool* live, const Headers& headers = Headers(), const Parameters& query_params = Parameters());
  Error IsServerReady(bool* ready, const Headers& headers = Headers(), const Parameters& query_params = Parameters());
  Error IsModelReady(bool* ready, const std::string& model_name, const std::string& model_version = "",
                     const Headers& headers = Headers(), const Parameters& query_params = Parameters());
  Error ServerMetadata(std::string* server_metadata, const Headers& headers = Headers(),
                       const Parameters& query_params = Parameters());
  Error ModelMetadata(std::string* model_metadata, const std::string& model_name,
                      const std::string& model_version = "", const Headers& headers = Headers(),
                      const Parameters& query_params = Parameters());
  Error ModelConfig(std::string* model_config, const std::string& model_name, const std::string& model_version = "",
                    const Headers& headers = Headers(), const Parameters& query_params = Parameters());
  Error ModelRepositoryIndex(std::string* repository_index, const Headers& headers = Headers(),
                             const Parameters& query_params = Parameters());
  Error LoadModel(const std::string& model_name, const Headers& headers = Headers(),
                  const Parameters& query_params = Parameters(), const std::string& config = std::string(),
                  const std::map<std::string, std::vector<char>>& files = {});
  Error UnloadModel(const std::string& model_name, const Headers& headers = Headers(),
                    const Parameters& query_params = Parameters());
  Error ModelInferenceStatistics(std::string* infer_stat, const std::string& model_name = "",
                                 const std::string& model_version = "", const Headers& headers = Headers(),
                                 const Parameters& query_params = Parameters());
  Error UpdateTraceSettings(std::string* response, const std::string& model_name = "",
                            const std::map<std::string, std::vector<std::string>>& settings = {},
                            const Headers& headers = Headers(), const Parameters& query_params = Parameters());
  Error GetTraceSettings(std::string* settings, const std::string& model_name = "",
                         const Headers& headers = Headers(), const Parameters& query_params = Parameters());
  Error UpdateLogSettings(std::string* response, const std::string& settings_json,
                          const Headers& headers = Headers(), const Parameters& query_params = Parameters());
  Error GetLogSettings(std::string* settings, const Headers& headers = Headers(),
                       const Parameters& query_params = Parameters());
  Error SystemSharedMemoryStatus(std::string* status, const std::string& region_name = "",
                                 const Headers& headers = Headers(), const Parameters& query_params = Parameters());
  Error RegisterSystemSharedMemory(const std::string& name, const std::string& key, size_t byte_size,
                                   size_t offset = 0, const Headers& headers = Headers(),
                                   const Parameters& query_params = Parameters());
  Error UnregisterSystemSharedMemory(const std::string& name = "", const Headers& headers = Headers(),
                                     const Parameters& query_params = Parameters());
  // Device shared memory: `raw_handle` is the 64-byte HIP IPC handle.
  Error CudaSharedMemoryStatus(std::string* status, const std::string& region_name = "",
                               const Headers& headers = Headers(), const Parameters& query_params = Parameters());
  Error RegisterCudaSharedMemory(const std::string& name, const void* raw_handle, size_t device_id, size_t byte_size,
                                 const Headers& headers = Headers(), const Parameters& query_params = Parameters());
  Error UnregisterCudaSharedMemory(const std::string& name = "", const Headers& headers = Headers(),
                                   const Parameters& query_params = Parameters());

  // Build / parse bodies without touching a socket.
  static Error GenerateRequestBody(std::vector<char>* request_body, size_t* header_length,
                                   const InferOptions& options, const std::vector<InferInput*>& inputs,
                                   const std::vector<const InferRequestedOutput*>& outputs = {});
  static Error ParseResponseBody(InferResult** result, const std::vector<char>& response_body,
                                 size_t header_length = 0);

  Error Infer(InferResult** result, const InferOptions& options, const std::vector<InferInput*>& inputs,
              const std::vector<const InferRequestedOutput*>& outputs = {}, const Headers& headers = Headers(),
              const Parameters& query_params = Parameters(),
              CompressionType request_compression_algorithm = CompressionType::NONE,
              CompressionType response_compression_algorithm = CompressionType::NONE);
  Error AsyncInfer(OnCompleteFn callback, const InferOptions& options, const std::vector<InferInput*>& inputs,
                   const std::vector<const InferRequestedOutput*>& outputs = {}, const Headers& headers = Headers(),
                   const Parameters& query_params = Parameters(),
                   CompressionType request_compression_algorithm = CompressionType::NONE,
                   CompressionType response_compression_algorithm = CompressionType::NONE);
  // options/outputs may hold one element (applied to every request) or one per request.
  Error InferMulti(std::vector<InferResult*>* results, const std::vector<InferOptions>& options,
                   const std::vector<std::vector<InferInput*>>& inputs,
                   const std::vector<std::vector<const InferRequestedOutput*>>& outputs = {},
                   const Headers& headers = Headers(), const Parameters& query_params = Parameters(),
                   CompressionType request_compression_algorithm = CompressionType::NONE,
                   CompressionType response_compression_algorithm = CompressionType::NONE);
  Error AsyncInferMulti(OnMultiCompleteFn callback, const std::vector<InferOptions>& options,
                        const std::vector<std::vector<InferInput*>>& inputs,
                        const std::vector<std::vector<const InferRequestedOutput*>>& outputs = {},
                        const Headers& headers = Headers(), const Parameters& query_params = Parameters(),
                        CompressionType request_compression_algorithm = CompressionType::NONE,
                        CompressionType response_compression_algorithm = CompressionType::NONE);

  // Requests submitted through AsyncInfer that have not completed yet.
  size_t AsyncInFlight() const;

 private:
  InferenceServerHttpClient(const HttpEndpoint& ep, bool verbose);
  Error Get(const std::string& path, const Headers& headers, const Parameters& query, std::string* body, int* status);
  Error Post(const std::string& path, const std::string& request, const Headers& headers, const Parameters& query,
             std::string* body, int* status);
  Error PrepareInfer(HttpRequestSpec* spec, const InferOptions& options, const std::vector<InferInput*>& inputs,
                     const std::vector<const InferRequestedOutput*>& outputs, const Headers& headers,
                     const Parameters& query_params, CompressionType req_alg, CompressionType resp_alg);

  HttpEndpoint ep_;
  std::unique_ptr<HttpConn> sync_conn_;
  std::unique_ptr<HttpReactor> reactor_;  // created on the first AsyncInfer
};

}  // namespace client
}  // namespace tca
