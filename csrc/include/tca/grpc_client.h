// C++ gRPC client for the KServe-v2 / Triton protocol
// (`inference.GRPCInferenceService`).
//
// Capability parity: reference `src/c++/library/grpc_client.h:100-640` /
// `grpc_client.cc` - every control RPC, Infer / AsyncInfer / InferMulti /
// AsyncInferMulti, one bidirectional stream per client (StartStream /
// AsyncStreamInfer / StopStream, decoupled models included), per-request
// timers feeding InferStat, keep-alive options, client timeouts, channel
// sharing between clients of one URL.
//
// Differences by design: transport and protobuf coding are in-tree (tca/h2.h,
// tca/pbwire.h) because the toolchain has neither grpc++ nor protobuf; control
// RPC answers are returned as JSON text with the .proto field names (the shape
// the HTTP client returns) instead of generated message objects; request
// compression (gzip / deflate) is done with zlib; TLS runs on the in-tree
// session layer (tca/tls.h, OpenSSL) with ALPN "h2".
#pragma once

#include <memory>
#include <string>
#include <vector>

#include "tca/common.h"
#include "tca/grpc_channel.h"

namespace tca {
namespace client {

// PEM file paths (reference `grpc_client.h:43-60`); an empty root_certificates uses
// the system trust store.
struct SslOptions {
  std::string root_certificates, private_key, certificate_chain;
};

enum class GrpcCompression { NONE, DEFLATE, GZIP };

class InferenceServerGrpcClient : public InferenceServerClient {
 public:
  ~InferenceServerGrpcClient() override;

  static Error Create(std::unique_ptr<InferenceServerGrpcClient>* client, const std::string& server_url,
                      bool verbose = false, bool use_ssl = false, const SslOptions& ssl_options = SslOptions(),
                      const KeepAliveOptions& keepalive_options = KeepAliveOptions(),
                      bool use_cached_channel = true);

  // The same with explicit channel settings (message size limits, user agent, keep-alive, HTTP/2 windows):
  // what the reference takes as grpc::ChannelArguments (`grpc_client.h:146-150`).
  static Error Create(std::unique_ptr<InferenceServerGrpcClient>* client, const std::string& server_url,
                      const ChannelOptions& channel_options, bool verbose = false, bool use_ssl = false,
                      const SslOptions& ssl_options = SslOptions(), bool use_cached_channel = true);

  Error IsServerLive(bool* live, const Headers& headers = Headers(), uint64_t timeout_us = 0);
  Error IsServerReady(bool* ready, const Headers& headers = Headers(), uint64_t timeout_us = 0);
  Error IsModelReady(bool* ready, const std::string& model_name, const std::string& model_version = "",
                     const Headers& headers = Headers(), uint64_t timeout_us = 0);
  // Every control call takes `timeout_us` last: the time the call may take, in microseconds (0 = no limit);
  // a call that runs out of it fails with "Deadline Exceeded" (reference `grpc_client.h:157-456`, where the
  // parameter is spelled timeout_ms and documented as microseconds).
  // The following return the response message as JSON text.
  Error ServerMetadata(std::string* server_metadata, const Headers& headers = Headers(), uint64_t timeout_us = 0);
  Error ModelMetadata(std::string* model_metadata, const std::string& model_name,
                      const std::string& model_version = "", const Headers& headers = Headers(), uint64_t timeout_us = 0);
  Error ModelConfig(std::string* model_config, const std::string& model_name, const std::string& model_version = "",
                    const Headers& headers = Headers(), uint64_t timeout_us = 0);
  Error ModelRepositoryIndex(std::string* repository_index, const Headers& headers = Headers(), uint64_t timeout_us = 0);
  Error LoadModel(const std::string& model_name, const Headers& headers = Headers(),
                  const std::string& config = std::string(),
                  const std::map<std::string, std::vector<char>>& files = {}, uint64_t timeout_us = 0);
  Error UnloadModel(const std::string& model_name, const Headers& headers = Headers(), uint64_t timeout_us = 0);
  Error ModelInferenceStatistics(std::string* infer_stat, const std::string& model_name = "",
                                 const std::string& model_version = "", const Headers& headers = Headers(), uint64_t timeout_us = 0);
  Error UpdateTraceSettings(std::string* response, const std::string& model_name = "",
                            const std::map<std::string, std::vector<std::string>>& settings = {},
                            const Headers& headers = Headers(), uint64_t timeout_us = 0);
  Error GetTraceSettings(std::string* settings, const std::string& model_name = "",
                         const Headers& headers = Headers(), uint64_t timeout_us = 0);
  Error UpdateLogSettings(std::string* response, const std::string& settings_json,
                          const Headers& headers = Headers(), uint64_t timeout_us = 0);
  Error GetLogSettings(std::string* settings, const Headers& headers = Headers(), uint64_t timeout_us = 0);
  Error SystemSharedMemoryStatus(std::string* status, const std::string& region_name = "",
                                 const Headers& headers = Headers(), uint64_t timeout_us = 0);
  Error RegisterSystemSharedMemory(const std::string& name, const std::string& key, size_t byte_size,
                                   size_t offset = 0, const Headers& headers = Headers(), uint64_t timeout_us = 0);
  Error UnregisterSystemSharedMemory(const std::string& name = "", const Headers& headers = Headers(), uint64_t timeout_us = 0);
  // Device shared memory: `raw_handle` is the 64-byte HIP IPC handle.
  Error CudaSharedMemoryStatus(std::string* status, const std::string& region_name = "",
                               const Headers& headers = Headers(), uint64_t timeout_us = 0);
  Error RegisterCudaSharedMemory(const std::string& name, const void* raw_handle, size_t device_id, size_t byte_size,
                                 const Headers& headers = Headers(), uint64_t timeout_us = 0);
  Error UnregisterCudaSharedMemory(const std::string& name = "", const Headers& headers = Headers(), uint64_t timeout_us = 0);

  Error Infer(InferResult** result, const InferOptions& options, const std::vector<InferInput*>& inputs,
              const std::vector<const InferRequestedOutput*>& outputs = {}, const Headers& headers = Headers(),
              GrpcCompression compression_algorithm = GrpcCompression::NONE);
  Error AsyncInfer(OnCompleteFn callback, const InferOptions& options, const std::vector<InferInput*>& inputs,
                   const std::vector<const InferRequestedOutput*>& outputs = {}, const Headers& headers = Headers(),
                   GrpcCompression compression_algorithm = GrpcCompression::NONE);
  Error InferMulti(std::vector<InferResult*>* results, const std::vector<InferOptions>& options,
                   const std::vector<std::vector<InferInput*>>& inputs,
                   const std::vector<std::vector<const InferRequestedOutput*>>& outputs = {},
                   const Headers& headers = Headers(),
                   GrpcCompression compression_algorithm = GrpcCompression::NONE);
  Error AsyncInferMulti(OnMultiCompleteFn callback, const std::vector<InferOptions>& options,
                        const std::vector<std::vector<InferInput*>>& inputs,
                        const std::vector<std::vector<const InferRequestedOutput*>>& outputs = {},
                        const Headers& headers = Headers(),
                        GrpcCompression compression_algorithm = GrpcCompression::NONE);

  // One bidirectional stream per client.  `callback` runs on the channel's
  // reactor thread once per response; `stream_timeout` (microseconds, 0 = none)
  // bounds the whole stream.
  Error StartStream(OnCompleteFn callback, bool enable_stats = true, uint32_t stream_timeout = 0,
                    const Headers& headers = Headers(),
                    GrpcCompression compression_algorithm = GrpcCompression::NONE);
  Error StopStream();
  Error AsyncStreamInfer(const InferOptions& options, const std::vector<InferInput*>& inputs,
                         const std::vector<const InferRequestedOutput*>& outputs = {});

  // Serialise / parse inference messages without a connection.
  static Error GenerateRequestBody(std::string* request, const InferOptions& options,
                                   const std::vector<InferInput*>& inputs,
                                   const std::vector<const InferRequestedOutput*>& outputs = {});
  static Error ParseResponseBody(InferResult** result, std::string&& response);

  // Number of live channels in the process-wide cache (reference
  // `grpc_client.cc:1711-1716`).
  static size_t GetNumCachedChannels();

 private:
  InferenceServerGrpcClient(std::shared_ptr<GrpcChannel> channel, bool verbose);
  Error Unary(const std::string& method, const std::string& request, const Headers& headers, uint64_t timeout_us,
              std::string* response, bool compressed = false);
  Error JsonCall(const std::string& method, const std::string& request_json, const Headers& headers,
                 std::string* response_json, uint64_t timeout_us = 0);

  struct StreamState;
  struct AsyncState;
  std::shared_ptr<GrpcChannel> channel_;
  std::shared_ptr<StreamState> stream_;
  // asynchronous calls that have not delivered their result yet: the destructor cancels them and
  // waits for their callbacks, so none runs on a client that is gone (the channel may outlive the
  // client: it is shared through the cache)
  std::shared_ptr<AsyncState> async_;
};

}  // namespace client
}  // namespace tca
