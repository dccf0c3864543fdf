// Table-driven protobuf <-> JSON conversion for the `inference.*` messages.
//
// The native gRPC client returns metadata / config / statistics / repository
// answers as JSON text with the .proto field names - the same shape the HTTP
// client returns - instead of generated message classes (the reference returns
// `inference::...Response` objects, `src/c++/library/grpc_client.h:178-330`;
// there is no protobuf C++ runtime in this toolchain).  Rules: 64-bit integers
// are JSON numbers, enums their value names, bytes base64, maps objects,
// fields at their default are omitted.
#pragma once

#include <string>

#include "tca/error.h"
#include "tca/json.h"

namespace tca {
namespace pb {

// Index of a message by full name ("inference.ModelConfigResponse"), -1 if unknown.
int FindMessage(const std::string& full_name);
// RPC method table of inference.GRPCInferenceService.
bool FindMethod(const std::string& method, int* request_msg, int* response_msg);

Error ToJson(int message, const void* data, size_t len, json::Value* out);
Error FromJson(int message, const json::Value& value, std::string* out);

}  // namespace pb
}  // namespace tca
