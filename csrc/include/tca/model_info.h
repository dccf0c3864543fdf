// Typed views of the metadata documents the control calls return.
//
// Both C++ clients hand back server / model metadata, model configuration, the repository
// index and statistics as JSON text (the gRPC client renders the protobuf answers with the
// proto field names, so the two protocols read alike; the reference returns protobuf messages
// from `src/c++/library/grpc_client.h:177-237` and JSON text from `http_client.h:200-260`).
// These structs are the part of those documents that programs branch on, parsed once, for
// either protocol (the gRPC `ModelConfigResponse` wrapper `{"config": {...}}` is unwrapped).
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "tca/error.h"
#include "tca/json.h"

namespace tca {

struct TensorInfo {
  std::string name;
  std::string datatype;        // metadata: "FP32"; configuration: "TYPE_FP32" is normalised to "FP32"
  std::vector<int64_t> shape;  // metadata: with the batch dimension (-1) when the model batches; configuration: dims
  std::string format;          // configuration inputs only: FORMAT_NONE | FORMAT_NHWC | FORMAT_NCHW
  bool optional = false;
};

struct ServerInfo {
  std::string name, version;
  std::vector<std::string> extensions;
};

struct ModelMetadataInfo {
  std::string name, platform;
  std::vector<std::string> versions;
  std::vector<TensorInfo> inputs, outputs;
};

struct ModelConfigInfo {
  std::string name, platform, backend;
  int64_t max_batch_size = 0;
  std::vector<TensorInfo> inputs, outputs;
  bool decoupled = false;
  bool sequence_batching = false;
  bool dynamic_batching = false;
};

struct RepositoryEntry {
  std::string name, version, state, reason;
};

namespace detail {

inline std::vector<int64_t> Int64List(const json::Value* v) {
  std::vector<int64_t> out;
  if (v != nullptr && v->IsArray())
    for (const auto& item : v->Items()) out.push_back(item.IsString() ? std::strtoll(item.AsString().c_str(), nullptr, 10) : item.AsInt());
  return out;
}

inline std::vector<TensorInfo> Tensors(const json::Value* list, bool config) {
  std::vector<TensorInfo> out;
  if (list == nullptr || !list->IsArray()) return out;
  for (const auto& item : list->Items()) {
    TensorInfo t;
    t.name = item.GetString("name");
    t.datatype = item.GetString(config ? "data_type" : "datatype");
    if (t.datatype.compare(0, 5, "TYPE_") == 0) t.datatype.erase(0, 5);
    if (t.datatype == "STRING") t.datatype = "BYTES";
    t.shape = Int64List(item.Find(config ? "dims" : "shape"));
    t.format = item.GetString("format", config ? "FORMAT_NONE" : "");
    t.optional = item.GetBool("optional", false);
    out.push_back(std::move(t));
  }
  return out;
}

}  // namespace detail

inline Error ParseServerMetadata(const std::string& text, ServerInfo* out) {
  json::Value doc;
  TCA_RETURN_IF_ERROR(json::Parse(text, &doc));
  out->name = doc.GetString("name");
  out->version = doc.GetString("version");
  out->extensions.clear();
  if (const json::Value* e = doc.Find("extensions"))
    for (const auto& item : e->Items()) out->extensions.push_back(item.AsString());
  return Error::Success();
}

inline Error ParseModelMetadata(const std::string& text, ModelMetadataInfo* out) {
  json::Value doc;
  TCA_RETURN_IF_ERROR(json::Parse(text, &doc));
  if (!doc.IsObject() || doc.Find("name") == nullptr) return Error("not a model metadata document");
  out->name = doc.GetString("name");
  out->platform = doc.GetString("platform");
  out->versions.clear();
  if (const json::Value* v = doc.Find("versions"))
    for (const auto& item : v->Items()) out->versions.push_back(item.AsString());
  out->inputs = detail::Tensors(doc.Find("inputs"), false);
  out->outputs = detail::Tensors(doc.Find("outputs"), false);
  return Error::Success();
}

inline Error ParseModelConfig(const std::string& text, ModelConfigInfo* out) {
  json::Value doc;
  TCA_RETURN_IF_ERROR(json::Parse(text, &doc));
  const json::Value* cfg = doc.Find("config") != nullptr ? doc.Find("config") : &doc;  // gRPC wraps it
  if (!cfg->IsObject() || cfg->Find("name") == nullptr) return Error("not a model configuration document");
  out->name = cfg->GetString("name");
  out->platform = cfg->GetString("platform");
  out->backend = cfg->GetString("backend");
  const json::Value* mb = cfg->Find("max_batch_size");
  out->max_batch_size = mb == nullptr ? 0 : (mb->IsString() ? std::strtoll(mb->AsString().c_str(), nullptr, 10) : mb->AsInt());
  out->inputs = detail::Tensors(cfg->Find("input"), true);
  out->outputs = detail::Tensors(cfg->Find("output"), true);
  const json::Value* policy = cfg->Find("model_transaction_policy");
  out->decoupled = policy != nullptr && policy->GetBool("decoupled", false);
  out->sequence_batching = cfg->Find("sequence_batching") != nullptr;
  out->dynamic_batching = cfg->Find("dynamic_batching") != nullptr;
  return Error::Success();
}

// HTTP: a JSON array of entries; gRPC: {"models": [...]}.
inline Error ParseRepositoryIndex(const std::string& text, std::vector<RepositoryEntry>* out) {
  json::Value doc;
  TCA_RETURN_IF_ERROR(json::Parse(text, &doc));
  const json::Value* list = doc.IsArray() ? &doc : doc.Find("models");
  out->clear();
  if (list == nullptr) return doc.IsObject() ? Error::Success() : Error("not a repository index document");
  for (const auto& item : list->Items()) {
    RepositoryEntry e;
    e.name = item.GetString("name");
    e.version = item.GetString("version");
    e.state = item.GetString("state");
    e.reason = item.GetString("reason");
    out->push_back(std::move(e));
  }
  return Error::Success();
}

}  // namespace tca
