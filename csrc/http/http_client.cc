#include "tca/http_client.h"

#include <atomic>
#include <cmath>
#include <cstring>
#include <iostream>
#include <sstream>

#include <zlib.h>

#include "tca/base64.h"
#include "tca/codec.h"

namespace tca {
namespace client {

namespace {

//------------------------------------------------------------------------------
// URL helpers
//------------------------------------------------------------------------------
std::string UrlEncode(const std::string& s, bool plus_for_space) {
  static const char* hex = "0123456789ABCDEF";
  std::string out;
  for (unsigned char c : s) {
    if (std::isalnum(c) || c == '-' || c == '_' || c == '.' || c == '~') {
      out.push_back(static_cast<char>(c));
    } else if (c == ' ' && plus_for_space) {
      out.push_back('+');
    } else {
      out.push_back('%');
      out.push_back(hex[c >> 4]);
      out.push_back(hex[c & 15]);
    }
  }
  return out;
}

std::string QueryString(const Parameters& q) {
  std::string out;
  for (const auto& kv : q) {
    out += out.empty() ? "?" : "&";
    out += UrlEncode(kv.first, true) + "=" + UrlEncode(kv.second, true);
  }
  return out;
}

std::string ModelPath(const std::string& model, const std::string& version, const std::string& suffix) {
  std::string p = "/v2/models/" + UrlEncode(model, false);
  if (!version.empty()) p += "/versions/" + version;
  return p + suffix;
}

//------------------------------------------------------------------------------
// compression (zlib): "deflate" on the wire is a zlib stream, gzip is windowBits 15+16
//------------------------------------------------------------------------------
Error Compress(bool gzip, const std::vector<std::pair<const uint8_t*, size_t>>& pieces, std::string* out) {
  z_stream zs;
  std::memset(&zs, 0, sizeof(zs));
  if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, gzip ? 15 + 16 : 15, 8, Z_DEFAULT_STRATEGY) != Z_OK)
    return Error("failed to initialize state for compression");
  out->clear();
  char buf[65536];
  for (size_t i = 0; i <= pieces.size(); ++i) {
    const bool last = i == pieces.size();
    zs.next_in = last ? nullptr : const_cast<Bytef*>(pieces[i].first);
    zs.avail_in = last ? 0 : static_cast<uInt>(pieces[i].second);
    do {
      zs.next_out = reinterpret_cast<Bytef*>(buf);
      zs.avail_out = sizeof(buf);
      int rc = deflate(&zs, last ? Z_FINISH : Z_NO_FLUSH);
      if (rc == Z_STREAM_ERROR) {
        deflateEnd(&zs);
        return Error("encountered inconsistent stream state during compression");
      }
      out->append(buf, sizeof(buf) - zs.avail_out);
    } while (zs.avail_out == 0);
  }
  deflateEnd(&zs);
  return Error::Success();
}

Error Decompress(const std::string& in, std::string* out) {
  z_stream zs;
  std::memset(&zs, 0, sizeof(zs));
  if (inflateInit2(&zs, 15 + 32) != Z_OK)  // +32: auto-detect zlib / gzip
    return Error("failed to initialize state for decompression");
  zs.next_in = reinterpret_cast<Bytef*>(const_cast<char*>(in.data()));
  zs.avail_in = static_cast<uInt>(in.size());
  out->clear();
  char buf[65536];
  int rc;
  do {
    zs.next_out = reinterpret_cast<Bytef*>(buf);
    zs.avail_out = sizeof(buf);
    rc = inflate(&zs, Z_NO_FLUSH);
    if (rc != Z_OK && rc != Z_STREAM_END) {
      inflateEnd(&zs);
      return Error("failed to decompress the response body");
    }
    out->append(buf, sizeof(buf) - zs.avail_out);
  } while (rc != Z_STREAM_END);
  inflateEnd(&zs);
  return Error::Success();
}

//------------------------------------------------------------------------------
// JSON <-> binary tensor conversion
//------------------------------------------------------------------------------
template <typename T>
void AppendValues(const std::string& raw, json::Value* arr) {
  const size_t n = raw.size() / sizeof(T);
  arr->Reserve(n);
  for (size_t i = 0; i < n; ++i) {
    T v;
    std::memcpy(&v, raw.data() + i * sizeof(T), sizeof(T));
    arr->Append(json::Value(v));
  }
}

// Binary tensor bytes -> JSON "data" array (reference: ConvertBinaryInputToJSON,
// `http_client.cc:606-678`).  FP16 / BF16 have no JSON representation.
Error BinaryToJson(const std::string& datatype, const std::string& raw, const std::string& name, json::Value* arr) {
  *arr = json::Value::Array();
  const DataType dt = ParseDataType(datatype);
  switch (dt) {
    case DataType::BOOL: {
      for (unsigned char c : raw) arr->Append(json::Value(c != 0));
      break;
    }
    case DataType::UINT8: for (unsigned char c : raw) arr->Append(json::Value(static_cast<unsigned>(c))); break;
    case DataType::INT8: for (char c : raw) arr->Append(json::Value(static_cast<int>(static_cast<signed char>(c)))); break;
    case DataType::UINT16: { const size_t n = raw.size() / 2; for (size_t i = 0; i < n; ++i) { uint16_t v; std::memcpy(&v, raw.data() + 2 * i, 2); arr->Append(json::Value(static_cast<unsigned>(v))); } break; }
    case DataType::INT16: { const size_t n = raw.size() / 2; for (size_t i = 0; i < n; ++i) { int16_t v; std::memcpy(&v, raw.data() + 2 * i, 2); arr->Append(json::Value(static_cast<int>(v))); } break; }
    case DataType::UINT32: AppendValues<uint32_t>(raw, arr); break;
    case DataType::INT32: AppendValues<int32_t>(raw, arr); break;
    case DataType::UINT64: { const size_t n = raw.size() / 8; for (size_t i = 0; i < n; ++i) { uint64_t v; std::memcpy(&v, raw.data() + 8 * i, 8); arr->Append(json::Value(static_cast<unsigned long long>(v))); } break; }
    case DataType::INT64: { const size_t n = raw.size() / 8; for (size_t i = 0; i < n; ++i) { int64_t v; std::memcpy(&v, raw.data() + 8 * i, 8); arr->Append(json::Value(static_cast<long long>(v))); } break; }
    case DataType::FP32: AppendValues<float>(raw, arr); break;
    case DataType::FP64: AppendValues<double>(raw, arr); break;
    case DataType::BYTES: {
      std::vector<std::string> items;
      TCA_RETURN_IF_ERROR(BytesDeserialize(reinterpret_cast<const uint8_t*>(raw.data()), raw.size(), -1, &items));
      for (auto& s : items) arr->Append(json::Value(std::move(s)));
      break;
    }
    case DataType::FP16:
      return Error("datatype FP16 of input '" + name + "' cannot be represented in JSON; send it as binary data");
    case DataType::BF16:
      return Error("datatype BF16 of input '" + name + "' cannot be represented in JSON; send it as binary data");
    default:
      return Error("invalid datatype '" + datatype + "' for input '" + name + "'");
  }
  return Error::Success();
}

template <typename T, typename Get>
void PackValues(const json::Value& data, std::string* out, Get get) {
  std::vector<const json::Value*> stack{&data};
  // flatten nested arrays depth-first (row-major)
  std::vector<const json::Value*> flat;
  while (!stack.empty()) {
    const json::Value* v = stack.back();
    stack.pop_back();
    if (v->IsArray()) {
      for (size_t i = v->Size(); i-- > 0;) stack.push_back(&(*v)[i]);
    } else {
      flat.push_back(v);
    }
  }
  out->resize(flat.size() * sizeof(T));
  for (size_t i = 0; i < flat.size(); ++i) {
    const T val = static_cast<T>(get(*flat[i]));
    std::memcpy(&(*out)[i * sizeof(T)], &val, sizeof(T));
  }
}

// JSON "data" array -> binary tensor bytes (reference: ConvertJSONOutputToBinary,
// `http_client.cc:1155-1281`; FP64 is packed as 8-byte doubles here).
Error JsonToBinary(const std::string& datatype, const json::Value& data, const std::string& name, std::string* out) {
  auto as_int = [](const json::Value& v) { return v.AsInt(); };
  auto as_uint = [](const json::Value& v) { return v.AsUInt(); };
  auto as_dbl = [](const json::Value& v) { return v.AsDouble(); };
  switch (ParseDataType(datatype)) {
    case DataType::BOOL: PackValues<uint8_t>(data, out, [](const json::Value& v) { return v.AsBool() ? 1 : 0; }); break;
    case DataType::UINT8: PackValues<uint8_t>(data, out, as_uint); break;
    case DataType::INT8: PackValues<int8_t>(data, out, as_int); break;
    case DataType::UINT16: PackValues<uint16_t>(data, out, as_uint); break;
    case DataType::INT16: PackValues<int16_t>(data, out, as_int); break;
    case DataType::UINT32: PackValues<uint32_t>(data, out, as_uint); break;
    case DataType::INT32: PackValues<int32_t>(data, out, as_int); break;
    case DataType::UINT64: PackValues<uint64_t>(data, out, as_uint); break;
    case DataType::INT64: PackValues<int64_t>(data, out, as_int); break;
    case DataType::FP32: PackValues<float>(data, out, as_dbl); break;
    case DataType::FP64: PackValues<double>(data, out, as_dbl); break;
    case DataType::BYTES: {
      out->clear();
      std::vector<const json::Value*> stack{&data};
      while (!stack.empty()) {
        const json::Value* v = stack.back();
        stack.pop_back();
        if (v->IsArray()) {
          for (size_t i = v->Size(); i-- > 0;) stack.push_back(&(*v)[i]);
        } else {
          const std::string& s = v->AsString();
          BytesAppend(s.data(), static_cast<uint32_t>(s.size()), out);
        }
      }
      break;
    }
    case DataType::FP16:
      return Error("datatype FP16 of output '" + name + "' cannot be parsed from JSON; request binary data");
    case DataType::BF16:
      return Error("datatype BF16 of output '" + name + "' cannot be parsed from JSON; request binary data");
    default:
      return Error("invalid datatype '" + datatype + "' for output '" + name + "'");
  }
  return Error::Success();
}

//------------------------------------------------------------------------------
// request JSON
//------------------------------------------------------------------------------
Error BuildRequestJson(const InferOptions& options, const std::vector<InferInput*>& inputs,
                       const std::vector<const InferRequestedOutput*>& outputs, std::string* json_out,
                       bool* any_binary) {
  json::Value doc = json::Value::Object();
  if (!options.request_id_.empty()) doc["id"] = options.request_id_;
  json::Value params = json::Value::Object();
  if (options.sequence_id_ != 0 || !options.sequence_id_str_.empty()) {
    if (!options.sequence_id_str_.empty()) params["sequence_id"] = options.sequence_id_str_;
    else params["sequence_id"] = static_cast<unsigned long long>(options.sequence_id_);
    params["sequence_start"] = options.sequence_start_;
    params["sequence_end"] = options.sequence_end_;
  }
  if (options.priority_ != 0) params["priority"] = static_cast<unsigned long long>(options.priority_);
  if (options.server_timeout_ != 0) params["timeout"] = static_cast<unsigned long long>(options.server_timeout_);
  for (const auto& kv : options.request_parameters) {
    static const char* reserved[] = {"sequence_id", "sequence_start", "sequence_end", "priority", "binary_data_output"};
    for (const char* r : reserved)
      if (kv.first == r) return Error("Parameter \"" + kv.first + "\" is a reserved parameter and cannot be specified.");
    switch (kv.second.type) {
      case RequestParameter::Type::STRING: params[kv.first] = kv.second.str_value; break;
      case RequestParameter::Type::INT: params[kv.first] = static_cast<long long>(kv.second.int_value); break;
      case RequestParameter::Type::UINT: params[kv.first] = static_cast<unsigned long long>(kv.second.uint_value); break;
      case RequestParameter::Type::BOOL: params[kv.first] = kv.second.bool_value; break;
      case RequestParameter::Type::DOUBLE: params[kv.first] = kv.second.double_value; break;
    }
  }
  *any_binary = false;
  json::Value ins = json::Value::Array();
  for (const InferInput* in : inputs) {
    json::Value t = json::Value::Object();
    t["name"] = in->Name();
    json::Value shape = json::Value::Array();
    for (int64_t d : in->Shape()) shape.Append(json::Value(static_cast<long long>(d)));
    t["shape"] = std::move(shape);
    t["datatype"] = in->Datatype();
    if (in->IsSharedMemory()) {
      std::string region;
      size_t size = 0, offset = 0;
      in->SharedMemoryInfo(&region, &size, &offset);
      json::Value p = json::Value::Object();
      p["shared_memory_region"] = region;
      p["shared_memory_byte_size"] = static_cast<unsigned long long>(size);
      if (offset != 0) p["shared_memory_offset"] = static_cast<unsigned long long>(offset);
      t["parameters"] = std::move(p);
    } else if (in->BinaryData()) {
      size_t size = 0;
      in->ByteSize(&size);
      json::Value p = json::Value::Object();
      p["binary_data_size"] = static_cast<unsigned long long>(size);
      t["parameters"] = std::move(p);
      *any_binary = true;
    } else {
      std::string raw;
      in->Gather(&raw);
      json::Value data;
      TCA_RETURN_IF_ERROR(BinaryToJson(in->Datatype(), raw, in->Name(), &data));
      t["data"] = std::move(data);
    }
    ins.Append(std::move(t));
  }
  doc["inputs"] = std::move(ins);
  if (!outputs.empty()) {
    json::Value outs = json::Value::Array();
    for (const InferRequestedOutput* o : outputs) {
      json::Value t = json::Value::Object();
      t["name"] = o->Name();
      json::Value p = json::Value::Object();
      if (o->ClassificationCount() != 0) p["classification"] = static_cast<unsigned long long>(o->ClassificationCount());
      if (o->IsSharedMemory()) {
        std::string region;
        size_t size = 0, offset = 0;
        o->SharedMemoryInfo(&region, &size, &offset);
        p["binary_data"] = false;
        p["shared_memory_region"] = region;
        p["shared_memory_byte_size"] = static_cast<unsigned long long>(size);
        if (offset != 0) p["shared_memory_offset"] = static_cast<unsigned long long>(offset);
      } else {
        p["binary_data"] = o->BinaryData();
      }
      t["parameters"] = std::move(p);
      outs.Append(std::move(t));
    }
    doc["outputs"] = std::move(outs);
  } else {
    params["binary_data_output"] = true;
  }
  if (params.Size() != 0) doc["parameters"] = std::move(params);
  *json_out = doc.Dump();
  return Error::Success();
}

}  // namespace

//==============================================================================
// InferResultHttp
//==============================================================================
class InferResultHttp : public InferResult {
 public:
  // Takes ownership of `body`.  `header_length` == 0 means the whole body is JSON.
  static Error Create(InferResult** result, std::string body, size_t header_length, int http_status,
                      const Error& transport_error) {
    auto* r = new InferResultHttp();
    r->body_ = std::move(body);
    r->Parse(header_length, http_status, transport_error);
    *result = r;
    return Error::Success();
  }

  Error ModelName(std::string* name) const override { TCA_RETURN_IF_ERROR(status_); *name = doc_.GetString("model_name"); return Error::Success(); }
  Error ModelVersion(std::string* version) const override { TCA_RETURN_IF_ERROR(status_); *version = doc_.GetString("model_version"); return Error::Success(); }
  Error Id(std::string* id) const override { TCA_RETURN_IF_ERROR(status_); *id = doc_.GetString("id"); return Error::Success(); }

  Error Shape(const std::string& output_name, std::vector<int64_t>* shape) const override {
    const json::Value* o = nullptr;
    TCA_RETURN_IF_ERROR(Find(output_name, &o));
    shape->clear();
    if (const json::Value* s = o->Find("shape"))
      for (const auto& d : s->Items()) shape->push_back(d.AsInt());
    return Error::Success();
  }
  Error Datatype(const std::string& output_name, std::string* datatype) const override {
    const json::Value* o = nullptr;
    TCA_RETURN_IF_ERROR(Find(output_name, &o));
    *datatype = o->GetString("datatype");
    return Error::Success();
  }
  Error RawData(const std::string& output_name, const uint8_t** buf, size_t* byte_size) const override {
    const json::Value* o = nullptr;
    TCA_RETURN_IF_ERROR(Find(output_name, &o));
    auto it = spans_.find(output_name);
    if (it != spans_.end()) {
      *buf = reinterpret_cast<const uint8_t*>(it->second.first);
      *byte_size = it->second.second;
      return Error::Success();
    }
    auto jt = converted_.find(output_name);
    if (jt != converted_.end()) {
      *buf = reinterpret_cast<const uint8_t*>(jt->second.data());
      *byte_size = jt->second.size();
      return Error::Success();
    }
    return Error("The response does not contain data for output '" + output_name + "' (it may live in shared memory)");
  }
  Error StringData(const std::string& output_name, std::vector<std::string>* string_result) const override {
    std::string dt;
    TCA_RETURN_IF_ERROR(Datatype(output_name, &dt));
    if (dt != "BYTES") return Error("This function supports tensors with datatype 'BYTES', requested datatype '" + dt + "'");
    const uint8_t* buf = nullptr;
    size_t size = 0;
    TCA_RETURN_IF_ERROR(RawData(output_name, &buf, &size));
    return BytesDeserialize(buf, size, -1, string_result);
  }
  Error IsFinalResponse(bool* is_final_response) const override { *is_final_response = true; return Error::Success(); }
  Error IsNullResponse(bool* is_null_response) const override { *is_null_response = false; return Error::Success(); }
  std::string DebugString() const override { return status_.IsOk() ? doc_.Dump() : status_.Message(); }
  Error RequestStatus() const override { return status_; }

 private:
  void Parse(size_t header_length, int http_status, const Error& transport_error) {
    if (!transport_error.IsOk()) {
      status_ = transport_error;
      return;
    }
    const size_t json_len = header_length == 0 ? body_.size() : std::min(header_length, body_.size());
    if (json_len == 0) {
      status_ = Error("client received an empty response from the server.", http_status);
      return;
    }
    Error e = json::Parse(body_.data(), json_len, &doc_);
    if (!e.IsOk()) {
      status_ = Error(e.Message(), http_status);
      return;
    }
    if (http_status != 200) {
      status_ = Error(doc_.GetString("error", "inference failed with HTTP status " + std::to_string(http_status)), http_status);
      return;
    }
    size_t cursor = json_len;
    if (const json::Value* outs = doc_.Find("outputs")) {
      for (const auto& o : outs->Items()) {
        const std::string name = o.GetString("name");
        const json::Value* p = o.Find("parameters");
        const json::Value* bs = p ? p->Find("binary_data_size") : nullptr;
        if (bs != nullptr) {
          const size_t n = static_cast<size_t>(bs->AsUInt());
          if (cursor + n > body_.size()) {
            status_ = Error("binary data of output '" + name + "' overruns the response body");
            return;
          }
          spans_[name] = {body_.data() + cursor, n};
          cursor += n;
        } else if (const json::Value* data = o.Find("data")) {
          std::string packed;
          Error ce = JsonToBinary(o.GetString("datatype"), *data, name, &packed);
          if (ce.IsOk()) converted_[name] = std::move(packed);
        }
      }
    }
  }
  Error Find(const std::string& name, const json::Value** out) const {
    TCA_RETURN_IF_ERROR(status_);
    if (const json::Value* outs = doc_.Find("outputs"))
      for (const auto& o : outs->Items())
        if (o.GetString("name") == name) {
          *out = &o;
          return Error::Success();
        }
    return Error("The response does not contain results for output name '" + name + "'");
  }

  std::string body_;
  json::Value doc_;
  Error status_;
  std::map<std::string, std::pair<const char*, size_t>> spans_;
  std::map<std::string, std::string> converted_;
};

//==============================================================================
// client
//==============================================================================
InferenceServerHttpClient::InferenceServerHttpClient(const HttpEndpoint& ep, bool verbose)
    : InferenceServerClient(verbose), ep_(ep), sync_conn_(new HttpConn(ep)) {}

InferenceServerHttpClient::~InferenceServerHttpClient() {
  exiting_.store(true);
  reactor_.reset();  // drains in-flight requests, then joins the reactor thread
}

Error InferenceServerHttpClient::Create(std::unique_ptr<InferenceServerHttpClient>* client, const std::string& server_url,
                                        bool verbose, const HttpSslOptions& ssl_options) {
  HttpEndpoint ep;
  TCA_RETURN_IF_ERROR(ParseHttpUrl(server_url, &ep));
  if (ep.https) {
    net::TlsOptions tls;
    tls.verify_peer = ssl_options.verify_peer != 0;
    tls.verify_host = ssl_options.verify_host != 0;
    tls.ca_file = ssl_options.ca_info;
    tls.cert_file = ssl_options.cert;
    tls.cert_der = ssl_options.cert_type == HttpSslOptions::CERT_DER;
    tls.key_file = ssl_options.key;
    tls.key_der = ssl_options.key_type == HttpSslOptions::KEY_DER;
    tls.alpn = {"http/1.1"};
    TCA_RETURN_IF_ERROR(net::TlsContext::Create(tls, &ep.tls));
  }
  client->reset(new InferenceServerHttpClient(ep, verbose));
  return Error::Success();
}

size_t InferenceServerHttpClient::AsyncInFlight() const { return reactor_ ? reactor_->InFlight() : 0; }

namespace {
Error StatusToError(int status, const std::string& body) {
  if (status == 200) return Error::Success();
  if (body.empty()) return Error("client received an empty response from the server.", status);
  json::Value doc;
  if (json::Parse(body, &doc).IsOk() && doc.Has("error")) return Error(doc.GetString("error"), status);
  return Error("an exception occurred in the client while decoding the response: " + body, status);
}
}  // namespace

Error InferenceServerHttpClient::Get(const std::string& path, const Headers& headers, const Parameters& query,
                                     std::string* body, int* status) {
  HttpRequestSpec spec;
  spec.method = "GET";
  spec.target = path + QueryString(query);
  spec.headers = headers;
  HttpResponseData resp;
  if (verbose_) std::cout << "GET " << spec.target << std::endl;
  TCA_RETURN_IF_ERROR(sync_conn_->Begin(&spec, &resp, nullptr));
  TCA_RETURN_IF_ERROR(sync_conn_->RunToCompletion());
  if (verbose_) std::cout << resp.status << " " << resp.body << std::endl;
  *status = resp.status;
  *body = std::move(resp.body);
  return Error::Success();
}

Error InferenceServerHttpClient::Post(const std::string& path, const std::string& request, const Headers& headers,
                                      const Parameters& query, std::string* body, int* status) {
  HttpRequestSpec spec;
  spec.method = "POST";
  spec.target = path + QueryString(query);
  spec.headers = headers;
  if (!request.empty()) {
    spec.AddOwned(request);
    if (!spec.headers.count("Content-Type")) spec.headers["Content-Type"] = "application/json";
  }
  HttpResponseData resp;
  if (verbose_) std::cout << "POST " << spec.target << "\n" << request << std::endl;
  TCA_RETURN_IF_ERROR(sync_conn_->Begin(&spec, &resp, nullptr));
  TCA_RETURN_IF_ERROR(sync_conn_->RunToCompletion());
  if (verbose_) std::cout << resp.status << " " << resp.body << std::endl;
  *status = resp.status;
  *body = std::move(resp.body);
  return Error::Success();
}

#define TCA_SIMPLE_GET(path_expr, out_ptr)                            \
  std::string body_;                                                  \
  int status_ = 0;                                                    \
  TCA_RETURN_IF_ERROR(Get((path_expr), headers, query_params, &body_, &status_)); \
  TCA_RETURN_IF_ERROR(StatusToError(status_, body_));                 \
  *(out_ptr) = std::move(body_);                                      \
  return Error::Success();

Error InferenceServerHttpClient::IsServerLive(bool* live, const Headers& headers, const Parameters& query_params) {
  std::string body;
  int status = 0;
  TCA_RETURN_IF_ERROR(Get("/v2/health/live", headers, query_params, &body, &status));
  *live = status == 200;
  return Error::Success();
}

Error InferenceServerHttpClient::IsServerReady(bool* ready, const Headers& headers, const Parameters& query_params) {
  std::string body;
  int status = 0;
  TCA_RETURN_IF_ERROR(Get("/v2/health/ready", headers, query_params, &body, &status));
  *ready = status == 200;
  return Error::Success();
}

Error InferenceServerHttpClient::IsModelReady(bool* ready, const std::string& model_name, const std::string& model_version,
                                              const Headers& headers, const Parameters& query_params) {
  std::string body;
  int status = 0;
  TCA_RETURN_IF_ERROR(Get(ModelPath(model_name, model_version, "/ready"), headers, query_params, &body, &status));
  *ready = status == 200;
  return Error::Success();
}

Error InferenceServerHttpClient::ServerMetadata(std::string* server_metadata, const Headers& headers,
                                                const Parameters& query_params) {
  TCA_SIMPLE_GET("/v2", server_metadata)
}

Error InferenceServerHttpClient::ModelMetadata(std::string* model_metadata, const std::string& model_name,
                                               const std::string& model_version, const Headers& headers,
                                               const Parameters& query_params) {
  TCA_SIMPLE_GET(ModelPath(model_name, model_version, ""), model_metadata)
}

Error InferenceServerHttpClient::ModelConfig(std::string* model_config, const std::string& model_name,
                                             const std::string& model_version, const Headers& headers,
                                             const Parameters& query_params) {
  TCA_SIMPLE_GET(ModelPath(model_name, model_version, "/config"), model_config)
}

Error InferenceServerHttpClient::ModelRepositoryIndex(std::string* repository_index, const Headers& headers,
                                                      const Parameters& query_params) {
  std::string body;
  int status = 0;
  TCA_RETURN_IF_ERROR(Post("/v2/repository/index", "", headers, query_params, &body, &status));
  TCA_RETURN_IF_ERROR(StatusToError(status, body));
  *repository_index = std::move(body);
  return Error::Success();
}

Error InferenceServerHttpClient::LoadModel(const std::string& model_name, const Headers& headers,
                                           const Parameters& query_params, const std::string& config,
                                           const std::map<std::string, std::vector<char>>& files) {
  json::Value doc = json::Value::Object();
  if (!config.empty() || !files.empty()) {
    json::Value params = json::Value::Object();
    if (!config.empty()) params["config"] = config;
    for (const auto& f : files) params[f.first] = Base64Encode(f.second.data(), f.second.size());
    doc["parameters"] = std::move(params);
  }
  std::string body;
  int status = 0;
  TCA_RETURN_IF_ERROR(Post("/v2/repository/models/" + UrlEncode(model_name, false) + "/load", doc.Dump(), headers,
                           query_params, &body, &status));
  return StatusToError(status, body);
}

Error InferenceServerHttpClient::UnloadModel(const std::string& model_name, const Headers& headers,
                                             const Parameters& query_params) {
  std::string body;
  int status = 0;
  TCA_RETURN_IF_ERROR(Post("/v2/repository/models/" + UrlEncode(model_name, false) + "/unload", "", headers,
                           query_params, &body, &status));
  return StatusToError(status, body);
}

Error InferenceServerHttpClient::ModelInferenceStatistics(std::string* infer_stat, const std::string& model_name,
                                                          const std::string& model_version, const Headers& headers,
                                                          const Parameters& query_params) {
  TCA_SIMPLE_GET(model_name.empty() ? std::string("/v2/models/stats") : ModelPath(model_name, model_version, "/stats"),
                 infer_stat)
}

Error InferenceServerHttpClient::UpdateTraceSettings(std::string* response, const std::string& model_name,
                                                     const std::map<std::string, std::vector<std::string>>& settings,
                                                     const Headers& headers, const Parameters& query_params) {
  json::Value doc = json::Value::Object();
  for (const auto& kv : settings) {
    if (kv.second.empty()) {
      doc[kv.first] = json::Value();  // null clears the setting
    } else if (kv.second.size() == 1 && kv.first != "trace_level") {
      doc[kv.first] = kv.second[0];
    } else {
      json::Value arr = json::Value::Array();
      for (const auto& v : kv.second) arr.Append(json::Value(v));
      doc[kv.first] = std::move(arr);
    }
  }
  const std::string path = model_name.empty() ? std::string("/v2/trace/setting")
                                              : "/v2/models/" + UrlEncode(model_name, false) + "/trace/setting";
  std::string body;
  int status = 0;
  TCA_RETURN_IF_ERROR(Post(path, doc.Dump(), headers, query_params, &body, &status));
  TCA_RETURN_IF_ERROR(StatusToError(status, body));
  if (response) *response = std::move(body);
  return Error::Success();
}

Error InferenceServerHttpClient::GetTraceSettings(std::string* settings, const std::string& model_name,
                                                  const Headers& headers, const Parameters& query_params) {
  TCA_SIMPLE_GET(model_name.empty() ? std::string("/v2/trace/setting")
                                    : "/v2/models/" + UrlEncode(model_name, false) + "/trace/setting",
                 settings)
}

Error InferenceServerHttpClient::UpdateLogSettings(std::string* response, const std::string& settings_json,
                                                   const Headers& headers, const Parameters& query_params) {
  std::string body;
  int status = 0;
  TCA_RETURN_IF_ERROR(Post("/v2/logging", settings_json, headers, query_params, &body, &status));
  TCA_RETURN_IF_ERROR(StatusToError(status, body));
  if (response) *response = std::move(body);
  return Error::Success();
}

Error InferenceServerHttpClient::GetLogSettings(std::string* settings, const Headers& headers,
                                                const Parameters& query_params) {
  TCA_SIMPLE_GET("/v2/logging", settings)
}

Error InferenceServerHttpClient::SystemSharedMemoryStatus(std::string* status_out, const std::string& region_name,
                                                          const Headers& headers, const Parameters& query_params) {
  TCA_SIMPLE_GET(region_name.empty() ? std::string("/v2/systemsharedmemory/status")
                                     : "/v2/systemsharedmemory/region/" + UrlEncode(region_name, false) + "/status",
                 status_out)
}

Error InferenceServerHttpClient::RegisterSystemSharedMemory(const std::string& name, const std::string& key,
                                                            size_t byte_size, size_t offset, const Headers& headers,
                                                            const Parameters& query_params) {
  json::Value doc = json::Value::Object();
  doc["key"] = key;
  doc["offset"] = static_cast<unsigned long long>(offset);
  doc["byte_size"] = static_cast<unsigned long long>(byte_size);
  std::string body;
  int status = 0;
  TCA_RETURN_IF_ERROR(Post("/v2/systemsharedmemory/region/" + UrlEncode(name, false) + "/register", doc.Dump(), headers,
                           query_params, &body, &status));
  return StatusToError(status, body);
}

Error InferenceServerHttpClient::UnregisterSystemSharedMemory(const std::string& name, const Headers& headers,
                                                              const Parameters& query_params) {
  const std::string path = name.empty() ? std::string("/v2/systemsharedmemory/unregister")
                                        : "/v2/systemsharedmemory/region/" + UrlEncode(name, false) + "/unregister";
  std::string body;
  int status = 0;
  TCA_RETURN_IF_ERROR(Post(path, "", headers, query_params, &body, &status));
  return StatusToError(status, body);
}

Error InferenceServerHttpClient::CudaSharedMemoryStatus(std::string* status_out, const std::string& region_name,
                                                        const Headers& headers, const Parameters& query_params) {
  TCA_SIMPLE_GET(region_name.empty() ? std::string("/v2/cudasharedmemory/status")
                                     : "/v2/cudasharedmemory/region/" + UrlEncode(region_name, false) + "/status",
                 status_out)
}

Error InferenceServerHttpClient::RegisterCudaSharedMemory(const std::string& name, const void* raw_handle,
                                                          size_t device_id, size_t byte_size, const Headers& headers,
                                                          const Parameters& query_params) {
  json::Value doc = json::Value::Object();
  json::Value handle = json::Value::Object();
  handle["b64"] = Base64Encode(raw_handle, 64);
  doc["raw_handle"] = std::move(handle);
  doc["device_id"] = static_cast<unsigned long long>(device_id);
  doc["byte_size"] = static_cast<unsigned long long>(byte_size);
  std::string body;
  int status = 0;
  TCA_RETURN_IF_ERROR(Post("/v2/cudasharedmemory/region/" + UrlEncode(name, false) + "/register", doc.Dump(), headers,
                           query_params, &body, &status));
  return StatusToError(status, body);
}

Error InferenceServerHttpClient::UnregisterCudaSharedMemory(const std::string& name, const Headers& headers,
                                                            const Parameters& query_params) {
  const std::string path = name.empty() ? std::string("/v2/cudasharedmemory/unregister")
                                        : "/v2/cudasharedmemory/region/" + UrlEncode(name, false) + "/unregister";
  std::string body;
  int status = 0;
  TCA_RETURN_IF_ERROR(Post(path, "", headers, query_params, &body, &status));
  return StatusToError(status, body);
}

//------------------------------------------------------------------------------
// inference
//------------------------------------------------------------------------------
Error InferenceServerHttpClient::GenerateRequestBody(std::vector<char>* request_body, size_t* header_length,
                                                     const InferOptions& options, const std::vector<InferInput*>& inputs,
                                                     const std::vector<const InferRequestedOutput*>& outputs) {
  std::string head;
  bool any_binary = false;
  TCA_RETURN_IF_ERROR(BuildRequestJson(options, inputs, outputs, &head, &any_binary));
  request_body->assign(head.begin(), head.end());
  *header_length = head.size();
  for (const InferInput* in : inputs) {
    if (in->IsSharedMemory() || !in->BinaryData()) continue;
    for (const auto& b : in->Buffers())
      request_body->insert(request_body->end(), reinterpret_cast<const char*>(b.first),
                           reinterpret_cast<const char*>(b.first) + b.second);
  }
  return Error::Success();
}

Error InferenceServerHttpClient::ParseResponseBody(InferResult** result, const std::vector<char>& response_body,
                                                   size_t header_length) {
  return InferResultHttp::Create(result, std::string(response_body.begin(), response_body.end()), header_length, 200,
                                 Error::Success());
}

Error InferenceServerHttpClient::PrepareInfer(HttpRequestSpec* spec, const InferOptions& options,
                                              const std::vector<InferInput*>& inputs,
                                              const std::vector<const InferRequestedOutput*>& outputs,
                                              const Headers& headers, const Parameters& query_params,
                                              CompressionType req_alg, CompressionType resp_alg) {
  std::string head;
  bool any_binary = false;
  TCA_RETURN_IF_ERROR(BuildRequestJson(options, inputs, outputs, &head, &any_binary));
  spec->method = "POST";
  spec->target = ModelPath(options.model_name_, options.model_version_, "/infer") + QueryString(query_params);
  spec->headers = headers;
  spec->timeout_us = options.client_timeout_;
  spec->headers[kInferHeaderContentLengthHTTPHeader] = std::to_string(head.size());
  spec->headers["Content-Type"] = any_binary ? "application/octet-stream" : "application/json";
  spec->AddOwned(std::move(head));
  for (const InferInput* in : inputs) {
    if (in->IsSharedMemory() || !in->BinaryData()) continue;
    for (const auto& b : in->Buffers()) spec->body.emplace_back(b.first, b.second);  // zero copy
  }
  if (req_alg != CompressionType::NONE) {
    std::string packed;
    TCA_RETURN_IF_ERROR(Compress(req_alg == CompressionType::GZIP, spec->body, &packed));
    spec->body.clear();
    spec->owned.clear();
    spec->AddOwned(std::move(packed));
    spec->headers["Content-Encoding"] = req_alg == CompressionType::GZIP ? "gzip" : "deflate";
  }
  if (resp_alg != CompressionType::NONE)
    spec->headers["Accept-Encoding"] = resp_alg == CompressionType::GZIP ? "gzip" : "deflate";
  return Error::Success();
}

namespace {
Error FinishInfer(const Error& transport, HttpResponseData* resp, InferResult** result) {
  size_t header_length = 0;
  std::string body = std::move(resp->body);
  Error terr = transport;
  if (terr.IsOk()) {
    if (const std::string* enc = resp->Header("content-encoding")) {
      if (*enc == "gzip" || *enc == "deflate") {
        std::string plain;
        Error de = Decompress(body, &plain);
        if (de.IsOk()) body = std::move(plain); else terr = de;
      }
    }
    if (const std::string* hl = resp->Header("inference-header-content-length"))
      header_length = static_cast<size_t>(std::strtoull(hl->c_str(), nullptr, 10));
  }
  return InferResultHttp::Create(result, std::move(body), header_length, resp->status, terr);
}
}  // namespace

Error InferenceServerHttpClient::Infer(InferResult** result, const InferOptions& options,
                                       const std::vector<InferInput*>& inputs,
                                       const std::vector<const InferRequestedOutput*>& outputs, const Headers& headers,
                                       const Parameters& query_params, CompressionType request_compression_algorithm,
                                       CompressionType response_compression_algorithm) {
  RequestTimers timers;
  timers.CaptureTimestamp(RequestTimers::Kind::REQUEST_START);
  HttpRequestSpec spec;
  TCA_RETURN_IF_ERROR(PrepareInfer(&spec, options, inputs, outputs, headers, query_params,
                                   request_compression_algorithm, response_compression_algorithm));
  HttpResponseData resp;
  Error transport = sync_conn_->Begin(&spec, &resp, &timers);
  if (transport.IsOk()) transport = sync_conn_->RunToCompletion();
  timers.CaptureTimestamp(RequestTimers::Kind::REQUEST_END);
  if (transport.IsOk()) UpdateInferStat(timers);
  TCA_RETURN_IF_ERROR(FinishInfer(transport, &resp, result));
  Error status = (*result)->RequestStatus();
  if (!status.IsOk()) {
    delete *result;
    *result = nullptr;
  }
  return status;
}

Error InferenceServerHttpClient::AsyncInfer(OnCompleteFn callback, const InferOptions& options,
                                            const std::vector<InferInput*>& inputs,
                                            const std::vector<const InferRequestedOutput*>& outputs,
                                            const Headers& headers, const Parameters& query_params,
                                            CompressionType request_compression_algorithm,
                                            CompressionType response_compression_algorithm) {
  if (callback == nullptr) return Error("Callback function must be provided along with AsyncInfer() call.");
  if (!reactor_) reactor_.reset(new HttpReactor(ep_));
  std::unique_ptr<HttpReactor::Op> op(new HttpReactor::Op());
  TCA_RETURN_IF_ERROR(PrepareInfer(&op->spec, options, inputs, outputs, headers, query_params,
                                   request_compression_algorithm, response_compression_algorithm));
  op->done = [this, callback](const Error& transport, HttpReactor::Op* done_op) {
    done_op->timers.CaptureTimestamp(RequestTimers::Kind::REQUEST_END);
    if (transport.IsOk()) UpdateInferStat(done_op->timers);
    InferResult* result = nullptr;
    FinishInfer(transport, &done_op->resp, &result);
    callback(result);  // the callee owns `result`
  };
  reactor_->Submit(std::move(op));
  return Error::Success();
}

namespace {
Error CheckMulti(size_t n_inputs, size_t n_options, size_t n_outputs) {
  if (n_inputs == 0) return Error("'inputs' must not be empty");
  if (n_options != 1 && n_options != n_inputs)
    return Error("'options' has " + std::to_string(n_options) + " elements, expecting 1 or " + std::to_string(n_inputs));
  if (n_outputs > 1 && n_outputs != n_inputs)
    return Error("'outputs' has " + std::to_string(n_outputs) + " elements, expecting 0, 1 or " + std::to_string(n_inputs));
  return Error::Success();
}
}  // namespace

Error InferenceServerHttpClient::InferMulti(std::vector<InferResult*>* results, const std::vector<InferOptions>& options,
                                            const std::vector<std::vector<InferInput*>>& inputs,
                                            const std::vector<std::vector<const InferRequestedOutput*>>& outputs,
                                            const Headers& headers, const Parameters& query_params,
                                            CompressionType request_compression_algorithm,
                                            CompressionType response_compression_algorithm) {
  TCA_RETURN_IF_ERROR(CheckMulti(inputs.size(), options.size(), outputs.size()));
  static const std::vector<const InferRequestedOutput*> none;
  results->clear();
  for (size_t i = 0; i < inputs.size(); ++i) {
    const InferOptions& opt = options.size() == 1 ? options[0] : options[i];
    const auto& outs = outputs.empty() ? none : (outputs.size() == 1 ? outputs[0] : outputs[i]);
    InferResult* r = nullptr;
    Error e = Infer(&r, opt, inputs[i], outs, headers, query_params, request_compression_algorithm,
                    response_compression_algorithm);
    if (!e.IsOk()) {
      for (InferResult* p : *results) delete p;
      results->clear();
      return e;
    }
    results->push_back(r);
  }
  return Error::Success();
}

Error InferenceServerHttpClient::AsyncInferMulti(OnMultiCompleteFn callback, const std::vector<InferOptions>& options,
                                                 const std::vector<std::vector<InferInput*>>& inputs,
                                                 const std::vector<std::vector<const InferRequestedOutput*>>& outputs,
                                                 const Headers& headers, const Parameters& query_params,
                                                 CompressionType request_compression_algorithm,
                                                 CompressionType response_compression_algorithm) {
  if (callback == nullptr) return Error("Callback function must be provided along with AsyncInferMulti() call.");
  TCA_RETURN_IF_ERROR(CheckMulti(inputs.size(), options.size(), outputs.size()));
  static const std::vector<const InferRequestedOutput*> none;
  struct Shared {
    std::atomic<size_t> remaining;
    std::vector<InferResult*> results;
    OnMultiCompleteFn cb;
  };
  auto shared = std::make_shared<Shared>();
  shared->remaining.store(inputs.size());
  shared->results.assign(inputs.size(), nullptr);
  shared->cb = std::move(callback);
  for (size_t i = 0; i < inputs.size(); ++i) {
    const InferOptions& opt = options.size() == 1 ? options[0] : options[i];
    const auto& outs = outputs.empty() ? none : (outputs.size() == 1 ? outputs[0] : outputs[i]);
    Error e = AsyncInfer(
        [shared, i](InferResult* r) {
          shared->results[i] = r;
          if (shared->remaining.fetch_sub(1) == 1) shared->cb(shared->results);
        },
        opt, inputs[i], outs, headers, query_params, request_compression_algorithm, response_compression_algorithm);
    if (!e.IsOk()) {
      // account for the requests that will never be issued so the callback still fires once
      const size_t missing = inputs.size() - i;
      if (shared->remaining.fetch_sub(missing) == missing) shared->cb(shared->results);
      return e;
    }
  }
  return Error::Success();
}

}  // namespace client
}  // namespace tca
