// Native gRPC client (see tca/grpc_client.h).
#include "tca/grpc_client.h"
#include "tca/sync.h"

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <vector>
#include <iostream>
#include <sstream>

#include "tca/base64.h"
#include "tca/grpc_compress.h"
#include "tca/json.h"
#include "tca/pbjson.h"
#include "tca/pbwire.h"

namespace tca {
namespace client {
namespace {

// ---- channel cache -----------------------------------------------------------
struct CacheEntry {
  std::weak_ptr<GrpcChannel> channel;
  size_t shares = 0;
};
std::mutex g_cache_mu;
std::map<std::string, CacheEntry> g_cache;

size_t MaxShareCount() {
  const char* v = std::getenv("TRITON_CLIENT_GRPC_CHANNEL_MAX_SHARE_COUNT");
  if (v != nullptr) {
    const long n = std::strtol(v, nullptr, 10);
    if (n > 0) return static_cast<size_t>(n);
  }
  return 6;
}

// ---- request encoding ----------------------------------------------------------
void PutParam(pb::Writer* w, uint32_t map_field, const std::string& key, const std::function<void(pb::Writer*)>& value) {
  std::string param;
  pb::Writer pw(&param);
  value(&pw);
  std::string entry;
  pb::Writer ew(&entry);
  ew.String(1, key);
  ew.Message(2, param);
  w->Message(map_field, entry);
}
void PutBoolParam(pb::Writer* w, uint32_t f, const std::string& k, bool v) {
  PutParam(w, f, k, [v](pb::Writer* p) { p->Bool(1, v); });
}
void PutIntParam(pb::Writer* w, uint32_t f, const std::string& k, int64_t v) {
  PutParam(w, f, k, [v](pb::Writer* p) { p->Int(2, v); });
}
void PutStringParam(pb::Writer* w, uint32_t f, const std::string& k, const std::string& v) {
  PutParam(w, f, k, [&v](pb::Writer* p) { p->String(3, v); });
}
void PutDoubleParam(pb::Writer* w, uint32_t f, const std::string& k, double v) {
  PutParam(w, f, k, [v](pb::Writer* p) { p->Double(4, v); });
}
void PutUIntParam(pb::Writer* w, uint32_t f, const std::string& k, uint64_t v) {
  PutParam(w, f, k, [v](pb::Writer* p) { p->UInt(5, v); });
}

Error EncodeInferRequest(std::string* out, const InferOptions& options, const std::vector<InferInput*>& inputs,
                         const std::vector<const InferRequestedOutput*>& outputs) {
  static const char* const kReserved[] = {"sequence_id", "sequence_start", "sequence_end", "priority",
                                          "binary_data_output"};
  size_t payload = 0;
  for (const InferInput* in : inputs) {
    size_t n = 0;
    in->ByteSize(&n);
    payload += n;
  }
  if (payload > MAX_GRPC_MESSAGE_SIZE) {
    return Error("Request has byte size " + std::to_string(payload) + " which exceeds gRPC's byte size limit of " +
                 std::to_string(MAX_GRPC_MESSAGE_SIZE) + ".");
  }
  out->clear();
  out->reserve(payload + 512);
  pb::Writer w(out);
  w.String(1, options.model_name_);
  if (!options.model_version_.empty()) w.String(2, options.model_version_);
  if (!options.request_id_.empty()) w.String(3, options.request_id_);
  if (options.sequence_id_ != 0 || !options.sequence_id_str_.empty()) {
    if (!options.sequence_id_str_.empty()) PutStringParam(&w, 4, "sequence_id", options.sequence_id_str_);
    else PutIntParam(&w, 4, "sequence_id", static_cast<int64_t>(options.sequence_id_));
    PutBoolParam(&w, 4, "sequence_start", options.sequence_start_);
    PutBoolParam(&w, 4, "sequence_end", options.sequence_end_);
  }
  if (options.priority_ != 0) PutUIntParam(&w, 4, "priority", options.priority_);
  if (options.server_timeout_ != 0) PutIntParam(&w, 4, "timeout", static_cast<int64_t>(options.server_timeout_));
  if (options.triton_enable_empty_final_response_) PutBoolParam(&w, 4, "triton_enable_empty_final_response", true);
  for (const auto& kv : options.request_parameters) {
    for (const char* r : kReserved) {
      if (kv.first == r)
        return Error("Parameter \"" + kv.first + "\" is a reserved parameter and cannot be specified.");
    }
    const RequestParameter& p = kv.second;
    switch (p.type) {
      case RequestParameter::Type::STRING: PutStringParam(&w, 4, kv.first, p.str_value); break;
      case RequestParameter::Type::INT: PutIntParam(&w, 4, kv.first, p.int_value); break;
      case RequestParameter::Type::UINT: PutUIntParam(&w, 4, kv.first, p.uint_value); break;
      case RequestParameter::Type::BOOL: PutBoolParam(&w, 4, kv.first, p.bool_value); break;
      case RequestParameter::Type::DOUBLE: PutDoubleParam(&w, 4, kv.first, p.double_value); break;
    }
  }
  for (const InferInput* in : inputs) {
    std::string tensor;
    pb::Writer tw(&tensor);
    tw.String(1, in->Name());
    tw.String(2, in->Datatype());
    if (!in->Shape().empty()) {
      std::string packed;
      pb::Writer sw(&packed);
      for (int64_t d : in->Shape()) sw.Varint(static_cast<uint64_t>(d));
      tw.Message(3, packed);
    }
    if (in->IsSharedMemory()) {
      std::string region;
      size_t byte_size = 0, offset = 0;
      in->SharedMemoryInfo(&region, &byte_size, &offset);
      PutStringParam(&tw, 4, "shared_memory_region", region);
      PutIntParam(&tw, 4, "shared_memory_byte_size", static_cast<int64_t>(byte_size));
      if (offset != 0) PutIntParam(&tw, 4, "shared_memory_offset", static_cast<int64_t>(offset));
    }
    w.Message(5, tensor);
  }
  for (const InferRequestedOutput* o : outputs) {
    std::string tensor;
    pb::Writer tw(&tensor);
    tw.String(1, o->Name());
    if (o->ClassificationCount() != 0) PutIntParam(&tw, 2, "classification", static_cast<int64_t>(o->ClassificationCount()));
    if (o->IsSharedMemory()) {
      std::string region;
      size_t byte_size = 0, offset = 0;
      o->SharedMemoryInfo(&region, &byte_size, &offset);
      PutStringParam(&tw, 2, "shared_memory_region", region);
      PutIntParam(&tw, 2, "shared_memory_byte_size", static_cast<int64_t>(byte_size));
      if (offset != 0) PutIntParam(&tw, 2, "shared_memory_offset", static_cast<int64_t>(offset));
    }
    w.Message(6, tensor);
  }
  for (const InferInput* in : inputs) {
    if (in->IsSharedMemory()) continue;
    size_t n = 0;
    in->ByteSize(&n);
    w.LenHeader(7, n);
    for (const auto& piece : in->Buffers()) out->append(reinterpret_cast<const char*>(piece.first), piece.second);
  }
  return Error::Success();
}

// ---- response decoding -----------------------------------------------------------
struct ParamValue {
  enum Kind { NONE, BOOL, INT, STRING, DOUBLE, UINT } kind = NONE;
  bool b = false;
  int64_t i = 0;
  uint64_t u = 0;
  double d = 0;
  std::string s;
};

bool ParseParamEntry(const uint8_t* data, size_t len, std::string* key, ParamValue* value) {
  pb::Reader r(data, len);
  uint32_t f, wt;
  while (r.Next(&f, &wt)) {
    if (f == 1 && wt == pb::kLen) {
      if (!r.String(key)) return false;
    } else if (f == 2 && wt == pb::kLen) {
      const uint8_t* d;
      size_t n;
      if (!r.Len(&d, &n)) return false;
      pb::Reader pr(d, n);
      uint32_t pf, pw;
      while (pr.Next(&pf, &pw)) {
        uint64_t v = 0;
        if (pf == 1 && pw == pb::kVarint && pr.Varint(&v)) { value->kind = ParamValue::BOOL; value->b = v != 0; }
        else if (pf == 2 && pw == pb::kVarint && pr.Varint(&v)) { value->kind = ParamValue::INT; value->i = static_cast<int64_t>(v); }
        else if (pf == 3 && pw == pb::kLen && pr.String(&value->s)) { value->kind = ParamValue::STRING; }
        else if (pf == 4 && pw == pb::kFixed64 && pr.Fixed64(&v)) { value->kind = ParamValue::DOUBLE; std::memcpy(&value->d, &v, 8); }
        else if (pf == 5 && pw == pb::kVarint && pr.Varint(&v)) { value->kind = ParamValue::UINT; value->u = v; }
        else if (!pr.Skip(pw)) return false;
      }
    } else if (!r.Skip(wt)) {
      return false;
    }
  }
  return !r.Failed();
}

class InferResultGrpc : public InferResult {
 public:
  struct Output {
    std::string name, datatype;
    std::vector<int64_t> shape;
    const uint8_t* raw = nullptr;
    size_t raw_size = 0;
    bool has_raw = false;
    std::vector<std::string> bytes_contents;  // typed contents fallback for BYTES
    std::string typed;                        // typed contents packed to little-endian raw form
    bool has_typed = false;
  };

  // `message` is a serialized ModelInferResponse (moved in, outputs are views into it).
  static InferResultGrpc* FromResponse(std::string&& message, const Error& status = Error::Success()) {
    auto* r = new InferResultGrpc();
    r->status_ = status;
    r->buf_ = std::move(message);
    if (!r->Parse(reinterpret_cast<const uint8_t*>(r->buf_.data()), r->buf_.size()) && status.IsOk())
      r->status_ = Error("failed to parse ModelInferResponse");
    return r;
  }
  // `message` is a ModelStreamInferResponse.
  static InferResultGrpc* FromStreamResponse(std::string&& message) {
    auto* r = new InferResultGrpc();
    r->buf_ = std::move(message);
    pb::Reader rd(r->buf_);
    uint32_t f, wt;
    std::string error_message;
    bool ok = true;
    while (rd.Next(&f, &wt)) {
      if (f == 1 && wt == pb::kLen) {
        ok = rd.String(&error_message) && ok;
      } else if (f == 2 && wt == pb::kLen) {
        const uint8_t* d;
        size_t n;
        ok = rd.Len(&d, &n) && r->Parse(d, n) && ok;
      } else if (!rd.Skip(wt)) {
        ok = false;
        break;
      }
    }
    if (!error_message.empty()) r->status_ = Error(error_message);
    else if (!ok) r->status_ = Error("failed to parse ModelStreamInferResponse");
    return r;
  }
  static InferResultGrpc* FromError(const Error& status) {
    auto* r = new InferResultGrpc();
    r->status_ = status;
    return r;
  }

  Error ModelName(std::string* name) const override { *name = model_name_; return Error::Success(); }
  Error ModelVersion(std::string* version) const override { *version = model_version_; return Error::Success(); }
  Error Id(std::string* id) const override { *id = id_; return Error::Success(); }
  Error Shape(const std::string& output_name, std::vector<int64_t>* shape) const override {
    const Output* o = Find(output_name);
    if (o == nullptr) return Missing(output_name);
    *shape = o->shape;
    return Error::Success();
  }
  Error Datatype(const std::string& output_name, std::string* datatype) const override {
    const Output* o = Find(output_name);
    if (o == nullptr) return Missing(output_name);
    *datatype = o->datatype;
    return Error::Success();
  }
  Error RawData(const std::string& output_name, const uint8_t** buf, size_t* byte_size) const override {
    const Output* o = Find(output_name);
    if (o == nullptr) return Missing(output_name);
    if (o->has_raw) {
      *buf = o->raw;
      *byte_size = o->raw_size;
    } else if (o->has_typed) {
      *buf = reinterpret_cast<const uint8_t*>(o->typed.data());
      *byte_size = o->typed.size();
    } else {
      *buf = nullptr;
      *byte_size = 0;
    }
    return Error::Success();
  }
  Error StringData(const std::string& output_name, std::vector<std::string>* string_result) const override {
    const Output* o = Find(output_name);
    if (o == nullptr) return Missing(output_name);
    string_result->clear();
    if (o->has_raw) {
      size_t off = 0;
      while (off + 4 <= o->raw_size) {
        uint32_t n;
        std::memcpy(&n, o->raw + off, 4);
        off += 4;
        if (off + n > o->raw_size) return Error("malformed BYTES tensor in output '" + output_name + "'");
        string_result->emplace_back(reinterpret_cast<const char*>(o->raw + off), n);
        off += n;
      }
    } else {
      *string_result = o->bytes_contents;
    }
    return Error::Success();
  }
  Error IsFinalResponse(bool* is_final_response) const override {
    if (is_final_response == nullptr) return Error("is_final_response cannot be nullptr");
    *is_final_response = is_final_;
    return Error::Success();
  }
  Error IsNullResponse(bool* is_null_response) const override {
    if (is_null_response == nullptr) return Error("is_null_response cannot be nullptr");
    *is_null_response = is_null_;
    return Error::Success();
  }
  std::string DebugString() const override {
    std::ostringstream os;
    os << "model_name: \"" << model_name_ << "\" model_version: \"" << model_version_ << "\" id: \"" << id_ << "\"";
    for (const auto& o : outputs_) {
      os << " outputs { name: \"" << o.name << "\" datatype: \"" << o.datatype << "\" shape: [";
      for (size_t i = 0; i < o.shape.size(); ++i) os << (i ? "," : "") << o.shape[i];
      os << "] bytes: " << (o.has_raw ? o.raw_size : o.typed.size()) << " }";
    }
    return os.str();
  }
  Error RequestStatus() const override { return status_; }

 private:
  const Output* Find(const std::string& name) const {
    for (const auto& o : outputs_)
      if (o.name == name) return &o;
    return nullptr;
  }
  static Error Missing(const std::string& name) {
    return Error("The response does not contain results for output name '" + name + "'");
  }

  bool ParseContents(const uint8_t* data, size_t len, Output* o) {
    pb::Reader r(data, len);
    uint32_t f, wt;
    auto put = [o](const void* p, size_t n) { o->typed.append(static_cast<const char*>(p), n); o->has_typed = true; };
    while (r.Next(&f, &wt)) {
      if (f == 8 && wt == pb::kLen) {  // bytes_contents
        std::string s;
        if (!r.String(&s)) return false;
        o->bytes_contents.push_back(std::move(s));
        continue;
      }
      // numeric repeated fields: packed or not
      const uint8_t* d = nullptr;
      size_t n = 0;
      const bool packed = wt == pb::kLen;
      if (packed && !r.Len(&d, &n)) return false;
      pb::Reader pr(packed ? d : nullptr, packed ? n : 0);
      auto each = [&](const std::function<bool(pb::Reader*)>& one) {
        if (!packed) return one(&r);
        while (!pr.Done())
          if (!one(&pr)) return false;
        return true;
      };
      const std::string& dt = o->datatype;
      bool ok = true;
      switch (f) {
        case 1: ok = each([&](pb::Reader* x) { uint64_t v; if (!x->Varint(&v)) return false; const uint8_t b = v != 0; put(&b, 1); return true; }); break;
        case 2: ok = each([&](pb::Reader* x) {
          uint64_t v; if (!x->Varint(&v)) return false;
          const int32_t i = static_cast<int32_t>(v);
          if (dt == "INT8") { const int8_t s = static_cast<int8_t>(i); put(&s, 1); }
          else if (dt == "INT16") { const int16_t s = static_cast<int16_t>(i); put(&s, 2); }
          else put(&i, 4);
          return true; }); break;
        case 3: ok = each([&](pb::Reader* x) { uint64_t v; if (!x->Varint(&v)) return false; const int64_t i = static_cast<int64_t>(v); put(&i, 8); return true; }); break;
        case 4: ok = each([&](pb::Reader* x) {
          uint64_t v; if (!x->Varint(&v)) return false;
          const uint32_t u = static_cast<uint32_t>(v);
          if (dt == "UINT8") { const uint8_t s = static_cast<uint8_t>(u); put(&s, 1); }
          else if (dt == "UINT16") { const uint16_t s = static_cast<uint16_t>(u); put(&s, 2); }
          else put(&u, 4);
          return true; }); break;
        case 5: ok = each([&](pb::Reader* x) { uint64_t v; if (!x->Varint(&v)) return false; put(&v, 8); return true; }); break;
        case 6: ok = each([&](pb::Reader* x) { uint32_t v; if (!x->Fixed32(&v)) return false; put(&v, 4); return true; }); break;
        case 7: ok = each([&](pb::Reader* x) { uint64_t v; if (!x->Fixed64(&v)) return false; put(&v, 8); return true; }); break;
        default: ok = packed ? true : r.Skip(wt); break;
      }
      if (!ok) return false;
    }
    return !r.Failed();
  }

  bool Parse(const uint8_t* data, size_t len) {
    pb::Reader r(data, len);
    uint32_t f, wt;
    std::vector<std::pair<const uint8_t*, size_t>> raws;
    while (r.Next(&f, &wt)) {
      if (wt != pb::kLen) {
        if (!r.Skip(wt)) return false;
        continue;
      }
      const uint8_t* d;
      size_t n;
      if (!r.Len(&d, &n)) return false;
      switch (f) {
        case 1: model_name_.assign(reinterpret_cast<const char*>(d), n); break;
        case 2: model_version_.assign(reinterpret_cast<const char*>(d), n); break;
        case 3: id_.assign(reinterpret_cast<const char*>(d), n); break;
        case 4: {
          std::string key;
          ParamValue v;
          if (!ParseParamEntry(d, n, &key, &v)) return false;
          if (key == "triton_final_response" && v.kind == ParamValue::BOOL) is_final_ = v.b;
          break;
        }
        case 5: {
          Output o;
          pb::Reader tr(d, n);
          uint32_t tf, tw;
          const uint8_t* contents = nullptr;
          size_t contents_len = 0;
          while (tr.Next(&tf, &tw)) {
            if (tf == 1 && tw == pb::kLen) { if (!tr.String(&o.name)) return false; }
            else if (tf == 2 && tw == pb::kLen) { if (!tr.String(&o.datatype)) return false; }
            else if (tf == 3 && tw == pb::kLen) {
              const uint8_t* sd;
              size_t sn;
              if (!tr.Len(&sd, &sn)) return false;
              pb::Reader sr(sd, sn);
              while (!sr.Done()) {
                uint64_t v;
                if (!sr.Varint(&v)) return false;
                o.shape.push_back(static_cast<int64_t>(v));
              }
            } else if (tf == 3 && tw == pb::kVarint) {
              uint64_t v;
              if (!tr.Varint(&v)) return false;
              o.shape.push_back(static_cast<int64_t>(v));
            } else if (tf == 5 && tw == pb::kLen) {
              if (!tr.Len(&contents, &contents_len)) return false;
            } else if (!tr.Skip(tw)) {
              return false;
            }
          }
          if (contents != nullptr && !ParseContents(contents, contents_len, &o)) return false;
          outputs_.push_back(std::move(o));
          break;
        }
        case 6: raws.emplace_back(d, n); break;
        default: break;
      }
    }
    if (r.Failed()) return false;
    // raw_output_contents are positional: entry i belongs to output i
    for (size_t i = 0; i < outputs_.size() && i < raws.size(); ++i) {
      outputs_[i].raw = raws[i].first;
      outputs_[i].raw_size = raws[i].second;
      outputs_[i].has_raw = true;
    }
    is_null_ = outputs_.empty() && is_final_;
    return true;
  }

  std::string buf_;
  std::string model_name_, model_version_, id_;
  std::vector<Output> outputs_;
  bool is_final_ = true;
  bool is_null_ = false;
  Error status_;
};

// Compresses `request` in place and names the algorithm in the call metadata.
Error ApplyCompression(GrpcCompression alg, std::string* request, Headers* metadata, bool* compressed) {
  *compressed = false;
  if (alg == GrpcCompression::NONE) return Error::Success();
  std::string packed;
  TCA_RETURN_IF_ERROR(grpcz::Compress(request->data(), request->size(), alg == GrpcCompression::GZIP, &packed));
  request->swap(packed);
  (*metadata)["grpc-encoding"] = alg == GrpcCompression::GZIP ? "gzip" : "deflate";
  *compressed = true;
  return Error::Success();
}

Error StatusToError(int status, const std::string& message) {
  if (status == kGrpcOk) return Error::Success();
  return Error(message.empty() ? ("gRPC status " + std::to_string(status)) : message, status);
}

}  // namespace

// ---- stream state -----------------------------------------------------------------
struct InferenceServerGrpcClient::StreamState {
  std::mutex mu;
  std::condition_variable cv;
  uint64_t call = 0;
  bool done = false;
  int status = 0;
  std::string message;
  bool enable_stats = true;
  GrpcCompression compression = GrpcCompression::NONE;
  OnCompleteFn callback;
  std::deque<RequestTimers> timers;  // FIFO, one per request in flight (non-decoupled accounting)
};

// ---- client --------------------------------------------------------------------------
struct InferenceServerGrpcClient::AsyncState {
  std::mutex mu;
  std::condition_variable cv;
  std::set<uint64_t> calls;  // started, result not delivered yet
  int pending = 0;           // calls whose completion handler has not returned yet
};

namespace {
// the AsyncState whose completion handler is running on this thread (a callback may destroy its client)
thread_local const void* t_completing = nullptr;
}  // namespace

InferenceServerGrpcClient::InferenceServerGrpcClient(std::shared_ptr<GrpcChannel> channel, bool verbose)
    : InferenceServerClient(verbose), channel_(std::move(channel)), async_(std::make_shared<AsyncState>()) {}

InferenceServerGrpcClient::~InferenceServerGrpcClient() {
  exiting_.store(true);
  StopStream();
  {
    // calls still in flight end now, as cancelled; wait until their handlers are through with this object
    std::vector<uint64_t> open;
    {
      std::lock_guard<std::mutex> lk(async_->mu);
      open.assign(async_->calls.begin(), async_->calls.end());
    }
    for (uint64_t id : open) channel_->Cancel(id);
    std::unique_lock<std::mutex> lk(async_->mu);
    const int own = t_completing == async_.get() ? 1 : 0;  // destroyed from inside one of its own callbacks
    tca::WaitFor(async_->cv, lk, std::chrono::seconds(30), [&] { return async_->pending <= own; });
  }
  channel_.reset();
  std::lock_guard<std::mutex> lk(g_cache_mu);
  for (auto it = g_cache.begin(); it != g_cache.end();) {
    if (it->second.channel.expired()) it = g_cache.erase(it);
    else ++it;
  }
}

Error InferenceServerGrpcClient::Create(std::unique_ptr<InferenceServerGrpcClient>* client, const std::string& server_url,
                                        bool verbose, bool use_ssl, const SslOptions& ssl_options,
                                        const KeepAliveOptions& keepalive_options, bool use_cached_channel) {
  ChannelOptions options;
  options.keepalive = keepalive_options;
  return Create(client, server_url, options, verbose, use_ssl, ssl_options, use_cached_channel);
}

Error InferenceServerGrpcClient::Create(std::unique_ptr<InferenceServerGrpcClient>* client, const std::string& server_url,
                                        const ChannelOptions& channel_options, bool verbose, bool use_ssl,
                                        const SslOptions& ssl_options, bool use_cached_channel) {
  std::shared_ptr<net::TlsContext> tls;
  // a channel is only shared between clients that asked for the same settings
  std::string cache_key = server_url + "|" + channel_options.CacheKey();
  if (use_ssl) {
    net::TlsOptions opt;
    opt.ca_file = ssl_options.root_certificates;
    opt.key_file = ssl_options.private_key;
    opt.cert_file = ssl_options.certificate_chain;
    opt.alpn = {"h2"};
    TCA_RETURN_IF_ERROR(net::TlsContext::Create(opt, &tls));
    // a secure channel is only shared between clients created with the same credentials
    cache_key += "|tls|" + opt.ca_file + "|" + opt.key_file + "|" + opt.cert_file;
  }
  std::shared_ptr<GrpcChannel> channel;
  if (use_cached_channel) {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    CacheEntry& entry = g_cache[cache_key];
    channel = entry.channel.lock();
    if (channel && entry.shares < MaxShareCount()) {
      ++entry.shares;
    } else {
      TCA_RETURN_IF_ERROR(GrpcChannel::Create(&channel, server_url, verbose, channel_options, tls));
      entry.channel = channel;
      entry.shares = 1;
    }
  } else {
    TCA_RETURN_IF_ERROR(GrpcChannel::Create(&channel, server_url, verbose, channel_options, tls));
  }
  client->reset(new InferenceServerGrpcClient(std::move(channel), verbose));
  return Error::Success();
}

size_t InferenceServerGrpcClient::GetNumCachedChannels() {
  std::lock_guard<std::mutex> lk(g_cache_mu);
  size_t n = 0;
  for (const auto& kv : g_cache)
    if (!kv.second.channel.expired()) ++n;
  return n;
}

Error InferenceServerGrpcClient::Unary(const std::string& method, const std::string& request, const Headers& headers,
                                       uint64_t timeout_us, std::string* response, bool compressed) {
  struct Sync {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    int status = 0;
    std::string message, response;
  };
  auto sync = std::make_shared<Sync>();
  GrpcChannel::CallEvents ev;
  ev.on_message = [sync](std::string&& m) { sync->response = std::move(m); };
  ev.on_done = [sync](int status, const std::string& message) {
    std::lock_guard<std::mutex> lk(sync->mu);
    sync->status = status;
    sync->message = message;
    sync->done = true;
    sync->cv.notify_all();
  };
  if (static_cast<int64_t>(request.size()) > channel_->Options().max_send_message_bytes)
    return Error("Sent message larger than max (" + std::to_string(request.size()) + " vs. " +
                     std::to_string(channel_->Options().max_send_message_bytes) + ")",
                 kGrpcResourceExhausted);
  const uint64_t call = channel_->StartCall(method, headers, timeout_us, std::move(ev));
  channel_->SendMessage(call, std::string(request), true, compressed);
  std::unique_lock<std::mutex> lk(sync->mu);
  sync->cv.wait(lk, [&] { return sync->done; });
  if (sync->status != kGrpcOk) return StatusToError(sync->status, sync->message);
  *response = std::move(sync->response);
  return Error::Success();
}

Error InferenceServerGrpcClient::JsonCall(const std::string& method, const std::string& request_json,
                                          const Headers& headers, std::string* response_json, uint64_t timeout_us) {
  int req_msg = -1, resp_msg = -1;
  if (!pb::FindMethod(method, &req_msg, &resp_msg)) return Error("unknown RPC method '" + method + "'");
  json::Value req;
  TCA_RETURN_IF_ERROR(json::Parse(request_json.data(), request_json.size(), &req));
  std::string wire, reply;
  TCA_RETURN_IF_ERROR(pb::FromJson(req_msg, req, &wire));
  TCA_RETURN_IF_ERROR(Unary(method, wire, headers, timeout_us, &reply));
  json::Value resp;
  TCA_RETURN_IF_ERROR(pb::ToJson(resp_msg, reply.data(), reply.size(), &resp));
  if (response_json != nullptr) *response_json = resp.Dump();
  if (verbose_) std::cout << method << ": " << resp.Dump() << std::endl;
  return Error::Success();
}

namespace {
std::string NameVersion(const std::string& name, const std::string& version, const char* name_key = "name") {
  json::Value v = json::Value::Object();
  if (!name.empty()) v[name_key] = name;
  if (!version.empty()) v["version"] = version;
  return v.Dump();
}
}  // namespace

Error InferenceServerGrpcClient::IsServerLive(bool* live, const Headers& headers, uint64_t timeout_us) {
  std::string out;
  TCA_RETURN_IF_ERROR(JsonCall("ServerLive", "{}", headers, &out, timeout_us));
  json::Value v;
  TCA_RETURN_IF_ERROR(json::Parse(out.data(), out.size(), &v));
  *live = v.GetBool("live");
  return Error::Success();
}

Error InferenceServerGrpcClient::IsServerReady(bool* ready, const Headers& headers, uint64_t timeout_us) {
  std::string out;
  TCA_RETURN_IF_ERROR(JsonCall("ServerReady", "{}", headers, &out, timeout_us));
  json::Value v;
  TCA_RETURN_IF_ERROR(json::Parse(out.data(), out.size(), &v));
  *ready = v.GetBool("ready");
  return Error::Success();
}

Error InferenceServerGrpcClient::IsModelReady(bool* ready, const std::string& model_name,
                                              const std::string& model_version, const Headers& headers, uint64_t timeout_us) {
  std::string out;
  TCA_RETURN_IF_ERROR(JsonCall("ModelReady", NameVersion(model_name, model_version), headers, &out, timeout_us));
  json::Value v;
  TCA_RETURN_IF_ERROR(json::Parse(out.data(), out.size(), &v));
  *ready = v.GetBool("ready");
  return Error::Success();
}

Error InferenceServerGrpcClient::ServerMetadata(std::string* server_metadata, const Headers& headers, uint64_t timeout_us) {
  return JsonCall("ServerMetadata", "{}", headers, server_metadata, timeout_us);
}

Error InferenceServerGrpcClient::ModelMetadata(std::string* model_metadata, const std::string& model_name,
                                               const std::string& model_version, const Headers& headers, uint64_t timeout_us) {
  return JsonCall("ModelMetadata", NameVersion(model_name, model_version), headers, model_metadata, timeout_us);
}

Error InferenceServerGrpcClient::ModelConfig(std::string* model_config, const std::string& model_name,
                                             const std::string& model_version, const Headers& headers, uint64_t timeout_us) {
  return JsonCall("ModelConfig", NameVersion(model_name, model_version), headers, model_config, timeout_us);
}

Error InferenceServerGrpcClient::ModelRepositoryIndex(std::string* repository_index, const Headers& headers, uint64_t timeout_us) {
  return JsonCall("RepositoryIndex", "{}", headers, repository_index, timeout_us);
}

Error InferenceServerGrpcClient::LoadModel(const std::string& model_name, const Headers& headers,
                                           const std::string& config,
                                           const std::map<std::string, std::vector<char>>& files, uint64_t timeout_us) {
  json::Value req = json::Value::Object();
  req["model_name"] = model_name;
  json::Value params = json::Value::Object();
  if (!config.empty()) {
    json::Value p = json::Value::Object();
    p["string_param"] = config;
    params["config"] = p;
  }
  for (const auto& kv : files) {
    json::Value p = json::Value::Object();
    p["bytes_param"] = Base64Encode(kv.second.data(), kv.second.size());
    params[kv.first] = p;
  }
  if (params.Members().size()) req["parameters"] = params;
  return JsonCall("RepositoryModelLoad", req.Dump(), headers, nullptr, timeout_us);
}

Error InferenceServerGrpcClient::UnloadModel(const std::string& model_name, const Headers& headers, uint64_t timeout_us) {
  json::Value req = json::Value::Object();
  req["model_name"] = model_name;
  return JsonCall("RepositoryModelUnload", req.Dump(), headers, nullptr, timeout_us);
}

Error InferenceServerGrpcClient::ModelInferenceStatistics(std::string* infer_stat, const std::string& model_name,
                                                          const std::string& model_version, const Headers& headers, uint64_t timeout_us) {
  return JsonCall("ModelStatistics", NameVersion(model_name, model_version), headers, infer_stat, timeout_us);
}

Error InferenceServerGrpcClient::UpdateTraceSettings(std::string* response, const std::string& model_name,
                                                     const std::map<std::string, std::vector<std::string>>& settings,
                                                     const Headers& headers, uint64_t timeout_us) {
  json::Value req = json::Value::Object();
  if (!model_name.empty()) req["model_name"] = model_name;
  json::Value s = json::Value::Object();
  for (const auto& kv : settings) {
    json::Value entry = json::Value::Object();
    json::Value values = json::Value::Array();
    for (const auto& v : kv.second) values.Append(json::Value(v));
    if (!kv.second.empty()) entry["value"] = values;
    s[kv.first] = entry;
  }
  if (s.Members().size()) req["settings"] = s;
  return JsonCall("TraceSetting", req.Dump(), headers, response, timeout_us);
}

Error InferenceServerGrpcClient::GetTraceSettings(std::string* settings, const std::string& model_name,
                                                  const Headers& headers, uint64_t timeout_us) {
  json::Value req = json::Value::Object();
  if (!model_name.empty()) req["model_name"] = model_name;
  return JsonCall("TraceSetting", req.Dump(), headers, settings, timeout_us);
}

Error InferenceServerGrpcClient::UpdateLogSettings(std::string* response, const std::string& settings_json,
                                                   const Headers& headers, uint64_t timeout_us) {
  // {"log_info": true, "log_verbose_level": 1, "log_format": "default"} -> typed SettingValue entries
  json::Value in;
  TCA_RETURN_IF_ERROR(json::Parse(settings_json.data(), settings_json.size(), &in));
  if (!in.IsObject()) return Error("log settings must be a JSON object");
  json::Value s = json::Value::Object();
  for (const auto& kv : in.Members()) {
    json::Value entry = json::Value::Object();
    if (kv.second.IsBool()) entry["bool_param"] = kv.second;
    else if (kv.second.IsInteger()) entry["uint32_param"] = kv.second;
    else entry["string_param"] = kv.second.IsString() ? kv.second : json::Value(kv.second.Dump());
    s[kv.first] = entry;
  }
  json::Value req = json::Value::Object();
  if (s.Members().size()) req["settings"] = s;
  return JsonCall("LogSettings", req.Dump(), headers, response, timeout_us);
}

Error InferenceServerGrpcClient::GetLogSettings(std::string* settings, const Headers& headers, uint64_t timeout_us) {
  return JsonCall("LogSettings", "{}", headers, settings, timeout_us);
}

Error InferenceServerGrpcClient::SystemSharedMemoryStatus(std::string* status, const std::string& region_name,
                                                          const Headers& headers, uint64_t timeout_us) {
  return JsonCall("SystemSharedMemoryStatus", NameVersion(region_name, ""), headers, status, timeout_us);
}

Error InferenceServerGrpcClient::RegisterSystemSharedMemory(const std::string& name, const std::string& key,
                                                            size_t byte_size, size_t offset, const Headers& headers, uint64_t timeout_us) {
  json::Value req = json::Value::Object();
  req["name"] = name;
  req["key"] = key;
  req["offset"] = static_cast<unsigned long long>(offset);
  req["byte_size"] = static_cast<unsigned long long>(byte_size);
  return JsonCall("SystemSharedMemoryRegister", req.Dump(), headers, nullptr, timeout_us);
}

Error InferenceServerGrpcClient::UnregisterSystemSharedMemory(const std::string& name, const Headers& headers, uint64_t timeout_us) {
  return JsonCall("SystemSharedMemoryUnregister", NameVersion(name, ""), headers, nullptr, timeout_us);
}

Error InferenceServerGrpcClient::CudaSharedMemoryStatus(std::string* status, const std::string& region_name,
                                                        const Headers& headers, uint64_t timeout_us) {
  return JsonCall("CudaSharedMemoryStatus", NameVersion(region_name, ""), headers, status, timeout_us);
}

Error InferenceServerGrpcClient::RegisterCudaSharedMemory(const std::string& name, const void* raw_handle,
                                                          size_t device_id, size_t byte_size, const Headers& headers, uint64_t timeout_us) {
  json::Value req = json::Value::Object();
  req["name"] = name;
  req["raw_handle"] = Base64Encode(raw_handle, 64);
  req["device_id"] = static_cast<long long>(device_id);
  req["byte_size"] = static_cast<unsigned long long>(byte_size);
  return JsonCall("CudaSharedMemoryRegister", req.Dump(), headers, nullptr, timeout_us);
}

Error InferenceServerGrpcClient::UnregisterCudaSharedMemory(const std::string& name, const Headers& headers, uint64_t timeout_us) {
  return JsonCall("CudaSharedMemoryUnregister", NameVersion(name, ""), headers, nullptr, timeout_us);
}

// ---- inference ---------------------------------------------------------------------------
Error InferenceServerGrpcClient::GenerateRequestBody(std::string* request, const InferOptions& options,
                                                     const std::vector<InferInput*>& inputs,
                                                     const std::vector<const InferRequestedOutput*>& outputs) {
  return EncodeInferRequest(request, options, inputs, outputs);
}

Error InferenceServerGrpcClient::ParseResponseBody(InferResult** result, std::string&& response) {
  InferResultGrpc* r = InferResultGrpc::FromResponse(std::move(response));
  *result = r;
  return r->RequestStatus();
}

Error InferenceServerGrpcClient::Infer(InferResult** result, const InferOptions& options,
                                       const std::vector<InferInput*>& inputs,
                                       const std::vector<const InferRequestedOutput*>& outputs, const Headers& headers,
                                       GrpcCompression compression_algorithm) {
  RequestTimers timers;
  timers.CaptureTimestamp(RequestTimers::Kind::REQUEST_START);
  timers.CaptureTimestamp(RequestTimers::Kind::SEND_START);
  std::string request;
  TCA_RETURN_IF_ERROR(EncodeInferRequest(&request, options, inputs, outputs));
  Headers metadata = headers;
  bool compressed = false;
  TCA_RETURN_IF_ERROR(ApplyCompression(compression_algorithm, &request, &metadata, &compressed));
  timers.CaptureTimestamp(RequestTimers::Kind::SEND_END);
  std::string response;
  const Error e = Unary("ModelInfer", request, metadata, options.client_timeout_, &response, compressed);
  timers.CaptureTimestamp(RequestTimers::Kind::RECV_START);
  if (!e.IsOk()) {
    *result = nullptr;
    return e;
  }
  InferResultGrpc* r = InferResultGrpc::FromResponse(std::move(response));
  timers.CaptureTimestamp(RequestTimers::Kind::RECV_END);
  timers.CaptureTimestamp(RequestTimers::Kind::REQUEST_END);
  UpdateInferStat(timers);
  *result = r;
  if (verbose_) std::cout << r->DebugString() << std::endl;
  return r->RequestStatus();
}

Error InferenceServerGrpcClient::AsyncInfer(OnCompleteFn callback, const InferOptions& options,
                                            const std::vector<InferInput*>& inputs,
                                            const std::vector<const InferRequestedOutput*>& outputs,
                                            const Headers& headers, GrpcCompression compression_algorithm) {
  if (callback == nullptr) return Error("Callback function must be provided along with AsyncInfer() call.");
  auto timers = std::make_shared<RequestTimers>();
  timers->CaptureTimestamp(RequestTimers::Kind::REQUEST_START);
  timers->CaptureTimestamp(RequestTimers::Kind::SEND_START);
  std::string request;
  TCA_RETURN_IF_ERROR(EncodeInferRequest(&request, options, inputs, outputs));
  Headers metadata = headers;
  bool compressed = false;
  TCA_RETURN_IF_ERROR(ApplyCompression(compression_algorithm, &request, &metadata, &compressed));
  auto response = std::make_shared<std::string>();
  GrpcChannel::CallEvents ev;
  ev.on_message = [response, timers](std::string&& m) {
    timers->CaptureTimestamp(RequestTimers::Kind::RECV_START);
    *response = std::move(m);
  };
  auto async = async_;
  auto call_id = std::make_shared<uint64_t>(0);  // 0 until StartCall has returned; written and read under async->mu
  auto finished = std::make_shared<bool>(false);
  {
    std::lock_guard<std::mutex> lk(async->mu);
    ++async->pending;
  }
  ev.on_done = [this, async, call_id, finished, callback, response, timers](int status, const std::string& message) {
    InferResultGrpc* r = status == kGrpcOk ? InferResultGrpc::FromResponse(std::move(*response))
                                           : InferResultGrpc::FromError(StatusToError(status, message));
    timers->CaptureTimestamp(RequestTimers::Kind::RECV_END);
    timers->CaptureTimestamp(RequestTimers::Kind::REQUEST_END);
    // `this` is alive: its destructor waits for `pending` to drain
    if (status == kGrpcOk && !exiting_.load()) UpdateInferStat(*timers);
    const void* outer = t_completing;
    t_completing = async.get();
    callback(r);  // may destroy the client: nothing below touches it
    t_completing = outer;
    std::lock_guard<std::mutex> lk(async->mu);
    *finished = true;
    if (*call_id) async->calls.erase(*call_id);
    --async->pending;
    async->cv.notify_all();
  };
  const uint64_t call = channel_->StartCall("ModelInfer", metadata, options.client_timeout_, std::move(ev));
  {
    std::lock_guard<std::mutex> lk(async->mu);
    *call_id = call;
    if (!*finished) async->calls.insert(call);
  }
  // the hand-over to the channel's thread ends the caller's send phase; stamped before it because the answer
  // (and the statistics update that reads the stamps) can arrive before SendMessage returns
  timers->CaptureTimestamp(RequestTimers::Kind::SEND_END);
  channel_->SendMessage(call, std::move(request), true, compressed);
  return Error::Success();
}

namespace {
Error CheckMulti(const std::vector<InferOptions>& options, const std::vector<std::vector<InferInput*>>& inputs,
                 const std::vector<std::vector<const InferRequestedOutput*>>& outputs) {
  if (inputs.empty()) return Error("'inputs' must hold at least one request");
  if (options.size() != 1 && options.size() != inputs.size())
    return Error("'options' must either contain 1 element or match size of 'inputs'");
  if (!outputs.empty() && outputs.size() != 1 && outputs.size() != inputs.size())
    return Error("'outputs' must either contain 0/1 element or match size of 'inputs'");
  return Error::Success();
}
}  // namespace

Error InferenceServerGrpcClient::InferMulti(std::vector<InferResult*>* results, const std::vector<InferOptions>& options,
                                            const std::vector<std::vector<InferInput*>>& inputs,
                                            const std::vector<std::vector<const InferRequestedOutput*>>& outputs,
                                            const Headers& headers, GrpcCompression compression_algorithm) {
  TCA_RETURN_IF_ERROR(CheckMulti(options, inputs, outputs));
  static const std::vector<const InferRequestedOutput*> kNoOutputs;
  results->clear();
  for (size_t i = 0; i < inputs.size(); ++i) {
    const InferOptions& opt = options.size() == 1 ? options[0] : options[i];
    const auto& out = outputs.empty() ? kNoOutputs : (outputs.size() == 1 ? outputs[0] : outputs[i]);
    InferResult* r = nullptr;
    const Error e = Infer(&r, opt, inputs[i], out, headers, compression_algorithm);
    if (!e.IsOk()) {
      delete r;
      for (InferResult* done : *results) delete done;
      results->clear();
      return e;
    }
    results->push_back(r);
  }
  return Error::Success();
}

Error InferenceServerGrpcClient::AsyncInferMulti(OnMultiCompleteFn callback, const std::vector<InferOptions>& options,
                                                 const std::vector<std::vector<InferInput*>>& inputs,
                                                 const std::vector<std::vector<const InferRequestedOutput*>>& outputs,
                                                 const Headers& headers, GrpcCompression compression_algorithm) {
  if (callback == nullptr) return Error("Callback function must be provided along with AsyncInferMulti() call.");
  TCA_RETURN_IF_ERROR(CheckMulti(options, inputs, outputs));
  static const std::vector<const InferRequestedOutput*> kNoOutputs;
  struct Gather {
    std::mutex mu;
    std::vector<InferResult*> results;
    size_t left;
    OnMultiCompleteFn callback;
  };
  auto gather = std::make_shared<Gather>();
  gather->results.resize(inputs.size(), nullptr);
  gather->left = inputs.size();
  gather->callback = std::move(callback);
  for (size_t i = 0; i < inputs.size(); ++i) {
    const InferOptions& opt = options.size() == 1 ? options[0] : options[i];
    const auto& out = outputs.empty() ? kNoOutputs : (outputs.size() == 1 ? outputs[0] : outputs[i]);
    auto one = [gather, i](InferResult* r) {
      bool last = false;
      {
        std::lock_guard<std::mutex> lk(gather->mu);
        gather->results[i] = r;
        last = --gather->left == 0;
      }
      if (last) gather->callback(gather->results);
    };
    const Error e = AsyncInfer(one, opt, inputs[i], out, headers, compression_algorithm);
    if (!e.IsOk()) one(InferResultGrpc::FromError(e));  // keeps the countdown consistent
  }
  return Error::Success();
}

// ---- streaming -------------------------------------------------------------------------------
Error InferenceServerGrpcClient::StartStream(OnCompleteFn callback, bool enable_stats, uint32_t stream_timeout,
                                             const Headers& headers, GrpcCompression compression_algorithm) {
  if (stream_ != nullptr) {
    return Error("cannot start another stream with one already running. 'InferenceServerClient' supports only a "
                 "single active stream at a given time.");
  }
  if (callback == nullptr) return Error("Callback function must be provided along with StartStream() call.");
  auto st = std::make_shared<StreamState>();
  st->compression = compression_algorithm;
  Headers metadata = headers;
  if (compression_algorithm != GrpcCompression::NONE)
    metadata["grpc-encoding"] = compression_algorithm == GrpcCompression::GZIP ? "gzip" : "deflate";
  st->enable_stats = enable_stats;
  st->callback = std::move(callback);
  GrpcChannel::CallEvents ev;
  ev.on_message = [this, st](std::string&& m) {
    InferResultGrpc* r = InferResultGrpc::FromStreamResponse(std::move(m));
    if (st->enable_stats) {
      RequestTimers t;
      bool have = false;
      {
        std::lock_guard<std::mutex> lk(st->mu);
        if (!st->timers.empty()) {
          t = st->timers.front();
          st->timers.pop_front();
          have = true;
        }
      }
      if (have) {
        t.CaptureTimestamp(RequestTimers::Kind::RECV_START);
        t.CaptureTimestamp(RequestTimers::Kind::RECV_END);
        t.CaptureTimestamp(RequestTimers::Kind::REQUEST_END);
        if (!exiting_.load()) UpdateInferStat(t);
      }
    }
    st->callback(r);
  };
  ev.on_done = [st](int status, const std::string& message) {
    if (status != kGrpcOk && status != kGrpcCancelled) {
      // surface a broken stream to the application, as one failed result
      st->callback(InferResultGrpc::FromError(StatusToError(status, message)));
    }
    std::lock_guard<std::mutex> lk(st->mu);
    st->status = status;
    st->message = message;
    st->done = true;
    st->cv.notify_all();
  };
  st->call = channel_->StartCall("ModelStreamInfer", metadata, stream_timeout, std::move(ev));
  stream_ = st;
  return Error::Success();
}

Error InferenceServerGrpcClient::StopStream() {
  if (stream_ == nullptr) return Error::Success();
  auto st = stream_;
  stream_.reset();
  channel_->CloseSend(st->call);
  std::unique_lock<std::mutex> lk(st->mu);
  if (!tca::WaitFor(st->cv, lk, std::chrono::seconds(30), [&] { return st->done; })) {
    lk.unlock();
    channel_->Cancel(st->call);
    lk.lock();
    tca::WaitFor(st->cv, lk, std::chrono::seconds(5), [&] { return st->done; });
  }
  return Error::Success();
}

Error InferenceServerGrpcClient::AsyncStreamInfer(const InferOptions& options, const std::vector<InferInput*>& inputs,
                                                  const std::vector<const InferRequestedOutput*>& outputs) {
  if (stream_ == nullptr) return Error("stream not available, use StartStream() to make one available.");
  {
    std::lock_guard<std::mutex> lk(stream_->mu);
    if (stream_->done) return StatusToError(stream_->status ? stream_->status : kGrpcUnavailable,
                                             stream_->message.empty() ? "the stream is closed" : stream_->message);
  }
  RequestTimers timers;
  timers.CaptureTimestamp(RequestTimers::Kind::REQUEST_START);
  timers.CaptureTimestamp(RequestTimers::Kind::SEND_START);
  std::string request;
  TCA_RETURN_IF_ERROR(EncodeInferRequest(&request, options, inputs, outputs));
  bool compressed = false;
  if (stream_->compression != GrpcCompression::NONE) {
    Headers unused;
    TCA_RETURN_IF_ERROR(ApplyCompression(stream_->compression, &request, &unused, &compressed));
  }
  timers.CaptureTimestamp(RequestTimers::Kind::SEND_END);
  if (stream_->enable_stats) {
    std::lock_guard<std::mutex> lk(stream_->mu);
    stream_->timers.push_back(timers);
  }
  channel_->SendMessage(stream_->call, std::move(request), false, compressed);
  return Error::Success();
}

}  // namespace client
}  // namespace tca
