// Table-driven protobuf <-> JSON codec (see tca/pbjson.h).
#include "tca/pbjson.h"

#include <cstring>

#include "tca/base64.h"
#include "tca/pbwire.h"

namespace tca {
namespace pb {
namespace {

struct EnumValue {
  int number;
  const char* name;
};
struct EnumDesc {
  const char* name;
  const EnumValue* values;
  int count;
};
struct FieldDesc {
  uint32_t number;
  const char* name;
  uint8_t type;  // descriptor.proto FieldDescriptorProto.Type
  uint8_t repeated;
  uint8_t is_map;
  int16_t type_index;  // message or enum index
};
struct MessageDesc {
  const char* name;
  const FieldDesc* fields;
  int count;
};
struct MethodDesc {
  const char* name;
  int request;
  int response;
};

#include "pb_schema.inc"

enum Type : uint8_t {
  T_DOUBLE = 1, T_FLOAT = 2, T_INT64 = 3, T_UINT64 = 4, T_INT32 = 5, T_FIXED64 = 6, T_FIXED32 = 7, T_BOOL = 8,
  T_STRING = 9, T_MESSAGE = 11, T_BYTES = 12, T_UINT32 = 13, T_ENUM = 14, T_SFIXED32 = 15, T_SFIXED64 = 16,
  T_SINT32 = 17, T_SINT64 = 18,
};

const FieldDesc* FindField(const MessageDesc& m, uint32_t number) {
  for (int i = 0; i < m.count; ++i)
    if (m.fields[i].number == number) return &m.fields[i];
  return nullptr;
}

int64_t ZigZag(uint64_t v) { return static_cast<int64_t>((v >> 1) ^ (~(v & 1) + 1)); }

json::Value EnumToJson(int enum_index, int64_t number) {
  const EnumDesc& e = kEnums[enum_index];
  for (int i = 0; i < e.count; ++i)
    if (e.values[i].number == number) return json::Value(e.values[i].name);
  return json::Value(static_cast<long long>(number));
}

// One scalar from the reader, according to the field type.
bool ReadScalar(const FieldDesc& f, Reader* r, uint32_t wire, json::Value* out) {
  uint64_t v = 0;
  uint32_t w = 0;
  switch (f.type) {
    case T_DOUBLE: {
      if (wire != kFixed64 || !r->Fixed64(&v)) return false;
      double d;
      std::memcpy(&d, &v, 8);
      *out = json::Value(d);
      return true;
    }
    case T_FLOAT: {
      if (wire != kFixed32 || !r->Fixed32(&w)) return false;
      float fl;
      std::memcpy(&fl, &w, 4);
      *out = json::Value(static_cast<double>(fl));
      return true;
    }
    case T_INT64: case T_INT32:
      if (wire != kVarint || !r->Varint(&v)) return false;
      *out = json::Value(static_cast<long long>(static_cast<int64_t>(v)));
      return true;
    case T_UINT64: case T_UINT32:
      if (wire != kVarint || !r->Varint(&v)) return false;
      *out = json::Value(static_cast<unsigned long long>(v));
      return true;
    case T_SINT32: case T_SINT64:
      if (wire != kVarint || !r->Varint(&v)) return false;
      *out = json::Value(static_cast<long long>(ZigZag(v)));
      return true;
    case T_FIXED64:
      if (wire != kFixed64 || !r->Fixed64(&v)) return false;
      *out = json::Value(static_cast<unsigned long long>(v));
      return true;
    case T_SFIXED64:
      if (wire != kFixed64 || !r->Fixed64(&v)) return false;
      *out = json::Value(static_cast<long long>(static_cast<int64_t>(v)));
      return true;
    case T_FIXED32:
      if (wire != kFixed32 || !r->Fixed32(&w)) return false;
      *out = json::Value(static_cast<unsigned long long>(w));
      return true;
    case T_SFIXED32:
      if (wire != kFixed32 || !r->Fixed32(&w)) return false;
      *out = json::Value(static_cast<long long>(static_cast<int32_t>(w)));
      return true;
    case T_BOOL:
      if (wire != kVarint || !r->Varint(&v)) return false;
      *out = json::Value(v != 0);
      return true;
    case T_ENUM:
      if (wire != kVarint || !r->Varint(&v)) return false;
      *out = EnumToJson(f.type_index, static_cast<int32_t>(v));
      return true;
    default:
      return false;
  }
}

uint32_t PackedWire(uint8_t type) {
  switch (type) {
    case T_DOUBLE: case T_FIXED64: case T_SFIXED64: return kFixed64;
    case T_FLOAT: case T_FIXED32: case T_SFIXED32: return kFixed32;
    default: return kVarint;
  }
}

Error Decode(int message, const uint8_t* data, size_t len, json::Value* out, int depth);

Error DecodeField(const FieldDesc& f, Reader* r, uint32_t wire, json::Value* slot, int depth) {
  if (f.type == T_STRING || f.type == T_BYTES) {
    const uint8_t* d;
    size_t n;
    if (wire != kLen || !r->Len(&d, &n)) return Error("protobuf: malformed string field '" + std::string(f.name) + "'");
    if (f.type == T_STRING) *slot = json::Value(std::string(reinterpret_cast<const char*>(d), n));
    else *slot = json::Value(Base64Encode(d, n));
    return Error::Success();
  }
  if (f.type == T_MESSAGE) {
    const uint8_t* d;
    size_t n;
    if (wire != kLen || !r->Len(&d, &n)) return Error("protobuf: malformed message field '" + std::string(f.name) + "'");
    return Decode(f.type_index, d, n, slot, depth + 1);
  }
  if (!ReadScalar(f, r, wire, slot)) return Error("protobuf: malformed field '" + std::string(f.name) + "'");
  return Error::Success();
}

std::string KeyToString(const json::Value& v) {
  if (v.IsString()) return v.AsString();
  if (v.IsBool()) return v.AsBool() ? "true" : "false";
  return v.Dump();
}

Error Decode(int message, const uint8_t* data, size_t len, json::Value* out, int depth) {
  if (depth > 64) return Error("protobuf: message nesting too deep");
  const MessageDesc& m = kMessages[message];
  *out = json::Value::Object();
  Reader r(data, len);
  uint32_t number, wire;
  while (r.Next(&number, &wire)) {
    const FieldDesc* f = FindField(m, number);
    if (f == nullptr) {
      if (!r.Skip(wire)) break;
      continue;
    }
    if (f->is_map) {
      const uint8_t* d;
      size_t n;
      if (wire != kLen || !r.Len(&d, &n)) return Error("protobuf: malformed map field '" + std::string(f->name) + "'");
      json::Value entry;
      TCA_RETURN_IF_ERROR(Decode(f->type_index, d, n, &entry, depth + 1));
      json::Value& obj = (*out)[f->name];
      if (!obj.IsObject()) obj = json::Value::Object();
      const json::Value* key = entry.Find("key");
      const json::Value* value = entry.Find("value");
      const std::string k = key ? KeyToString(*key) : std::string();
      if (value != nullptr) {
        obj[k] = *value;
      } else {  // value at its default: synthesise it from the entry type
        const FieldDesc* vf = FindField(kMessages[f->type_index], 2);
        if (vf && vf->type == T_MESSAGE) obj[k] = json::Value::Object();
        else if (vf && (vf->type == T_STRING || vf->type == T_BYTES)) obj[k] = json::Value("");
        else if (vf && vf->type == T_BOOL) obj[k] = json::Value(false);
        else obj[k] = json::Value(0);
      }
      continue;
    }
    if (f->repeated) {
      json::Value& arr = (*out)[f->name];
      if (!arr.IsArray()) arr = json::Value::Array();
      const bool packable = f->type != T_STRING && f->type != T_BYTES && f->type != T_MESSAGE;
      if (packable && wire == kLen) {
        const uint8_t* d;
        size_t n;
        if (!r.Len(&d, &n)) return Error("protobuf: malformed packed field '" + std::string(f->name) + "'");
        Reader pr(d, n);
        while (!pr.Done()) {
          json::Value item;
          if (!ReadScalar(*f, &pr, PackedWire(f->type), &item))
            return Error("protobuf: malformed packed field '" + std::string(f->name) + "'");
          arr.Append(std::move(item));
        }
      } else {
        json::Value item;
        TCA_RETURN_IF_ERROR(DecodeField(*f, &r, wire, &item, depth));
        arr.Append(std::move(item));
      }
      continue;
    }
    json::Value item;
    TCA_RETURN_IF_ERROR(DecodeField(*f, &r, wire, &item, depth));
    (*out)[f->name] = std::move(item);
  }
  if (r.Failed()) return Error(std::string("protobuf: malformed ") + m.name);
  return Error::Success();
}

// ---- encoding ----
bool JsonToInt(const json::Value& v, int64_t* out) {
  if (v.IsBool()) { *out = v.AsBool(); return true; }
  if (v.IsNumber()) { *out = v.IsInteger() ? v.AsInt() : static_cast<int64_t>(v.AsDouble()); return true; }
  if (v.IsString()) {
    char* end = nullptr;
    const long long r = std::strtoll(v.AsString().c_str(), &end, 10);
    if (end == v.AsString().c_str() || *end) return false;
    *out = r;
    return true;
  }
  return false;
}

Error Encode(int message, const json::Value& value, std::string* out, int depth);

Error EncodeScalar(const FieldDesc& f, const json::Value& v, Writer* w, bool packed_element) {
  const auto bad = [&]() { return Error("cannot convert JSON to field '" + std::string(f.name) + "'"); };
  int64_t i = 0;
  switch (f.type) {
    case T_DOUBLE: {
      if (!v.IsNumber()) return bad();
      const double d = v.AsDouble();
      uint64_t bits;
      std::memcpy(&bits, &d, 8);
      if (!packed_element) w->Tag(f.number, kFixed64);
      for (int b = 0; b < 8; ++b) w->out()->push_back(static_cast<char>(bits >> (8 * b)));
      return Error::Success();
    }
    case T_FLOAT: {
      if (!v.IsNumber()) return bad();
      const float d = static_cast<float>(v.AsDouble());
      uint32_t bits;
      std::memcpy(&bits, &d, 4);
      if (!packed_element) w->Tag(f.number, kFixed32);
      for (int b = 0; b < 4; ++b) w->out()->push_back(static_cast<char>(bits >> (8 * b)));
      return Error::Success();
    }
    case T_ENUM: {
      if (v.IsString()) {
        const EnumDesc& e = kEnums[f.type_index];
        bool found = false;
        for (int k = 0; k < e.count; ++k)
          if (v.AsString() == e.values[k].name) { i = e.values[k].number; found = true; break; }
        if (!found && !JsonToInt(v, &i)) return bad();
      } else if (!JsonToInt(v, &i)) {
        return bad();
      }
      if (!packed_element) w->Tag(f.number, kVarint);
      w->Varint(static_cast<uint64_t>(i));
      return Error::Success();
    }
    case T_INT64: case T_INT32: case T_UINT64: case T_UINT32: case T_BOOL:
      if (v.kind() == json::Value::Kind::UInt) i = static_cast<int64_t>(v.AsUInt());
      else if (!JsonToInt(v, &i)) return bad();
      if (!packed_element) w->Tag(f.number, kVarint);
      w->Varint(static_cast<uint64_t>(i));
      return Error::Success();
    case T_SINT32: case T_SINT64:
      if (!JsonToInt(v, &i)) return bad();
      if (!packed_element) w->Tag(f.number, kVarint);
      w->Varint((static_cast<uint64_t>(i) << 1) ^ static_cast<uint64_t>(i >> 63));
      return Error::Success();
    case T_FIXED64: case T_SFIXED64:
      if (!JsonToInt(v, &i)) return bad();
      if (!packed_element) w->Tag(f.number, kFixed64);
      for (int b = 0; b < 8; ++b) w->out()->push_back(static_cast<char>(static_cast<uint64_t>(i) >> (8 * b)));
      return Error::Success();
    case T_FIXED32: case T_SFIXED32:
      if (!JsonToInt(v, &i)) return bad();
      if (!packed_element) w->Tag(f.number, kFixed32);
      for (int b = 0; b < 4; ++b) w->out()->push_back(static_cast<char>(static_cast<uint32_t>(i) >> (8 * b)));
      return Error::Success();
    default:
      return bad();
  }
}

Error EncodeField(const FieldDesc& f, const json::Value& v, Writer* w, int depth) {
  if (f.type == T_STRING) {
    w->String(f.number, v.IsString() ? v.AsString() : v.Dump());
    return Error::Success();
  }
  if (f.type == T_BYTES) {
    if (!v.IsString()) return Error("field '" + std::string(f.name) + "' expects base64 text");
    std::string raw;
    if (!Base64Decode(v.AsString(), &raw)) return Error("field '" + std::string(f.name) + "' holds invalid base64");
    w->String(f.number, raw);
    return Error::Success();
  }
  if (f.type == T_MESSAGE) {
    std::string body;
    TCA_RETURN_IF_ERROR(Encode(f.type_index, v, &body, depth + 1));
    w->Message(f.number, body);
    return Error::Success();
  }
  return EncodeScalar(f, v, w, false);
}

Error Encode(int message, const json::Value& value, std::string* out, int depth) {
  if (depth > 64) return Error("protobuf: message nesting too deep");
  const MessageDesc& m = kMessages[message];
  if (!value.IsObject()) return Error(std::string("JSON object expected for ") + m.name);
  Writer w(out);
  for (const auto& member : value.Members()) {
    const FieldDesc* f = nullptr;
    for (int i = 0; i < m.count; ++i)
      if (member.first == m.fields[i].name) { f = &m.fields[i]; break; }
    if (f == nullptr) return Error("unknown field '" + member.first + "' for " + m.name);
    const json::Value& v = member.second;
    if (f->is_map) {
      if (!v.IsObject()) return Error("JSON object expected for map field '" + member.first + "'");
      const MessageDesc& entry = kMessages[f->type_index];
      const FieldDesc* kf = FindField(entry, 1);
      const FieldDesc* vf = FindField(entry, 2);
      for (const auto& kv : v.Members()) {
        std::string body;
        Writer ew(&body);
        if (kf->type == T_STRING) {
          ew.String(1, kv.first);
        } else {
          TCA_RETURN_IF_ERROR(EncodeScalar(*kf, json::Value(kv.first), &ew, false));
        }
        TCA_RETURN_IF_ERROR(EncodeField(*vf, kv.second, &ew, depth + 1));
        w.Message(f->number, body);
      }
    } else if (f->repeated) {
      if (!v.IsArray()) return Error("JSON array expected for repeated field '" + member.first + "'");
      const bool packable = f->type != T_STRING && f->type != T_BYTES && f->type != T_MESSAGE;
      if (packable) {
        std::string body;
        Writer pw(&body);
        for (const auto& item : v.Items()) TCA_RETURN_IF_ERROR(EncodeScalar(*f, item, &pw, true));
        if (!body.empty()) w.Message(f->number, body);
      } else {
        for (const auto& item : v.Items()) TCA_RETURN_IF_ERROR(EncodeField(*f, item, &w, depth));
      }
    } else {
      TCA_RETURN_IF_ERROR(EncodeField(*f, v, &w, depth));
    }
  }
  return Error::Success();
}

}  // namespace

int FindMessage(const std::string& full_name) {
  for (int i = 0; i < kMessageCount; ++i)
    if (full_name == kMessages[i].name) return i;
  return -1;
}

bool FindMethod(const std::string& method, int* request_msg, int* response_msg) {
  for (int i = 0; i < kMethodCount; ++i) {
    if (method == kMethods[i].name) {
      *request_msg = kMethods[i].request;
      *response_msg = kMethods[i].response;
      return true;
    }
  }
  return false;
}

Error ToJson(int message, const void* data, size_t len, json::Value* out) {
  if (message < 0 || message >= kMessageCount) return Error("unknown protobuf message");
  return Decode(message, static_cast<const uint8_t*>(data), len, out, 0);
}

Error FromJson(int message, const json::Value& value, std::string* out) {
  if (message < 0 || message >= kMessageCount) return Error("unknown protobuf message");
  out->clear();
  return Encode(message, value, out, 0);
}

}  // namespace pb
}  // namespace tca
