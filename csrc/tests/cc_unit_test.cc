// Hermetic unit tests of the native core: JSON, base64, BYTES / FP16 codec,
// shm helpers, URL parsing and - through the public static
// GenerateRequestBody / ParseResponseBody - the JSON<->binary tensor conversion
// for every datatype including the FP16 / BF16 / invalid error cases
// (intent of the reference's protected-member tests,
// `src/c++/tests/cc_client_test.cc:1641-2171`).
#include <cmath>
#include <cstring>

#include "minitest.h"
#include "tca/base64.h"
#include "tca/codec.h"
#include "tca/http_client.h"
#include "tca/json.h"
#include "tca/model_info.h"
#include "tca/shm_utils.h"

namespace tc = tca::client;

TEST(JsonRoundTrip) {
  tca::json::Value v;
  ASSERT_OK(tca::json::Parse(R"({"a":[1,-2,3.5,true,null,"x\né😀"],"b":{"c":18446744073709551615},"n":NaN,"i":-Infinity})", &v));
  EXPECT_EQ(v["a"].Size(), size_t(6));
  EXPECT_EQ(v["a"][0].AsInt(), 1);
  EXPECT_EQ(v["a"][1].AsInt(), -2);
  EXPECT_TRUE(v["a"][2].AsDouble() == 3.5);
  EXPECT_TRUE(v["a"][3].AsBool());
  EXPECT_TRUE(v["a"][4].IsNull());
  EXPECT_EQ(v["a"][5].AsString(), std::string("x\n\xc3\xa9\xf0\x9f\x98\x80"));
  EXPECT_EQ(v["b"]["c"].AsUInt(), 18446744073709551615ull);
  EXPECT_TRUE(std::isnan(v["n"].AsDouble()));
  EXPECT_TRUE(std::isinf(v["i"].AsDouble()) && v["i"].AsDouble() < 0);
  tca::json::Value again;
  ASSERT_OK(tca::json::Parse(v.Dump(), &again));
  EXPECT_EQ(again.Dump(), v.Dump());
  tca::json::Value bad;
  EXPECT_TRUE(!tca::json::Parse("{\"a\":}", &bad).IsOk());
  EXPECT_TRUE(!tca::json::Parse("[1,2", &bad).IsOk());
  EXPECT_TRUE(!tca::json::Parse("{} x", &bad).IsOk());
}

TEST(TypedMetadataViews) {
  // HTTP form of the documents
  tca::ModelMetadataInfo md;
  ASSERT_OK(tca::ParseModelMetadata(
      R"({"name":"resnet50","versions":["1"],"platform":"pytorch_rocm","inputs":[{"name":"input","datatype":"FP16","shape":[-1,3,224,224]}],"outputs":[{"name":"output","datatype":"FP32","shape":[-1,1000]}]})",
      &md));
  EXPECT_EQ(md.name, std::string("resnet50"));
  EXPECT_EQ(md.versions.size(), size_t(1));
  EXPECT_EQ(md.inputs[0].shape.size(), size_t(4));
  EXPECT_EQ(md.inputs[0].shape[0], int64_t(-1));
  EXPECT_EQ(md.outputs[0].datatype, std::string("FP32"));
  tca::ModelConfigInfo cfg;
  ASSERT_OK(tca::ParseModelConfig(
      R"({"name":"resnet50","platform":"pytorch_rocm","max_batch_size":128,"input":[{"name":"input","data_type":"TYPE_FP16","format":"FORMAT_NCHW","dims":[3,224,224]}],"output":[{"name":"output","data_type":"TYPE_FP32","dims":[1000]}],"dynamic_batching":{}})",
      &cfg));
  EXPECT_EQ(cfg.max_batch_size, int64_t(128));
  EXPECT_EQ(cfg.inputs[0].datatype, std::string("FP16"));
  EXPECT_EQ(cfg.inputs[0].format, std::string("FORMAT_NCHW"));
  EXPECT_TRUE(cfg.dynamic_batching && !cfg.decoupled && !cfg.sequence_batching);
  // gRPC form: wrapped configuration, 64-bit integers as strings, STRING for BYTES
  ASSERT_OK(tca::ParseModelConfig(
      R"({"config":{"name":"repeat","max_batch_size":"0","input":[{"name":"IN","data_type":"TYPE_STRING","dims":["-1"]}],"model_transaction_policy":{"decoupled":true}}})",
      &cfg));
  EXPECT_EQ(cfg.name, std::string("repeat"));
  EXPECT_EQ(cfg.max_batch_size, int64_t(0));
  EXPECT_EQ(cfg.inputs[0].datatype, std::string("BYTES"));
  EXPECT_EQ(cfg.inputs[0].shape[0], int64_t(-1));
  EXPECT_TRUE(cfg.decoupled);
  std::vector<tca::RepositoryEntry> index;
  ASSERT_OK(tca::ParseRepositoryIndex(R"([{"name":"a","version":"1","state":"READY"},{"name":"b"}])", &index));
  EXPECT_EQ(index.size(), size_t(2));
  EXPECT_EQ(index[0].state, std::string("READY"));
  ASSERT_OK(tca::ParseRepositoryIndex(R"({"models":[{"name":"c","state":"UNAVAILABLE","reason":"x"}]})", &index));
  EXPECT_EQ(index[0].reason, std::string("x"));
  tca::ServerInfo server;
  ASSERT_OK(tca::ParseServerMetadata(R"({"name":"triton","version":"2.0","extensions":["classification","binary_tensor_data"]})", &server));
  EXPECT_EQ(server.extensions.size(), size_t(2));
  EXPECT_TRUE(!tca::ParseModelMetadata("[1,2]", &md).IsOk());
  EXPECT_TRUE(!tca::ParseModelConfig("{\"config\":{}}", &cfg).IsOk());
}

TEST(Base64) {
  const std::string raw("\x00\x01\xfe\xffhello", 9);
  const std::string enc = tca::Base64Encode(raw);
  EXPECT_EQ(enc, std::string("AAH+/2hlbGxv"));
  std::string dec;
  EXPECT_TRUE(tca::Base64Decode(enc, &dec));
  EXPECT_EQ(dec, raw);
  EXPECT_TRUE(tca::Base64Decode("AAH+\n/2hl bGxv", &dec) && dec == raw);  // whitespace tolerated
  EXPECT_EQ(tca::Base64Encode("a"), std::string("YQ=="));
  EXPECT_EQ(tca::Base64Encode("ab"), std::string("YWI="));
  EXPECT_TRUE(!tca::Base64Decode("@@@@", &dec));
}

TEST(BytesCodec) {
  std::vector<std::string> items = {"ab", "", std::string("x\0y", 3)};
  const std::string wire = tca::BytesSerialize(items);
  EXPECT_EQ(wire.size(), size_t(4 * 3 + 5));
  EXPECT_EQ(static_cast<int>(wire[0]), 2);
  std::vector<std::string> back;
  ASSERT_OK(tca::BytesDeserialize(reinterpret_cast<const uint8_t*>(wire.data()), wire.size(), -1, &back));
  EXPECT_TRUE(back == items);
  ASSERT_OK(tca::BytesDeserialize(reinterpret_cast<const uint8_t*>(wire.data()), wire.size(), 2, &back));
  EXPECT_EQ(back.size(), size_t(2));
  EXPECT_TRUE(!tca::BytesDeserialize(reinterpret_cast<const uint8_t*>(wire.data()), wire.size() - 1, -1, &back).IsOk());
  EXPECT_EQ(tca::DataTypeByteSize("FP16"), size_t(2));
  EXPECT_EQ(tca::DataTypeByteSize("BYTES"), size_t(0));
  EXPECT_TRUE(tca::ParseDataType("NOPE") == tca::DataType::INVALID);
}

TEST(HalfAndBf16) {
  const float vals[] = {0.f, 1.f, -2.5f, 65504.f, 6.1e-5f, 3.0e-7f, 1e6f};
  for (float f : vals) {
    const float back = tca::HalfToFloat(tca::FloatToHalf(f));
    if (std::fabs(f) <= 65504.f) EXPECT_TRUE(std::fabs(back - f) <= std::fabs(f) * 1e-3f + 6e-8f);
    else EXPECT_TRUE(std::isinf(back));
  }
  float src[3] = {1.0f, 3.14159274f, -1.00390625f};
  uint16_t b[3];
  float out[3];
  tca::Fp32ToBf16Truncate(src, b, 3);
  tca::Bf16ToFp32(b, out, 3);
  EXPECT_TRUE(out[0] == 1.0f && out[1] == 3.140625f && out[2] == -1.0f);
}

TEST(ShmUtils) {
  const std::string key = "/tca_cc_unit_shm";
  tca::UnlinkSharedMemoryRegion(key);
  int fd = -1;
  ASSERT_OK(tca::CreateSharedMemoryRegion(key, 128, &fd));
  void* addr = nullptr;
  ASSERT_OK(tca::MapSharedMemory(fd, 0, 128, &addr));
  ASSERT_OK(tca::CloseSharedMemory(fd));
  std::memset(addr, 7, 128);
  tca::ShmRegion view;
  ASSERT_OK(view.Attach(key, 64, 64));
  EXPECT_EQ(static_cast<int>(static_cast<unsigned char*>(view.data())[0]), 7);
  EXPECT_TRUE(!view.Attach(key, 100, 64).IsOk());
  ASSERT_OK(tca::UnmapSharedMemory(addr, 128));
  ASSERT_OK(tca::UnlinkSharedMemoryRegion(key));
  EXPECT_TRUE(!tca::UnlinkSharedMemoryRegion(key).IsOk());
}

TEST(UrlParsing) {
  tc::HttpEndpoint ep;
  ASSERT_OK(tc::ParseHttpUrl("localhost:8000", &ep));
  EXPECT_EQ(ep.host, std::string("localhost"));
  EXPECT_EQ(ep.port, 8000);
  ASSERT_OK(tc::ParseHttpUrl("10.0.0.1:80/base/path/", &ep));
  EXPECT_EQ(ep.base_path, std::string("/base/path"));
  ASSERT_OK(tc::ParseHttpUrl("[::1]:9000", &ep));
  EXPECT_EQ(ep.host, std::string("::1"));
  ASSERT_OK(tc::ParseHttpUrl("http://localhost:8000", &ep));
  EXPECT_TRUE(!ep.https);
  EXPECT_EQ(ep.port, 8000);
  ASSERT_OK(tc::ParseHttpUrl("https://example.org/v2prefix", &ep));
  EXPECT_TRUE(ep.https);
  EXPECT_EQ(ep.port, 443);
  EXPECT_EQ(ep.host, std::string("example.org"));
  EXPECT_EQ(ep.base_path, std::string("/v2prefix"));
  EXPECT_TRUE(!tc::ParseHttpUrl("https://:8000", &ep).IsOk());
}

namespace {
template <typename T>
std::vector<uint8_t> Bytes(const std::vector<T>& v) {
  std::vector<uint8_t> out(v.size() * sizeof(T));
  std::memcpy(out.data(), v.data(), out.size());
  return out;
}

// Builds a request with one JSON (non-binary) input and returns the "data" array text.
tca::Error JsonDataOf(const std::string& datatype, const std::vector<int64_t>& shape, const std::vector<uint8_t>& raw,
                      std::string* data_json) {
  tc::InferInput* in = nullptr;
  TCA_RETURN_IF_ERROR(tc::InferInput::Create(&in, "IN", shape, datatype));
  std::unique_ptr<tc::InferInput> guard(in);
  in->AppendRaw(raw.data(), raw.size());
  in->SetBinaryData(false);
  std::vector<char> body;
  size_t header = 0;
  tc::InferOptions opt("m");
  TCA_RETURN_IF_ERROR(tc::InferenceServerHttpClient::GenerateRequestBody(&body, &header, opt, {in}));
  tca::json::Value doc;
  TCA_RETURN_IF_ERROR(tca::json::Parse(body.data(), header, &doc));
  *data_json = doc["inputs"][0]["data"].Dump();
  return tca::Error::Success();
}
}  // namespace

TEST(BinaryInputToJsonEveryDatatype) {
  std::string j;
  ASSERT_OK(JsonDataOf("BOOL", {3}, {1, 0, 1}, &j));
  EXPECT_EQ(j, std::string("[true,false,true]"));
  ASSERT_OK(JsonDataOf("UINT8", {2}, {1, 255}, &j));
  EXPECT_EQ(j, std::string("[1,255]"));
  ASSERT_OK(JsonDataOf("INT8", {2}, {1, 0xFF}, &j));
  EXPECT_EQ(j, std::string("[1,-1]"));
  ASSERT_OK(JsonDataOf("UINT16", {2}, Bytes<uint16_t>({1, 65535}), &j));
  EXPECT_EQ(j, std::string("[1,65535]"));
  ASSERT_OK(JsonDataOf("INT16", {2}, Bytes<int16_t>({-2, 300}), &j));
  EXPECT_EQ(j, std::string("[-2,300]"));
  ASSERT_OK(JsonDataOf("UINT32", {1}, Bytes<uint32_t>({4000000000u}), &j));
  EXPECT_EQ(j, std::string("[4000000000]"));
  ASSERT_OK(JsonDataOf("INT32", {2}, Bytes<int32_t>({-7, 8}), &j));
  EXPECT_EQ(j, std::string("[-7,8]"));
  ASSERT_OK(JsonDataOf("UINT64", {1}, Bytes<uint64_t>({18446744073709551615ull}), &j));
  EXPECT_EQ(j, std::string("[18446744073709551615]"));
  ASSERT_OK(JsonDataOf("INT64", {1}, Bytes<int64_t>({-9000000000000000000ll}), &j));
  EXPECT_EQ(j, std::string("[-9000000000000000000]"));
  ASSERT_OK(JsonDataOf("FP32", {2}, Bytes<float>({1.5f, -0.25f}), &j));
  EXPECT_EQ(j, std::string("[1.5,-0.25]"));
  ASSERT_OK(JsonDataOf("FP64", {1}, Bytes<double>({2.5}), &j));
  EXPECT_EQ(j, std::string("[2.5]"));
  const std::string bytes_wire = tca::BytesSerialize({"ab", "c"});
  ASSERT_OK(JsonDataOf("BYTES", {2}, std::vector<uint8_t>(bytes_wire.begin(), bytes_wire.end()), &j));
  EXPECT_EQ(j, std::string("[\"ab\",\"c\"]"));
  EXPECT_ERROR_CONTAINS(JsonDataOf("FP16", {1}, {0, 0}, &j), "FP16");
  EXPECT_ERROR_CONTAINS(JsonDataOf("BF16", {1}, {0, 0}, &j), "BF16");
  tc::InferInput* bad = nullptr;
  EXPECT_ERROR_CONTAINS(tc::InferInput::Create(&bad, "IN", {1}, "NOPE"), "invalid datatype");
}

TEST(JsonOutputToBinaryEveryDatatype) {
  const std::string head = R"({"model_name":"m","model_version":"1","outputs":[
    {"name":"b","datatype":"BOOL","shape":[2],"data":[true,false]},
    {"name":"u8","datatype":"UINT8","shape":[2],"data":[1,255]},
    {"name":"i16","datatype":"INT16","shape":[2],"data":[-2,300]},
    {"name":"i32","datatype":"INT32","shape":[2,2],"data":[[1,2],[3,-4]]},
    {"name":"u64","datatype":"UINT64","shape":[1],"data":[18446744073709551615]},
    {"name":"f32","datatype":"FP32","shape":[2],"data":[1.5,2]},
    {"name":"f64","datatype":"FP64","shape":[2],"data":[0.1,-3]},
    {"name":"s","datatype":"BYTES","shape":[2],"data":["ab","c"]},
    {"name":"h","datatype":"FP16","shape":[1],"data":[1.0]}]})";
  std::vector<char> body(head.begin(), head.end());
  tc::InferResult* res = nullptr;
  ASSERT_OK(tc::InferenceServerHttpClient::ParseResponseBody(&res, body, 0));
  std::unique_ptr<tc::InferResult> guard(res);
  ASSERT_OK(res->RequestStatus());
  const uint8_t* buf = nullptr;
  size_t n = 0;
  ASSERT_OK(res->RawData("b", &buf, &n));
  EXPECT_TRUE(n == 2 && buf[0] == 1 && buf[1] == 0);
  ASSERT_OK(res->RawData("u8", &buf, &n));
  EXPECT_TRUE(n == 2 && buf[1] == 255);
  ASSERT_OK(res->RawData("i16", &buf, &n));
  EXPECT_TRUE(n == 4 && reinterpret_cast<const int16_t*>(buf)[0] == -2 && reinterpret_cast<const int16_t*>(buf)[1] == 300);
  ASSERT_OK(res->RawData("i32", &buf, &n));
  EXPECT_TRUE(n == 16 && tca::LoadAs<int32_t>(buf, 3) == -4);
  ASSERT_OK(res->RawData("u64", &buf, &n));
  EXPECT_TRUE(n == 8 && *reinterpret_cast<const uint64_t*>(buf) == 18446744073709551615ull);
  ASSERT_OK(res->RawData("f32", &buf, &n));
  EXPECT_TRUE(n == 8 && tca::LoadAs<float>(buf, 0) == 1.5f && tca::LoadAs<float>(buf, 1) == 2.f);
  ASSERT_OK(res->RawData("f64", &buf, &n));
  EXPECT_TRUE(n == 16 && reinterpret_cast<const double*>(buf)[0] == 0.1);  // 8-byte doubles, not floats
  std::vector<std::string> strs;
  ASSERT_OK(res->StringData("s", &strs));
  EXPECT_TRUE(strs.size() == 2 && strs[0] == "ab" && strs[1] == "c");
  EXPECT_TRUE(!res->RawData("h", &buf, &n).IsOk());  // FP16 has no JSON form
  std::vector<int64_t> shape;
  ASSERT_OK(res->Shape("i32", &shape));
  EXPECT_TRUE(shape.size() == 2 && shape[0] == 2 && shape[1] == 2);
  EXPECT_ERROR_CONTAINS(res->Shape("nope", &shape), "does not contain results");
  EXPECT_ERROR_CONTAINS(res->StringData("i32", &strs), "BYTES");
}

TEST(MixedBinaryResponse) {
  const std::string head = R"({"model_name":"m","model_version":"1","id":"7","outputs":[
    {"name":"A","datatype":"INT32","shape":[2],"parameters":{"binary_data_size":8}},
    {"name":"J","datatype":"INT32","shape":[1],"data":[9]},
    {"name":"B","datatype":"BYTES","shape":[1],"parameters":{"binary_data_size":6}}]})";
  std::vector<char> body(head.begin(), head.end());
  const int32_t a[2] = {5, -6};
  body.insert(body.end(), reinterpret_cast<const char*>(a), reinterpret_cast<const char*>(a) + 8);
  const std::string s = tca::BytesSerialize({"hi"});
  body.insert(body.end(), s.begin(), s.end());
  tc::InferResult* res = nullptr;
  ASSERT_OK(tc::InferenceServerHttpClient::ParseResponseBody(&res, body, head.size()));
  std::unique_ptr<tc::InferResult> guard(res);
  const uint8_t* buf = nullptr;
  size_t n = 0;
  ASSERT_OK(res->RawData("A", &buf, &n));
  // tensors sit at arbitrary offsets inside the response body: no aligned loads
  auto load_i32 = [](const uint8_t* p) {
    int32_t v;
    std::memcpy(&v, p, sizeof(v));
    return v;
  };
  EXPECT_TRUE(n == 8 && load_i32(buf + 4) == -6);
  ASSERT_OK(res->RawData("J", &buf, &n));
  EXPECT_TRUE(n == 4 && load_i32(buf) == 9);
  std::vector<std::string> strs;
  ASSERT_OK(res->StringData("B", &strs));
  EXPECT_TRUE(strs.size() == 1 && strs[0] == "hi");
  std::string id;
  ASSERT_OK(res->Id(&id));
  EXPECT_EQ(id, std::string("7"));
}

TEST(RequestBodyLayout) {
  tc::InferInput* in0 = nullptr;
  tc::InferInput* in1 = nullptr;
  ASSERT_OK(tc::InferInput::Create(&in0, "INPUT0", {1, 4}, "INT32"));
  ASSERT_OK(tc::InferInput::Create(&in1, "INPUT1", {2}, "BYTES"));
  std::unique_ptr<tc::InferInput> g0(in0), g1(in1);
  const int32_t a[4] = {1, 2, 3, 4};
  in0->AppendRaw(reinterpret_cast<const uint8_t*>(a), 8);  // scatter-gather: two pieces
  in0->AppendRaw(reinterpret_cast<const uint8_t*>(a) + 8, 8);
  ASSERT_OK(in1->AppendFromString({"xy", "z"}));
  tc::InferRequestedOutput* out = nullptr;
  ASSERT_OK(tc::InferRequestedOutput::Create(&out, "OUTPUT0", 3));
  std::unique_ptr<tc::InferRequestedOutput> go(out);
  EXPECT_ERROR_CONTAINS(out->SetSharedMemory("r", 8), "classification");
  tc::InferOptions opt("simple");
  opt.request_id_ = "id1";
  opt.sequence_id_ = 42;
  opt.sequence_start_ = true;
  opt.priority_ = 3;
  opt.server_timeout_ = 1000;
  tc::RequestParameter p;
  p.type = tc::RequestParameter::Type::BOOL;
  p.bool_value = true;
  opt.request_parameters["flag"] = p;
  std::vector<char> body;
  size_t header = 0;
  ASSERT_OK(tc::InferenceServerHttpClient::GenerateRequestBody(&body, &header, opt, {in0, in1}, {out}));
  tca::json::Value doc;
  ASSERT_OK(tca::json::Parse(body.data(), header, &doc));
  EXPECT_EQ(doc["id"].AsString(), std::string("id1"));
  EXPECT_EQ(doc["parameters"]["sequence_id"].AsInt(), 42);
  EXPECT_TRUE(doc["parameters"]["sequence_start"].AsBool() && !doc["parameters"]["sequence_end"].AsBool());
  EXPECT_EQ(doc["parameters"]["priority"].AsInt(), 3);
  EXPECT_EQ(doc["parameters"]["timeout"].AsInt(), 1000);
  EXPECT_TRUE(doc["parameters"]["flag"].AsBool());
  EXPECT_EQ(doc["inputs"][0]["parameters"]["binary_data_size"].AsInt(), 16);
  EXPECT_EQ(doc["inputs"][1]["parameters"]["binary_data_size"].AsInt(), 4 + 2 + 4 + 1);
  EXPECT_EQ(doc["outputs"][0]["parameters"]["classification"].AsInt(), 3);
  EXPECT_EQ(body.size(), header + 16 + 11);
  EXPECT_TRUE(std::memcmp(body.data() + header, a, 16) == 0);
  p.type = tc::RequestParameter::Type::INT;
  opt.request_parameters["priority"] = p;
  EXPECT_ERROR_CONTAINS(tc::InferenceServerHttpClient::GenerateRequestBody(&body, &header, opt, {in0}), "reserved");
  // shared memory input
  in0->SetSharedMemory("region", 16, 32);
  tc::InferOptions opt2("simple");
  ASSERT_OK(tc::InferenceServerHttpClient::GenerateRequestBody(&body, &header, opt2, {in0}));
  ASSERT_OK(tca::json::Parse(body.data(), header, &doc));
  EXPECT_EQ(doc["inputs"][0]["parameters"]["shared_memory_region"].AsString(), std::string("region"));
  EXPECT_EQ(doc["inputs"][0]["parameters"]["shared_memory_offset"].AsInt(), 32);
  EXPECT_TRUE(doc["parameters"]["binary_data_output"].AsBool());
  EXPECT_EQ(body.size(), header);
}

TEST(RequestTimersAndStat) {
  tc::RequestTimers t;
  using K = tc::RequestTimers::Kind;
  EXPECT_TRUE(t.Duration(K::REQUEST_START, K::REQUEST_END) == UINT64_MAX);
  t.CaptureTimestamp(K::REQUEST_START);
  t.CaptureTimestamp(K::REQUEST_END);
  EXPECT_TRUE(t.Duration(K::REQUEST_START, K::REQUEST_END) != UINT64_MAX);
  EXPECT_TRUE(t.Duration(K::REQUEST_END, K::REQUEST_START) == UINT64_MAX || t.Timestamp(K::REQUEST_START) == t.Timestamp(K::REQUEST_END));
}

int main(int argc, char** argv) { return minitest::RunAllTests(argc > 1 ? argv[1] : nullptr); }
