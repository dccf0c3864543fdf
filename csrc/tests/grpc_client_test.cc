// Tests of the native gRPC stack.
//
// Unit part (no server): HPACK primitives and the Huffman code against the
// published RFC 7541 Appendix C vectors, an in-memory HTTP/2 client <-> server
// exchange with flow control, protobuf <-> JSON round trips, request encoding
// / response parsing.  Server-backed part (`grpc_client_test -u host:port`,
// against the in-repo v2 server, which is a grpcio server - so this doubles as
// an interoperability test with the stock gRPC implementation): the
// reference's typed client suite (`src/c++/tests/cc_client_test.cc:298-1639`)
// for the gRPC flavour, plus streaming, sequences, decoupled responses,
// timeouts and the channel cache.
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>

#include "minitest.h"
#include "tca/grpc_client.h"
#include "tca/sync.h"
#include "tca/h2.h"
#include "tca/json.h"
#include "tca/pbjson.h"
#include "tca/pbwire.h"
#include "tca/shm_utils.h"

namespace tc = tca::client;
namespace h2 = tca::h2;

static std::string g_url;

namespace {

std::string Hex(const std::string& s) {
  static const char* d = "0123456789abcdef";
  std::string out;
  for (unsigned char c : s) {
    out.push_back(d[c >> 4]);
    out.push_back(d[c & 15]);
  }
  return out;
}

std::string FromHex(const std::string& h) {
  std::string out;
  for (size_t i = 0; i + 1 < h.size(); i += 2) out.push_back(static_cast<char>(std::stoi(h.substr(i, 2), nullptr, 16)));
  return out;
}

}  // namespace

// ---------------------------------------------------------------- HPACK / Huffman
TEST(HuffmanRfcVectors) {
  // RFC 7541 Appendix C.4 / C.6 request and response examples
  const std::pair<const char*, const char*> vectors[] = {
      {"www.example.com", "f1e3c2e5f23a6ba0ab90f4ff"},
      {"no-cache", "a8eb10649cbf"},
      {"custom-key", "25a849e95ba97d7f"},
      {"custom-value", "25a849e95bb8e8b4bf"},
      {"302", "6402"},
      {"private", "aec3771a4b"},
      {"Mon, 21 Oct 2013 20:13:21 GMT", "d07abe941054d444a8200595040b8166e082a62d1bff"},
      {"https://www.example.com", "9d29ad171863c78f0b97c8e9ae82ae43d3"},
      {"307", "640eff"},
      {"gzip", "9bd9ab"},
  };
  for (const auto& v : vectors) {
    std::string enc;
    h2::HuffmanEncode(v.first, &enc);
    EXPECT_EQ(Hex(enc), std::string(v.second));
    std::string dec;
    const std::string raw = FromHex(v.second);
    EXPECT_OK(h2::HuffmanDecode(reinterpret_cast<const uint8_t*>(raw.data()), raw.size(), &dec));
    EXPECT_EQ(dec, std::string(v.first));
  }
  // every byte value survives a round trip
  std::string all;
  for (int i = 0; i < 256; ++i) all.push_back(static_cast<char>(i));
  std::string enc, dec;
  h2::HuffmanEncode(all, &enc);
  EXPECT_OK(h2::HuffmanDecode(reinterpret_cast<const uint8_t*>(enc.data()), enc.size(), &dec));
  EXPECT_TRUE(dec == all);
  // bad padding (zeros) is rejected
  const std::string bad = FromHex("f1e3c2e5f23a6ba0ab90f400");
  EXPECT_TRUE(!h2::HuffmanDecode(reinterpret_cast<const uint8_t*>(bad.data()), bad.size(), &dec).IsOk());
}

TEST(HpackRfcRequestSequence) {
  // RFC 7541 C.4.1 - C.4.3: three requests on one connection, Huffman coded, with
  // incremental indexing (the second and third reference the dynamic table)
  h2::HpackDecoder dec;
  {
    const std::string block = FromHex("828684418cf1e3c2e5f23a6ba0ab90f4ff");
    h2::HeaderList h;
    EXPECT_OK(dec.Decode(reinterpret_cast<const uint8_t*>(block.data()), block.size(), &h));
    EXPECT_EQ(h.size(), size_t(4));
    if (h.size() == 4) {
      EXPECT_EQ(h[0].name + "=" + h[0].value, std::string(":method=GET"));
      EXPECT_EQ(h[3].name + "=" + h[3].value, std::string(":authority=www.example.com"));
    }
  }
  {
    const std::string block = FromHex("828684be5886a8eb10649cbf");
    h2::HeaderList h;
    EXPECT_OK(dec.Decode(reinterpret_cast<const uint8_t*>(block.data()), block.size(), &h));
    EXPECT_EQ(h.size(), size_t(5));
    if (h.size() == 5) {
      EXPECT_EQ(h[3].value, std::string("www.example.com"));  // from the dynamic table
      EXPECT_EQ(h[4].name + "=" + h[4].value, std::string("cache-control=no-cache"));
    }
  }
  {
    const std::string block = FromHex("828785bf408825a849e95ba97d7f8925a849e95bb8e8b4bf");
    h2::HeaderList h;
    EXPECT_OK(dec.Decode(reinterpret_cast<const uint8_t*>(block.data()), block.size(), &h));
    EXPECT_EQ(h.size(), size_t(5));
    if (h.size() == 5) {
      EXPECT_EQ(h[1].name + "=" + h[1].value, std::string(":scheme=https"));
      EXPECT_EQ(h[2].value, std::string("/index.html"));
      EXPECT_EQ(h[4].name + "=" + h[4].value, std::string("custom-key=custom-value"));
    }
  }
}

TEST(HpackEncoderRoundTrip) {
  h2::HeaderList in = {{":method", "POST"}, {":path", "/inference.GRPCInferenceService/ModelInfer"},
                       {"content-type", "application/grpc"}, {"x-custom", std::string(300, 'v')}, {"te", "trailers"}};
  std::string block;
  h2::HpackEncoder::Encode(in, &block);
  h2::HpackDecoder dec;
  h2::HeaderList out;
  EXPECT_OK(dec.Decode(reinterpret_cast<const uint8_t*>(block.data()), block.size(), &out));
  EXPECT_EQ(out.size(), in.size());
  for (size_t i = 0; i < in.size() && i < out.size(); ++i) {
    EXPECT_EQ(out[i].name, in[i].name);
    EXPECT_EQ(out[i].value, in[i].value);
  }
}

// ---------------------------------------------------------------- HTTP/2 engine
namespace {
// Moves bytes between two in-memory connections until both are quiet.
void Pump(h2::Connection* a, h2::Connection* b) {
  for (int round = 0; round < 20000 && (a->WantsWrite() || b->WantsWrite()); ++round) {
    for (auto pair : {std::make_pair(a, b), std::make_pair(b, a)}) {
      h2::Connection* from = pair.first;
      h2::Connection* to = pair.second;
      if (!from->WantsWrite()) continue;
      // deliver in odd-sized pieces to exercise partial-frame buffering
      const std::string chunk = from->Output().substr(from->OutputOffset(), 1000 + 37 * round % 5000);
      from->ConsumeOutput(chunk.size());
      EXPECT_OK(to->Receive(reinterpret_cast<const uint8_t*>(chunk.data()), chunk.size()));
    }
  }
}
}  // namespace

TEST(Http2LoopbackWithFlowControl) {
  std::string got_request, got_response;
  h2::HeaderList req_headers, resp_headers, trailers;
  bool req_done = false, resp_done = false;
  h2::Connection* server_ptr = nullptr;

  h2::Connection::Callbacks scb;
  scb.on_headers = [&](uint32_t, h2::HeaderList&& h, bool) { req_headers = std::move(h); };
  scb.on_data = [&](uint32_t stream, const uint8_t* d, size_t n, bool end) {
    got_request.append(reinterpret_cast<const char*>(d), n);
    if (end) {
      req_done = true;
      EXPECT_OK(server_ptr->SendHeaders(stream, {{":status", "200"}, {"content-type", "application/grpc"}}, false));
      EXPECT_OK(server_ptr->SendData(stream, std::string(3 << 20, 'r'), false));  // larger than every window
      EXPECT_OK(server_ptr->SendHeaders(stream, {{"grpc-status", "0"}}, true));   // must queue behind the data
    }
  };
  h2::Connection server(h2::Connection::Role::SERVER, scb);
  server_ptr = &server;
  server.SetLocalWindow(65535, 65535, 16384);  // small windows: many WINDOW_UPDATE round trips

  h2::Connection::Callbacks ccb;
  ccb.on_headers = [&](uint32_t, h2::HeaderList&& h, bool end) {
    if (resp_headers.empty()) resp_headers = std::move(h);
    else trailers = std::move(h);
    if (end) resp_done = true;
  };
  ccb.on_data = [&](uint32_t, const uint8_t* d, size_t n, bool) { got_response.append(reinterpret_cast<const char*>(d), n); };
  h2::Connection client(h2::Connection::Role::CLIENT, ccb);
  client.SetLocalWindow(100000, 200000, 16384);

  client.Start();
  server.Start();
  const uint32_t stream = client.OpenStream({{":method", "POST"}, {":scheme", "http"}, {":path", "/x"}}, false);
  EXPECT_EQ(stream, 1u);
  std::string body(1 << 20, 'q');
  body[12345] = 'Z';
  EXPECT_OK(client.SendData(stream, std::string(body), true));
  EXPECT_TRUE(client.QueuedBytes() > 0);  // 1 MiB does not fit the initial 64 KiB window
  Pump(&client, &server);
  EXPECT_TRUE(req_done);
  EXPECT_TRUE(got_request == body);
  EXPECT_EQ(req_headers.size(), size_t(3));
  EXPECT_TRUE(resp_done);
  EXPECT_EQ(got_response.size(), size_t(3 << 20));
  EXPECT_EQ(trailers.size(), size_t(1));
  EXPECT_EQ(client.OpenStreams(), size_t(0));
  EXPECT_EQ(server.OpenStreams(), size_t(0));
  EXPECT_EQ(client.QueuedBytes(), size_t(0));
}

TEST(Http2ProtocolErrors) {
  h2::Connection server(h2::Connection::Role::SERVER, {});
  server.Start();
  const std::string junk = "GET / HTTP/1.1\r\n\r\n";
  EXPECT_TRUE(!server.Receive(reinterpret_cast<const uint8_t*>(junk.data()), junk.size()).IsOk());
  EXPECT_TRUE(server.GoawaySent());
}

// ---------------------------------------------------------------- protobuf codec
TEST(PbJsonRoundTrip) {
  const int msg = tca::pb::FindMessage("inference.ModelConfig");
  EXPECT_TRUE(msg >= 0);
  const std::string text =
      R"({"name":"m","platform":"p","max_batch_size":8,"input":[{"name":"IN","data_type":"TYPE_FP16",)"
      R"("format":"FORMAT_NCHW","dims":[3,224,224]}],"output":[{"name":"OUT","data_type":"TYPE_FP32","dims":[1000],)"
      R"("label_filename":"labels.txt"}],"parameters":{"k":{"string_value":"v"}},)"
      R"("dynamic_batching":{"preferred_batch_size":[8,16],"max_queue_delay_microseconds":100}})";
  tca::json::Value v;
  ASSERT_OK(tca::json::Parse(text.data(), text.size(), &v));
  std::string wire;
  ASSERT_OK(tca::pb::FromJson(msg, v, &wire));
  tca::json::Value back;
  ASSERT_OK(tca::pb::ToJson(msg, wire.data(), wire.size(), &back));
  EXPECT_EQ(back.GetString("name"), std::string("m"));
  EXPECT_EQ(back.GetInt("max_batch_size"), int64_t(8));
  EXPECT_EQ(back["input"][0].GetString("data_type"), std::string("TYPE_FP16"));
  EXPECT_EQ(back["input"][0]["dims"].Size(), size_t(3));
  EXPECT_EQ(back["input"][0]["dims"][2].AsInt(), int64_t(224));
  EXPECT_EQ(back["parameters"]["k"].GetString("string_value"), std::string("v"));
  EXPECT_EQ(back["dynamic_batching"]["preferred_batch_size"][1].AsInt(), int64_t(16));
  tca::json::Value bad;
  tca::json::Parse("{\"nope\":1}", 10, &bad);
  EXPECT_TRUE(!tca::pb::FromJson(msg, bad, &wire).IsOk());
}

TEST(InferRequestEncodingAndResponseParsing) {
  tc::InferInput* in = nullptr;
  tc::InferInput::Create(&in, "INPUT0", {1, 4}, "INT32");
  std::unique_ptr<tc::InferInput> holder(in);
  const int32_t data[4] = {1, -2, 3, -4};
  in->AppendRaw(reinterpret_cast<const uint8_t*>(data), 8);
  in->AppendRaw(reinterpret_cast<const uint8_t*>(data) + 8, 8);  // two pieces, gathered into one raw entry
  tc::InferRequestedOutput* out = nullptr;
  tc::InferRequestedOutput::Create(&out, "OUTPUT0", 3);
  std::unique_ptr<tc::InferRequestedOutput> oholder(out);
  tc::InferOptions opt("simple");
  opt.request_id_ = "rid";
  opt.sequence_id_ = 42;
  opt.sequence_start_ = true;
  opt.priority_ = 7;
  std::string wire;
  ASSERT_OK(tc::InferenceServerGrpcClient::GenerateRequestBody(&wire, opt, {in}, {out}));
  tca::json::Value v;
  ASSERT_OK(tca::pb::ToJson(tca::pb::FindMessage("inference.ModelInferRequest"), wire.data(), wire.size(), &v));
  EXPECT_EQ(v.GetString("model_name"), std::string("simple"));
  EXPECT_EQ(v.GetString("id"), std::string("rid"));
  EXPECT_EQ(v["parameters"]["sequence_id"].GetInt("int64_param"), int64_t(42));
  EXPECT_TRUE(v["parameters"]["sequence_start"].GetBool("bool_param"));
  EXPECT_EQ(v["parameters"]["priority"].GetInt("uint64_param"), int64_t(7));
  EXPECT_EQ(v["inputs"][0].GetString("datatype"), std::string("INT32"));
  EXPECT_EQ(v["inputs"][0]["shape"][1].AsInt(), int64_t(4));
  EXPECT_EQ(v["outputs"][0]["parameters"]["classification"].GetInt("int64_param"), int64_t(3));
  EXPECT_EQ(v["raw_input_contents"].Size(), size_t(1));
  // reserved parameter names are refused
  opt.request_parameters["sequence_id"] = tc::RequestParameter();
  EXPECT_ERROR_CONTAINS(tc::InferenceServerGrpcClient::GenerateRequestBody(&wire, opt, {in}, {out}), "reserved");

  // response with one raw and one typed-contents output
  const std::string resp_json =
      R"({"model_name":"simple","model_version":"1","id":"rid","outputs":[{"name":"A","datatype":"INT32","shape":[1,2]},)"
      R"({"name":"B","datatype":"BYTES","shape":[2],"contents":{"bytes_contents":["eA==","eXo="]}}],)"
      R"("raw_output_contents":["BQAAAAYAAAA="]})";
  tca::json::Value rv;
  ASSERT_OK(tca::json::Parse(resp_json.data(), resp_json.size(), &rv));
  std::string resp_wire;
  ASSERT_OK(tca::pb::FromJson(tca::pb::FindMessage("inference.ModelInferResponse"), rv, &resp_wire));
  tc::InferResult* result = nullptr;
  ASSERT_OK(tc::InferenceServerGrpcClient::ParseResponseBody(&result, std::move(resp_wire)));
  std::unique_ptr<tc::InferResult> rh(result);
  const uint8_t* buf = nullptr;
  size_t n = 0;
  EXPECT_OK(result->RawData("A", &buf, &n));
  EXPECT_EQ(n, size_t(8));
  if (n == 8) EXPECT_EQ(tca::LoadAs<int32_t>(buf, 1), 6);
  std::vector<std::string> strings;
  EXPECT_OK(result->StringData("B", &strings));
  EXPECT_EQ(strings.size(), size_t(2));
  if (strings.size() == 2) EXPECT_EQ(strings[1], std::string("yz"));
  std::vector<int64_t> shape;
  EXPECT_OK(result->Shape("A", &shape));
  EXPECT_EQ(shape.size(), size_t(2));
  EXPECT_TRUE(!result->RawData("missing", &buf, &n).IsOk());
}

// ---------------------------------------------------------------- server-backed
namespace {

struct Request {
  std::vector<int32_t> in0, in1;
  std::unique_ptr<tc::InferInput> i0, i1;
  std::vector<tc::InferInput*> inputs;
};

std::unique_ptr<Request> MakeRequest(int seed) {
  std::unique_ptr<Request> r(new Request());
  r->in0.resize(16);
  r->in1.resize(16);
  for (int i = 0; i < 16; ++i) {
    r->in0[i] = i + seed;
    r->in1[i] = 1 + seed;
  }
  tc::InferInput* p = nullptr;
  tc::InferInput::Create(&p, "INPUT0", {1, 16}, "INT32");
  r->i0.reset(p);
  tc::InferInput::Create(&p, "INPUT1", {1, 16}, "INT32");
  r->i1.reset(p);
  r->i0->AppendRaw(reinterpret_cast<const uint8_t*>(r->in0.data()), 64);
  r->i1->AppendRaw(reinterpret_cast<const uint8_t*>(r->in1.data()), 64);
  r->inputs = {r->i0.get(), r->i1.get()};
  return r;
}

void CheckSums(tc::InferResult* res, const Request& req, bool swapped = false) {
  EXPECT_OK(res->RequestStatus());
  const uint8_t* buf = nullptr;
  size_t n = 0;
  EXPECT_OK(res->RawData(swapped ? "OUTPUT1" : "OUTPUT0", &buf, &n));
  EXPECT_EQ(n, size_t(64));
  for (int i = 0; n == 64 && i < 16; ++i) EXPECT_EQ(tca::LoadAs<int32_t>(buf, i), req.in0[i] + req.in1[i]);
  EXPECT_OK(res->RawData(swapped ? "OUTPUT0" : "OUTPUT1", &buf, &n));
  EXPECT_EQ(n, size_t(64));
  for (int i = 0; n == 64 && i < 16; ++i) EXPECT_EQ(tca::LoadAs<int32_t>(buf, i), req.in0[i] - req.in1[i]);
}

std::unique_ptr<tc::InferenceServerGrpcClient> NewClient(bool cached = true) {
  std::unique_ptr<tc::InferenceServerGrpcClient> c;
  tca::Error e = tc::InferenceServerGrpcClient::Create(&c, g_url, false, false, tc::SslOptions(), tc::KeepAliveOptions(),
                                                       cached);
  if (!e.IsOk()) std::cout << "client creation failed: " << e.Message() << std::endl;
  return c;
}

}  // namespace

#define SERVER_TEST(name)                      \
  static void server_##name();                 \
  TEST(name) {                                 \
    if (g_url.empty()) {                       \
      std::printf("  (skipped: no -u url)\n"); \
      return;                                  \
    }                                          \
    server_##name();                           \
  }                                            \
  static void server_##name()

SERVER_TEST(GrpcHealthAndMetadata) {
  auto c = NewClient();
  bool b = false;
  EXPECT_OK(c->IsServerLive(&b));
  EXPECT_TRUE(b);
  b = false;
  EXPECT_OK(c->IsServerReady(&b));
  EXPECT_TRUE(b);
  b = false;
  EXPECT_OK(c->IsModelReady(&b, "simple"));
  EXPECT_TRUE(b);
  EXPECT_OK(c->IsModelReady(&b, "no_such_model"));
  EXPECT_TRUE(!b);
  std::string text;
  EXPECT_OK(c->ServerMetadata(&text, {{"my-header", "1"}}));
  tca::json::Value v;
  ASSERT_OK(tca::json::Parse(text.data(), text.size(), &v));
  EXPECT_EQ(v.GetString("name"), std::string("triton"));
  EXPECT_TRUE(v["extensions"].Size() > 0);
  EXPECT_OK(c->ModelMetadata(&text, "simple"));
  ASSERT_OK(tca::json::Parse(text.data(), text.size(), &v));
  EXPECT_EQ(v.GetString("name"), std::string("simple"));
  EXPECT_EQ(v["inputs"].Size(), size_t(2));
  EXPECT_EQ(v["inputs"][0]["shape"][1].AsInt(), int64_t(16));
  EXPECT_OK(c->ModelConfig(&text, "simple"));
  ASSERT_OK(tca::json::Parse(text.data(), text.size(), &v));
  EXPECT_EQ(v["config"].GetString("name"), std::string("simple"));
  EXPECT_EQ(v["config"]["input"][0].GetString("data_type"), std::string("TYPE_INT32"));
  EXPECT_ERROR_CONTAINS(c->ModelMetadata(&text, "wrong_model_name"), "Request for unknown model");
  EXPECT_OK(c->ModelRepositoryIndex(&text));
  ASSERT_OK(tca::json::Parse(text.data(), text.size(), &v));
  EXPECT_TRUE(v["models"].Size() >= 8);
  EXPECT_OK(c->ModelInferenceStatistics(&text, "simple"));
  ASSERT_OK(tca::json::Parse(text.data(), text.size(), &v));
  EXPECT_EQ(v["model_stats"][0].GetString("name"), std::string("simple"));
}

SERVER_TEST(GrpcInferSync) {
  auto c = NewClient();
  auto req = MakeRequest(3);
  tc::InferOptions opt("simple");
  opt.request_id_ = "sync-1";
  tc::InferResult* res = nullptr;
  ASSERT_OK(c->Infer(&res, opt, req->inputs));
  std::unique_ptr<tc::InferResult> holder(res);
  CheckSums(res, *req);
  std::string s;
  res->Id(&s);
  EXPECT_EQ(s, std::string("sync-1"));
  res->ModelName(&s);
  EXPECT_EQ(s, std::string("simple"));
  // requested outputs + classification
  tc::InferRequestedOutput* o = nullptr;
  tc::InferRequestedOutput::Create(&o, "OUTPUT0", 2);
  std::unique_ptr<tc::InferRequestedOutput> oh(o);
  ASSERT_OK(c->Infer(&res, opt, req->inputs, {o}));
  holder.reset(res);
  std::vector<std::string> cls;
  EXPECT_OK(res->StringData("OUTPUT0", &cls));
  EXPECT_EQ(cls.size(), size_t(2));
  const uint8_t* buf;
  size_t n;
  EXPECT_TRUE(!res->RawData("OUTPUT1", &buf, &n).IsOk());
  // versioned model: 2 and 3 swap the outputs
  tc::InferOptions v3("onnx_int32_int32_int32");
  v3.model_version_ = "3";
  ASSERT_OK(c->Infer(&res, v3, req->inputs));
  holder.reset(res);
  CheckSums(res, *req, true);
  // errors
  tc::InferOptions unknown("wrong_model_name");
  EXPECT_ERROR_CONTAINS(c->Infer(&res, unknown, req->inputs), "Request for unknown model");
  tc::InferStat stat;
  EXPECT_OK(c->ClientInferStat(&stat));
  EXPECT_EQ(stat.completed_request_count, size_t(3));
  EXPECT_TRUE(stat.cumulative_total_request_time_ns > 0);
}

SERVER_TEST(GrpcStringTensors) {
  auto c = NewClient();
  tc::InferInput *a = nullptr, *b = nullptr;
  tc::InferInput::Create(&a, "INPUT0", {1, 16}, "BYTES");
  tc::InferInput::Create(&b, "INPUT1", {1, 16}, "BYTES");
  std::unique_ptr<tc::InferInput> ah(a), bh(b);
  std::vector<std::string> v0, v1;
  for (int i = 0; i < 16; ++i) {
    v0.push_back(std::to_string(i * 3));
    v1.push_back("2");
  }
  a->AppendFromString(v0);
  b->AppendFromString(v1);
  tc::InferResult* res = nullptr;
  ASSERT_OK(c->Infer(&res, tc::InferOptions("simple_string"), {a, b}));
  std::unique_ptr<tc::InferResult> holder(res);
  std::vector<std::string> sum, diff;
  EXPECT_OK(res->StringData("OUTPUT0", &sum));
  EXPECT_OK(res->StringData("OUTPUT1", &diff));
  EXPECT_EQ(sum.size(), size_t(16));
  for (size_t i = 0; i < sum.size() && i < diff.size(); ++i) {
    EXPECT_EQ(sum[i], std::to_string(i * 3 + 2));
    EXPECT_EQ(diff[i], std::to_string(static_cast<int>(i) * 3 - 2));
  }
  // identity with an embedded NUL
  tc::InferInput* id = nullptr;
  tc::InferInput::Create(&id, "INPUT0", {1, 1}, "BYTES");
  std::unique_ptr<tc::InferInput> idh(id);
  const std::string with_nul("ab\0cd", 5);
  id->AppendFromString({with_nul});
  ASSERT_OK(c->Infer(&res, tc::InferOptions("simple_identity"), {id}));
  holder.reset(res);
  std::vector<std::string> echoed;
  EXPECT_OK(res->StringData("OUTPUT0", &echoed));
  EXPECT_EQ(echoed.size(), size_t(1));
  if (echoed.size() == 1) EXPECT_TRUE(echoed[0] == with_nul);
}

SERVER_TEST(GrpcLargeTensorsBothWays) {
  // 4 MiB each way: many DATA frames, window updates in both directions
  auto c = NewClient();
  const size_t count = 1 << 20;
  std::vector<int32_t> data(count);
  for (size_t i = 0; i < count; ++i) data[i] = static_cast<int32_t>(i * 2654435761u);
  tc::InferInput* in = nullptr;
  tc::InferInput::Create(&in, "INPUT0", {1, static_cast<int64_t>(count)}, "INT32");
  std::unique_ptr<tc::InferInput> ih(in);
  in->AppendRaw(reinterpret_cast<const uint8_t*>(data.data()), count * 4);
  tc::InferResult* res = nullptr;
  ASSERT_OK(c->Infer(&res, tc::InferOptions("custom_identity_int32"), {in}));
  std::unique_ptr<tc::InferResult> holder(res);
  const uint8_t* buf = nullptr;
  size_t n = 0;
  EXPECT_OK(res->RawData("OUTPUT0", &buf, &n));
  EXPECT_EQ(n, count * 4);
  if (n == count * 4) EXPECT_TRUE(std::memcmp(buf, data.data(), n) == 0);
}

SERVER_TEST(GrpcAsyncInferManyInFlight) {
  auto c = NewClient();
  constexpr int kN = 64;
  std::vector<std::unique_ptr<Request>> reqs;
  std::mutex mu;
  std::condition_variable cv;
  int done = 0;
  std::vector<std::unique_ptr<tc::InferResult>> results(kN);
  for (int i = 0; i < kN; ++i) reqs.push_back(MakeRequest(i));
  for (int i = 0; i < kN; ++i) {
    tc::InferOptions opt("simple");
    opt.request_id_ = std::to_string(i);
    ASSERT_OK(c->AsyncInfer(
        [&, i](tc::InferResult* r) {
          std::lock_guard<std::mutex> lk(mu);
          results[i].reset(r);
          ++done;
          cv.notify_all();
        },
        opt, reqs[i]->inputs));
  }
  {
    std::unique_lock<std::mutex> lk(mu);
    EXPECT_TRUE(tca::WaitFor(cv, lk, std::chrono::seconds(30), [&] { return done == kN; }));
  }
  for (int i = 0; i < kN; ++i) {
    ASSERT_OK(results[i] ? tca::Error::Success() : tca::Error("missing result"));
    CheckSums(results[i].get(), *reqs[i]);
    std::string id;
    results[i]->Id(&id);
    EXPECT_EQ(id, std::to_string(i));
  }
  tc::InferStat stat;
  c->ClientInferStat(&stat);
  EXPECT_EQ(stat.completed_request_count, size_t(kN));
}

SERVER_TEST(GrpcInferMultiVariants) {
  auto c = NewClient();
  std::vector<std::unique_ptr<Request>> reqs;
  std::vector<std::vector<tc::InferInput*>> inputs;
  for (int i = 0; i < 4; ++i) {
    reqs.push_back(MakeRequest(i * 5));
    inputs.push_back(reqs.back()->inputs);
  }
  std::vector<tc::InferOptions> one{tc::InferOptions("simple")};
  std::vector<tc::InferResult*> results;
  ASSERT_OK(c->InferMulti(&results, one, inputs));
  EXPECT_EQ(results.size(), size_t(4));
  for (size_t i = 0; i < results.size(); ++i) {
    CheckSums(results[i], *reqs[i]);
    delete results[i];
  }
  std::vector<tc::InferOptions> mismatch{tc::InferOptions("simple"), tc::InferOptions("simple")};
  EXPECT_ERROR_CONTAINS(c->InferMulti(&results, mismatch, inputs), "'options' must either contain 1 element");
  // async flavour
  std::mutex mu;
  std::condition_variable cv;
  bool done = false;
  std::vector<tc::InferResult*> async_results;
  ASSERT_OK(c->AsyncInferMulti(
      [&](std::vector<tc::InferResult*> r) {
        std::lock_guard<std::mutex> lk(mu);
        async_results = std::move(r);
        done = true;
        cv.notify_all();
      },
      one, inputs));
  std::unique_lock<std::mutex> lk(mu);
  EXPECT_TRUE(tca::WaitFor(cv, lk, std::chrono::seconds(30), [&] { return done; }));
  EXPECT_EQ(async_results.size(), size_t(4));
  for (size_t i = 0; i < async_results.size(); ++i) {
    CheckSums(async_results[i], *reqs[i]);
    delete async_results[i];
  }
}

SERVER_TEST(GrpcSystemSharedMemory) {
  auto c = NewClient();
  EXPECT_OK(c->UnregisterSystemSharedMemory());
  const std::string key = "/tca_grpc_cc_test_" + std::to_string(getpid());
  int fd = -1;
  void* base = nullptr;
  ASSERT_OK(tca::CreateSharedMemoryRegion(key, 256, &fd));
  ASSERT_OK(tca::MapSharedMemory(fd, 0, 256, &base));
  tca::CloseSharedMemory(fd);
  int32_t* p = static_cast<int32_t*>(base);
  for (int i = 0; i < 16; ++i) {
    p[i] = i;
    p[16 + i] = 2;
  }
  EXPECT_OK(c->RegisterSystemSharedMemory("cc_grpc_region", key, 256));
  std::string status;
  EXPECT_OK(c->SystemSharedMemoryStatus(&status));
  EXPECT_TRUE(status.find("cc_grpc_region") != std::string::npos);
  tc::InferInput *a = nullptr, *b = nullptr;
  tc::InferInput::Create(&a, "INPUT0", {1, 16}, "INT32");
  tc::InferInput::Create(&b, "INPUT1", {1, 16}, "INT32");
  std::unique_ptr<tc::InferInput> ah(a), bh(b);
  a->SetSharedMemory("cc_grpc_region", 64, 0);
  b->SetSharedMemory("cc_grpc_region", 64, 64);
  tc::InferRequestedOutput *o0 = nullptr, *o1 = nullptr;
  tc::InferRequestedOutput::Create(&o0, "OUTPUT0");
  tc::InferRequestedOutput::Create(&o1, "OUTPUT1");
  std::unique_ptr<tc::InferRequestedOutput> o0h(o0), o1h(o1);
  o0->SetSharedMemory("cc_grpc_region", 64, 128);
  o1->SetSharedMemory("cc_grpc_region", 64, 192);
  tc::InferResult* res = nullptr;
  ASSERT_OK(c->Infer(&res, tc::InferOptions("simple"), {a, b}, {o0, o1}));
  std::unique_ptr<tc::InferResult> holder(res);
  for (int i = 0; i < 16; ++i) {
    EXPECT_EQ(p[32 + i], i + 2);
    EXPECT_EQ(p[48 + i], i - 2);
  }
  EXPECT_OK(c->UnregisterSystemSharedMemory("cc_grpc_region"));
  tca::UnmapSharedMemory(base, 256);
  tca::UnlinkSharedMemoryRegion(key);
}

SERVER_TEST(GrpcModelControlAndSettings) {
  auto c = NewClient();
  EXPECT_OK(c->UnloadModel("simple_identity"));
  bool ready = true;
  EXPECT_OK(c->IsModelReady(&ready, "simple_identity"));
  EXPECT_TRUE(!ready);
  EXPECT_OK(c->LoadModel("simple_identity"));
  EXPECT_OK(c->IsModelReady(&ready, "simple_identity"));
  EXPECT_TRUE(ready);
  EXPECT_ERROR_CONTAINS(c->LoadModel("wrong_model_name"), "failed to load");
  EXPECT_OK(c->LoadModel("onnx_int32_int32_int32", tc::Headers(), R"({"backend":"onnxruntime","max_batch_size":4})"));
  std::string text;
  EXPECT_OK(c->ModelConfig(&text, "onnx_int32_int32_int32"));
  EXPECT_TRUE(text.find("\"max_batch_size\":4") != std::string::npos);
  EXPECT_OK(c->LoadModel("onnx_int32_int32_int32"));
  EXPECT_OK(c->UpdateTraceSettings(&text, "", {{"trace_rate", {"7"}}, {"trace_level", {"TIMESTAMPS"}}}));
  EXPECT_TRUE(text.find("\"trace_rate\":{\"value\":[\"7\"]}") != std::string::npos);
  EXPECT_OK(c->GetTraceSettings(&text));
  EXPECT_TRUE(text.find("TIMESTAMPS") != std::string::npos);
  EXPECT_OK(c->UpdateTraceSettings(&text, "", {{"trace_rate", {}}, {"trace_level", {}}}));
  EXPECT_OK(c->UpdateLogSettings(&text, R"({"log_verbose_level":2,"log_info":false})"));
  EXPECT_OK(c->GetLogSettings(&text));
  EXPECT_TRUE(text.find("\"log_verbose_level\":{\"uint32_param\":2}") != std::string::npos);
  EXPECT_OK(c->UpdateLogSettings(&text, R"({"log_verbose_level":0,"log_info":true})"));
}

SERVER_TEST(GrpcStreamSequencesInterleaved) {
  auto c = NewClient();
  std::mutex mu;
  std::condition_variable cv;
  std::map<std::string, std::vector<int32_t>> got;
  int received = 0;
  ASSERT_OK(c->StartStream([&](tc::InferResult* r) {
    std::unique_ptr<tc::InferResult> holder(r);
    std::lock_guard<std::mutex> lk(mu);
    ++received;
    if (r->RequestStatus().IsOk()) {
      std::string id;
      r->Id(&id);
      const uint8_t* buf;
      size_t n;
      if (r->RawData("OUTPUT", &buf, &n).IsOk() && n == 4) got[id.substr(0, id.find('_'))].push_back(tca::LoadAs<int32_t>(buf));
    }
    cv.notify_all();
  }));
  EXPECT_ERROR_CONTAINS(c->StartStream([](tc::InferResult* r) { delete r; }), "cannot start another stream");
  const std::vector<int32_t> values = {0, 11, 7, 5, 3, 2, 0, 1};
  for (size_t i = 0; i < values.size(); ++i) {
    for (uint64_t seq : {1001ull, 1002ull}) {
      int32_t v = seq == 1001 ? values[i] : -values[i];
      tc::InferInput* in = nullptr;
      tc::InferInput::Create(&in, "INPUT", {1, 1}, "INT32");
      std::unique_ptr<tc::InferInput> ih(in);
      in->AppendRaw(reinterpret_cast<const uint8_t*>(&v), 4);
      tc::InferOptions opt("simple_sequence");
      opt.sequence_id_ = seq;
      opt.sequence_start_ = i == 0;
      opt.sequence_end_ = i + 1 == values.size();
      opt.request_id_ = std::to_string(seq) + "_" + std::to_string(i);
      ASSERT_OK(c->AsyncStreamInfer(opt, {in}));  // the request is serialised before the call returns
    }
  }
  {
    std::unique_lock<std::mutex> lk(mu);
    EXPECT_TRUE(tca::WaitFor(cv, lk, std::chrono::seconds(30), [&] { return received == 16; }));
  }
  EXPECT_OK(c->StopStream());
  EXPECT_EQ(got["1001"].size(), size_t(8));
  EXPECT_EQ(got["1002"].size(), size_t(8));
  if (got["1001"].size() == 8 && got["1002"].size() == 8) {
    EXPECT_EQ(got["1001"][0], 1);   // start flag adds one
    EXPECT_EQ(got["1002"][0], 1);
    for (size_t i = 1; i < 8; ++i) {
      EXPECT_EQ(got["1001"][i], values[i]);
      EXPECT_EQ(got["1002"][i], -values[i]);
    }
  }
  tc::InferInput* in = nullptr;
  tc::InferInput::Create(&in, "INPUT", {1, 1}, "INT32");
  std::unique_ptr<tc::InferInput> ih(in);
  EXPECT_ERROR_CONTAINS(c->AsyncStreamInfer(tc::InferOptions("simple_sequence"), {in}), "stream not available");
}

SERVER_TEST(GrpcDecoupledRepeat) {
  auto c = NewClient();
  std::mutex mu;
  std::condition_variable cv;
  std::vector<int32_t> outs;
  int finals = 0;
  ASSERT_OK(c->StartStream(
      [&](tc::InferResult* r) {
        std::unique_ptr<tc::InferResult> holder(r);
        std::lock_guard<std::mutex> lk(mu);
        bool is_null = false, is_final = false;
        r->IsNullResponse(&is_null);
        r->IsFinalResponse(&is_final);
        const uint8_t* buf;
        size_t n;
        if (!is_null && r->RawData("OUT", &buf, &n).IsOk() && n == 4) outs.push_back(tca::LoadAs<int32_t>(buf));
        if (is_final) ++finals;
        cv.notify_all();
      },
      false));
  const int32_t in_values[5] = {10, 11, 12, 13, 14};
  const uint32_t delays[5] = {1, 1, 1, 1, 1};
  const uint32_t wait = 1;
  tc::InferInput *in = nullptr, *delay = nullptr, *w = nullptr;
  tc::InferInput::Create(&in, "IN", {5}, "INT32");
  tc::InferInput::Create(&delay, "DELAY", {5}, "UINT32");
  tc::InferInput::Create(&w, "WAIT", {1}, "UINT32");
  std::unique_ptr<tc::InferInput> h1(in), h2_(delay), h3(w);
  in->AppendRaw(reinterpret_cast<const uint8_t*>(in_values), 20);
  delay->AppendRaw(reinterpret_cast<const uint8_t*>(delays), 20);
  w->AppendRaw(reinterpret_cast<const uint8_t*>(&wait), 4);
  tc::InferOptions opt("repeat_int32");
  opt.triton_enable_empty_final_response_ = true;
  ASSERT_OK(c->AsyncStreamInfer(opt, {in, delay, w}));
  {
    std::unique_lock<std::mutex> lk(mu);
    EXPECT_TRUE(tca::WaitFor(cv, lk, std::chrono::seconds(30), [&] { return finals == 1; }));
  }
  EXPECT_OK(c->StopStream());
  EXPECT_EQ(outs.size(), size_t(5));
  for (size_t i = 0; i < outs.size(); ++i) EXPECT_EQ(outs[i], in_values[i]);
  // unary ModelInfer refuses decoupled models
  tc::InferResult* res = nullptr;
  EXPECT_ERROR_CONTAINS(c->Infer(&res, tc::InferOptions("repeat_int32"), {in, delay, w}), "decoupled");
}

SERVER_TEST(GrpcClientTimeout) {
  auto c = NewClient();
  const int32_t data[16] = {0};
  tc::InferInput* in = nullptr;
  tc::InferInput::Create(&in, "INPUT0", {1, 16}, "INT32");
  std::unique_ptr<tc::InferInput> ih(in);
  in->AppendRaw(reinterpret_cast<const uint8_t*>(data), 64);
  tc::InferOptions opt("custom_identity_int32");
  opt.client_timeout_ = 100000;  // 100 ms against a 1500 ms execution delay
  tc::RequestParameter delay;
  delay.type = tc::RequestParameter::Type::INT;
  delay.int_value = 1500;
  opt.request_parameters["execute_delay_ms"] = delay;
  tc::InferResult* res = nullptr;
  const tca::Error e = c->Infer(&res, opt, {in});
  EXPECT_TRUE(!e.IsOk());
  EXPECT_EQ(e.Code(), int(tc::kGrpcDeadlineExceeded));
  EXPECT_ERROR_CONTAINS(e, "Deadline Exceeded");
  // the connection stays usable
  tc::InferOptions ok("custom_identity_int32");
  ASSERT_OK(c->Infer(&res, ok, {in}));
  delete res;
}

SERVER_TEST(GrpcChannelCacheAndConnectFailure) {
  setenv("TRITON_CLIENT_GRPC_CHANNEL_MAX_SHARE_COUNT", "2", 1);
  const size_t before = tc::InferenceServerGrpcClient::GetNumCachedChannels();
  {
    auto a = NewClient(), b = NewClient(), c3 = NewClient();
    // two clients share the first channel, the third opens another
    EXPECT_TRUE(tc::InferenceServerGrpcClient::GetNumCachedChannels() >= std::max<size_t>(before, 1));
    bool live = false;
    EXPECT_OK(a->IsServerLive(&live));
    EXPECT_OK(b->IsServerLive(&live));
    EXPECT_OK(c3->IsServerLive(&live));
    auto own = NewClient(false);
    EXPECT_OK(own->IsServerLive(&live));
  }
  unsetenv("TRITON_CLIENT_GRPC_CHANNEL_MAX_SHARE_COUNT");
  std::unique_ptr<tc::InferenceServerGrpcClient> dead;
  ASSERT_OK(tc::InferenceServerGrpcClient::Create(&dead, "127.0.0.1:1", false, false, tc::SslOptions(),
                                                  tc::KeepAliveOptions(), false));
  bool live = true;
  const tca::Error e = dead->IsServerLive(&live);
  EXPECT_TRUE(!e.IsOk());
  EXPECT_EQ(e.Code(), int(tc::kGrpcUnavailable));
  // a TLS client against the cleartext port: the handshake fails, the call reports UNAVAILABLE
  std::unique_ptr<tc::InferenceServerGrpcClient> ssl;
  ASSERT_OK(tc::InferenceServerGrpcClient::Create(&ssl, g_url, false, true));
  const tca::Error tls_err = ssl->IsServerLive(&live);
  EXPECT_TRUE(!tls_err.IsOk());
  EXPECT_EQ(tls_err.Code(), int(tc::kGrpcUnavailable));
  // unreadable credentials are reported at Create
  tc::SslOptions bad_ssl;
  bad_ssl.root_certificates = "/nonexistent/ca.pem";
  std::unique_ptr<tc::InferenceServerGrpcClient> bad;
  EXPECT_ERROR_CONTAINS(tc::InferenceServerGrpcClient::Create(&bad, g_url, false, true, bad_ssl), "CA certificates");
}

SERVER_TEST(GrpcChannelOptionsAndControlDeadlines) {
  // channel settings by field and by gRPC argument name
  tc::ChannelOptions opt;
  ASSERT_OK(opt.SetInt("grpc.max_receive_message_length", 48));
  ASSERT_OK(opt.SetInt("grpc.keepalive_time_ms", 10000));
  ASSERT_OK(opt.SetString("grpc.primary_user_agent", "cc-test/1"));
  EXPECT_EQ(opt.max_receive_message_bytes, int64_t(48));
  EXPECT_EQ(opt.keepalive.keepalive_time_ms, 10000);
  EXPECT_ERROR_CONTAINS(opt.SetInt("grpc.no_such_argument", 1), "unsupported");
  std::unique_ptr<tc::InferenceServerGrpcClient> small;
  ASSERT_OK(tc::InferenceServerGrpcClient::Create(&small, g_url, opt));
  bool live = false;
  EXPECT_OK(small->IsServerLive(&live));  // a few bytes: fits
  EXPECT_TRUE(live);
  std::string text;
  const tca::Error big = small->ModelConfig(&text, "simple");  // hundreds of bytes: refused by the client
  EXPECT_TRUE(!big.IsOk());
  EXPECT_EQ(big.Code(), int(tc::kGrpcResourceExhausted));
  EXPECT_ERROR_CONTAINS(big, "Received message larger than max");
  tc::ChannelOptions tight;
  tight.max_send_message_bytes = 32;
  std::unique_ptr<tc::InferenceServerGrpcClient> mute;
  ASSERT_OK(tc::InferenceServerGrpcClient::Create(&mute, g_url, tight));
  auto req = MakeRequest(1);
  tc::InferResult* res = nullptr;
  EXPECT_ERROR_CONTAINS(mute->Infer(&res, tc::InferOptions("simple"), req->inputs), "Sent message larger than max");
  // HTTP/2 windows of a client's own choosing still move a large tensor both ways
  tc::ChannelOptions narrow;
  narrow.http2_stream_window_bytes = 65535;
  narrow.http2_connection_window_bytes = 65535;
  narrow.http2_max_frame_bytes = 16384;
  std::unique_ptr<tc::InferenceServerGrpcClient> slow;
  ASSERT_OK(tc::InferenceServerGrpcClient::Create(&slow, g_url, narrow, false, false, tc::SslOptions(), false));
  std::vector<int32_t> payload(1 << 18);
  for (size_t i = 0; i < payload.size(); ++i) payload[i] = static_cast<int32_t>(i * 7);
  tc::InferInput* in = nullptr;
  ASSERT_OK(tc::InferInput::Create(&in, "INPUT0", {1, static_cast<int64_t>(payload.size())}, "INT32"));
  std::unique_ptr<tc::InferInput> in_holder(in);
  in->AppendRaw(reinterpret_cast<const uint8_t*>(payload.data()), payload.size() * 4);
  ASSERT_OK(slow->Infer(&res, tc::InferOptions("custom_identity_int32"), {in}));
  std::unique_ptr<tc::InferResult> holder(res);
  const uint8_t* buf = nullptr;
  size_t n = 0;
  ASSERT_OK(res->RawData("OUTPUT0", &buf, &n));
  EXPECT_EQ(n, payload.size() * 4);
  EXPECT_TRUE(std::memcmp(buf, payload.data(), n) == 0);
  tc::ChannelOptions bad_frame;
  bad_frame.http2_max_frame_bytes = 100;
  std::unique_ptr<tc::InferenceServerGrpcClient> never;
  EXPECT_ERROR_CONTAINS(tc::InferenceServerGrpcClient::Create(&never, g_url, bad_frame), "http2_max_frame_bytes");
  // control calls take a deadline (microseconds); 1 us cannot be met, a second can
  auto c = NewClient();
  const tca::Error late = c->ServerMetadata(&text, tc::Headers(), 1);
  EXPECT_EQ(late.Code(), int(tc::kGrpcDeadlineExceeded));
  EXPECT_ERROR_CONTAINS(late, "Deadline Exceeded");
  EXPECT_OK(c->ServerMetadata(&text, tc::Headers(), 1000000));
  EXPECT_OK(c->IsModelReady(&live, "simple", "", tc::Headers(), 1000000));
}

int main(int argc, char** argv) {
  const char* filter = nullptr;
  for (int i = 1; i < argc; ++i) {
    if (std::strcmp(argv[i], "-u") == 0 && i + 1 < argc) g_url = argv[++i];
    else filter = argv[i];
  }
  return minitest::RunAllTests(filter);
}
