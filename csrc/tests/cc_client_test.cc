// Server-backed tests of the C++ HTTP client (run against the in-repo v2
// server: `cc_client_test -u host:port`).  Mirrors the intent of the
// reference's typed gtest suite (`src/c++/tests/cc_client_test.cc:298-1639`):
// InferMulti / AsyncInferMulti argument combinations, load with config / file
// override against `onnx_int32_int32_int32`, trace settings.
#include <unistd.h>

#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>

#include "minitest.h"
#include "tca/http_client.h"
#include "tca/shm_utils.h"

namespace tc = tca::client;

static std::string g_url = "localhost:8000";

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

void CheckResult(tc::InferResult* res, const Request& req, bool want0, bool want1, bool swapped = false) {
  EXPECT_OK(res->RequestStatus());
  const uint8_t* buf = nullptr;
  size_t n = 0;
  const char* add_name = swapped ? "OUTPUT1" : "OUTPUT0";
  const char* sub_name = swapped ? "OUTPUT0" : "OUTPUT1";
  if ((want0 && !swapped) || (want1 && swapped)) {
    EXPECT_OK(res->RawData(add_name, &buf, &n));
    EXPECT_EQ(n, size_t(64));
    for (int i = 0; n == 64 && i < 16; ++i) EXPECT_EQ(tca::LoadAs<int32_t>(buf, i), req.in0[i] + req.in1[i]);
  }
  if ((want1 && !swapped) || (want0 && swapped)) {
    EXPECT_OK(res->RawData(sub_name, &buf, &n));
    EXPECT_EQ(n, size_t(64));
    for (int i = 0; n == 64 && i < 16; ++i) EXPECT_EQ(tca::LoadAs<int32_t>(buf, i), req.in0[i] - req.in1[i]);
  }
  if (!want0) EXPECT_TRUE(!res->RawData("OUTPUT0", &buf, &n).IsOk());
  if (!want1) EXPECT_TRUE(!res->RawData("OUTPUT1", &buf, &n).IsOk());
}

std::unique_ptr<tc::InferenceServerHttpClient> NewClient() {
  std::unique_ptr<tc::InferenceServerHttpClient> c;
  tca::Error e = tc::InferenceServerHttpClient::Create(&c, g_url);
  if (!e.IsOk()) std::cout << "client creation failed: " << e.Message() << std::endl;
  return c;
}

struct Outputs {
  std::unique_ptr<tc::InferRequestedOutput> o0, o1;
  Outputs() {
    tc::InferRequestedOutput* p = nullptr;
    tc::InferRequestedOutput::Create(&p, "OUTPUT0");
    o0.reset(p);
    tc::InferRequestedOutput::Create(&p, "OUTPUT1");
    o1.reset(p);
  }
};

// Runs fn either through InferMulti or AsyncInferMulti and hands back results.
tca::Error RunMulti(bool async, tc::InferenceServerHttpClient* c, const std::vector<tc::InferOptions>& options,
                    const std::vector<std::vector<tc::InferInput*>>& inputs,
                    const std::vector<std::vector<const tc::InferRequestedOutput*>>& outputs,
                    std::vector<tc::InferResult*>* results) {
  if (!async) return c->InferMulti(results, options, inputs, outputs);
  std::mutex mu;
  std::condition_variable cv;
  bool done = false;
  tca::Error e = c->AsyncInferMulti(
      [&](std::vector<tc::InferResult*> r) {
        std::lock_guard<std::mutex> lk(mu);
        *results = std::move(r);
        done = true;
        cv.notify_one();
      },
      options, inputs, outputs);
  if (!e.IsOk()) return e;
  std::unique_lock<std::mutex> lk(mu);
  cv.wait(lk, [&] { return done; });
  return tca::Error::Success();
}

void MultiCases(bool async) {
  auto c = NewClient();
  if (!c) { ++minitest::Failures(); return; }
  std::vector<std::unique_ptr<Request>> reqs;
  std::vector<std::vector<tc::InferInput*>> inputs;
  for (int i = 0; i < 3; ++i) {
    reqs.push_back(MakeRequest(i * 10));
    inputs.push_back(reqs.back()->inputs);
  }
  Outputs outs;
  auto free_results = [](std::vector<tc::InferResult*>* r) {
    for (auto* p : *r) delete p;
    r->clear();
  };
  std::vector<tc::InferResult*> results;
  // DifferentOutputs: per-request output lists
  {
    std::vector<tc::InferOptions> options(3, tc::InferOptions("simple"));
    std::vector<std::vector<const tc::InferRequestedOutput*>> outputs = {{outs.o0.get()}, {outs.o1.get()}, {outs.o0.get(), outs.o1.get()}};
    ASSERT_OK(RunMulti(async, c.get(), options, inputs, outputs, &results));
    EXPECT_EQ(results.size(), size_t(3));
    if (results.size() == 3) {
      CheckResult(results[0], *reqs[0], true, false);
      CheckResult(results[1], *reqs[1], false, true);
      CheckResult(results[2], *reqs[2], true, true);
    }
    free_results(&results);
  }
  // DifferentOptions: versions of the versioned model (2 and 3 swap the outputs)
  {
    std::vector<tc::InferOptions> options;
    for (const char* v : {"1", "2", "3"}) {
      options.emplace_back("onnx_int32_int32_int32");
      options.back().model_version_ = v;
      options.back().request_id_ = std::string("req_") + v;
    }
    ASSERT_OK(RunMulti(async, c.get(), options, inputs, {}, &results));
    EXPECT_EQ(results.size(), size_t(3));
    if (results.size() == 3) {
      CheckResult(results[0], *reqs[0], true, true, false);
      CheckResult(results[1], *reqs[1], true, true, true);
      CheckResult(results[2], *reqs[2], true, true, true);
      std::string id;
      EXPECT_OK(results[1]->Id(&id));
      EXPECT_EQ(id, std::string("req_2"));
      std::string ver;
      EXPECT_OK(results[2]->ModelVersion(&ver));
      EXPECT_EQ(ver, std::string("3"));
    }
    free_results(&results);
  }
  // OneOption + OneOutput applied to every request
  {
    std::vector<tc::InferOptions> options(1, tc::InferOptions("simple"));
    std::vector<std::vector<const tc::InferRequestedOutput*>> outputs = {{outs.o1.get()}};
    ASSERT_OK(RunMulti(async, c.get(), options, inputs, outputs, &results));
    EXPECT_EQ(results.size(), size_t(3));
    for (size_t i = 0; i < results.size(); ++i) CheckResult(results[i], *reqs[i], false, true);
    free_results(&results);
  }
  // NoOutput: everything comes back
  {
    std::vector<tc::InferOptions> options(1, tc::InferOptions("simple"));
    ASSERT_OK(RunMulti(async, c.get(), options, inputs, {}, &results));
    for (size_t i = 0; i < results.size(); ++i) CheckResult(results[i], *reqs[i], true, true);
    free_results(&results);
  }
  // MismatchOptions / MismatchOutputs
  {
    std::vector<tc::InferOptions> options(2, tc::InferOptions("simple"));
    EXPECT_ERROR_CONTAINS(RunMulti(async, c.get(), options, inputs, {}, &results), "'options' has 2 elements");
    std::vector<tc::InferOptions> one(1, tc::InferOptions("simple"));
    std::vector<std::vector<const tc::InferRequestedOutput*>> outputs = {{outs.o0.get()}, {outs.o1.get()}};
    EXPECT_ERROR_CONTAINS(RunMulti(async, c.get(), one, inputs, outputs, &results), "'outputs' has 2 elements");
  }
}

}  // namespace

TEST(HealthAndMetadata) {
  auto c = NewClient();
  ASSERT_OK(c ? tca::Error::Success() : tca::Error("no client"));
  bool live = false, ready = false, mready = false;
  ASSERT_OK(c->IsServerLive(&live));
  ASSERT_OK(c->IsServerReady(&ready));
  ASSERT_OK(c->IsModelReady(&mready, "simple"));
  EXPECT_TRUE(live && ready && mready);
  ASSERT_OK(c->IsModelReady(&mready, "no_such_model"));
  EXPECT_TRUE(!mready);
  std::string md;
  ASSERT_OK(c->ServerMetadata(&md));
  tca::json::Value doc;
  ASSERT_OK(tca::json::Parse(md, &doc));
  EXPECT_EQ(doc.GetString("name"), std::string("triton"));
  ASSERT_OK(c->ModelMetadata(&md, "simple"));
  ASSERT_OK(tca::json::Parse(md, &doc));
  EXPECT_EQ(doc["inputs"][0].GetString("name"), std::string("INPUT0"));
  ASSERT_OK(c->ModelConfig(&md, "simple"));
  ASSERT_OK(tca::json::Parse(md, &doc));
  EXPECT_EQ(doc.GetInt("max_batch_size"), 8);
  EXPECT_ERROR_CONTAINS(c->ModelMetadata(&md, "wrong_model_name"), "Request for unknown model");
  ASSERT_OK(c->ModelRepositoryIndex(&md));
  ASSERT_OK(tca::json::Parse(md, &doc));
  EXPECT_TRUE(doc.Size() >= 8);
  ASSERT_OK(c->ModelInferenceStatistics(&md, "simple"));
  ASSERT_OK(tca::json::Parse(md, &doc));
  EXPECT_EQ(doc["model_stats"][0].GetString("name"), std::string("simple"));
}

TEST(InferSyncVariants) {
  auto c = NewClient();
  ASSERT_OK(c ? tca::Error::Success() : tca::Error("no client"));
  auto req = MakeRequest(3);
  Outputs outs;
  tc::InferOptions opt("simple");
  opt.request_id_ = "abc";
  tc::InferResult* res = nullptr;
  ASSERT_OK(c->Infer(&res, opt, req->inputs, {outs.o0.get(), outs.o1.get()}, {{"my-header", "1"}}, {{"q", "a b"}}));
  CheckResult(res, *req, true, true);
  std::string id, name;
  EXPECT_OK(res->Id(&id));
  EXPECT_OK(res->ModelName(&name));
  EXPECT_EQ(id, std::string("abc"));
  EXPECT_EQ(name, std::string("simple"));
  delete res;
  // JSON tensors both ways
  req->i0->SetBinaryData(false);
  outs.o0->SetBinaryData(false);
  ASSERT_OK(c->Infer(&res, opt, req->inputs, {outs.o0.get(), outs.o1.get()}));
  CheckResult(res, *req, true, true);
  delete res;
  req->i0->SetBinaryData(true);
  outs.o0->SetBinaryData(true);
  // compression both directions
  using CT = tc::InferenceServerHttpClient::CompressionType;
  for (CT alg : {CT::GZIP, CT::DEFLATE}) {
    ASSERT_OK(c->Infer(&res, opt, req->inputs, {}, {}, {}, alg, alg));
    CheckResult(res, *req, true, true);
    delete res;
  }
  // errors
  tc::InferOptions bad("wrong_model_name");
  EXPECT_ERROR_CONTAINS(c->Infer(&res, bad, req->inputs), "Request for unknown model");
  tc::InferStat stat;
  ASSERT_OK(c->ClientInferStat(&stat));
  EXPECT_TRUE(stat.completed_request_count >= 4 && stat.cumulative_total_request_time_ns > 0);
  EXPECT_TRUE(stat.cumulative_send_time_ns > 0 && stat.cumulative_receive_time_ns > 0);
}

TEST(StringTensors) {
  auto c = NewClient();
  ASSERT_OK(c ? tca::Error::Success() : tca::Error("no client"));
  std::vector<std::string> a, b;
  for (int i = 0; i < 16; ++i) {
    a.push_back(std::to_string(i));
    b.push_back("1");
  }
  tc::InferInput *i0 = nullptr, *i1 = nullptr;
  ASSERT_OK(tc::InferInput::Create(&i0, "INPUT0", {1, 16}, "BYTES"));
  ASSERT_OK(tc::InferInput::Create(&i1, "INPUT1", {1, 16}, "BYTES"));
  std::unique_ptr<tc::InferInput> g0(i0), g1(i1);
  ASSERT_OK(i0->AppendFromString(a));
  ASSERT_OK(i1->AppendFromString(b));
  tc::InferResult* res = nullptr;
  ASSERT_OK(c->Infer(&res, tc::InferOptions("simple_string"), {i0, i1}));
  std::unique_ptr<tc::InferResult> guard(res);
  std::vector<std::string> out0, out1;
  ASSERT_OK(res->StringData("OUTPUT0", &out0));
  ASSERT_OK(res->StringData("OUTPUT1", &out1));
  EXPECT_EQ(out0.size(), size_t(16));
  for (int i = 0; i < 16 && out0.size() == 16; ++i) {
    EXPECT_EQ(out0[i], std::to_string(i + 1));
    EXPECT_EQ(out1[i], std::to_string(i - 1));
  }
}

TEST(AsyncInferManyInFlight) {
  auto c = NewClient();
  ASSERT_OK(c ? tca::Error::Success() : tca::Error("no client"));
  constexpr int kN = 64;
  std::vector<std::unique_ptr<Request>> reqs;
  for (int i = 0; i < kN; ++i) reqs.push_back(MakeRequest(i));
  std::mutex mu;
  std::condition_variable cv;
  int done = 0;
  std::vector<tc::InferResult*> results(kN, nullptr);
  for (int i = 0; i < kN; ++i) {
    tc::InferOptions opt("simple");
    opt.request_id_ = std::to_string(i);
    ASSERT_OK(c->AsyncInfer(
        [&, i](tc::InferResult* r) {
          std::lock_guard<std::mutex> lk(mu);
          results[i] = r;
          if (++done == kN) cv.notify_one();
        },
        opt, reqs[i]->inputs));
  }
  {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return done == kN; });
  }
  for (int i = 0; i < kN; ++i) {
    CheckResult(results[i], *reqs[i], true, true);
    delete results[i];
  }
  EXPECT_EQ(c->AsyncInFlight(), size_t(0));
  EXPECT_ERROR_CONTAINS(c->AsyncInfer(nullptr, tc::InferOptions("simple"), reqs[0]->inputs), "Callback function must be provided");
}

TEST(InferMulti) { MultiCases(false); }
TEST(AsyncInferMulti) { MultiCases(true); }

TEST(SystemSharedMemory) {
  auto c = NewClient();
  ASSERT_OK(c ? tca::Error::Success() : tca::Error("no client"));
  c->UnregisterSystemSharedMemory();
  tca::ShmRegion in, out;
  const std::string kin = "/tca_cc_in_" + std::to_string(getpid()), kout = "/tca_cc_out_" + std::to_string(getpid());
  ASSERT_OK(in.Create(kin, 128));
  ASSERT_OK(out.Create(kout, 128));
  auto* a = static_cast<int32_t*>(in.data());
  for (int i = 0; i < 16; ++i) { a[i] = i; a[16 + i] = 1; }
  ASSERT_OK(c->RegisterSystemSharedMemory("input_data", kin, 128));
  ASSERT_OK(c->RegisterSystemSharedMemory("output_data", kout, 128));
  std::string status;
  ASSERT_OK(c->SystemSharedMemoryStatus(&status));
  EXPECT_TRUE(status.find("input_data") != std::string::npos && status.find("output_data") != std::string::npos);
  tc::InferInput *i0 = nullptr, *i1 = nullptr;
  ASSERT_OK(tc::InferInput::Create(&i0, "INPUT0", {1, 16}, "INT32"));
  ASSERT_OK(tc::InferInput::Create(&i1, "INPUT1", {1, 16}, "INT32"));
  std::unique_ptr<tc::InferInput> g0(i0), g1(i1);
  i0->SetSharedMemory("input_data", 64, 0);
  i1->SetSharedMemory("input_data", 64, 64);
  Outputs outs;
  outs.o0->SetSharedMemory("output_data", 64, 0);
  outs.o1->SetSharedMemory("output_data", 64, 64);
  tc::InferResult* res = nullptr;
  ASSERT_OK(c->Infer(&res, tc::InferOptions("simple"), {i0, i1}, {outs.o0.get(), outs.o1.get()}));
  delete res;
  auto* o = static_cast<int32_t*>(out.data());
  for (int i = 0; i < 16; ++i) {
    EXPECT_EQ(o[i], i + 1);
    EXPECT_EQ(o[16 + i], i - 1);
  }
  ASSERT_OK(c->UnregisterSystemSharedMemory("input_data"));
  ASSERT_OK(c->UnregisterSystemSharedMemory());
  EXPECT_ERROR_CONTAINS(c->Infer(&res, tc::InferOptions("simple"), {i0, i1}), "Unable to find shared memory region");
}

TEST(LoadWithConfigAndFileOverride) {
  auto c = NewClient();
  ASSERT_OK(c ? tca::Error::Success() : tca::Error("no client"));
  const std::string model = "onnx_int32_int32_int32";
  std::string cfg;
  ASSERT_OK(c->LoadModel(model, {}, {}, R"({"max_batch_size": 16, "backend": "onnxruntime"})"));
  ASSERT_OK(c->ModelConfig(&cfg, model));
  tca::json::Value doc;
  ASSERT_OK(tca::json::Parse(cfg, &doc));
  EXPECT_EQ(doc.GetInt("max_batch_size"), 16);
  // file override requires a config override
  std::map<std::string, std::vector<char>> files = {{"file:1/model.onnx", {'o', 'n', 'n', 'x'}}};
  EXPECT_ERROR_CONTAINS(c->LoadModel(model, {}, {}, "", files), "failed to load");
  ASSERT_OK(c->LoadModel(model, {}, {}, R"({"max_batch_size": 8})", files));
  ASSERT_OK(c->ModelConfig(&cfg, model));
  ASSERT_OK(tca::json::Parse(cfg, &doc));
  EXPECT_EQ(doc.GetInt("max_batch_size"), 8);
  EXPECT_ERROR_CONTAINS(c->LoadModel("wrong_model_name"), "failed to load");
  ASSERT_OK(c->UnloadModel(model));
  bool ready = true;
  ASSERT_OK(c->IsModelReady(&ready, model));
  EXPECT_TRUE(!ready);
  ASSERT_OK(c->LoadModel(model));
  ASSERT_OK(c->IsModelReady(&ready, model));
  EXPECT_TRUE(ready);
}

TEST(TraceAndLogSettings) {
  auto c = NewClient();
  ASSERT_OK(c ? tca::Error::Success() : tca::Error("no client"));
  std::string resp;
  ASSERT_OK(c->UpdateTraceSettings(&resp, "", {{"trace_level", {"TIMESTAMPS"}}, {"trace_rate", {"1"}}}));
  tca::json::Value doc;
  ASSERT_OK(tca::json::Parse(resp, &doc));
  EXPECT_EQ(doc["trace_level"][0].AsString(), std::string("TIMESTAMPS"));
  EXPECT_EQ(doc["trace_rate"][0].AsString(), std::string("1"));
  ASSERT_OK(c->UpdateTraceSettings(&resp, "simple", {{"trace_rate", {"7"}}}));
  ASSERT_OK(c->GetTraceSettings(&resp, "simple"));
  ASSERT_OK(tca::json::Parse(resp, &doc));
  EXPECT_EQ(doc["trace_rate"][0].AsString(), std::string("7"));
  ASSERT_OK(c->UpdateTraceSettings(&resp, "simple", {{"trace_rate", {}}}));  // clear -> global value
  ASSERT_OK(tca::json::Parse(resp, &doc));
  EXPECT_EQ(doc["trace_rate"][0].AsString(), std::string("1"));
  ASSERT_OK(c->UpdateTraceSettings(&resp, "", {{"trace_level", {"OFF"}}, {"trace_rate", {"1000"}}}));
  ASSERT_OK(c->UpdateLogSettings(&resp, R"({"log_verbose_level": 1})"));
  ASSERT_OK(c->GetLogSettings(&resp));
  ASSERT_OK(tca::json::Parse(resp, &doc));
  EXPECT_EQ(doc.GetInt("log_verbose_level"), 1);
  ASSERT_OK(c->UpdateLogSettings(&resp, R"({"log_verbose_level": 0})"));
}

int main(int argc, char** argv) {
  const char* filter = nullptr;
  for (int i = 1; i < argc; ++i) {
    if (std::strcmp(argv[i], "-u") == 0 && i + 1 < argc) g_url = argv[++i];
    else filter = argv[i];
  }
  return minitest::RunAllTests(filter);
}
