// Decoupled model over a gRPC stream: ONE request to `repeat_int32` yields one
// response per input element plus an empty final response.  (Reference:
// `src/c++/examples/simple_grpc_custom_repeat.cc:138-286`.)
#include <condition_variable>
#include <mutex>

#include "example_utils.h"
#include "tca/sync.h"

int main(int argc, char** argv) {
  CommonFlags flags = ParseCommonFlags(argc, argv);
  int repeat = 10, delay_ms = 10;
  for (size_t i = 0; i + 1 < flags.rest.size(); ++i) {
    if (flags.rest[i] == "-r") repeat = std::atoi(flags.rest[i + 1].c_str());
    if (flags.rest[i] == "-d") delay_ms = std::atoi(flags.rest[i + 1].c_str());
  }
  auto client = MakeGrpcClient(flags);
  std::mutex mu;
  std::condition_variable cv;
  std::vector<int32_t> outs;
  bool final_seen = false;
  std::string failure;
  // per-request statistics make no sense for a 1 -> N exchange: disable them
  FAIL_IF_ERR(client->StartStream(
                  [&](tc::InferResult* r) {
                    auto holder = Own(r);
                    std::lock_guard<std::mutex> lk(mu);
                    const tca::Error status = r->RequestStatus();
                    if (!status.IsOk()) failure = status.Message();
                    bool is_final = false, is_null = false;
                    r->IsFinalResponse(&is_final);
                    r->IsNullResponse(&is_null);
                    const uint8_t* buf = nullptr;
                    size_t n = 0;
                    if (!is_null && r->RawData("OUT", &buf, &n).IsOk() && n == 4)
                      outs.push_back(tca::LoadAs<int32_t>(buf));
                    if (is_final || !status.IsOk()) final_seen = true;
                    cv.notify_all();
                  },
                  false),
              "unable to start the stream");
  std::vector<int32_t> values(repeat);
  std::vector<uint32_t> delays(repeat, static_cast<uint32_t>(delay_ms));
  for (int i = 0; i < repeat; ++i) values[i] = 100 + i;
  const uint32_t wait = static_cast<uint32_t>(delay_ms);
  auto in = NewInput("IN", {repeat}, "INT32");
  auto delay = NewInput("DELAY", {repeat}, "UINT32");
  auto w = NewInput("WAIT", {1}, "UINT32");
  in->AppendRaw(reinterpret_cast<const uint8_t*>(values.data()), values.size() * 4);
  delay->AppendRaw(reinterpret_cast<const uint8_t*>(delays.data()), delays.size() * 4);
  w->AppendRaw(reinterpret_cast<const uint8_t*>(&wait), 4);
  tc::InferOptions options("repeat_int32");
  options.triton_enable_empty_final_response_ = true;
  FAIL_IF_ERR(client->AsyncStreamInfer(options, {in.get(), delay.get(), w.get()}), "stream request");
  {
    std::unique_lock<std::mutex> lk(mu);
    if (!tca::WaitFor(cv, lk, std::chrono::seconds(60), [&] { return final_seen; })) FAIL("timed out");
  }
  FAIL_IF_ERR(client->StopStream(), "unable to stop the stream");
  if (!failure.empty()) FAIL(failure);
  if (static_cast<int>(outs.size()) != repeat) FAIL("expected " << repeat << " responses, got " << outs.size());
  for (int i = 0; i < repeat; ++i) {
    std::cout << outs[i] << " : " << values[i] << std::endl;
    if (outs[i] != values[i]) FAIL("mismatch at " << i);
  }
  std::cout << "PASS : Decoupled API" << std::endl;
  return 0;
}
