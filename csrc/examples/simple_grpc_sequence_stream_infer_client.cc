// Two sequences interleaved on ONE bidirectional gRPC stream; responses arrive
// through the stream callback.  (Reference:
// `src/c++/examples/simple_grpc_sequence_stream_infer_client.cc`.)
#include <condition_variable>
#include <map>
#include <mutex>

#include "example_utils.h"
#include "tca/sync.h"

int main(int argc, char** argv) {
  CommonFlags flags = ParseCommonFlags(argc, argv);
  bool dyna = false;
  for (const auto& a : flags.rest) dyna |= a == "-d";
  auto client = MakeGrpcClient(flags);
  const std::string model = dyna ? "simple_dyna_sequence" : "simple_sequence";
  const std::vector<int32_t> values = {11, 7, 5, 3, 2, 0, 1};
  const uint64_t id0 = 1000, id1 = 1001;
  std::mutex mu;
  std::condition_variable cv;
  std::map<std::string, std::vector<int32_t>> got;
  size_t received = 0;
  std::string failure;
  FAIL_IF_ERR(client->StartStream(
                  [&](tc::InferResult* r) {
                    auto holder = Own(r);
                    std::lock_guard<std::mutex> lk(mu);
                    ++received;
                    const tca::Error status = r->RequestStatus();
                    if (!status.IsOk()) {
                      failure = status.Message();
                    } else {
                      std::string id;
                      r->Id(&id);
                      const uint8_t* buf = nullptr;
                      size_t n = 0;
                      if (r->RawData("OUTPUT", &buf, &n).IsOk() && n == 4)
                        got[id.substr(0, id.find('_'))].push_back(tca::LoadAs<int32_t>(buf));
                    }
                    cv.notify_all();
                  },
                  true, static_cast<uint32_t>(flags.client_timeout_us), flags.headers),
              "unable to start the stream");
  const size_t total = 2 * (values.size() + 1);
  for (size_t i = 0; i <= values.size(); ++i) {
    for (uint64_t seq : {id0, id1}) {
      const bool start = i == 0, end = i == values.size();
      int32_t v = seq == id0 ? (start ? 0 : values[i - 1]) : (start ? 100 : -values[i - 1]);
      auto in = NewInput("INPUT", {1, 1}, "INT32");
      in->AppendRaw(reinterpret_cast<const uint8_t*>(&v), sizeof(v));
      tc::InferOptions options(model);
      options.sequence_id_ = seq;
      options.sequence_start_ = start;
      options.sequence_end_ = end;
      options.request_id_ = std::to_string(seq) + "_" + std::to_string(i);
      FAIL_IF_ERR(client->AsyncStreamInfer(options, {in.get()}), "stream request");
    }
  }
  {
    std::unique_lock<std::mutex> lk(mu);
    if (!tca::WaitFor(cv, lk, std::chrono::seconds(30), [&] { return received == total; })) FAIL("timed out");
  }
  FAIL_IF_ERR(client->StopStream(), "unable to stop the stream");
  if (!failure.empty()) FAIL(failure);
  const auto& r0 = got[std::to_string(id0)];
  const auto& r1 = got[std::to_string(id1)];
  if (r0.size() != values.size() + 1 || r1.size() != values.size() + 1) FAIL("missing responses");
  for (size_t i = 0; i <= values.size(); ++i) {
    int32_t e0 = i == 0 ? 1 : values[i - 1];
    int32_t e1 = i == 0 ? 101 : -values[i - 1];
    if (dyna && i == values.size()) {
      e0 += static_cast<int32_t>(id0);
      e1 += static_cast<int32_t>(id1);
    }
    std::cout << "[" << i << "] " << r0[i] << " : " << r1[i] << std::endl;
    if (r0[i] != e0 || r1[i] != e1) FAIL("[ expected ] " << e0 << " : " << e1);
  }
  std::cout << "PASS : Sequence" << std::endl;
  return 0;
}
