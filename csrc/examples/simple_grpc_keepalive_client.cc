// gRPC client with explicit keep-alive options (HTTP/2 PING probes).
// (Reference: `src/c++/examples/simple_grpc_keepalive_client.cc`.)
#include <thread>

#include "example_utils.h"

int main(int argc, char** argv) {
  CommonFlags flags = ParseCommonFlags(argc, argv);
  tc::KeepAliveOptions keepalive;
  keepalive.keepalive_time_ms = 200;  // probe after 200 ms of silence
  keepalive.keepalive_timeout_ms = 2000;
  keepalive.keepalive_permit_without_calls = true;
  for (size_t i = 0; i + 1 < flags.rest.size(); ++i) {
    if (flags.rest[i] == "--grpc-keepalive-time") keepalive.keepalive_time_ms = std::atoi(flags.rest[i + 1].c_str());
    if (flags.rest[i] == "--grpc-keepalive-timeout") keepalive.keepalive_timeout_ms = std::atoi(flags.rest[i + 1].c_str());
  }
  auto client = MakeGrpcClient(flags, keepalive, /*cached=*/false);
  std::vector<int32_t> a(16), b(16, 1);
  for (int i = 0; i < 16; ++i) a[i] = i;
  auto in0 = NewInput("INPUT0", {1, 16}, "INT32");
  auto in1 = NewInput("INPUT1", {1, 16}, "INT32");
  in0->AppendRaw(reinterpret_cast<const uint8_t*>(a.data()), 64);
  in1->AppendRaw(reinterpret_cast<const uint8_t*>(b.data()), 64);
  tc::InferResult* result = nullptr;
  FAIL_IF_ERR(client->Infer(&result, tc::InferOptions("simple"), {in0.get(), in1.get()}), "first request");
  delete result;
  // idle long enough for several probes: the connection must survive them
  std::this_thread::sleep_for(std::chrono::milliseconds(900));
  FAIL_IF_ERR(client->Infer(&result, tc::InferOptions("simple"), {in0.get(), in1.get()}), "request after idling");
  auto holder = Own(result);
  ValidateSimple(result, a, b);
  std::cout << "PASS : KeepAlive" << std::endl;
  return 0;
}
