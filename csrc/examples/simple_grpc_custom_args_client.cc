// Channel settings of the gRPC client.  The reference passes raw grpc::ChannelArguments
// (`src/c++/examples/simple_grpc_custom_args_client.cc`); with the in-tree transport the same
// settings are a tca::client::ChannelOptions - fields, or SetInt / SetString with the gRPC
// argument names - next to channel caching and the TRITON_CLIENT_GRPC_CHANNEL_MAX_SHARE_COUNT
// environment variable.
#include "example_utils.h"

int main(int argc, char** argv) {
  CommonFlags flags = ParseCommonFlags(argc, argv);
  setenv("TRITON_CLIENT_GRPC_CHANNEL_MAX_SHARE_COUNT", "2", 1);
  const size_t before = tc::InferenceServerGrpcClient::GetNumCachedChannels();
  auto c1 = MakeGrpcClient(flags);
  auto c2 = MakeGrpcClient(flags);
  auto c3 = MakeGrpcClient(flags);  // third client of the URL: over the share limit, new channel
  auto own = MakeGrpcClient(flags, tc::KeepAliveOptions(), /*cached=*/false);
  if (tc::InferenceServerGrpcClient::GetNumCachedChannels() < before + 1) FAIL("channel cache is empty");
  // what the reference's example sets: message size limits, keep-alive, a user agent
  tc::ChannelOptions args;
  FAIL_IF_ERR(args.SetInt("grpc.max_send_message_length", 1 << 20), "max send message length");
  FAIL_IF_ERR(args.SetInt("grpc.max_receive_message_length", 1 << 20), "max receive message length");
  FAIL_IF_ERR(args.SetInt("grpc.keepalive_time_ms", 2 << 20), "keep-alive time");
  FAIL_IF_ERR(args.SetInt("grpc.keepalive_timeout_ms", 20000), "keep-alive timeout");
  FAIL_IF_ERR(args.SetInt("grpc.keepalive_permit_without_calls", 0), "keep-alive without calls");
  FAIL_IF_ERR(args.SetInt("grpc.http2.max_pings_without_data", 2), "pings without data");
  FAIL_IF_ERR(args.SetString("grpc.primary_user_agent", "simple_grpc_custom_args_client"), "user agent");
  std::unique_ptr<tc::InferenceServerGrpcClient> custom;
  FAIL_IF_ERR(tc::InferenceServerGrpcClient::Create(&custom, flags.url, args, flags.verbose), "unable to create client with channel options");
  std::vector<int32_t> a(16), b(16, 1);
  for (int i = 0; i < 16; ++i) a[i] = i;
  auto in0 = NewInput("INPUT0", {1, 16}, "INT32");
  auto in1 = NewInput("INPUT1", {1, 16}, "INT32");
  in0->AppendRaw(reinterpret_cast<const uint8_t*>(a.data()), 64);
  in1->AppendRaw(reinterpret_cast<const uint8_t*>(b.data()), 64);
  for (tc::InferenceServerGrpcClient* c : {c1.get(), c2.get(), c3.get(), own.get(), custom.get()}) {
    tc::InferResult* result = nullptr;
    FAIL_IF_ERR(c->Infer(&result, tc::InferOptions("simple"), {in0.get(), in1.get()}), "unable to run model");
    auto holder = Own(result);
    ValidateSimple(result, a, b);
  }
  std::cout << "PASS : Custom Args" << std::endl;
  return 0;
}
