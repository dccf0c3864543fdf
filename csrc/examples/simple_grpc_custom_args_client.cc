// Channel behaviour knobs of the gRPC client: a private (non-cached) channel
// per client and the share-count limit of the process-wide channel cache.  The
// reference passes raw grpc::ChannelArguments (`src/c++/examples/
// simple_grpc_custom_args_client.cc`); with the in-tree transport the
// equivalent knobs are keep-alive options, channel caching and the
// TRITON_CLIENT_GRPC_CHANNEL_MAX_SHARE_COUNT environment variable.
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
  std::vector<int32_t> a(16), b(16, 1);
  for (int i = 0; i < 16; ++i) a[i] = i;
  auto in0 = NewInput("INPUT0", {1, 16}, "INT32");
  auto in1 = NewInput("INPUT1", {1, 16}, "INT32");
  in0->AppendRaw(reinterpret_cast<const uint8_t*>(a.data()), 64);
  in1->AppendRaw(reinterpret_cast<const uint8_t*>(b.data()), 64);
  for (tc::InferenceServerGrpcClient* c : {c1.get(), c2.get(), c3.get(), own.get()}) {
    tc::InferResult* result = nullptr;
    FAIL_IF_ERR(c->Infer(&result, tc::InferOptions("simple"), {in0.get(), in1.get()}), "unable to run model");
    auto holder = Own(result);
    ValidateSimple(result, a, b);
  }
  std::cout << "PASS : Custom Args" << std::endl;
  return 0;
}
