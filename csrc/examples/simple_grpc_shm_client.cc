// System shared memory over gRPC: both inputs in one region, both outputs in
// another, addressed by offsets.  (Reference:
// `src/c++/examples/simple_grpc_shm_client.cc:165-239`.)
#include <unistd.h>

#include "example_utils.h"
#include "tca/shm_utils.h"

int main(int argc, char** argv) {
  CommonFlags flags = ParseCommonFlags(argc, argv);
  auto client = MakeGrpcClient(flags);
  FAIL_IF_ERR(client->UnregisterSystemSharedMemory(), "unregister all");
  const std::string in_key = "/grpc_input_simple_" + std::to_string(getpid());
  const std::string out_key = "/grpc_output_simple_" + std::to_string(getpid());
  int fd = -1;
  void *in_base = nullptr, *out_base = nullptr;
  FAIL_IF_ERR(tca::CreateSharedMemoryRegion(in_key, 128, &fd), "create input region");
  FAIL_IF_ERR(tca::MapSharedMemory(fd, 0, 128, &in_base), "map input region");
  FAIL_IF_ERR(tca::CloseSharedMemory(fd), "close");
  FAIL_IF_ERR(tca::CreateSharedMemoryRegion(out_key, 128, &fd), "create output region");
  FAIL_IF_ERR(tca::MapSharedMemory(fd, 0, 128, &out_base), "map output region");
  FAIL_IF_ERR(tca::CloseSharedMemory(fd), "close");
  int32_t* in = static_cast<int32_t*>(in_base);
  std::vector<int32_t> a(16), b(16, 1);
  for (int i = 0; i < 16; ++i) {
    a[i] = i;
    in[i] = a[i];
    in[16 + i] = b[i];
  }
  FAIL_IF_ERR(client->RegisterSystemSharedMemory("input_data", in_key, 128), "register input region");
  FAIL_IF_ERR(client->RegisterSystemSharedMemory("output_data", out_key, 128), "register output region");
  auto in0 = NewInput("INPUT0", {1, 16}, "INT32");
  auto in1 = NewInput("INPUT1", {1, 16}, "INT32");
  FAIL_IF_ERR(in0->SetSharedMemory("input_data", 64, 0), "INPUT0 shm");
  FAIL_IF_ERR(in1->SetSharedMemory("input_data", 64, 64), "INPUT1 shm");
  auto out0 = NewOutput("OUTPUT0");
  auto out1 = NewOutput("OUTPUT1");
  FAIL_IF_ERR(out0->SetSharedMemory("output_data", 64, 0), "OUTPUT0 shm");
  FAIL_IF_ERR(out1->SetSharedMemory("output_data", 64, 64), "OUTPUT1 shm");
  tc::InferResult* result = nullptr;
  FAIL_IF_ERR(client->Infer(&result, tc::InferOptions("simple"), {in0.get(), in1.get()}, {out0.get(), out1.get()}),
              "unable to run model");
  delete result;
  const int32_t* out = static_cast<const int32_t*>(out_base);
  for (int i = 0; i < 16; ++i) {
    std::cout << a[i] << " + " << b[i] << " = " << out[i] << std::endl;
    if (out[i] != a[i] + b[i] || out[16 + i] != a[i] - b[i]) FAIL("incorrect result at " << i);
  }
  std::string status;
  FAIL_IF_ERR(client->SystemSharedMemoryStatus(&status), "status");
  std::cout << status << std::endl;
  FAIL_IF_ERR(client->UnregisterSystemSharedMemory("input_data"), "unregister");
  FAIL_IF_ERR(client->UnregisterSystemSharedMemory("output_data"), "unregister");
  tca::UnmapSharedMemory(in_base, 128);
  tca::UnmapSharedMemory(out_base, 128);
  tca::UnlinkSharedMemoryRegion(in_key);
  tca::UnlinkSharedMemoryRegion(out_key);
  std::cout << "PASS : System Shared Memory" << std::endl;
  return 0;
}
