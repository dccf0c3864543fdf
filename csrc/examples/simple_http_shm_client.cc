// System shared memory over HTTP: both inputs of `simple` live in one POSIX shm
// region (at offsets 0 and 64), both outputs are written by the server into a
// second region.  (Reference: `src/c++/examples/simple_http_shm_client.cc`.)
#include <unistd.h>

#include "example_utils.h"
#include "tca/shm_utils.h"

int main(int argc, char** argv) {
  CommonFlags flags = ParseCommonFlags(argc, argv);
  auto client = MakeClient(flags);
  FAIL_IF_ERR(client->UnregisterSystemSharedMemory(), "unable to unregister all system shared memory regions");
  const std::string tag = std::to_string(getpid());
  const std::string in_key = "/input_simple_" + tag, out_key = "/output_simple_" + tag;
  constexpr size_t kBytes = 64;
  tca::ShmRegion in_region, out_region;
  FAIL_IF_ERR(in_region.Create(in_key, 2 * kBytes), "creating input region");
  FAIL_IF_ERR(out_region.Create(out_key, 2 * kBytes), "creating output region");
  int32_t* in = static_cast<int32_t*>(in_region.data());
  std::vector<int32_t> a(16), b(16, 1);
  for (int i = 0; i < 16; ++i) {
    a[i] = i;
    in[i] = a[i];
    in[16 + i] = b[i];
  }
  FAIL_IF_ERR(client->RegisterSystemSharedMemory("input_data", in_key, 2 * kBytes), "registering input region");
  FAIL_IF_ERR(client->RegisterSystemSharedMemory("output_data", out_key, 2 * kBytes), "registering output region");

  auto in0 = NewInput("INPUT0", {1, 16}, "INT32");
  auto in1 = NewInput("INPUT1", {1, 16}, "INT32");
  FAIL_IF_ERR(in0->SetSharedMemory("input_data", kBytes, 0), "INPUT0 shm");
  FAIL_IF_ERR(in1->SetSharedMemory("input_data", kBytes, kBytes), "INPUT1 shm");
  auto out0 = NewOutput("OUTPUT0");
  auto out1 = NewOutput("OUTPUT1");
  FAIL_IF_ERR(out0->SetSharedMemory("output_data", kBytes, 0), "OUTPUT0 shm");
  FAIL_IF_ERR(out1->SetSharedMemory("output_data", kBytes, kBytes), "OUTPUT1 shm");

  tc::InferResult* result = nullptr;
  FAIL_IF_ERR(client->Infer(&result, tc::InferOptions("simple"), {in0.get(), in1.get()}, {out0.get(), out1.get()},
                            flags.headers),
              "unable to run model");
  delete result;
  const int32_t* out = static_cast<const int32_t*>(out_region.data());
  for (int i = 0; i < 16; ++i) {
    if (out[i] != a[i] + b[i]) FAIL("incorrect sum in shared memory");
    if (out[16 + i] != a[i] - b[i]) FAIL("incorrect difference in shared memory");
  }
  std::string status;
  FAIL_IF_ERR(client->SystemSharedMemoryStatus(&status), "status");
  if (status.find("input_data") == std::string::npos) FAIL("status does not list the input region");
  FAIL_IF_ERR(client->UnregisterSystemSharedMemory("input_data"), "unregister input");
  FAIL_IF_ERR(client->UnregisterSystemSharedMemory("output_data"), "unregister output");
  std::cout << "PASS : System Shared Memory" << std::endl;
  return 0;
}
