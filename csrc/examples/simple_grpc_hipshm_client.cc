// Device shared memory over gRPC on MI355X: input and output tensors of `simple`
// live in GPU memory allocated by this process (hipMalloc), exported with
// hipIpcGetMemHandle and registered with the server, which opens the handle in
// its own process and reads / writes the GPU memory directly.  The inference
// request carries only region names and byte counts.
// (Reference: `src/c++/examples/simple_http_cudashm_client.cc:106-115,196-319`.)
#include <hip/hip_runtime_api.h>

#include "example_utils.h"

#define HIP_OK(X)                                                                      \
  do {                                                                                 \
    hipError_t e__ = (X);                                                              \
    if (e__ != hipSuccess) FAIL(#X << " failed: " << hipGetErrorString(e__));           \
  } while (0)

int main(int argc, char** argv) {
  CommonFlags flags = ParseCommonFlags(argc, argv);
  int device = 0;
  for (size_t i = 0; i + 1 < flags.rest.size(); ++i)
    if (flags.rest[i] == "-g") device = std::atoi(flags.rest[i + 1].c_str());
  auto client = MakeGrpcClient(flags);
  FAIL_IF_ERR(client->UnregisterCudaSharedMemory(), "unable to unregister all device shared memory regions");
  HIP_OK(hipSetDevice(device));
  constexpr size_t kBytes = 64;
  std::vector<int32_t> a(16), b(16, 1);
  for (int i = 0; i < 16; ++i) a[i] = i;
  void *in_dev = nullptr, *out_dev = nullptr;
  HIP_OK(hipMalloc(&in_dev, 2 * kBytes));
  HIP_OK(hipMalloc(&out_dev, 2 * kBytes));
  HIP_OK(hipMemcpy(in_dev, a.data(), kBytes, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(static_cast<char*>(in_dev) + kBytes, b.data(), kBytes, hipMemcpyHostToDevice));
  HIP_OK(hipMemset(out_dev, 0, 2 * kBytes));
  hipIpcMemHandle_t in_handle, out_handle;
  HIP_OK(hipIpcGetMemHandle(&in_handle, in_dev));
  HIP_OK(hipIpcGetMemHandle(&out_handle, out_dev));
  FAIL_IF_ERR(client->RegisterCudaSharedMemory("input_data", &in_handle, device, 2 * kBytes), "registering input region");
  FAIL_IF_ERR(client->RegisterCudaSharedMemory("output_data", &out_handle, device, 2 * kBytes), "registering output region");

  auto in0 = NewInput("INPUT0", {1, 16}, "INT32");
  auto in1 = NewInput("INPUT1", {1, 16}, "INT32");
  in0->SetSharedMemory("input_data", kBytes, 0);
  in1->SetSharedMemory("input_data", kBytes, kBytes);
  auto out0 = NewOutput("OUTPUT0");
  auto out1 = NewOutput("OUTPUT1");
  out0->SetSharedMemory("output_data", kBytes, 0);
  out1->SetSharedMemory("output_data", kBytes, kBytes);
  tc::InferResult* result = nullptr;
  FAIL_IF_ERR(client->Infer(&result, tc::InferOptions("simple"), {in0.get(), in1.get()}, {out0.get(), out1.get()},
                            flags.headers),
              "unable to run model");
  delete result;
  std::vector<int32_t> sum(16), diff(16);
  HIP_OK(hipMemcpy(sum.data(), out_dev, kBytes, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(diff.data(), static_cast<char*>(out_dev) + kBytes, kBytes, hipMemcpyDeviceToHost));
  for (int i = 0; i < 16; ++i) {
    if (sum[i] != a[i] + b[i]) FAIL("incorrect sum in device shared memory");
    if (diff[i] != a[i] - b[i]) FAIL("incorrect difference in device shared memory");
  }
  std::string status;
  FAIL_IF_ERR(client->CudaSharedMemoryStatus(&status), "status");
  if (status.find("output_data") == std::string::npos) FAIL("status does not list the output region");
  FAIL_IF_ERR(client->UnregisterCudaSharedMemory("input_data"), "unregister input");
  FAIL_IF_ERR(client->UnregisterCudaSharedMemory("output_data"), "unregister output");
  HIP_OK(hipFree(in_dev));
  HIP_OK(hipFree(out_dev));
  std::cout << "PASS : HIP Shared Memory" << std::endl;
  return 0;
}
