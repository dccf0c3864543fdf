// The same InferInput / InferRequestedOutput objects are used alternately with
// data on the wire and with system shared memory, sync and async.  (Reference:
// `src/c++/examples/reuse_infer_objects_client.cc`.)
#include <unistd.h>

#include <condition_variable>
#include <mutex>

#include "example_utils.h"
#include "tca/shm_utils.h"

int main(int argc, char** argv) {
  CommonFlags flags = ParseCommonFlags(argc, argv);
  auto client = MakeClient(flags);
  client->UnregisterSystemSharedMemory();
  const std::string tag = std::to_string(getpid());
  tca::ShmRegion in_region, out_region;
  FAIL_IF_ERR(in_region.Create("/reuse_in_" + tag, 128), "input region");
  FAIL_IF_ERR(out_region.Create("/reuse_out_" + tag, 128), "output region");
  std::vector<int32_t> a(16), b(16, 1);
  int32_t* shm_in = static_cast<int32_t*>(in_region.data());
  for (int i = 0; i < 16; ++i) {
    a[i] = i;
    shm_in[i] = a[i];
    shm_in[16 + i] = b[i];
  }
  FAIL_IF_ERR(client->RegisterSystemSharedMemory("input_data", "/reuse_in_" + tag, 128), "register input");
  FAIL_IF_ERR(client->RegisterSystemSharedMemory("output_data", "/reuse_out_" + tag, 128), "register output");
  auto in0 = NewInput("INPUT0", {1, 16}, "INT32");
  auto in1 = NewInput("INPUT1", {1, 16}, "INT32");
  auto out0 = NewOutput("OUTPUT0");
  auto out1 = NewOutput("OUTPUT1");
  std::vector<tc::InferInput*> inputs = {in0.get(), in1.get()};
  std::vector<const tc::InferRequestedOutput*> outputs = {out0.get(), out1.get()};
  tc::InferOptions options("simple");

  for (int round = 0; round < 4; ++round) {
    const bool use_shm = round % 2 == 1;
    const bool async = round >= 2;
    if (use_shm) {
      in0->SetSharedMemory("input_data", 64, 0);
      in1->SetSharedMemory("input_data", 64, 64);
      out0->SetSharedMemory("output_data", 64, 0);
      out1->SetSharedMemory("output_data", 64, 64);
      std::memset(out_region.data(), 0, 128);
    } else {
      in0->Reset();
      in1->Reset();
      in0->AppendRaw(reinterpret_cast<uint8_t*>(a.data()), 64);
      in1->AppendRaw(reinterpret_cast<uint8_t*>(b.data()), 64);
      out0->UnsetSharedMemory();
      out1->UnsetSharedMemory();
    }
    std::unique_ptr<tc::InferResult> result;
    if (async) {
      std::mutex mu;
      std::condition_variable cv;
      bool done = false;
      FAIL_IF_ERR(client->AsyncInfer(
                      [&](tc::InferResult* r) {
                        std::lock_guard<std::mutex> lk(mu);
                        result.reset(r);
                        done = true;
                        cv.notify_one();
                      },
                      options, inputs, outputs),
                  "async request");
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return done; });
    } else {
      tc::InferResult* r = nullptr;
      FAIL_IF_ERR(client->Infer(&r, options, inputs, outputs), "sync request");
      result.reset(r);
    }
    FAIL_IF_ERR(result->RequestStatus(), "inference");
    if (use_shm) {
      const int32_t* out = static_cast<const int32_t*>(out_region.data());
      for (int i = 0; i < 16; ++i)
        if (out[i] != a[i] + b[i] || out[16 + i] != a[i] - b[i]) FAIL("incorrect result in shared memory, round " << round);
    } else {
      ValidateSimple(result.get(), a, b);
    }
  }
  client->UnregisterSystemSharedMemory();
  std::cout << "PASS : Reuse Infer Objects" << std::endl;
  return 0;
}
