// Stream-ordered flags for hand-offs between GPUs / processes without a host round trip.
//
// A flag is a 32-bit counter in fine-grained device memory (hipExtMallocWithFlags with
// hipDeviceMallocFinegrained, exported to the other processes as a HIP IPC handle): coherent
// between GPUs while kernels run.  A producer enqueues "set flag to v" behind its work on a
// stream; a consumer enqueues "wait until flag >= v" in front of its work.  Two
// implementations with the same meaning:
//   * command-processor packets (hipStreamWriteValue32 / hipStreamWaitValue32): no compute unit
//     involved; availability is a device attribute;
//   * one-lane kernels: a system-scope release store, and an acquire-load poll loop that gives
//     up after `timeout_ms` (it then raises the error word instead of spinning forever, so the
//     grid always drains).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

namespace {

thread_local std::string g_flag_err;

int Fail(const char* what, hipError_t e) {
  g_flag_err = std::string(what) + ": " + hipGetErrorString(e);
  return -1000 - static_cast<int>(e);
}

__global__ void FlagSetKernel(uint32_t* flag, uint32_t value) {
  __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// wall_clock64() ticks at 100 MHz on gfx950: timeout_ticks = timeout_ms * 100000
__global__ void FlagWaitKernel(const uint32_t* flag, uint32_t value, uint64_t timeout_ticks, uint32_t* error) {
  const uint64_t start = wall_clock64();
  while (static_cast<int32_t>(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - value) < 0) {
    if (wall_clock64() - start > timeout_ticks) {
      if (error != nullptr) __hip_atomic_store(error, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    __builtin_amdgcn_s_sleep(32);
  }
}

}  // namespace

extern "C" {

const char* tca_flag_error() { return g_flag_err.c_str(); }

int tca_can_stream_wait_value(int device, int* yes) {
  int v = 0;
  const hipError_t e = hipDeviceGetAttribute(&v, hipDeviceAttributeCanUseStreamWaitValue, device);
  if (e != hipSuccess) return Fail("hipDeviceGetAttribute", e);
  *yes = v;
  return 0;
}

// mode 0: kernels, mode 1: command-processor packets
int tca_flag_set(void* stream, void* flag, uint32_t value, int mode) {
  auto st = static_cast<hipStream_t>(stream);
  if (mode == 1) {
    const hipError_t e = hipStreamWriteValue32(st, flag, value, 0);
    return e == hipSuccess ? 0 : Fail("hipStreamWriteValue32", e);
  }
  hipLaunchKernelGGL(FlagSetKernel, dim3(1), dim3(1), 0, st, static_cast<uint32_t*>(flag), value);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : Fail("FlagSetKernel", e);
}

int tca_flag_wait(void* stream, void* flag, uint32_t value, int mode, uint32_t timeout_ms, void* error_word) {
  auto st = static_cast<hipStream_t>(stream);
  if (mode == 1) {
    const hipError_t e = hipStreamWaitValue32(st, flag, value, hipStreamWaitValueGte, 0xFFFFFFFFu);
    return e == hipSuccess ? 0 : Fail("hipStreamWaitValue32", e);
  }
  hipLaunchKernelGGL(FlagWaitKernel, dim3(1), dim3(1), 0, st, static_cast<const uint32_t*>(flag), value,
                     static_cast<uint64_t>(timeout_ms) * 100000ull, static_cast<uint32_t*>(error_word));
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : Fail("FlagWaitKernel", e);
}

}  // extern "C"
