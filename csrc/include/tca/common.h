// Data model shared by the native clients: inference options, input / output
// descriptions, result interface, per-request timers and cumulative statistics.
//
// Capability parity: reference `src/c++/library/common.h:93-673` and
// `common.cc:55-289` (InferStat, InferOptions, InferInput with scatter-gather
// buffers and BYTES serialisation, InferRequestedOutput, InferResult,
// RequestTimers).  Differences by design: exit flags are atomics, InferInput
// owns the bytes produced by AppendFromString (no dangling string storage), and
// the cursor used while sending lives in the request, not in the shared
// InferInput, so one InferInput can be part of several in-flight requests.
#pragma once

#include <cstring>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "tca/error.h"

namespace tca {

// Element `index` of a tensor buffer as returned by InferResult::RawData.  Tensor
// bytes sit at arbitrary offsets inside a response body (after a JSON header, after
// a BYTES tensor), so they must not be read through a typed pointer.
template <typename T>
inline T LoadAs(const uint8_t* buf, size_t index = 0) {
  T value;
  std::memcpy(&value, buf + index * sizeof(T), sizeof(T));
  return value;
}

namespace client {

constexpr char kInferHeaderContentLengthHTTPHeader[] = "Inference-Header-Content-Length";
constexpr char kContentLengthHTTPHeader[] = "Content-Length";
constexpr size_t MAX_GRPC_MESSAGE_SIZE = INT32_MAX;

using Headers = std::map<std::string, std::string>;
using Parameters = std::map<std::string, std::string>;

//==============================================================================
// Cumulative client-side statistics (all durations in nanoseconds).
struct InferStat {
  size_t completed_request_count = 0;
  uint64_t cumulative_total_request_time_ns = 0;
  uint64_t cumulative_send_time_ns = 0;
  uint64_t cumulative_receive_time_ns = 0;
};

//==============================================================================
// Six timestamps per request; durations are derived from pairs of them.
class RequestTimers {
 public:
  enum class Kind { REQUEST_START, REQUEST_END, SEND_START, SEND_END, RECV_START, RECV_END, COUNT__ };

  RequestTimers() { Reset(); }
  void Reset() { for (auto& t : ts_) t = 0; }
  uint64_t Timestamp(Kind k) const { return ts_[static_cast<size_t>(k)]; }
  uint64_t CaptureTimestamp(Kind k) {
    ts_[static_cast<size_t>(k)] = static_cast<uint64_t>(
        std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count());
    return ts_[static_cast<size_t>(k)];
  }
  // Duration between two stamps, or uint64 max when they are unset / reversed.
  uint64_t Duration(Kind start, Kind end) const {
    const uint64_t s = Timestamp(start), e = Timestamp(end);
    return (s == 0 || e == 0 || s > e) ? UINT64_MAX : e - s;
  }

 private:
  uint64_t ts_[static_cast<size_t>(Kind::COUNT__)];
};

//==============================================================================
// A typed request parameter (custom "parameters" of the protocol).
struct RequestParameter {
  enum class Type { STRING, INT, UINT, BOOL, DOUBLE } type = Type::STRING;
  std::string str_value;
  int64_t int_value = 0;
  uint64_t uint_value = 0;
  bool bool_value = false;
  double double_value = 0.0;
};

struct InferOptions {
  explicit InferOptions(std::string model_name) : model_name_(std::move(model_name)) {}
  std::string model_name_;
  std::string model_version_;
  std::string request_id_;
  uint64_t sequence_id_ = 0;        // numeric correlation id (0 = none) ...
  std::string sequence_id_str_;     // ... or a string one
  bool sequence_start_ = false;
  bool sequence_end_ = false;
  uint64_t priority_ = 0;
  uint64_t server_timeout_ = 0;     // microseconds, queue policy timeout on the server
  uint64_t client_timeout_ = 0;     // microseconds, 0 = wait forever
  bool triton_enable_empty_final_response_ = false;
  std::map<std::string, RequestParameter> request_parameters;
};

//==============================================================================
class InferInput {
 public:
  static Error Create(InferInput** infer_input, const std::string& name, const std::vector<int64_t>& dims,
                      const std::string& datatype);

  const std::string& Name() const { return name_; }
  const std::string& Datatype() const { return datatype_; }
  const std::vector<int64_t>& Shape() const { return shape_; }
  Error SetShape(const std::vector<int64_t>& dims) { shape_ = dims; return Error::Success(); }

  // Forget all data and any shared memory setting.
  Error Reset();
  // Reference (do not copy) `input_bytes` more bytes at `input`; the caller keeps
  // them alive until the request completed.
  Error AppendRaw(const uint8_t* input, size_t input_byte_size);
  Error AppendRaw(const std::vector<uint8_t>& input) { return AppendRaw(input.data(), input.size()); }
  // BYTES tensors: serialises each string as <u32 length><bytes>; the
  // serialised form is owned by this object.
  Error AppendFromString(const std::vector<std::string>& input);
  Error SetSharedMemory(const std::string& region_name, size_t byte_size, size_t offset = 0);
  // When false the data is embedded in the request JSON (HTTP only).
  Error SetBinaryData(bool binary_data) { binary_data_ = binary_data; return Error::Success(); }

  bool IsSharedMemory() const { return io_type_ == IOType::SHARED_MEMORY; }
  bool BinaryData() const { return binary_data_; }
  Error SharedMemoryInfo(std::string* name, size_t* byte_size, size_t* offset) const;
  Error ByteSize(size_t* byte_size) const { *byte_size = total_bytes_; return Error::Success(); }
  // (pointer, length) pieces in order.
  const std::vector<std::pair<const uint8_t*, size_t>>& Buffers() const { return bufs_; }
  // Copy out everything into one contiguous string (used for JSON conversion).
  void Gather(std::string* out) const;

 private:
  InferInput(std::string name, std::vector<int64_t> dims, std::string datatype)
      : name_(std::move(name)), shape_(std::move(dims)), datatype_(std::move(datatype)) {}
  enum class IOType { NONE, RAW, SHARED_MEMORY };
  std::string name_;
  std::vector<int64_t> shape_;
  std::string datatype_;
  IOType io_type_ = IOType::NONE;
  bool binary_data_ = true;
  size_t total_bytes_ = 0;
  std::vector<std::pair<const uint8_t*, size_t>> bufs_;
  std::deque<std::string> owned_;  // deque: growth never moves earlier strings
  std::string shm_name_;
  size_t shm_byte_size_ = 0;
  size_t shm_offset_ = 0;
};

//==============================================================================
class InferRequestedOutput {
 public:
  static Error Create(InferRequestedOutput** infer_output, const std::string& name, size_t class_count = 0,
                      const std::string& datatype = "");
  const std::string& Name() const { return name_; }
  size_t ClassificationCount() const { return class_count_; }
  Error SetSharedMemory(const std::string& region_name, size_t byte_size, size_t offset = 0);
  Error UnsetSharedMemory();
  bool IsSharedMemory() const { return shm_; }
  Error SharedMemoryInfo(std::string* name, size_t* byte_size, size_t* offset) const;
  Error SetBinaryData(bool binary_data) { binary_data_ = binary_data; return Error::Success(); }
  bool BinaryData() const { return binary_data_; }

 private:
  InferRequestedOutput(std::string name, size_t class_count) : name_(std::move(name)), class_count_(class_count) {}
  std::string name_;
  size_t class_count_ = 0;
  bool binary_data_ = true;
  bool shm_ = false;
  std::string shm_name_;
  size_t shm_byte_size_ = 0;
  size_t shm_offset_ = 0;
};

//==============================================================================
class InferResult {
 public:
  virtual ~InferResult() = default;
  virtual Error ModelName(std::string* name) const = 0;
  virtual Error ModelVersion(std::string* version) const = 0;
  virtual Error Id(std::string* id) const = 0;
  virtual Error Shape(const std::string& output_name, std::vector<int64_t>* shape) const = 0;
  virtual Error Datatype(const std::string& output_name, std::string* datatype) const = 0;
  // Zero-copy view of an output's bytes inside the response buffer.
  virtual Error RawData(const std::string& output_name, const uint8_t** buf, size_t* byte_size) const = 0;
  // BYTES outputs decoded into strings.
  virtual Error StringData(const std::string& output_name, std::vector<std::string>* string_result) const = 0;
  virtual Error IsFinalResponse(bool* is_final_response) const = 0;
  virtual Error IsNullResponse(bool* is_null_response) const = 0;
  virtual std::string DebugString() const = 0;
  // OK, or the error the server reported for this request.
  virtual Error RequestStatus() const = 0;
};

using OnCompleteFn = std::function<void(InferResult*)>;
using OnMultiCompleteFn = std::function<void(std::vector<InferResult*>)>;

//==============================================================================
// Base of the protocol clients: statistics and the shutdown flag.
class InferenceServerClient {
 public:
  explicit InferenceServerClient(bool verbose) : verbose_(verbose) {}
  virtual ~InferenceServerClient() = default;
  Error ClientInferStat(InferStat* infer_stat) const;

 protected:
  Error UpdateInferStat(const RequestTimers& timer);
  const bool verbose_;
  std::atomic<bool> exiting_{false};
  mutable std::mutex stat_mutex_;
  InferStat infer_stat_;
};

}  // namespace client
}  // namespace tca
