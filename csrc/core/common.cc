#include "tca/common.h"

#include "tca/codec.h"

namespace tca {
namespace client {

Error InferInput::Create(InferInput** infer_input, const std::string& name, const std::vector<int64_t>& dims,
                         const std::string& datatype) {
  if (ParseDataType(datatype) == DataType::INVALID)
    return Error("invalid datatype '" + datatype + "' for input '" + name + "'");
  *infer_input = new InferInput(name, dims, datatype);
  return Error::Success();
}

Error InferInput::Reset() {
  bufs_.clear();
  owned_.clear();
  total_bytes_ = 0;
  io_type_ = IOType::NONE;
  shm_name_.clear();
  shm_byte_size_ = shm_offset_ = 0;
  return Error::Success();
}

Error InferInput::AppendRaw(const uint8_t* input, size_t input_byte_size) {
  if (io_type_ == IOType::SHARED_MEMORY) Reset();
  io_type_ = IOType::RAW;
  bufs_.emplace_back(input, input_byte_size);
  total_bytes_ += input_byte_size;
  return Error::Success();
}

Error InferInput::AppendFromString(const std::vector<std::string>& input) {
  if (datatype_ != "BYTES") return Error("AppendFromString is only valid for BYTES tensors, '" + name_ + "' is " + datatype_);
  owned_.emplace_back(BytesSerialize(input));
  const std::string& blob = owned_.back();
  return AppendRaw(reinterpret_cast<const uint8_t*>(blob.data()), blob.size());
}

Error InferInput::SetSharedMemory(const std::string& region_name, size_t byte_size, size_t offset) {
  Reset();
  io_type_ = IOType::SHARED_MEMORY;
  shm_name_ = region_name;
  shm_byte_size_ = byte_size;
  shm_offset_ = offset;
  return Error::Success();
}

Error InferInput::SharedMemoryInfo(std::string* name, size_t* byte_size, size_t* offset) const {
  if (io_type_ != IOType::SHARED_MEMORY) return Error("The input '" + name_ + "' has not set with the shared memory.");
  *name = shm_name_;
  *byte_size = shm_byte_size_;
  *offset = shm_offset_;
  return Error::Success();
}

void InferInput::Gather(std::string* out) const {
  out->clear();
  out->reserve(total_bytes_);
  for (const auto& b : bufs_) out->append(reinterpret_cast<const char*>(b.first), b.second);
}

Error InferRequestedOutput::Create(InferRequestedOutput** infer_output, const std::string& name, size_t class_count,
                                   const std::string&) {
  *infer_output = new InferRequestedOutput(name, class_count);
  return Error::Success();
}

Error InferRequestedOutput::SetSharedMemory(const std::string& region_name, size_t byte_size, size_t offset) {
  if (class_count_ != 0) return Error("shared memory can't be set on classification output");
  shm_ = true;
  shm_name_ = region_name;
  shm_byte_size_ = byte_size;
  shm_offset_ = offset;
  return Error::Success();
}

Error InferRequestedOutput::UnsetSharedMemory() {
  shm_ = false;
  shm_name_.clear();
  shm_byte_size_ = shm_offset_ = 0;
  return Error::Success();
}

Error InferRequestedOutput::SharedMemoryInfo(std::string* name, size_t* byte_size, size_t* offset) const {
  if (!shm_) return Error("The output '" + name_ + "' has not set with the shared memory.");
  *name = shm_name_;
  *byte_size = shm_byte_size_;
  *offset = shm_offset_;
  return Error::Success();
}

Error InferenceServerClient::ClientInferStat(InferStat* infer_stat) const {
  std::lock_guard<std::mutex> lk(stat_mutex_);
  *infer_stat = infer_stat_;
  return Error::Success();
}

Error InferenceServerClient::UpdateInferStat(const RequestTimers& timer) {
  using K = RequestTimers::Kind;
  const uint64_t total = timer.Duration(K::REQUEST_START, K::REQUEST_END);
  if (total == UINT64_MAX) return Error("Clock total time invalid");
  const uint64_t send = timer.Duration(K::SEND_START, K::SEND_END);
  if (send == UINT64_MAX) return Error("Clock send time invalid");
  const uint64_t recv = timer.Duration(K::RECV_START, K::RECV_END);
  if (recv == UINT64_MAX) return Error("Clock receive time invalid");
  std::lock_guard<std::mutex> lk(stat_mutex_);
  infer_stat_.completed_request_count++;
  infer_stat_.cumulative_total_request_time_ns += total;
  infer_stat_.cumulative_send_time_ns += send;
  infer_stat_.cumulative_receive_time_ns += recv;
  return Error::Success();
}

}  // namespace client
}  // namespace tca
