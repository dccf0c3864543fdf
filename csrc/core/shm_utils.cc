#include "tca/shm_utils.h"

#include <cerrno>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace tca {

namespace {
std::string Why() { return std::string(std::strerror(errno)); }
}  // namespace

Error CreateSharedMemoryRegion(const std::string& shm_key, size_t byte_size, int* shm_fd) {
  *shm_fd = shm_open(shm_key.c_str(), O_RDWR | O_CREAT, S_IRUSR | S_IWUSR);
  if (*shm_fd == -1)
    return Error("unable to get shared memory descriptor for shared-memory key '" + shm_key + "': " + Why());
  if (ftruncate(*shm_fd, static_cast<off_t>(byte_size)) == -1) {
    std::string why = Why();
    close(*shm_fd);
    *shm_fd = -1;
    return Error("unable to initialize shared-memory key '" + shm_key + "' to requested size " +
                 std::to_string(byte_size) + " bytes: " + why);
  }
  return Error::Success();
}

Error OpenSharedMemoryRegion(const std::string& shm_key, int* shm_fd) {
  *shm_fd = shm_open(shm_key.c_str(), O_RDWR, S_IRUSR | S_IWUSR);
  if (*shm_fd == -1)
    return Error("unable to open shared-memory key '" + shm_key + "': " + Why());
  return Error::Success();
}

Error MapSharedMemory(int shm_fd, size_t offset, size_t byte_size, void** shm_addr) {
  *shm_addr = mmap(nullptr, byte_size, PROT_READ | PROT_WRITE, MAP_SHARED, shm_fd, static_cast<off_t>(offset));
  if (*shm_addr == MAP_FAILED) {
    *shm_addr = nullptr;
    return Error("unable to process address space or shared-memory descriptor " + std::to_string(shm_fd) + ": " + Why());
  }
  return Error::Success();
}

Error CloseSharedMemory(int shm_fd) {
  if (close(shm_fd) == -1)
    return Error("unable to close shared-memory descriptor " + std::to_string(shm_fd) + ": " + Why());
  return Error::Success();
}

Error UnlinkSharedMemoryRegion(const std::string& shm_key) {
  if (shm_unlink(shm_key.c_str()) == -1)
    return Error("unable to unlink shared memory for key '" + shm_key + "': " + Why());
  return Error::Success();
}

Error UnmapSharedMemory(void* shm_addr, size_t byte_size) {
  if (munmap(shm_addr, byte_size) == -1)
    return Error("unable to munmap shared memory region: " + Why());
  return Error::Success();
}

Error ShmRegion::Create(const std::string& key, size_t byte_size) {
  Reset();
  int fd = -1;
  TCA_RETURN_IF_ERROR(CreateSharedMemoryRegion(key, byte_size, &fd));
  Error e = MapSharedMemory(fd, 0, byte_size, &addr_);
  CloseSharedMemory(fd);
  if (!e.IsOk()) {
    shm_unlink(key.c_str());
    return e;
  }
  key_ = key;
  size_ = map_size_ = byte_size;
  map_base_ = addr_;
  owner_ = true;
  return Error::Success();
}

Error ShmRegion::Attach(const std::string& key, size_t offset, size_t byte_size) {
  Reset();
  int fd = -1;
  TCA_RETURN_IF_ERROR(OpenSharedMemoryRegion(key, &fd));
  struct stat st;
  if (fstat(fd, &st) == -1 || static_cast<size_t>(st.st_size) < offset + byte_size) {
    CloseSharedMemory(fd);
    return Error("shared-memory key '" + key + "' is smaller than offset + byte_size");
  }
  // mmap offsets must be page aligned: map from the enclosing page and bias.
  const size_t page = static_cast<size_t>(sysconf(_SC_PAGESIZE));
  const size_t aligned = offset / page * page;
  const size_t bias = offset - aligned;
  Error e = MapSharedMemory(fd, aligned, byte_size + bias, &map_base_);
  CloseSharedMemory(fd);
  TCA_RETURN_IF_ERROR(e);
  key_ = key;
  addr_ = static_cast<char*>(map_base_) + bias;
  size_ = byte_size;
  map_size_ = byte_size + bias;
  owner_ = false;
  return Error::Success();
}

void ShmRegion::Reset() {
  if (map_base_ != nullptr) munmap(map_base_, map_size_);
  if (owner_ && !key_.empty()) shm_unlink(key_.c_str());
  map_base_ = addr_ = nullptr;
  size_ = map_size_ = 0;
  owner_ = false;
  key_.clear();
}

}  // namespace tca
