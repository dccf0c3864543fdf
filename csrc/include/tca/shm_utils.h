// POSIX system shared-memory helpers for C++ clients.
//
// Parity: reference `src/c++/library/shm_utils.{h,cc}` (five free functions
// returning Error).  `ShmRegion` adds an RAII owner on top, which the reference
// leaves to each example.
#pragma once

#include <cstddef>
#include <string>

#include "tca/error.h"

namespace tca {

// shm_open(O_RDWR|O_CREAT, 0600) + ftruncate(byte_size); returns the fd.
Error CreateSharedMemoryRegion(const std::string& shm_key, size_t byte_size, int* shm_fd);
// shm_open(O_RDWR) of an existing object (used by the server side).
Error OpenSharedMemoryRegion(const std::string& shm_key, int* shm_fd);
// mmap(PROT_READ|PROT_WRITE, MAP_SHARED) of [offset, offset+byte_size).
Error MapSharedMemory(int shm_fd, size_t offset, size_t byte_size, void** shm_addr);
Error CloseSharedMemory(int shm_fd);
Error UnlinkSharedMemoryRegion(const std::string& shm_key);
Error UnmapSharedMemory(void* shm_addr, size_t byte_size);

class ShmRegion {
 public:
  ShmRegion() = default;
  ~ShmRegion() { Reset(); }
  ShmRegion(const ShmRegion&) = delete;
  ShmRegion& operator=(const ShmRegion&) = delete;

  // Creates (owner == true, unlinks on destruction) or attaches to `key`.
  Error Create(const std::string& key, size_t byte_size);
  Error Attach(const std::string& key, size_t offset, size_t byte_size);
  void Reset();

  void* data() const { return addr_; }
  size_t size() const { return size_; }
  const std::string& key() const { return key_; }

 private:
  std::string key_;
  void* addr_ = nullptr;
  size_t size_ = 0;
  size_t map_size_ = 0;
  void* map_base_ = nullptr;
  bool owner_ = false;
};

}  // namespace tca
