// Resource growth check (`memory_leak_test -u host:port -r N [-R]`): N
// inferences on `custom_identity_int32`, either with a fresh client per request
// or (-R) reusing one client, then compares the process RSS and the number of
// open file descriptors before/after.  Intent: reference
// `src/c++/tests/memory_leak_test.cc:125-324` (which leaves the verdict to
// valgrind; here the test asserts bounded growth itself).
#include <dirent.h>
#include <unistd.h>

#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>

#include "tca/http_client.h"

namespace tc = tca::client;

static long RssKb() {
  std::ifstream f("/proc/self/statm");
  long size = 0, rss = 0;
  f >> size >> rss;
  return rss * (sysconf(_SC_PAGESIZE) / 1024);
}

static int OpenFds() {
  int n = 0;
  if (DIR* d = opendir("/proc/self/fd")) {
    while (readdir(d)) ++n;
    closedir(d);
  }
  return n;
}

int main(int argc, char** argv) {
  std::string url = "localhost:8000";
  int reps = 500;
  bool reuse = false;
  for (int i = 1; i < argc; ++i) {
    if (!std::strcmp(argv[i], "-u") && i + 1 < argc) url = argv[++i];
    else if (!std::strcmp(argv[i], "-r") && i + 1 < argc) reps = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "-R")) reuse = true;
  }
  int32_t data[16];
  for (int i = 0; i < 16; ++i) data[i] = i;
  auto run_one = [&](tc::InferenceServerHttpClient* c) -> bool {
    tc::InferInput* in = nullptr;
    if (!tc::InferInput::Create(&in, "INPUT0", {1, 16}, "INT32").IsOk()) return false;
    std::unique_ptr<tc::InferInput> guard(in);
    in->AppendRaw(reinterpret_cast<const uint8_t*>(data), sizeof(data));
    tc::InferResult* res = nullptr;
    tca::Error e = c->Infer(&res, tc::InferOptions("custom_identity_int32"), {in});
    if (!e.IsOk()) {
      std::cerr << "inference failed: " << e.Message() << std::endl;
      return false;
    }
    const uint8_t* buf = nullptr;
    size_t n = 0;
    bool ok = res->RawData("OUTPUT0", &buf, &n).IsOk() && n == sizeof(data) && !std::memcmp(buf, data, n);
    delete res;
    return ok;
  };
  std::unique_ptr<tc::InferenceServerHttpClient> shared;
  if (reuse && !tc::InferenceServerHttpClient::Create(&shared, url).IsOk()) return 1;
  // warm up allocators before sampling
  for (int i = 0; i < 20; ++i) {
    std::unique_ptr<tc::InferenceServerHttpClient> c;
    if (!reuse && !tc::InferenceServerHttpClient::Create(&c, url).IsOk()) return 1;
    if (!run_one(reuse ? shared.get() : c.get())) return 1;
  }
  const long rss0 = RssKb();
  const int fds0 = OpenFds();
  for (int i = 0; i < reps; ++i) {
    std::unique_ptr<tc::InferenceServerHttpClient> c;
    if (!reuse && !tc::InferenceServerHttpClient::Create(&c, url).IsOk()) return 1;
    if (!run_one(reuse ? shared.get() : c.get())) return 1;
  }
  const long rss1 = RssKb();
  const int fds1 = OpenFds();
  std::cout << "rss " << rss0 << " -> " << rss1 << " kB, fds " << fds0 << " -> " << fds1 << std::endl;
  if (fds1 > fds0 + 2) {
    std::cerr << "file descriptors leaked" << std::endl;
    return 1;
  }
  if (rss1 - rss0 > 4096 + reps / 4) {
    std::cerr << "resident memory grew by " << (rss1 - rss0) << " kB over " << reps << " requests" << std::endl;
    return 1;
  }
  std::cout << "PASS: memory_leak_test" << std::endl;
  return 0;
}
