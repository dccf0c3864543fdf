// Resource growth check (`memory_leak_test -u host:port [-i http|grpc] -r N [-R]`): N
// inferences on `custom_identity_int32`, either with a fresh client per request
// or (-R) reusing one client, then compares the process RSS and the number of
// open file descriptors before/after.  A fresh client per request can run the
// machine out of ephemeral ports (connections linger in TIME_WAIT): creating the
// client is retried with a pause, as the reference does (`:134-153`).  Intent: reference
// `src/c++/tests/memory_leak_test.cc:125-183,210-324` (which leaves the verdict to
// valgrind; here the test asserts bounded growth itself).
#include <dirent.h>
#include <unistd.h>

#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>

#include <chrono>
#include <thread>

#include "tca/grpc_client.h"
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

// Client of either protocol behind one face.
struct AnyClient {
  std::unique_ptr<tc::InferenceServerHttpClient> http;
  std::unique_ptr<tc::InferenceServerGrpcClient> grpc;
  tca::Error Infer(tc::InferResult** res, const tc::InferOptions& opt, const std::vector<tc::InferInput*>& in) {
    return http ? http->Infer(res, opt, in) : grpc->Infer(res, opt, in);
  }
};

// Retries while the failure looks like socket exhaustion (reference `:134-153`: 5 tries, 60 s apart; a
// shorter pause here, TIME_WAIT sockets free up continuously).
static bool MakeClient(AnyClient* c, const std::string& url, bool use_grpc) {
  for (int attempt = 0; attempt < 5; ++attempt) {
    // a channel of its own for every client: a cached one would make "new client" mean nothing
    const tca::Error e = use_grpc ? tc::InferenceServerGrpcClient::Create(&c->grpc, url, false, false, tc::SslOptions(),
                                                                           tc::KeepAliveOptions(), false)
                                  : tc::InferenceServerHttpClient::Create(&c->http, url);
    if (e.IsOk()) return true;
    std::cerr << "creating a client failed (" << e.Message() << "), retrying" << std::endl;
    std::this_thread::sleep_for(std::chrono::seconds(5));
  }
  return false;
}

int main(int argc, char** argv) {
  std::string url, protocol = "http";
  int reps = 500;
  bool reuse = false;
  for (int i = 1; i < argc; ++i) {
    if (!std::strcmp(argv[i], "-u") && i + 1 < argc) url = argv[++i];
    else if (!std::strcmp(argv[i], "-i") && i + 1 < argc) protocol = argv[++i];
    else if (!std::strcmp(argv[i], "-r") && i + 1 < argc) reps = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "-R")) reuse = true;
  }
  const bool use_grpc = protocol == "grpc" || protocol == "GRPC";
  if (!use_grpc && protocol != "http" && protocol != "HTTP") {
    std::cerr << "unknown protocol '" << protocol << "' (http | grpc)" << std::endl;
    return 2;
  }
  if (url.empty()) url = use_grpc ? "localhost:8001" : "localhost:8000";
  int32_t data[16];
  for (int i = 0; i < 16; ++i) data[i] = i;
  auto run_one = [&](AnyClient* c) -> bool {
    tc::InferInput* in = nullptr;
    if (!tc::InferInput::Create(&in, "INPUT0", {1, 16}, "INT32").IsOk()) return false;
    std::unique_ptr<tc::InferInput> guard(in);
    in->AppendRaw(reinterpret_cast<const uint8_t*>(data), sizeof(data));
    tc::InferResult* res = nullptr;
    tca::Error e = c->Infer(&res, tc::InferOptions("custom_identity_int32"), {in});
    for (int attempt = 0; !e.IsOk() && attempt < 5 && e.Message().find("connect") != std::string::npos; ++attempt) {
      std::this_thread::sleep_for(std::chrono::seconds(5));  // out of sockets: wait for TIME_WAIT entries to expire
      e = c->Infer(&res, tc::InferOptions("custom_identity_int32"), {in});
    }
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
  AnyClient shared;
  if (reuse && !MakeClient(&shared, url, use_grpc)) return 1;
  // warm up allocators before sampling
  for (int i = 0; i < 20; ++i) {
    AnyClient c;
    if (!reuse && !MakeClient(&c, url, use_grpc)) return 1;
    if (!run_one(reuse ? &shared : &c)) return 1;
  }
  const long rss0 = RssKb();
  const int fds0 = OpenFds();
  for (int i = 0; i < reps; ++i) {
    AnyClient c;
    if (!reuse && !MakeClient(&c, url, use_grpc)) return 1;
    if (!run_one(reuse ? &shared : &c)) return 1;
  }
  const long rss1 = RssKb();
  const int fds1 = OpenFds();
  std::cout << protocol << ": rss " << rss0 << " -> " << rss1 << " kB, fds " << fds0 << " -> " << fds1 << std::endl;
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
