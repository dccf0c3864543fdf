// The native execution path under concurrency, for the race / memory-error detectors:
// worker threads run gRPC- and HTTP-shaped `simple` requests (wire tensors and system shared
// memory) through one NativeExecutor while other threads keep replacing the model, registering
// and unregistering the regions and collecting statistics.  Every answered request is checked;
// a declined one is fine (the model or a region was just away), a wrong one is not.
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstring>
#include <iostream>
#include <thread>
#include <vector>

#include "tca/native_executor.h"
#include "tca/pbwire.h"

using tca::server::NativeExecutor;
namespace pb = tca::pb;

namespace {

std::atomic<int> g_failures{0};
void Fail(const std::string& what) {
  ++g_failures;
  std::cerr << "FAIL: " << what << std::endl;
}

NativeExecutor::ModelSpec Simple() {
  NativeExecutor::ModelSpec m;
  m.name = "simple";
  m.version = "1";
  m.max_batch = 8;
  m.kernel = "add_sub";
  m.inputs = {{"INPUT0", "INT32", {16}}, {"INPUT1", "INT32", {16}}};
  m.outputs = {{"OUTPUT0", "INT32", {16}}, {"OUTPUT1", "INT32", {16}}};
  return m;
}

void Param(pb::Writer* t, uint32_t field, const char* key, const std::string* s, int64_t i) {
  std::string entry, value;
  pb::Writer ew(&entry), vw(&value);
  if (s != nullptr) vw.String(3, *s);
  else vw.Int(2, i);
  ew.String(1, key);
  ew.Message(2, value);
  t->Message(field, entry);
}

std::string GrpcRequest(const int32_t* a, const int32_t* b, int batch, bool shm) {
  std::string out;
  pb::Writer w(&out);
  w.String(1, "simple");
  w.String(3, "id-1");
  const std::string region = "stress_region";
  for (int k = 0; k < 2; ++k) {
    std::string t;
    pb::Writer tw(&t);
    tw.String(1, k == 0 ? "INPUT0" : "INPUT1");
    tw.String(2, "INT32");
    std::string shape;
    pb::Writer sw(&shape);
    sw.Varint(static_cast<uint64_t>(batch));
    sw.Varint(16);
    tw.Message(3, shape);
    if (shm) {
      Param(&tw, 4, "shared_memory_region", &region, 0);
      Param(&tw, 4, "shared_memory_byte_size", nullptr, batch * 64);
      if (k == 1) Param(&tw, 4, "shared_memory_offset", nullptr, 512);
    }
    w.Message(5, t);
  }
  if (shm) {
    for (int k = 0; k < 2; ++k) {
      std::string t;
      pb::Writer tw(&t);
      tw.String(1, k == 0 ? "OUTPUT0" : "OUTPUT1");
      Param(&tw, 2, "shared_memory_region", &region, 0);
      Param(&tw, 2, "shared_memory_byte_size", nullptr, batch * 64);
      Param(&tw, 2, "shared_memory_offset", nullptr, 1024 + 512 * k);
      w.Message(6, t);
    }
  } else {
    w.Bytes(7, a, static_cast<size_t>(batch) * 64);
    w.Bytes(7, b, static_cast<size_t>(batch) * 64);
  }
  return out;
}

// raw_output_contents of a ModelInferResponse
std::vector<std::string> RawOutputs(const std::string& response) {
  std::vector<std::string> raws;
  pb::Reader r(response);
  uint32_t f, wt;
  while (r.Next(&f, &wt)) {
    if (f == 6 && wt == pb::kLen) {
      std::string s;
      r.String(&s);
      raws.push_back(std::move(s));
    } else if (!r.Skip(wt)) {
      break;
    }
  }
  return raws;
}

bool CheckSumDiff(const void* sum, const void* diff, const int32_t* a, const int32_t* b, int n) {
  for (int i = 0; i < n; ++i) {
    int32_t s, d;
    std::memcpy(&s, static_cast<const char*>(sum) + 4 * i, 4);
    std::memcpy(&d, static_cast<const char*>(diff) + 4 * i, 4);
    if (s != a[i] + b[i] || d != a[i] - b[i]) return false;
  }
  return true;
}

}  // namespace

int main() {
  const char* key = "/tca_native_executor_test";
  shm_unlink(key);
  const int fd = shm_open(key, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, 4096) != 0) {
    std::cerr << "cannot create the shared memory object" << std::endl;
    return 1;
  }
  auto* mem = static_cast<int32_t*>(mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
  close(fd);
  // inputs for the shm requests: INPUT0 at 0, INPUT1 at 512 (batch up to 8 x 64 bytes each)
  for (int i = 0; i < 128; ++i) {
    mem[i] = i * 3;
    mem[128 + i] = 1000 - i;
  }
  NativeExecutor ex;
  if (!ex.SetModel(Simple()).IsOk() || !ex.RegisterRegion("stress_region", key, 0, 4096).IsOk()) {
    std::cerr << "setup failed" << std::endl;
    return 1;
  }
  std::atomic<bool> stop{false};
  std::atomic<uint64_t> answered{0}, declined{0}, collected{0};
  std::vector<std::thread> threads;
  for (int t = 0; t < 6; ++t) {
    threads.emplace_back([&, t] {
      int32_t a[128], b[128];
      for (int i = 0; i < 128; ++i) {
        a[i] = t * 1000 + i;
        b[i] = 7 * i - t;
      }
      const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(1500);
      for (int round = 0; std::chrono::steady_clock::now() < deadline; ++round) {
        const int batch = 1 + (round + t) % 8;
        const bool shm = (round % 3) == 0 && t == 0;  // one writer for the shared output windows
        if (round % 2 == 0) {
          std::string response;
          if (!ex.TryGrpc(GrpcRequest(a, b, batch, shm), &response)) {
            ++declined;
            continue;
          }
          ++answered;
          if (shm) {
            if (!CheckSumDiff(mem + 256, mem + 384, mem, mem + 128, batch * 16)) Fail("grpc shm result");
          } else {
            const auto raws = RawOutputs(response);
            if (raws.size() != 2 || raws[0].size() != static_cast<size_t>(batch) * 64 ||
                !CheckSumDiff(raws[0].data(), raws[1].data(), a, b, batch * 16))
              Fail("grpc wire result");
          }
        } else {
          const std::string head =
              "{\"id\":\"h\",\"parameters\":{\"binary_data_output\":true},\"inputs\":[{\"name\":\"INPUT0\",\"datatype\":\"INT32\",\"shape\":[" +
              std::to_string(batch) + ",16],\"parameters\":{\"binary_data_size\":" + std::to_string(batch * 64) +
              "}},{\"name\":\"INPUT1\",\"datatype\":\"INT32\",\"shape\":[" + std::to_string(batch) +
              ",16],\"parameters\":{\"binary_data_size\":" + std::to_string(batch * 64) + "}}]}";
          std::string body = head;
          body.append(reinterpret_cast<const char*>(a), static_cast<size_t>(batch) * 64);
          body.append(reinterpret_cast<const char*>(b), static_cast<size_t>(batch) * 64);
          NativeExecutor::HeaderList headers;
          std::vector<std::string> parts;
          if (!ex.TryHttp("simple", "", body, static_cast<int64_t>(head.size()), &headers, &parts)) {
            ++declined;
            continue;
          }
          ++answered;
          if (parts.size() != 3 || headers.size() != 2 || std::stoul(headers[0].second) != parts[0].size() ||
              !CheckSumDiff(parts[1].data(), parts[2].data(), a, b, batch * 16))
            Fail("http result");
        }
      }
    });
  }
  std::thread churn([&] {
    for (int i = 0; !stop.load(); ++i) {
      if (i % 4 == 0) ex.RemoveModel("simple");
      else ex.SetModel(Simple());
      if (i % 5 == 0) ex.UnregisterRegion(i % 10 == 0 ? "" : "stress_region");
      else if (!ex.RegisterRegion("stress_region", key, 0, 4096).IsOk()) Fail("re-registering the region");
      NativeExecutor::Stats st;
      if (ex.TakeStats("simple", &st)) collected += st.requests;
      std::this_thread::sleep_for(std::chrono::milliseconds(3));
    }
  });
  for (auto& t : threads) t.join();
  stop.store(true);
  churn.join();
  // requests declined by a missing model never reach the counters; everything answered does, except what
  // was counted on a model instance that was replaced before its statistics were collected
  ex.SetModel(Simple());
  std::string response;
  int32_t a[16] = {1}, b[16] = {2};
  if (!ex.TryGrpc(GrpcRequest(a, b, 1, false), &response)) Fail("final request declined");
  NativeExecutor::Stats st;
  if (!ex.TakeStats("simple", &st) || st.requests != 1 || st.by_batch[1].count != 1) Fail("final statistics");
  // streams: an ordinary kernel answers with one wrapped response, the decoupled kernel with one per element
  {
    std::vector<std::string> stream;
    if (!ex.TryGrpcStream(GrpcRequest(a, b, 1, false), &stream) || stream.size() != 1) Fail("stream: simple");
    NativeExecutor::ModelSpec rep;
    rep.name = "repeat_int32";
    rep.version = "1";
    rep.kernel = "repeat";
    rep.inputs = {{"IN", "INT32", {-1}}, {"DELAY", "UINT32", {-1}}, {"WAIT", "UINT32", {1}}};
    rep.outputs = {{"OUT", "INT32", {1}}, {"IDX", "UINT32", {1}}};
    if (!ex.SetModel(rep).IsOk()) Fail("registering the decoupled kernel");
    auto repeat_request = [&](uint32_t delay_of_second) {
      std::string out;
      pb::Writer w(&out);
      w.String(1, "repeat_int32");
      w.String(3, "r");
      std::string entry, value;
      pb::Writer ew(&entry), vw(&value);
      vw.Bool(1, true);
      ew.String(1, "triton_enable_empty_final_response");
      ew.Message(2, value);
      w.Message(4, entry);
      const int32_t in[3] = {5, 6, 7};
      const uint32_t delay[3] = {0, delay_of_second, 0}, wait[1] = {0};
      const char* names[3] = {"IN", "DELAY", "WAIT"};
      const char* types[3] = {"INT32", "UINT32", "UINT32"};
      const uint64_t counts[3] = {3, 3, 1};
      for (int k = 0; k < 3; ++k) {
        std::string t, shape;
        pb::Writer tw(&t), sw(&shape);
        tw.String(1, names[k]);
        tw.String(2, types[k]);
        sw.Varint(counts[k]);
        tw.Message(3, shape);
        w.Message(5, t);
      }
      w.Bytes(7, in, 12);
      w.Bytes(7, delay, 12);
      w.Bytes(7, wait, 4);
      return out;
    };
    if (!ex.TryGrpcStream(repeat_request(0), &stream) || stream.size() != 4) Fail("stream: repeat with final marker");
    if (ex.TryGrpcStream(repeat_request(3), &stream)) Fail("stream: a delayed response was taken natively");
    std::string unary;
    if (ex.TryGrpc(repeat_request(0), &unary)) Fail("unary call on the decoupled kernel accepted");
    const std::string full = repeat_request(0);
    for (size_t cut = 0; cut < full.size(); cut += 2) ex.TryGrpcStream(full.substr(0, cut), &stream);
  }
  // malformed input must be declined, not crash
  for (size_t cut = 0; cut < response.size(); cut += 3) {
    std::string junk = GrpcRequest(a, b, 1, false).substr(0, cut);
    ex.TryGrpc(junk, &response);
  }
  NativeExecutor::HeaderList headers;
  std::vector<std::string> parts;
  if (ex.TryHttp("simple", "", "{\"inputs\":[{\"name\":\"INPUT0\"", -1, &headers, &parts)) Fail("truncated JSON accepted");
  if (ex.TryHttp("simple", "", "{\"inputs\":[]}", 400, &headers, &parts)) Fail("header length past the body accepted");
  munmap(mem, 4096);
  shm_unlink(key);
  std::cout << "answered " << answered.load() << ", declined " << declined.load() << ", statistics collected for "
            << collected.load() << ", failures " << g_failures.load() << std::endl;
  if (g_failures.load() != 0 || answered.load() == 0) return 1;
  std::cout << "PASS : native executor" << std::endl;
  return 0;
}
