// The native execution path under concurrency, for the race / memory-error detectors:
// worker threads run gRPC- and HTTP-shaped `simple` requests (wire tensors and system shared
// memory) through one NativeExecutor while other threads keep replacing the model, registering
// and unregistering the regions and collecting statistics.  Every answered request is checked;
// a declined one is fine (the model or a region was just away), a wrong one is not.
// A second part drives the device BYTES path (deferred answers from the executor's batch thread)
// against a host stand-in for the HIP pipeline while the region is unregistered and re-registered.
// A third part serves a "graph" model (requests merged along the batch axis, padded up to the next captured
// size, gathered / launched / scattered on instance streams and answered by completer threads) against
// host stand-ins for streams, events, graphs and the batched copy.
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
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

// ---- device BYTES path, with a stand-in for the HIP pipeline that works on host memory ----------------
// The executor only sees the pipeline's C entry points; these two do on the host what the kernels do on the
// device (walk the length-prefixed streams, add / subtract the decimals, pack the results), so "device
// addresses" are plain pointers here.
struct FakePipe {
  int n = 0, op = 0, count = 0;
  std::vector<uint32_t> codes;
  std::vector<uint64_t> produced;
  std::atomic<int> batches{0}, largest{0};
};

bool WalkStream(const uint8_t* p, uint64_t size, int count, std::vector<std::string>* out) {
  uint64_t at = 0;
  for (int e = 0; e < count; ++e) {
    if (at + 4 > size) return false;
    uint32_t len;
    std::memcpy(&len, p + at, 4);
    at += 4;
    if (at + len > size) return false;
    out->emplace_back(reinterpret_cast<const char*>(p + at), len);
    at += len;
  }
  return at == size;
}

bool PackStream(const std::vector<std::string>& items, uint8_t* dst, uint64_t cap, uint64_t* total) {
  uint64_t at = 0;
  for (const auto& s : items) {
    if (at + 4 + s.size() > cap) return false;
    const uint32_t len = static_cast<uint32_t>(s.size());
    std::memcpy(dst + at, &len, 4);
    std::memcpy(dst + at + 4, s.data(), s.size());
    at += 4 + s.size();
  }
  *total = at;
  return true;
}

int FakeSubmit(void* pipe, int op, int n, int count, const uint64_t* in_ptrs, const uint64_t* in_bytes,
               const uint64_t* out_ptrs, const uint64_t* out_caps, void*) {
  auto* p = static_cast<FakePipe*>(pipe);
  p->n = n;
  p->codes.assign(static_cast<size_t>(n), 0);
  p->produced.assign(2 * static_cast<size_t>(n), 0);
  ++p->batches;
  if (n > p->largest.load()) p->largest.store(n);
  const int arity = op == 0 ? 2 : 1;
  for (int i = 0; i < n; ++i) {
    std::vector<std::string> in[2];
    bool ok = true;
    for (int k = 0; k < arity; ++k)
      ok = ok && WalkStream(reinterpret_cast<const uint8_t*>(in_ptrs[i * arity + k]), in_bytes[i * arity + k], count, &in[k]);
    if (!ok) {
      p->codes[static_cast<size_t>(i)] = 1;
      continue;
    }
    std::vector<std::string> out[2];
    if (op == 0) {
      for (int e = 0; e < count; ++e) {
        const long long x = std::stoll(in[0][static_cast<size_t>(e)]), y = std::stoll(in[1][static_cast<size_t>(e)]);
        out[0].push_back(std::to_string(x + y));
        out[1].push_back(std::to_string(x - y));
      }
    } else {
      out[0] = in[0];
    }
    for (int k = 0; k < arity; ++k)
      if (!PackStream(out[k], reinterpret_cast<uint8_t*>(out_ptrs[i * arity + k]), out_caps[i * arity + k],
                      &p->produced[2 * static_cast<size_t>(i) + static_cast<size_t>(k)]))
        p->codes[static_cast<size_t>(i)] = 3;
  }
  return 0;
}

int FakeWait(void* pipe, int n, uint32_t* codes, uint64_t* out_bytes) {
  auto* p = static_cast<FakePipe*>(pipe);
  if (n != p->n) return -1;
  std::this_thread::sleep_for(std::chrono::microseconds(30));  // the length of a real batch
  for (int i = 0; i < n; ++i) {
    codes[i] = p->codes[static_cast<size_t>(i)];
    out_bytes[2 * i] = p->produced[2 * static_cast<size_t>(i)];
    out_bytes[2 * i + 1] = p->produced[2 * static_cast<size_t>(i) + 1];
  }
  return 0;
}

std::string StringRequest(const std::string& region, int64_t in_size, int64_t slot, int64_t out_cap, bool name_outputs = true) {
  std::string out;
  pb::Writer w(&out);
  w.String(1, "simple_string");
  w.String(3, "s-1");
  for (int k = 0; k < 2; ++k) {
    std::string t, shape;
    pb::Writer tw(&t), sw(&shape);
    tw.String(1, k == 0 ? "INPUT0" : "INPUT1");
    tw.String(2, "BYTES");
    sw.Varint(1);
    sw.Varint(16);
    tw.Message(3, shape);
    Param(&tw, 4, "shared_memory_region", &region, 0);
    Param(&tw, 4, "shared_memory_byte_size", nullptr, in_size);
    Param(&tw, 4, "shared_memory_offset", nullptr, slot + 256 * k);
    w.Message(5, t);
  }
  for (int k = 0; name_outputs && k < 2; ++k) {
    std::string t;
    pb::Writer tw(&t);
    tw.String(1, k == 0 ? "OUTPUT0" : "OUTPUT1");
    Param(&tw, 2, "shared_memory_region", &region, 0);
    Param(&tw, 2, "shared_memory_byte_size", nullptr, out_cap);
    Param(&tw, 2, "shared_memory_offset", nullptr, slot + 512 + 256 * k);
    w.Message(6, t);
  }
  return out;
}

// Worker threads queue `simple_string` requests whose tensors sit in a registered "device" region while another
// thread unregisters and re-registers the region; every answer is checked, every refusal must hand the request back.
void DeviceBytesPath() {
  NativeExecutor ex;
  FakePipe fake;
  NativeExecutor::DevicePipe pipe;
  pipe.submit = FakeSubmit;
  pipe.wait = FakeWait;
  pipe.pipe = &fake;
  pipe.max_requests = 8;
  pipe.max_count = 4096;
  pipe.max_stream_bytes = 1 << 20;
  NativeExecutor::ModelSpec m;
  m.name = "simple_string";
  m.version = "1";
  m.max_batch = 8;
  m.kernel = "bytes_add_sub";
  m.inputs = {{"INPUT0", "BYTES", {16}}, {"INPUT1", "BYTES", {16}}};
  m.outputs = {{"OUTPUT0", "BYTES", {16}}, {"OUTPUT1", "BYTES", {16}}};
  if (!ex.SetModel(m).IsOk()) Fail("device bytes: registering the model");
  std::string probe = StringRequest("dev", 10, 0, 256), response;
  if (ex.TryGrpcAsync(&probe, &response, [](bool, std::string&&) {}) != NativeExecutor::Outcome::kDeclined)
    Fail("device bytes: accepted without a pipeline");
  if (!ex.EnableDeviceBytes(pipe).IsOk()) Fail("device bytes: enabling");
  constexpr int kThreads = 4;
  constexpr size_t kSlot = 1024;
  std::vector<uint8_t> memory(kThreads * kSlot);
  if (!ex.RegisterDeviceRegion("dev", reinterpret_cast<uint64_t>(memory.data()), memory.size(), 0).IsOk())
    Fail("device bytes: registering the region");
  {
    // a stream too short for 16 elements: the device notices, and the request comes back for the general path
    const std::string copy = probe;
    std::atomic<int> verdict{-1};
    std::string returned;
    if (ex.TryGrpcAsync(&probe, &response, [&](bool ok, std::string&& p) {
          returned = std::move(p);
          verdict.store(ok ? 1 : 0);
        }) != NativeExecutor::Outcome::kDeferred)
      Fail("device bytes: a well-formed request for registered regions was not queued");
    ex.Quiesce();
    if (verdict.load() != 0 || returned != copy) Fail("device bytes: a malformed stream was not handed back");
  }
  std::atomic<bool> stop{false};
  std::atomic<long> answered{0}, refused{0}, declined{0};
  std::vector<std::thread> workers;
  for (int t = 0; t < kThreads; ++t) {
    workers.emplace_back([&, t] {
      uint8_t* slot = memory.data() + static_cast<size_t>(t) * kSlot;
      std::mutex mu;
      std::condition_variable cv;
      for (int round = 0; !stop.load(); ++round) {
        std::vector<std::string> a, b;
        for (int e = 0; e < 16; ++e) {
          a.push_back(std::to_string(round * 31 + e - 7 * t));
          b.push_back(std::to_string(e * 3 - round));
        }
        uint64_t size_a = 0, size_b = 0;
        PackStream(a, slot, 256, &size_a);
        PackStream(b, slot + 256, 256, &size_b);
        if (size_a != size_b) {  // one byte_size field serves both inputs in this request shape: pad by choosing equal widths
          continue;
        }
        const bool small = round % 7 == 3;  // output range too small: the device refuses, the request comes back
        std::string request = StringRequest("dev", static_cast<int64_t>(size_a), static_cast<int64_t>(t) * kSlot, small ? 8 : 256);
        const std::string copy = request;
        bool done = false, ok = false;
        std::string payload;
        const auto outcome = ex.TryGrpcAsync(&request, &response, [&](bool good, std::string&& p) {
          std::lock_guard<std::mutex> lk(mu);
          ok = good;
          payload = std::move(p);
          done = true;
          cv.notify_one();
        });
        if (outcome == NativeExecutor::Outcome::kDeclined) {
          if (request != copy) Fail("device bytes: a declined request was consumed");
          ++declined;
          continue;
        }
        if (outcome != NativeExecutor::Outcome::kDeferred) {
          Fail("device bytes: answered inline");
          continue;
        }
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return done; });
        }
        if (!ok) {
          if (!small || payload != copy) Fail("device bytes: refusal without cause, or without the request");
          ++refused;
          continue;
        }
        if (small) Fail("device bytes: a result that does not fit was reported as written");
        std::vector<std::string> sum, diff;
        uint64_t want_sum = 0, want_diff = 0;
        std::vector<std::string> es, ed;
        for (int e = 0; e < 16; ++e) {
          es.push_back(std::to_string(std::stoll(a[static_cast<size_t>(e)]) + std::stoll(b[static_cast<size_t>(e)])));
          ed.push_back(std::to_string(std::stoll(a[static_cast<size_t>(e)]) - std::stoll(b[static_cast<size_t>(e)])));
        }
        std::vector<uint8_t> expect(512);
        PackStream(es, expect.data(), 256, &want_sum);
        PackStream(ed, expect.data() + 256, 256, &want_diff);
        if (std::memcmp(slot + 512, expect.data(), want_sum) != 0 || std::memcmp(slot + 768, expect.data() + 256, want_diff) != 0)
          Fail("device bytes: wrong result");
        if (payload.find("simple_string") == std::string::npos || payload.find("OUTPUT1") == std::string::npos)
          Fail("device bytes: response without its model or outputs");
        ++answered;
      }
    });
  }
  std::thread churn([&] {
    while (!stop.load()) {
      std::this_thread::sleep_for(std::chrono::milliseconds(3));
      if (!ex.UnregisterDeviceRegion("dev", 5000)) Fail("device bytes: the region never became idle");
      std::this_thread::sleep_for(std::chrono::microseconds(300));
      if (!ex.RegisterDeviceRegion("dev", reinterpret_cast<uint64_t>(memory.data()), memory.size(), 0).IsOk())
        Fail("device bytes: registering the region again");
    }
  });
  std::this_thread::sleep_for(std::chrono::milliseconds(1500));
  stop.store(true);
  for (auto& w : workers) w.join();
  churn.join();
  ex.Quiesce();
  NativeExecutor::Stats st;
  if (!ex.TakeStats("simple_string", &st) || st.requests != static_cast<uint64_t>(answered.load()))
    Fail("device bytes: statistics do not match the answered requests");
  {
    // the HTTP form of the same request: answered with a JSON header that echoes the output regions, or handed back
    auto http_body = [&](int64_t in_size, int64_t out_cap) {
      std::string b = "{\"id\":\"h1\",\"inputs\":[";
      for (int k = 0; k < 2; ++k) {
        if (k) b += ",";
        b += "{\"name\":\"INPUT" + std::to_string(k) + "\",\"datatype\":\"BYTES\",\"shape\":[1,16],\"parameters\":{"
             "\"shared_memory_region\":\"dev\",\"shared_memory_byte_size\":" + std::to_string(in_size) +
             ",\"shared_memory_offset\":" + std::to_string(256 * k) + "}}";
      }
      b += "],\"outputs\":[";
      for (int k = 0; k < 2; ++k) {
        if (k) b += ",";
        b += "{\"name\":\"OUTPUT" + std::to_string(k) + "\",\"parameters\":{\"shared_memory_region\":\"dev\","
             "\"shared_memory_byte_size\":" + std::to_string(out_cap) + ",\"shared_memory_offset\":" +
             std::to_string(512 + 256 * k) + "}}";
      }
      return b + "]}";
    };
    std::vector<std::string> items;
    for (int e = 0; e < 16; ++e) items.push_back(std::to_string(e + 1));
    uint64_t size = 0;
    PackStream(items, memory.data(), 256, &size);
    PackStream(items, memory.data() + 256, 256, &size);
    if (!ex.Defers("simple_string")) Fail("device bytes: the model is not reported as deferring");
    for (const bool small : {false, true}) {
      std::string body = http_body(static_cast<int64_t>(size), small ? 8 : 256);
      const std::string copy = body;
      NativeExecutor::HeaderList headers;
      std::vector<std::string> parts, got;
      std::atomic<int> verdict{-1};
      if (ex.TryHttpAsync("simple_string", "", &body, -1, &headers, &parts,
                          [&](bool ok, NativeExecutor::HeaderList&&, std::vector<std::string>&& p) {
                            got = std::move(p);
                            verdict.store(ok ? 1 : 0);
                          }) != NativeExecutor::Outcome::kDeferred)
        Fail("device bytes: the HTTP request was not queued");
      ex.Quiesce();
      if (small) {
        if (verdict.load() != 0 || got.size() != 1 || got[0] != copy) Fail("device bytes: HTTP refusal without the body");
      } else if (verdict.load() != 1 || got.size() != 1 || got[0].find("\"shared_memory_region\":\"dev\"") == std::string::npos ||
                 got[0].find("\"id\":\"h1\"") == std::string::npos) {
        Fail("device bytes: HTTP answer");
      } else {
        std::vector<std::string> doubled;
        for (int e = 0; e < 16; ++e) doubled.push_back(std::to_string(2 * (e + 1)));
        std::vector<uint8_t> expect(256);
        uint64_t want = 0;
        PackStream(doubled, expect.data(), 256, &want);
        if (std::memcmp(memory.data() + 512, expect.data(), want) != 0) Fail("device bytes: HTTP result");
      }
    }
  }
  std::string unnamed = StringRequest("dev", 40, 0, 256, false);
  if (ex.TryGrpcAsync(&unnamed, &response, [](bool, std::string&&) {}) != NativeExecutor::Outcome::kDeclined)
    Fail("device bytes: outputs without regions must take the general path");
  ex.DisableDeviceBytes();
  std::string late = StringRequest("dev", 40, 0, 256);
  if (ex.TryGrpcAsync(&late, &response, [](bool, std::string&&) {}) != NativeExecutor::Outcome::kDeclined)
    Fail("device bytes: accepted after the pipeline was disabled");
  std::cout << "device bytes: answered " << answered.load() << ", refused by the device " << refused.load() << ", declined "
            << declined.load() << ", batches " << fake.batches.load() << " (largest " << fake.largest.load() << ")" << std::endl;
  if (answered.load() == 0 || refused.load() == 0) Fail("device bytes: nothing was answered or nothing refused");
}


// ---- host stand-ins for the HIP side of a graph model --------------------------------------------------
// A stream is a FIFO of closures run by a worker thread, an event a flag the FIFO sets, a graph a function
// over its two static buffers: out[row] = {sum of the row's 4 inputs + 1, 2 x first input}.
struct FakeStream {
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::function<void()>> work;
  bool stop = false;
  std::thread th;
  FakeStream() : th([this] { Run(); }) {}
  ~FakeStream() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv.notify_all();
    th.join();
  }
  void Push(std::function<void()> fn) {
    {
      std::lock_guard<std::mutex> lk(mu);
      work.push_back(std::move(fn));
    }
    cv.notify_all();
  }
  void Run() {
    for (;;) {
      std::function<void()> fn;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return stop || !work.empty(); });
        if (work.empty()) return;
        fn = std::move(work.front());
        work.pop_front();
      }
      fn();
    }
  }
};
struct FakeEvent {
  std::mutex mu;
  std::condition_variable cv;
  uint64_t recorded = 0, reached = 0;
};
struct FakeGraph {
  int batch = 0;
  std::vector<float> in, out;  // [batch][4], [batch][2]
  std::atomic<long> launches{0};
};
std::atomic<int> g_fail_launches{0};  // > 0: the next graph launches report an error

int FakeGraphLaunch(void* exec, void* stream) {
  if (g_fail_launches.load() > 0 && g_fail_launches.fetch_sub(1) > 0) return -5;
  auto* g = static_cast<FakeGraph*>(exec);
  static_cast<FakeStream*>(stream)->Push([g] {
    std::this_thread::sleep_for(std::chrono::microseconds(40));  // a forward pass takes a while
    for (int r = 0; r < g->batch; ++r) {
      const float* x = &g->in[static_cast<size_t>(r) * 4];
      g->out[static_cast<size_t>(r) * 2] = x[0] + x[1] + x[2] + x[3] + 1.0f;
      g->out[static_cast<size_t>(r) * 2 + 1] = 2.0f * x[0];
    }
    ++g->launches;
  });
  return 0;
}
int FakeEventCreate(void** event) {
  *event = new FakeEvent();
  return 0;
}
int FakeEventRecord(void* event, void* stream) {
  auto* e = static_cast<FakeEvent*>(event);
  uint64_t ticket;
  {
    std::lock_guard<std::mutex> lk(e->mu);
    ticket = ++e->recorded;
  }
  static_cast<FakeStream*>(stream)->Push([e, ticket] {
    std::lock_guard<std::mutex> lk(e->mu);
    e->reached = ticket;
    e->cv.notify_all();
  });
  return 0;
}
int FakeEventSync(void* event) {
  auto* e = static_cast<FakeEvent*>(event);
  std::unique_lock<std::mutex> lk(e->mu);
  const uint64_t want = e->recorded;
  e->cv.wait(lk, [&] { return e->reached >= want; });
  return 0;
}
// the staging thread's periodic drain: by contract the stream has nothing queued when it is called (it follows
// a wait for the event behind the round's copies)
std::atomic<int> g_stream_syncs{0};
std::atomic<int> g_stream_syncs_busy{0};
int FakeStreamSync(void* stream) {
  auto* s = static_cast<FakeStream*>(stream);
  std::lock_guard<std::mutex> lk(s->mu);
  if (!s->work.empty()) g_stream_syncs_busy.fetch_add(1);
  g_stream_syncs.fetch_add(1);
  return 0;
}
int FakeEventDestroy(void* event) {
  delete static_cast<FakeEvent*>(event);
  return 0;
}
int FakeBatchedCopy(const void* const* srcs, void* const* dsts, const uint64_t* bytes, int n, void* stream) {
  struct Triple {
    const void* src;
    void* dst;
    uint64_t bytes;
  };
  std::vector<Triple> items;
  for (int i = 0; i < n; ++i) items.push_back({srcs[i], dsts[i], bytes[i]});  // the caller reuses its arrays
  static_cast<FakeStream*>(stream)->Push([items] {
    for (const auto& t : items) std::memcpy(t.dst, t.src, t.bytes);
  });
  return 0;
}

// Tensors inside the message: `in` travels as raw contents; the output comes back inside the response
// (`out_region` empty) or goes to the region.
std::string GraphWireRequest(const float* in, int batch, const std::string& out_region, int64_t out_off, int64_t out_size) {
  std::string out;
  pb::Writer w(&out);
  w.String(1, "classifier");
  w.String(3, "g-7");
  std::string t, shape;
  pb::Writer tw(&t), sw(&shape);
  tw.String(1, "input");
  tw.String(2, "FP32");
  sw.Varint(static_cast<uint64_t>(batch));
  sw.Varint(4);
  tw.Message(3, shape);
  w.Message(5, t);
  if (!out_region.empty()) {
    std::string o;
    pb::Writer ow(&o);
    ow.String(1, "output");
    Param(&ow, 2, "shared_memory_region", &out_region, 0);
    Param(&ow, 2, "shared_memory_byte_size", nullptr, out_size);
    Param(&ow, 2, "shared_memory_offset", nullptr, out_off);
    w.Message(6, o);
  }
  w.Bytes(7, in, static_cast<size_t>(batch) * 16);
  return out;
}

std::string GraphRequest(const std::string& region, int batch, int64_t in_off, int64_t out_off, int64_t out_size,
                         bool name_output = true) {
  std::string out;
  pb::Writer w(&out);
  w.String(1, "classifier");
  w.String(3, "g-7");
  std::string t, shape;
  pb::Writer tw(&t), sw(&shape);
  tw.String(1, "input");
  tw.String(2, "FP32");
  sw.Varint(static_cast<uint64_t>(batch));
  sw.Varint(4);
  tw.Message(3, shape);
  Param(&tw, 4, "shared_memory_region", &region, 0);
  Param(&tw, 4, "shared_memory_byte_size", nullptr, static_cast<int64_t>(batch) * 16);
  if (in_off) Param(&tw, 4, "shared_memory_offset", nullptr, in_off);
  w.Message(5, t);
  if (name_output) {
    std::string o;
    pb::Writer ow(&o);
    ow.String(1, "output");
    Param(&ow, 2, "shared_memory_region", &region, 0);
    Param(&ow, 2, "shared_memory_byte_size", nullptr, out_size);
    Param(&ow, 2, "shared_memory_offset", nullptr, out_off);
    w.Message(6, o);
  }
  return out;
}

// Worker threads keep requests of batch 1..8 in flight against a model with graphs for 1, 2, 4, 8, 16 on two
// instances while the region is unregistered and re-registered; every answer is checked row by row, every
// refusal hands the request back, statistics count requests, batches and padding.
void GraphPath() {
  NativeExecutor ex;
  NativeExecutor::ModelSpec m;
  m.name = "classifier";
  m.version = "1";
  m.max_batch = 16;
  m.kernel = "graph";
  m.inputs = {{"input", "FP32", {4}}};
  m.outputs = {{"output", "FP32", {2}}};
  if (!ex.SetModel(m).IsOk()) Fail("graph: registering the model");
  constexpr int kThreads = 6;
  constexpr size_t kSlot = 512;  // inputs at 0 (8 x 16 bytes), outputs at 256 (8 x 8 bytes)
  std::vector<uint8_t> memory(kThreads * kSlot + 64);
  uint8_t* base = memory.data() + (16 - reinterpret_cast<uintptr_t>(memory.data()) % 16) % 16;
  std::string probe = GraphRequest("dev", 2, 0, 256, 64), response;
  if (ex.TryGrpcAsync(&probe, &response, [](bool, std::string&&) {}) != NativeExecutor::Outcome::kDeclined)
    Fail("graph: accepted without graphs");
  FakeStream streams[2];
  std::vector<std::unique_ptr<FakeGraph>> graphs;
  NativeExecutor::GraphPipe pipe;
  pipe.graph_launch = FakeGraphLaunch;
  pipe.event_create = FakeEventCreate;
  pipe.event_record = FakeEventRecord;
  pipe.event_sync = FakeEventSync;
  pipe.event_destroy = FakeEventDestroy;
  pipe.batched_copy = FakeBatchedCopy;
  pipe.stream_sync = FakeStreamSync;
  pipe.streams = {&streams[0], &streams[1]};
  pipe.in_row_bytes = 16;
  pipe.out_row_bytes = 8;
  pipe.max_batch = 16;
  pipe.fill_window_us = 200;
  pipe.split_min_batch = 4;
  // "page-locked" memory for tensors inside messages: 5 pieces of 256 bytes - too few for six threads, so
  // some requests find it taken and take the general path
  std::vector<uint8_t> arena(5 * 256 + 256);
  pipe.host_arena = (reinterpret_cast<uintptr_t>(arena.data()) + 255) / 256 * 256;
  pipe.host_arena_bytes = 5 * 256;
  pipe.host_chunk_bytes = 256;
  // ... moved on to "device" memory by the staging thread on a stream of its own
  std::vector<uint8_t> device_arena(5 * 256 + 256);
  FakeStream copy_stream;
  pipe.device_arena = (reinterpret_cast<uintptr_t>(device_arena.data()) + 255) / 256 * 256;
  pipe.copy_stream = &copy_stream;
  for (int inst = 0; inst < 2; ++inst)
    for (int b : {1, 2, 4, 8, 16}) {
      graphs.emplace_back(new FakeGraph());
      FakeGraph* g = graphs.back().get();
      g->batch = b;
      g->in.assign(static_cast<size_t>(b) * 4, -1000.f);
      g->out.assign(static_cast<size_t>(b) * 2, 0.f);
      NativeExecutor::GraphPipe::Graph e;
      e.instance = inst;
      e.batch = b;
      e.exec = g;
      e.in_ptr = reinterpret_cast<uint64_t>(g->in.data());
      e.out_ptr = reinterpret_cast<uint64_t>(g->out.data());
      pipe.graphs.push_back(e);
    }
  {
    NativeExecutor::GraphPipe small = pipe;
    small.graphs.pop_back();  // instance 1 loses its largest graph
    if (ex.EnableGraphModel("classifier", small).IsOk()) Fail("graph: accepted an instance without a graph for the largest batch");
    if (ex.EnableGraphModel("nobody", pipe).IsOk()) Fail("graph: accepted a model that is not registered");
  }
  if (!ex.EnableGraphModel("classifier", pipe).IsOk()) Fail("graph: enabling");
  if (!ex.Defers("classifier")) Fail("graph: the model is not reported as deferring");
  if (ex.TryGrpcAsync(&probe, &response, [](bool, std::string&&) {}) != NativeExecutor::Outcome::kDeclined)
    Fail("graph: accepted a request for a region that is not registered");
  if (!ex.RegisterDeviceRegion("dev", reinterpret_cast<uint64_t>(base), kThreads * kSlot, 0).IsOk())
    Fail("graph: registering the region");
  {
    const float two_rows[12] = {0};
    std::string short_raw = GraphWireRequest(two_rows, 3, "", 0, 0);  // says 3 rows ...
    short_raw.resize(short_raw.size() - 16);                              // ... (and the length prefix is off now)
    if (ex.TryGrpcAsync(&short_raw, &response, [](bool, std::string&&) {}) != NativeExecutor::Outcome::kDeclined)
      Fail("graph: a damaged message must take the general path");
    std::string tiny = GraphRequest("dev", 4, 0, 256, 16);
    if (ex.TryGrpcAsync(&tiny, &response, [](bool, std::string&&) {}) != NativeExecutor::Outcome::kDeclined)
      Fail("graph: an output range that cannot hold the result must take the general path");
    std::string odd = GraphRequest("dev", 1, 4, 256, 64);
    if (ex.TryGrpcAsync(&odd, &response, [](bool, std::string&&) {}) != NativeExecutor::Outcome::kDeclined)
      Fail("graph: a misaligned input must take the general path");
    std::string big = GraphRequest("dev", 17, 0, 256, 4096);
    if (ex.TryGrpcAsync(&big, &response, [](bool, std::string&&) {}) != NativeExecutor::Outcome::kDeclined)
      Fail("graph: a batch beyond the model's limit must take the general path");
  }
  // a system shared-memory region with the same per-thread windows
  const char* sys_key = "/tca_native_graph_test";
  shm_unlink(sys_key);
  const int sys_fd = shm_open(sys_key, O_CREAT | O_RDWR, 0600);
  if (sys_fd < 0 || ftruncate(sys_fd, kThreads * kSlot) != 0) Fail("graph: creating the system region");
  uint8_t* sys_base = static_cast<uint8_t*>(mmap(nullptr, kThreads * kSlot, PROT_READ | PROT_WRITE, MAP_SHARED, sys_fd, 0));
  if (sys_base == MAP_FAILED || !ex.RegisterRegion("sys", sys_key, 0, kThreads * kSlot).IsOk()) Fail("graph: registering the system region");
  std::atomic<bool> stop{false};
  std::atomic<long> answered{0}, refused{0}, declined{0}, images{0}, wire_answers{0}, sys_answers{0};
  std::vector<std::thread> workers;
  for (int t = 0; t < kThreads; ++t) {
    workers.emplace_back([&, t] {
      uint8_t* slot = base + static_cast<size_t>(t) * kSlot;
      std::mutex mu;
      std::condition_variable cv;
      std::string resp;
      for (int round = 0; !stop.load(); ++round) {
        const int batch = 1 + (round * 5 + t) % 8;
        float in[8 * 4];
        for (int i = 0; i < batch * 4; ++i) in[i] = static_cast<float>((round * 131 + i * 7 + t * 1009) % 4001) - 2000.f;
        std::memcpy(slot, in, static_cast<size_t>(batch) * 16);
        std::memset(slot + 256, 0xEE, 64);
        // a third of the requests carry the input inside the message, half of those the output too
        // ... and every fifth request has both tensors in the system region
        const int form = round % 5 == 4 ? 3 : round % 3 == 1 ? 1 + (round / 3) % 2 : 0;
        uint8_t* sys_slot = sys_base + static_cast<size_t>(t) * kSlot;
        if (form == 3) {
          std::memcpy(sys_slot, in, static_cast<size_t>(batch) * 16);
          std::memset(sys_slot + 256, 0xEE, 64);
        }
        std::string request =
            form == 0 ? GraphRequest("dev", batch, static_cast<int64_t>(t) * kSlot, static_cast<int64_t>(t) * kSlot + 256, 64)
            : form == 3 ? GraphRequest("sys", batch, static_cast<int64_t>(t) * kSlot, static_cast<int64_t>(t) * kSlot + 256, 64)
                        : GraphWireRequest(in, batch, form == 1 ? "" : "dev", static_cast<int64_t>(t) * kSlot + 256, 64);
        const std::string copy = request;
        bool done = false, ok = false;
        std::string payload;
        const auto outcome = ex.TryGrpcAsync(&request, &resp, [&](bool good, std::string&& p) {
          std::lock_guard<std::mutex> lk(mu);
          ok = good;
          payload = std::move(p);
          done = true;
          cv.notify_one();
        });
        if (outcome == NativeExecutor::Outcome::kDeclined) {
          if (request != copy) Fail("graph: a declined request was consumed");
          ++declined;
          continue;
        }
        if (outcome != NativeExecutor::Outcome::kDeferred) {
          Fail("graph: answered inline");
          continue;
        }
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return done; });
        }
        if (!ok) {
          if (payload != copy) Fail("graph: a refusal without the request");
          ++refused;
          continue;
        }
        float out[16];
        if (form == 1) {
          const auto raws = RawOutputs(payload);
          if (raws.size() != 1 || raws[0].size() != static_cast<size_t>(batch) * 8) {
            Fail("graph: a response without its output rows");
            continue;
          }
          std::memcpy(out, raws[0].data(), raws[0].size());
          ++wire_answers;
        } else if (form == 3) {
          std::memcpy(out, sys_slot + 256, static_cast<size_t>(batch) * 8);
          ++sys_answers;
        } else {
          std::memcpy(out, slot + 256, static_cast<size_t>(batch) * 8);
        }
        for (int r = 0; r < batch; ++r) {
          const float* x = in + r * 4;
          if (out[2 * r] != x[0] + x[1] + x[2] + x[3] + 1.0f || out[2 * r + 1] != 2.0f * x[0]) {
            Fail("graph: wrong row (merged batches must keep every request's rows apart)");
            break;
          }
        }
        if (payload.find("classifier") == std::string::npos || payload.find("output") == std::string::npos ||
            payload.find("g-7") == std::string::npos)
          Fail("graph: response without its model, output or id");
        ++answered;
        images += batch;
      }
    });
  }
  std::thread churn([&] {
    int round = 0;
    while (!stop.load()) {
      std::this_thread::sleep_for(std::chrono::milliseconds(4));
      if (!ex.UnregisterDeviceRegion("dev", 5000)) Fail("graph: the region never became idle");
      std::this_thread::sleep_for(std::chrono::microseconds(300));
      if (!ex.RegisterDeviceRegion("dev", reinterpret_cast<uint64_t>(base), kThreads * kSlot, 0).IsOk())
        Fail("graph: registering the region again");
      if (++round % 16 == 0) g_fail_launches.store(2);  // two launches fail: their requests come back
    }
  });
  std::this_thread::sleep_for(std::chrono::milliseconds(1500));
  stop.store(true);
  for (auto& w : workers) w.join();
  churn.join();
  ex.Quiesce();
  NativeExecutor::GraphCounters counters;
  if (!ex.GraphStats("classifier", &counters)) Fail("graph: no counters");
  NativeExecutor::Stats st;
  if (!ex.TakeStats("classifier", &st) || st.inferences != static_cast<uint64_t>(images.load()))
    Fail("graph: statistics do not match the answered requests");
  if (counters.requests < static_cast<uint64_t>(answered.load()) || counters.batches == 0 || counters.batches > counters.requests)
    Fail("graph: batch counters out of range");
  long launches = 0;
  for (const auto& g : graphs) launches += g->launches.load();
  if (static_cast<uint64_t>(launches) != counters.batches) Fail("graph: launches counted do not match the graphs' own count");
  if (sys_answers.load() == 0) Fail("graph: no request with tensors in system shared memory was served");
  ex.UnregisterRegion("sys");
  munmap(sys_base, kThreads * kSlot);
  close(sys_fd);
  shm_unlink(sys_key);
  if (wire_answers.load() == 0 || counters.wire_requests < static_cast<uint64_t>(wire_answers.load()) || counters.arena_full == 0)
    Fail("graph: requests with tensors inside the messages were not served, or the staging memory never ran out");
  if (g_stream_syncs.load() == 0 || g_stream_syncs_busy.load() != 0)
    Fail("graph: the staging thread never drained its copy stream, or drained it with copies still queued");
  // a request queued while the model goes away is answered or handed back, never dropped
  {
    std::atomic<int> verdicts{0};
    std::memset(base, 0, 64);
    for (int i = 0; i < 20; ++i) {
      // every request has an output range of its own: two instances may scatter at the same time
      const int64_t slot = static_cast<int64_t>(i % kThreads) * kSlot;
      std::string request = GraphRequest("dev", 1, slot, slot + 256 + 8 * (i / kThreads), 8);
      if (ex.TryGrpcAsync(&request, &response, [&](bool, std::string&&) { ++verdicts; }) != NativeExecutor::Outcome::kDeferred)
        ++verdicts;
    }
    ex.DisableGraphModel("classifier");
    if (verdicts.load() != 20) Fail("graph: requests were dropped when the model was disabled");
  }
  std::string late = GraphRequest("dev", 1, 0, 256, 64);
  if (ex.TryGrpcAsync(&late, &response, [](bool, std::string&&) {}) != NativeExecutor::Outcome::kDeclined)
    Fail("graph: accepted after the model was disabled");
  std::cout << "graph model: answered " << answered.load() << " requests (" << images.load() << " batch entries) in "
            << counters.batches << " batches, " << counters.padded_images << " padded rows, refused " << refused.load()
            << ", declined " << declined.load() << std::endl;
  if (answered.load() == 0 || counters.padded_images == 0 || refused.load() == 0) Fail("graph: nothing answered, padded or refused");
}

// Stateful kernel: threads run whole sequences through the gRPC path while another one walks sequences of its
// own through SequenceStep (what the Python core does for requests that took the general path).
std::string SequenceRequest(const char* model, int32_t value, uint64_t id, bool start, bool end) {
  std::string out;
  pb::Writer w(&out);
  w.String(1, model);
  auto param = [&](const char* key, bool is_bool, uint64_t v) {
    std::string entry, value_msg;
    pb::Writer ew(&entry), vw(&value_msg);
    if (is_bool) vw.Bool(1, v != 0);
    else vw.Int(2, static_cast<int64_t>(v));
    ew.String(1, key);
    ew.Message(2, value_msg);
    w.Message(4, entry);
  };
  param("sequence_id", false, id);
  if (start) param("sequence_start", true, 1);
  if (end) param("sequence_end", true, 1);
  std::string t, shape;
  pb::Writer tw(&t), sw(&shape);
  tw.String(1, "INPUT");
  tw.String(2, "INT32");
  sw.Varint(1);
  sw.Varint(1);
  tw.Message(3, shape);
  w.Message(5, t);
  w.Bytes(7, &value, 4);
  return out;
}

void SequencePath() {
  NativeExecutor ex;
  NativeExecutor::ModelSpec m;
  m.name = "simple_dyna_sequence";
  m.version = "1";
  m.max_batch = 1;
  m.kernel = "dyna_sequence";
  m.inputs = {{"INPUT", "INT32", {1}}};
  m.outputs = {{"OUTPUT", "INT32", {1}}};
  if (!ex.SetModel(m).IsOk()) Fail("sequence: registering the model");
  std::string response;
  if (ex.TryGrpc(SequenceRequest("simple_dyna_sequence", 1, 5, false, false), &response)) Fail("sequence: answered without a start");
  if (ex.TryGrpc(SequenceRequest("simple_dyna_sequence", 1, 0, true, false), &response)) Fail("sequence: id 0 accepted");
  std::vector<std::thread> threads;
  std::atomic<long> steps{0};
  for (int t = 0; t < 4; ++t) {
    threads.emplace_back([&, t] {
      std::string resp;
      for (int round = 0; round < 2000; ++round) {
        const uint64_t id = 1000 + static_cast<uint64_t>(t) * 100000 + static_cast<uint64_t>(round);
        for (int k = 0; k < 4; ++k) {
          const bool start = k == 0, end = k == 3;
          if (t == 3) {  // the general path's view of the same table
            if (!ex.SequenceStep("simple_dyna_sequence", id, start, end)) Fail("sequence: step refused");
            continue;
          }
          if (!ex.TryGrpc(SequenceRequest("simple_dyna_sequence", 10 * k, id, start, end), &resp)) {
            Fail("sequence: request declined");
            continue;
          }
          auto raws = RawOutputs(resp);
          int32_t got = 0;
          if (raws.size() == 1 && raws[0].size() == 4) std::memcpy(&got, raws[0].data(), 4);
          const int32_t want = 10 * k + (start ? 1 : 0) + (end ? static_cast<int32_t>(id) : 0);
          if (got != want) Fail("sequence: wrong value");
          ++steps;
        }
      }
    });
  }
  for (auto& t : threads) t.join();
  if (ex.OpenSequences("simple_dyna_sequence") != 0) Fail("sequence: sequences left open");
  std::cout << "sequences: " << steps.load() << " requests answered" << std::endl;
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
  {
    // mutated requests (the parsers read bytes from the network): flipped bits, overwritten and inserted bytes.
    // Whatever comes out - an answer or a refusal - no access may leave the message (the memory-error
    // detectors watch this loop).
    uint64_t rng = 0x9e3779b97f4a7c15ull;
    auto next = [&rng]() {
      rng ^= rng << 13;
      rng ^= rng >> 7;
      rng ^= rng << 17;
      return rng;
    };
    const std::string seeds[3] = {GrpcRequest(a, b, 1, false), GrpcRequest(a, b, 1, true),
                                  SequenceRequest("simple", 3, 9, true, false)};
    const std::string http_seed =
        "{\"id\":\"7\",\"inputs\":[{\"name\":\"INPUT0\",\"datatype\":\"INT32\",\"shape\":[1,16],\"parameters\":{\"binary_data_size\":64}},"
        "{\"name\":\"INPUT1\",\"datatype\":\"INT32\",\"shape\":[1,16],\"parameters\":{\"binary_data_size\":64}}],"
        "\"outputs\":[{\"name\":\"OUTPUT0\",\"parameters\":{\"binary_data\":true}}]}";
    std::vector<std::string> stream;
    for (int round = 0; round < 20000; ++round) {
      std::string m = seeds[next() % 3];
      const int edits = 1 + static_cast<int>(next() % 4);
      for (int k = 0; k < edits && !m.empty(); ++k) {
        const size_t at = next() % m.size();
        switch (next() % 4) {
          case 0: m[at] = static_cast<char>(m[at] ^ (1 << (next() % 8))); break;
          case 1: m[at] = static_cast<char>(next()); break;
          case 2: m.insert(at, 1, static_cast<char>(next())); break;
          default: m.erase(at, 1 + next() % 3); break;
        }
      }
      ex.TryGrpc(m, &response);
      ex.TryGrpcStream(m, &stream);
      std::string body = http_seed;
      const size_t json_len = body.size();
      body.append(128, '\1');
      const size_t at = next() % json_len;
      body[at] = static_cast<char>(next() % 3 == 0 ? next() : body[at] ^ (1 << (next() % 7)));
      ex.TryHttp("simple", "", body, static_cast<int64_t>(next() % 5 == 0 ? next() % 400 : json_len), &headers, &parts);
    }
  }
  if (ex.TryHttp("simple", "", "{\"inputs\":[{\"name\":\"INPUT0\"", -1, &headers, &parts)) Fail("truncated JSON accepted");
  if (ex.TryHttp("simple", "", "{\"inputs\":[]}", 400, &headers, &parts)) Fail("header length past the body accepted");
  munmap(mem, 4096);
  shm_unlink(key);
  DeviceBytesPath();
  SequencePath();
  GraphPath();
  std::cout << "answered " << answered.load() << ", declined " << declined.load() << ", statistics collected for "
            << collected.load() << ", failures " << g_failures.load() << std::endl;
  if (g_failures.load() != 0 || answered.load() == 0) return 1;
  std::cout << "PASS : native executor" << std::endl;
  return 0;
}
