// Native load generator core: keeps `-c N` inference requests outstanding over
// HTTP or gRPC from epoll reactor threads (no interpreter in the loop) and reports
// throughput and latency percentiles as a JSON line.  The Python analyzer uses
// it for high concurrency levels where per-request Python cost would bound the
// offered load (`python -m triton_client_amd.perf --native`).
//
//   tca_loadgen -u host:port -m model [-x version] [-i http|grpc] -c 64 -d 5.0 --warmup 1.0
//       --count N            issue and measure exactly N requests instead of -d seconds
//       --warmup-count W     (with --count) W unmeasured requests first
//       --wait-go            (with --count) print READY after the warm-up and start the
//                            measured phase when a line arrives on stdin
//       --connections M      spread the slots over M clients (gRPC: channels; HTTP: clients with a
//                            reactor thread and a connection pool each) (default 1)
//       --trace FILE         write one CSV row per measured request:
//                            start_ns (since the measured phase began), latency_ns, slot, ok
//       --shape NAME:DATATYPE:d0,d1,...      wire tensors (floating-point types: pseudo-random finite values,
//                                            --wire-data zero for zeros; other types zero filled)
//       --shm-input NAME:DATATYPE:dims:REGION:BYTES[:OFFSET]   registered shm input
//       --shm-output NAME:REGION:BYTES[:OFFSET]
//       --streaming          gRPC: requests travel on one bidirectional stream per connection
//                            (decoupled models: any number of responses per request; a request
//                            is complete at its final response).  Reports responses/sec as well.
//       --sequence-length L  stateful models: every slot sends sequences of L requests (start flag on
//                            the first, end flag on the last, one correlation id per sequence; ids
//                            count up from --sequence-id-base, default 1); a slot whose request
//                            fails starts a new sequence.  A timed run ends every sequence it began.
//       -b batch (only scales the reported infer/sec)
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <mutex>
#include <sstream>
#include <thread>

#include "tca/codec.h"
#include "tca/sync.h"
#include "tca/grpc_client.h"
#include "tca/http_client.h"

namespace tc = tca::client;

namespace {

std::vector<std::string> Split(const std::string& s, char sep) {
  std::vector<std::string> out;
  std::stringstream ss(s);
  std::string item;
  while (std::getline(ss, item, sep)) out.push_back(item);
  return out;
}

std::vector<int64_t> Dims(const std::string& s) {
  std::vector<int64_t> out;
  for (const auto& d : Split(s, ',')) out.push_back(std::atoll(d.c_str()));
  return out;
}

uint64_t NowNs() {
  return static_cast<uint64_t>(
      std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count());
}

struct Slot {
  std::vector<std::unique_ptr<tc::InferInput>> inputs;
  std::vector<std::unique_ptr<tc::InferRequestedOutput>> outputs;
  std::vector<std::vector<uint8_t>> storage;
  std::vector<tc::InferInput*> in_ptrs;
  std::vector<const tc::InferRequestedOutput*> out_ptrs;
  uint64_t t0 = 0;
  // sequences (--sequence-length): position of the request in flight and the sequences this slot has begun
  int seq_pos = 0;
  uint64_t seq_no = 0;
};

}  // namespace

int main(int argc, char** argv) {
  std::string url = "localhost:8000", model, version, protocol = "http";
  int concurrency = 1, batch = 1, connections = 1;
  long long count = 0, warmup_count = 0;
  bool wait_go = false, streaming = false;
  int sequence_length = 0;
  uint64_t sequence_id_base = 1;
  double duration = 5.0, warmup = 1.0;
  std::vector<std::string> shapes, shm_in, shm_out;
  bool wire_random = true;  // --wire-data zero|random
  std::string trace_path;
  for (int i = 1; i < argc; ++i) {
    auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
    if (!std::strcmp(argv[i], "-u")) url = next();
    else if (!std::strcmp(argv[i], "-m")) model = next();
    else if (!std::strcmp(argv[i], "-x")) version = next();
    else if (!std::strcmp(argv[i], "-i")) protocol = next();
    else if (!std::strcmp(argv[i], "--count")) count = std::atoll(next());
    else if (!std::strcmp(argv[i], "--warmup-count")) warmup_count = std::atoll(next());
    else if (!std::strcmp(argv[i], "--wait-go")) wait_go = true;
    else if (!std::strcmp(argv[i], "--connections")) connections = std::max(1, std::atoi(next()));
    else if (!std::strcmp(argv[i], "--trace")) trace_path = next();
    else if (!std::strcmp(argv[i], "--streaming")) streaming = true;
    else if (!std::strcmp(argv[i], "--sequence-length")) sequence_length = std::max(0, std::atoi(next()));
    else if (!std::strcmp(argv[i], "--sequence-id-base")) sequence_id_base = std::max(1ull, std::strtoull(next(), nullptr, 10));
    else if (!std::strcmp(argv[i], "-c")) concurrency = std::atoi(next());
    else if (!std::strcmp(argv[i], "-b")) batch = std::atoi(next());
    else if (!std::strcmp(argv[i], "-d")) duration = std::atof(next());
    else if (!std::strcmp(argv[i], "--warmup")) warmup = std::atof(next());
    else if (!std::strcmp(argv[i], "--shape")) shapes.push_back(next());
    else if (!std::strcmp(argv[i], "--wire-data")) wire_random = std::strcmp(next(), "zero") != 0;
    else if (!std::strcmp(argv[i], "--shm-input")) shm_in.push_back(next());
    else if (!std::strcmp(argv[i], "--shm-output")) shm_out.push_back(next());
    else { std::cerr << "unknown argument " << argv[i] << std::endl; return 2; }
  }
  if (model.empty() || concurrency < 1) { std::cerr << "usage: tca_loadgen -u url -m model -c N ..." << std::endl; return 2; }
  for (auto& ch : protocol) ch = static_cast<char>(std::tolower(static_cast<unsigned char>(ch)));
  const bool use_grpc = protocol == "grpc";
  if (!use_grpc && protocol != "http") { std::cerr << "-i must be http or grpc" << std::endl; return 2; }
  if (streaming && !use_grpc) { std::cerr << "--streaming needs -i grpc" << std::endl; return 2; }
  std::vector<std::unique_ptr<tc::InferenceServerHttpClient>> http_clients;
  std::vector<std::unique_ptr<tc::InferenceServerGrpcClient>> grpc_clients;
  tca::Error e;
  if (use_grpc) {
    for (int k = 0; k < connections; ++k) {
      std::unique_ptr<tc::InferenceServerGrpcClient> g;
      e = tc::InferenceServerGrpcClient::Create(&g, url, false, false, tc::SslOptions(), tc::KeepAliveOptions(),
                                                /*use_cached_channel=*/false);
      if (!e.IsOk()) { std::cerr << e.Message() << std::endl; return 1; }
      grpc_clients.push_back(std::move(g));
    }
  } else {
    for (int k = 0; k < connections; ++k) {
      std::unique_ptr<tc::InferenceServerHttpClient> h;
      e = tc::InferenceServerHttpClient::Create(&h, url);
      if (!e.IsOk()) { std::cerr << e.Message() << std::endl; return 1; }
      http_clients.push_back(std::move(h));
    }
  }

  // one request template per slot; `{slot}` in region names is replaced by the slot index
  std::vector<std::unique_ptr<Slot>> slots;
  for (int s = 0; s < concurrency; ++s) {
    std::unique_ptr<Slot> slot(new Slot());
    auto subst = [&](std::string name) {
      const size_t pos = name.find("{slot}");
      if (pos != std::string::npos) name.replace(pos, 6, std::to_string(s));
      return name;
    };
    // offsets: a number, or "<stride>*slot[+<base>]" for slots that share one region
    auto slot_offset = [&](const std::string& text) -> uint64_t {
      const size_t star = text.find("*slot");
      if (star == std::string::npos) return std::strtoull(text.c_str(), nullptr, 10);
      uint64_t off = std::strtoull(text.substr(0, star).c_str(), nullptr, 10) * static_cast<uint64_t>(s);
      const size_t plus = text.find('+', star);
      if (plus != std::string::npos) off += std::strtoull(text.substr(plus + 1).c_str(), nullptr, 10);
      return off;
    };
    for (const auto& spec : shapes) {
      auto parts = Split(spec, ':');
      if (parts.size() != 3) { std::cerr << "bad --shape " << spec << std::endl; return 2; }
      auto dims = Dims(parts[2]);
      tc::InferInput* in = nullptr;
      e = tc::InferInput::Create(&in, parts[0], dims, parts[1]);
      if (!e.IsOk()) { std::cerr << e.Message() << std::endl; return 1; }
      const int64_t count = tca::ElementCount(dims);
      size_t bytes = static_cast<size_t>(count) * tca::DataTypeByteSize(parts[1]);
      if (parts[1] == "BYTES") {
        // decimal strings "0".."99" (models such as simple_string parse them), <u32 length><bytes> each
        std::vector<uint8_t> packed;
        for (int64_t k = 0; k < count; ++k) {
          const std::string text = std::to_string((k * 7 + s) % 100);
          const uint32_t len = static_cast<uint32_t>(text.size());
          for (int b = 0; b < 4; ++b) packed.push_back(static_cast<uint8_t>(len >> (8 * b)));
          packed.insert(packed.end(), text.begin(), text.end());
        }
        bytes = packed.size();
        slot->storage.push_back(std::move(packed));
      } else {
        slot->storage.emplace_back(bytes, 0);
        // Floating-point tensors carry finite pseudo-random values of magnitude 1/8 ... 2 and either sign (a model
        // fed zeros does less switching work: ResNet-50 on an MI355X then runs at 2.39 GHz instead of 2.09 and 6 %
        // faster, which says nothing about real inputs); integer tensors stay zero - models read delays, counts
        // and indices from them.
        if (wire_random && (parts[1] == "FP16" || parts[1] == "BF16" || parts[1] == "FP32")) {
          uint64_t state = 0x9E3779B97F4A7C15ull * static_cast<uint64_t>(s + 1) + slot->storage.size();
          auto next = [&state]() {
            state ^= state << 13;
            state ^= state >> 7;
            state ^= state << 17;
            return state;
          };
          uint8_t* p = slot->storage.back().data();
          if (parts[1] == "FP32") {
            for (int64_t k = 0; k < count; ++k) {
              const uint64_t r = next();
              const uint32_t v = (static_cast<uint32_t>(r & 1) << 31) | ((124u + static_cast<uint32_t>((r >> 1) & 3)) << 23) |
                                 static_cast<uint32_t>((r >> 8) & 0x7FFFFF);
              std::memcpy(p + 4 * k, &v, 4);
            }
          } else {
            const bool bf16 = parts[1] == "BF16";
            for (int64_t k = 0; k < count; ++k) {
              const uint64_t r = next();
              const uint16_t v = bf16 ? static_cast<uint16_t>(((r & 1) << 15) | ((124u + ((r >> 1) & 3)) << 7) | ((r >> 8) & 0x7F))
                                      : static_cast<uint16_t>(((r & 1) << 15) | ((12u + ((r >> 1) & 3)) << 10) | ((r >> 8) & 0x3FF));
              std::memcpy(p + 2 * k, &v, 2);
            }
          }
        }
      }
      in->AppendRaw(slot->storage.back().data(), bytes);
      slot->inputs.emplace_back(in);
      slot->in_ptrs.push_back(in);
    }
    for (const auto& spec : shm_in) {
      auto parts = Split(spec, ':');
      if (parts.size() < 5) { std::cerr << "bad --shm-input " << spec << std::endl; return 2; }
      tc::InferInput* in = nullptr;
      e = tc::InferInput::Create(&in, parts[0], Dims(parts[2]), parts[1]);
      if (!e.IsOk()) { std::cerr << e.Message() << std::endl; return 1; }
      in->SetSharedMemory(subst(parts[3]), std::strtoull(parts[4].c_str(), nullptr, 10),
                          parts.size() > 5 ? slot_offset(parts[5]) : 0);
      slot->inputs.emplace_back(in);
      slot->in_ptrs.push_back(in);
    }
    for (const auto& spec : shm_out) {
      auto parts = Split(spec, ':');
      if (parts.size() < 3) { std::cerr << "bad --shm-output " << spec << std::endl; return 2; }
      tc::InferRequestedOutput* out = nullptr;
      tc::InferRequestedOutput::Create(&out, parts[0]);
      out->SetSharedMemory(subst(parts[1]), std::strtoull(parts[2].c_str(), nullptr, 10),
                           parts.size() > 3 ? slot_offset(parts[3]) : 0);
      slot->outputs.emplace_back(out);
      slot->out_ptrs.push_back(out);
    }
    slots.push_back(std::move(slot));
  }

  tc::InferOptions options(model);
  options.model_version_ = version;
  std::mutex mu;
  std::condition_variable cv;
  std::vector<uint64_t> latencies;
  latencies.reserve(1 << 20);
  std::atomic<uint64_t> errors{0};
  std::string first_error;
  // phase state (guarded by mu)
  bool record = false;      // completions of this phase go into `latencies`
  bool open_ended = false;  // duration phase: re-issue until `stopping`
  bool stopping = false;
  struct TraceRow {
    uint64_t start_ns, latency_ns;
    int slot;
    bool ok;
  };
  const bool tracing = !trace_path.empty();
  std::vector<TraceRow> trace_rows;
  long long to_issue = 0;   // count phase: requests still to be issued
  long long to_finish = 0;  // count phase: completions still owed
  int outstanding = 0;
  uint64_t last_done_ns = 0;
  std::atomic<uint64_t> responses{0};  // streaming: non-empty responses of the measured phase

  std::function<void(int)> issue;
  // one request of slot `s` is complete (unary: its response; streaming: its final response)
  auto complete = [&](int s, const tca::Error& status, uint64_t t1) {
    bool again = false;
    {
      std::lock_guard<std::mutex> lk(mu);
      --outstanding;
      last_done_ns = t1;
      if (!status.IsOk()) {
        ++errors;
        if (first_error.empty()) first_error = status.Message();
      } else if (record) {
        latencies.push_back(t1 - slots[s]->t0);
      }
      if (record && tracing) trace_rows.push_back({slots[s]->t0, t1 - slots[s]->t0, s, status.IsOk()});
      if (sequence_length > 0) {
        Slot* slot = slots[s].get();
        if (!status.IsOk() || ++slot->seq_pos == sequence_length) {  // next request opens a new sequence
          slot->seq_pos = 0;
          ++slot->seq_no;
        }
      }
      if (open_ended) {
        again = !stopping || (sequence_length > 0 && slots[s]->seq_pos != 0 && status.IsOk());  // a sequence is ended, not dropped
      } else {
        --to_finish;
        if (to_issue > 0) {
          --to_issue;
          again = true;
        }
      }
      if (again) ++outstanding;
      if (!again && (outstanding == 0 || (!open_ended && to_finish <= 0))) cv.notify_all();
    }
    if (again) issue(s);  // re-submitted from the reactor thread: zero idle time per slot
  };
  if (streaming) {
    // the stream callback sees every response of every slot on that connection: the request id
    // names the slot, the final flag ends the request
    for (auto& g : grpc_clients) {
      e = g->StartStream([&](tc::InferResult* r) {
        const uint64_t t1 = NowNs();
        tca::Error status = r->RequestStatus();
        std::string id;
        bool is_final = true, is_null = false;
        r->Id(&id);
        r->IsFinalResponse(&is_final);
        r->IsNullResponse(&is_null);
        delete r;
        if (status.IsOk() && !is_null && record) responses.fetch_add(1, std::memory_order_relaxed);
        if (!status.IsOk() || is_final) {
          const int s = id.empty() ? 0 : std::atoi(id.c_str());
          if (s >= 0 && s < static_cast<int>(slots.size())) complete(s, status, t1);
        }
      }, /*enable_stats=*/false);
      if (!e.IsOk()) { std::cerr << e.Message() << std::endl; return 1; }
    }
  }
  issue = [&](int s) {
    Slot* slot = slots[s].get();
    slot->t0 = NowNs();
    tca::Error se;
    tc::InferOptions per_request = options;
    if (sequence_length > 0) {
      per_request.sequence_id_ = sequence_id_base + slot->seq_no * static_cast<uint64_t>(concurrency) + static_cast<uint64_t>(s);
      per_request.sequence_start_ = slot->seq_pos == 0;
      per_request.sequence_end_ = slot->seq_pos == sequence_length - 1;
    }
    if (streaming) {
      per_request.request_id_ = std::to_string(s);
      per_request.triton_enable_empty_final_response_ = true;
      se = grpc_clients[s % grpc_clients.size()]->AsyncStreamInfer(per_request, slot->in_ptrs, slot->out_ptrs);
    } else {
      auto on_done = [&, s](tc::InferResult* r) {
        const uint64_t t1 = NowNs();
        tca::Error status = r->RequestStatus();
        delete r;
        complete(s, status, t1);
      };
      se = use_grpc ? grpc_clients[s % grpc_clients.size()]->AsyncInfer(on_done, per_request, slot->in_ptrs, slot->out_ptrs)
                    : http_clients[s % http_clients.size()]->AsyncInfer(on_done, per_request, slot->in_ptrs, slot->out_ptrs);
    }
    if (!se.IsOk()) {
      std::lock_guard<std::mutex> lk(mu);
      ++errors;
      if (first_error.empty()) first_error = se.Message();
      --outstanding;
      if (!open_ended) --to_finish;
      cv.notify_all();
    }
  };

  // Runs exactly `n` requests (never more than `concurrency` outstanding) and
  // returns the time from the first submission to the last completion.
  auto run_count = [&](long long n, bool measured) -> double {
    const int first_wave = static_cast<int>(std::min<long long>(n, concurrency));
    uint64_t t_start;
    {
      std::lock_guard<std::mutex> lk(mu);
      record = measured;
      open_ended = false;
      stopping = false;
      to_finish = n;
      to_issue = n - first_wave;
      outstanding = first_wave;
      t_start = NowNs();
    }
    for (int s = 0; s < first_wave; ++s) issue(s);
    std::unique_lock<std::mutex> lk(mu);
    tca::WaitFor(cv, lk, std::chrono::seconds(900), [&] { return to_finish <= 0 && outstanding == 0; });
    return (last_done_ns - t_start) / 1e9;
  };
  // Time-based: warm up, then record for `seconds`.
  auto run_duration = [&](double warm_s, double seconds) -> double {
    {
      std::lock_guard<std::mutex> lk(mu);
      record = false;
      open_ended = true;
      stopping = false;
      outstanding = concurrency;
    }
    for (int s = 0; s < concurrency; ++s) issue(s);
    std::this_thread::sleep_for(std::chrono::duration<double>(warm_s));
    uint64_t t0;
    {
      std::lock_guard<std::mutex> lk(mu);
      record = true;
      t0 = NowNs();
    }
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    uint64_t t1;
    {
      std::lock_guard<std::mutex> lk(mu);
      record = false;
      t1 = NowNs();
      stopping = true;
    }
    std::unique_lock<std::mutex> lk(mu);
    tca::WaitFor(cv, lk, std::chrono::seconds(60), [&] { return outstanding == 0; });
    return (t1 - t0) / 1e9;
  };

  double secs = 0;
  if (count > 0) {
    if (warmup_count > 0) run_count(warmup_count, false);
    if (wait_go) {
      // two-phase protocol for a driver that brackets the measured phase with its
      // own barriers: announce readiness, start on the first line read from stdin
      std::printf("READY\n");
      std::fflush(stdout);
      std::string line;
      if (!std::getline(std::cin, line)) return 1;
    }
    secs = run_count(count, true);
  } else {
    secs = run_duration(warmup, duration);
  }
  std::vector<uint64_t> lat;
  {
    std::lock_guard<std::mutex> lk(mu);
    lat = latencies;
  }
  if (tracing) {
    std::FILE* fh = std::fopen(trace_path.c_str(), "w");
    if (fh == nullptr) {
      std::cerr << "cannot write " << trace_path << std::endl;
      return 1;
    }
    std::lock_guard<std::mutex> lk(mu);
    uint64_t origin = ~0ull;
    for (const auto& r : trace_rows) origin = std::min(origin, r.start_ns);
    std::fprintf(fh, "start_ns,latency_ns,slot,ok\n");
    for (const auto& r : trace_rows)
      std::fprintf(fh, "%llu,%llu,%d,%d\n", static_cast<unsigned long long>(r.start_ns - origin),
                   static_cast<unsigned long long>(r.latency_ns), r.slot, r.ok ? 1 : 0);
    std::fclose(fh);
  }
  std::sort(lat.begin(), lat.end());
  auto pct = [&](double p) -> double {
    if (lat.empty()) return 0.0;
    const size_t idx = std::min(lat.size() - 1, static_cast<size_t>(p / 100.0 * (lat.size() - 1) + 0.5));
    return lat[idx] / 1e3;
  };
  double sum = 0;
  for (uint64_t v : lat) sum += static_cast<double>(v);
  std::printf(
      "{\"requests\": %zu, \"errors\": %llu, \"duration_s\": %.4f, \"requests_per_sec\": %.2f, \"infer_per_sec\": %.2f, "
      "\"avg_us\": %.1f, \"p50_us\": %.1f, \"p90_us\": %.1f, \"p95_us\": %.1f, \"p99_us\": %.1f, \"concurrency\": %d, "
      "\"protocol\": \"%s\", \"streaming\": %s, \"responses_per_sec\": %.2f, \"first_error\": \"%s\"}\n",
      lat.size(), static_cast<unsigned long long>(errors.load()), secs, lat.size() / secs, lat.size() * batch / secs,
      lat.empty() ? 0.0 : sum / lat.size() / 1e3, pct(50), pct(90), pct(95), pct(99), concurrency,
      protocol.c_str(), streaming ? "true" : "false", streaming ? responses.load() / secs : lat.size() / secs,
      first_error.substr(0, 200).c_str());
  return 0;
}
