// Native execution path of the in-repo server for models whose kernel exists in C++.
//
// The serving fronts hand every inference request to TryGrpc / TryHttp first, on the
// reactor thread that received it.  When the model is registered here and the request
// is completely ordinary (no request parameters, tensors on the wire or in system
// shared memory, shapes and types that match the model) it is decoded, executed and
// answered without the Python interpreter; anything else - including every request
// that would be an error - returns false and takes the general path, which also
// produces the error texts.  Microsecond-scale CPU models are bound by the
// interpreter lock otherwise (about 25 us of Python per request).
//
// Kernels: "add_sub" (OUTPUT0 = INPUT0 + INPUT1, OUTPUT1 = INPUT0 - INPUT1, any integer
// or floating type; the `simple` model family the reference's examples and perf tools
// are written against, e.g. `src/c++/examples/simple_http_infer_client.cc:127-160`),
// "identity" (OUTPUT0 = INPUT0) and, over gRPC streams, the decoupled "repeat" (one
// response per input element, as `src/python/examples/simple_grpc_custom_repeat.py` uses).
//
// Stateful models ("sequence" on `simple_sequence`: OUTPUT = INPUT, + 1 on the first request of a
// sequence; "dyna_sequence" on `simple_dyna_sequence`: the same plus the correlation id on the last
// one; `src/python/examples/simple_grpc_sequence_stream_infer_client.py` drives both): the table of
// open sequences lives here and the Python core consults the same table (SequenceStep), so a
// sequence may mix requests answered here and requests that took the general path.
//
// BYTES models whose tensors stay in device shared memory ("bytes_add_sub" on `simple_string`,
// "bytes_identity" on `simple_identity`; the reference drives them through
// `src/python/examples/simple_grpc_cudashm_string_client.py`-style callers): the request is
// decoded on the reactor thread and queued; one batch thread hands whole groups of requests
// to the device pipeline (prefix walk, decimal arithmetic, packing into the callers' regions:
// three kernel launches per group) and answers them from there (TryGrpcAsync).  The pipeline
// itself lives in the HIP library; this file only sees its C entry points (DevicePipe).
//
// GPU models whose forward pass is a set of captured HIP graphs ("graph" on `resnet50`; the model
// contract is the image classifier the reference's `src/python/examples/image_client.py:59-150` and
// perf tools drive: one fixed-size input, one vector output, a batch dimension): a request whose
// tensors are in device shared memory is decoded on the reactor thread and queued; a batch thread
// merges requests along the batch axis with the serving policy of the dynamic batcher (absorb
// arrivals while the GPU is full, fill window, burst window, a cap that follows the load, halves on
// two idle instances), gathers the inputs into the static buffer of the smallest captured graph
// that holds the batch (padding: rows past the batch are computed and ignored), launches the graph
// and scatters the outputs into the callers' regions - three launches per batch, no interpreter -
// and one completer thread per execution instance waits for the batch's event and answers.  The
// graphs, their buffers and the streams stay owned by the Python model; this file only sees HIP
// through the C entry points in GraphPipe.
//
// Statistics are accumulated here and collected by the Python core (TakeStats), so the
// model statistics endpoints keep counting requests that never reached Python.
#pragma once

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <unordered_set>
#include <utility>
#include <vector>

#include "tca/error.h"

namespace tca {
namespace server {

class NativeExecutor {
 public:
  struct TensorSpec {
    std::string name, datatype;
    std::vector<int64_t> dims;  // without the batch dimension; -1 = any
  };
  struct ModelSpec {
    std::string name, version;  // version = the one unversioned requests resolve to
    int max_batch = 0;
    std::vector<TensorSpec> inputs, outputs;
    std::string kernel;  // "add_sub" | "identity" | "repeat" | "bytes_add_sub" | "bytes_identity" | "graph" ...
  };
  // Captured forward graphs of one GPU model and the HIP entry points to run them (all owned by the
  // caller, alive until DisableGraphModel has returned).
  struct GraphPipe {
    int (*graph_launch)(void* exec, void* stream) = nullptr;
    int (*event_create)(void** event) = nullptr;
    int (*event_record)(void* event, void* stream) = nullptr;
    int (*event_sync)(void* event) = nullptr;
    // optional: the staging thread drains its copy stream with it every few rounds.  The HIP runtime keeps about
    // a kilobyte per host-to-device hipMemcpyAsync until the STREAM is synchronised (waiting for an event that
    // follows the copies does not release it): 16 MB per second at 9k staged requests per second without this.
    int (*stream_sync)(void* stream) = nullptr;
    int (*event_destroy)(void* event) = nullptr;
    int (*set_device)(int device) = nullptr;
    int (*batched_copy)(const void* const* srcs, void* const* dsts, const uint64_t* bytes, int n, void* stream) = nullptr;
    int device = 0;
    std::vector<void*> streams;  // one per execution instance
    struct Graph {
      int instance = 0, batch = 0;
      void* exec = nullptr;
      uint64_t in_ptr = 0, out_ptr = 0;  // static buffers the graph reads / writes: [batch][row bytes]
    };
    std::vector<Graph> graphs;
    uint64_t in_row_bytes = 0, out_row_bytes = 0;  // per batch entry
    int max_batch = 0;
    int max_inflight = 0;  // batches enqueued on the GPU at once (0: two per instance)
    // batching policy, microseconds (see server/batcher.py for what each one is for)
    int fill_window_us = 0, burst_gap_us = 300, burst_window_us = 100;
    int split_min_batch = 16;
    bool adaptive_cap = true;
    // With few images outstanding (at most cap_limit x max_batch, seen as a decaying peak) merged batches are capped so
    // that every instance gets cap_depth of them; beyond the limit everything queued merges.
    double cap_depth = 1.0, cap_limit = 0.8;
    // Requests whose tensors travel inside the messages: page-locked host memory that the device addresses too
    // (the gather / scatter kernels read and write it over the bus).  A request's input bytes are copied into
    // a piece of it by the thread that received the message, its output rows are taken from there into the
    // response.  0 bytes: such requests take the general path.  Owned by the caller like the graphs.
    uint64_t host_arena = 0, host_arena_bytes = 0;
    uint64_t host_chunk_bytes = 1u << 20;  // allocation unit
    // Optional second stage: device memory of the same size and a stream of its own.  A staging thread moves
    // the input bytes of arriving requests from the host arena to the same offset of the device arena (one
    // copy-kernel launch for everything that arrived meanwhile) before the requests join the batch queue, so
    // the transfer over the bus overlaps the forward passes instead of running in front of them on the
    // instances' streams.  Without it the gather kernel reads the host arena directly.
    uint64_t device_arena = 0;
    void* copy_stream = nullptr;
    // When set, the staging thread moves each request's bytes with this (hipMemcpyAsync, host to device: the
    // DMA engines, no compute unit involved) instead of the copy kernel.
    int (*memcpy_async)(void* dst, const void* src, uint64_t bytes, int kind, void* stream) = nullptr;
  };
  // C entry points of the device BYTES pipeline (`csrc/hip/bytes_models.hip`) plus one pipeline
  // object and the stream its batches run on; all owned by the caller, alive until
  // DisableDeviceBytes has returned.
  struct DevicePipe {
    int (*submit)(void* pipe, int op, int n, int count, const uint64_t* in_ptrs, const uint64_t* in_bytes,
                  const uint64_t* out_ptrs, const uint64_t* out_caps, void* stream) = nullptr;
    int (*wait)(void* pipe, int n, uint32_t* codes, uint64_t* out_bytes) = nullptr;
    void* pipe = nullptr;
    void* stream = nullptr;
    int device = 0, max_requests = 0, max_count = 0;
    uint64_t max_stream_bytes = 0;
  };
  enum class Outcome { kDeclined, kAnswered, kDeferred };
  // ok: `payload` is the serialized response.  Not ok: the device refused the request (malformed stream,
  // output range too small...): `payload` is the request, which the caller hands to the general path.
  using Completion = std::function<void(bool ok, std::string&& payload)>;
  using HeaderList = std::vector<std::pair<std::string, std::string>>;
  // The HTTP form: ok: response headers and parts.  Not ok: `parts` holds the request body alone.
  using HttpCompletion = std::function<void(bool ok, HeaderList&& headers, std::vector<std::string>&& parts)>;
  struct BatchStats {
    uint64_t count = 0, input_ns = 0, infer_ns = 0, output_ns = 0;
  };
  struct Stats {
    uint64_t requests = 0, inferences = 0, queue_ns = 0, last_ms = 0;
    std::map<int, BatchStats> by_batch;
  };

  Error SetModel(const ModelSpec& spec);
  void RemoveModel(const std::string& name);
  // A window [offset, offset + size) of the POSIX shared-memory object `key`.
  Error RegisterRegion(const std::string& name, const std::string& key, uint64_t offset, uint64_t size);
  void UnregisterRegion(const std::string& name);  // "" = all
  void SetEnabled(bool on) { enabled_.store(on); }

  NativeExecutor();
  ~NativeExecutor();
  NativeExecutor(const NativeExecutor&) = delete;
  NativeExecutor& operator=(const NativeExecutor&) = delete;

  // Starts (or, with other arguments, restarts) the batch thread of the device BYTES path.
  Error EnableDeviceBytes(const DevicePipe& pipe);
  // Answers what is queued, then stops the batch thread.
  void DisableDeviceBytes();
  // A registered device shared-memory region: `size` bytes at device address `address` on GPU `device`.
  Error RegisterDeviceRegion(const std::string& name, uint64_t address, uint64_t size, int device);
  // Stops new requests on the region ("" = all), then waits for the queued ones that name it.  False: still
  // in use after `timeout_ms` (the region stays registered).
  bool UnregisterDeviceRegion(const std::string& name, int timeout_ms);
  // Blocks until no deferred request is queued or running.
  void Quiesce();
  // Starts (or restarts) the batch and completer threads that serve `model` (registered with the "graph"
  // kernel) from its captured graphs.
  Error EnableGraphModel(const std::string& model, const GraphPipe& pipe);
  // Answers what is queued for `model` ("" = every graph model), then stops its threads.
  void DisableGraphModel(const std::string& model);
  // Batches launched, requests merged into them and rows computed beyond the merged batches (padding), per model.
  // (wire_requests: requests of a graph model whose input or output travelled inside the messages)
  struct GraphCounters {
    uint64_t batches = 0, requests = 0, images = 0, padded_images = 0, wire_requests = 0, arena_full = 0;
  };
  bool GraphStats(const std::string& model, GraphCounters* out);
  // Stream capture elsewhere in the process (a model capturing one more graph) and graph launches from the
  // batch threads here must not overlap: a launch into a capturing stream would be recorded into that graph,
  // and under the global capture mode a launch on any stream may fail the capture.  BeginCapture waits for
  // the enqueue sequences in progress and holds new ones back until the matching EndCapture.
  void BeginCapture();
  void EndCapture();

  // `request` = serialized ModelInferRequest; true: `response` holds the serialized ModelInferResponse.
  bool TryGrpc(const std::string& request, std::string* response);
  // TryGrpc plus the device BYTES path.  kDeferred: `*request` has been moved into the queue and `done`
  // runs later on the batch thread.
  Outcome TryGrpcAsync(std::string* request, std::string* response, Completion done);
  // One message of a ModelStreamInfer call; true: `responses` holds the serialized ModelStreamInferResponse
  // messages to send, in order (one for the ordinary kernels, one per element - plus the empty final one
  // when the request asks for it - for the decoupled "repeat" kernel).
  bool TryGrpcStream(const std::string& request, std::vector<std::string>* responses);
  // `body` = JSON header (+ binary tensors when `header_length` >= 0); true: answer in headers / parts.
  bool TryHttp(const std::string& model, const std::string& version, const std::string& body, int64_t header_length,
               HeaderList* headers, std::vector<std::string>* parts);

  // One request of sequence `id` (numeric correlation id, not 0) of a model with a sequence kernel: opens the
  // sequence on `start`, closes it on `end`.  False: no such model, or the sequence was never started.
  bool SequenceStep(const std::string& name, uint64_t id, bool start, bool end);
  // Sequences of `name` that have been started and not ended (-1: no such model).
  long OpenSequences(const std::string& name);

  // Whether requests for `model` may be deferred (a device BYTES kernel): only those need TryGrpcAsync / TryHttpAsync.
  bool Defers(const std::string& model);
  // TryHttp plus the device BYTES path.  kDeferred: `*body` has been moved into the queue and `done` runs later
  // on the batch thread.
  Outcome TryHttpAsync(const std::string& model, const std::string& version, std::string* body, int64_t header_length,
                       HeaderList* headers, std::vector<std::string>* parts, HttpCompletion done);

  // Moves the statistics accumulated for `name` since the last call into `out`; false if there are none.
  bool TakeStats(const std::string& name, Stats* out);
  uint64_t Handled() const { return handled_.load(); }
  uint64_t Declined() const { return declined_.load(); }

 private:
  struct Region {
    uint8_t* base = nullptr;  // first byte of the window
    uint64_t size = 0;
    void* map = nullptr;
    uint64_t map_size = 0;
    ~Region();
  };
  struct Model {
    ModelSpec spec;
    int kernel = 0;
    size_t itemsize = 0;
    std::mutex stats_mu;
    Stats stats;
    std::mutex seq_mu;
    std::unordered_set<uint64_t> open_sequences;
    bool Step(uint64_t id, bool start, bool end);
  };
  struct DeviceRegion {
    uint64_t address = 0, size = 0;
    int device = 0;
    std::atomic<int> inflight{0};
  };
  struct Request;
  struct Result;
  struct DeviceJob;
  struct GraphJob;
  class GraphRunner;
  friend class GraphRunner;
  bool PrepareGraphJob(Request* req, const std::shared_ptr<Model>& model, const std::shared_ptr<GraphRunner>& runner,
                       GraphJob* job);
  std::shared_ptr<GraphRunner> FindRunner(const std::string& model);
  // Validates and queues a request for a graph model (`bytes`: the request message / body, moved on kDeferred).
  Outcome SubmitGraph(Request* req, const std::shared_ptr<Model>& model, std::string* bytes, Completion done,
                      HttpCompletion http_done);
  void QuiesceGraphModels();
  void EnterLaunch();  // the batch threads, around every gather / launch / scatter / event sequence
  void LeaveLaunch();
  void FinishGraphJob(GraphJob* job, bool ok, uint64_t t_exec, uint64_t t_enq, uint64_t t_done, bool first);
  bool PrepareDeviceJob(Request* req, const std::shared_ptr<Model>& model, DeviceJob* job);
  void DeviceLoop();
  void FinishDeviceJob(DeviceJob* job, bool ok);
  static void BuildGrpcResponse(const Request& req, const std::vector<Result>& results, int final_flag,
                                std::string* response);
  bool Run(Request* req, const std::shared_ptr<Model>& model);
  bool RunRepeat(Request* req, const std::shared_ptr<Model>& model, std::vector<std::string>* responses);
  bool ParseGrpc(const std::string& request, bool stream, Request* out, std::shared_ptr<Model>* model);
  bool ParseHttp(const std::string& model_name, const std::string& version, const std::string& body, int64_t header_length,
                 Request* out, std::shared_ptr<Model>* model);
  static void BuildHttpResponse(const Request& req, std::vector<Result>* results, HeaderList* headers,
                                std::vector<std::string>* parts);
  std::shared_ptr<Model> FindModel(const std::string& name, const std::string& version);

  std::atomic<bool> enabled_{true};
  std::atomic<uint64_t> handled_{0}, declined_{0};
  std::shared_mutex mu_;
  std::map<std::string, std::shared_ptr<Model>> models_;
  std::map<std::string, std::shared_ptr<Region>> regions_;
  std::map<std::string, std::shared_ptr<DeviceRegion>> device_regions_;
  std::map<std::string, std::shared_ptr<GraphRunner>> runners_;  // graph models being served
  std::mutex gate_mu_;
  std::condition_variable gate_cv_;
  int capturing_ = 0, launching_ = 0;

  // device BYTES path
  DevicePipe pipe_;
  std::atomic<bool> pipe_on_{false};
  std::thread batch_thread_;
  std::mutex q_mu_;
  std::condition_variable q_cv_, idle_cv_;
  std::deque<std::unique_ptr<DeviceJob>> queue_;
  size_t running_ = 0;  // taken off the queue, not yet answered
  bool stop_ = false;
};

}  // namespace server
}  // namespace tca
