// Native execution path (see tca/native_executor.h).
#include "tca/native_executor.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <thread>

#include "tca/json.h"
#include "tca/pbwire.h"
#include "tca/sync.h"

namespace tca {
namespace server {
namespace {

enum Kernel { kAddSub = 1, kIdentity = 2, kRepeat = 3, kBytesAddSub = 4, kBytesIdentity = 5, kSequence = 6, kDynaSequence = 7, kGraph = 8 };

size_t ItemSize(const std::string& dt) {
  if (dt == "INT8" || dt == "UINT8" || dt == "BOOL") return 1;
  if (dt == "INT16" || dt == "UINT16" || dt == "FP16" || dt == "BF16") return 2;
  if (dt == "INT32" || dt == "UINT32" || dt == "FP32") return 4;
  if (dt == "INT64" || dt == "UINT64" || dt == "FP64") return 8;
  return 0;
}

inline uint64_t NowNs() {
  return static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(
                                   std::chrono::steady_clock::now().time_since_epoch()).count());
}

// Integer arithmetic wraps like numpy's fixed-width types do.
template <typename T, typename U>
void AddSubT(const uint8_t* a, const uint8_t* b, uint8_t* sum, uint8_t* diff, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    T x, y;
    std::memcpy(&x, a + i * sizeof(T), sizeof(T));
    std::memcpy(&y, b + i * sizeof(T), sizeof(T));
    const T s = static_cast<T>(static_cast<U>(x) + static_cast<U>(y));
    const T d = static_cast<T>(static_cast<U>(x) - static_cast<U>(y));
    if (sum) std::memcpy(sum + i * sizeof(T), &s, sizeof(T));
    if (diff) std::memcpy(diff + i * sizeof(T), &d, sizeof(T));
  }
}

bool AddSub(const std::string& dt, const uint8_t* a, const uint8_t* b, uint8_t* sum, uint8_t* diff, size_t n) {
  if (dt == "INT32") AddSubT<int32_t, uint32_t>(a, b, sum, diff, n);
  else if (dt == "INT8") AddSubT<int8_t, uint8_t>(a, b, sum, diff, n);
  else if (dt == "INT16") AddSubT<int16_t, uint16_t>(a, b, sum, diff, n);
  else if (dt == "INT64") AddSubT<int64_t, uint64_t>(a, b, sum, diff, n);
  else if (dt == "UINT8") AddSubT<uint8_t, uint8_t>(a, b, sum, diff, n);
  else if (dt == "UINT16") AddSubT<uint16_t, uint16_t>(a, b, sum, diff, n);
  else if (dt == "UINT32") AddSubT<uint32_t, uint32_t>(a, b, sum, diff, n);
  else if (dt == "UINT64") AddSubT<uint64_t, uint64_t>(a, b, sum, diff, n);
  else if (dt == "FP32") AddSubT<float, float>(a, b, sum, diff, n);
  else if (dt == "FP64") AddSubT<double, double>(a, b, sum, diff, n);
  else return false;
  return true;
}

struct ShmRef {
  bool set = false;
  std::string region;
  int64_t size = 0, offset = 0;
};

}  // namespace

struct NativeExecutor::Result {
  std::string name, datatype;
  std::vector<int64_t> shape;
  std::string bytes;  // when not in shared memory
  ShmRef shm;
  size_t shm_bytes = 0;
};

struct NativeExecutor::Request {
  struct Input {
    std::string name, datatype;
    std::vector<int64_t> shape;
    const uint8_t* data = nullptr;  // wire bytes, or resolved shm window
    size_t size = 0;
    ShmRef shm;
    bool has_data = false;
  };
  struct Output {
    std::string name;
    ShmRef shm;
    bool binary = true;  // HTTP only
  };
  std::string model, version, id;
  bool want_final = false;  // streams: triton_enable_empty_final_response
  bool has_sequence = false;  // any of sequence_id / sequence_start / sequence_end was given
  uint64_t sequence_id = 0;
  bool sequence_start = false, sequence_end = false;
  std::vector<Input> inputs;
  std::vector<Output> outputs;  // empty = all
  std::vector<Result> results;
  std::string version_used;
  std::vector<std::shared_ptr<Region>> held;  // regions in use stay mapped until the answer is built
  uint64_t t_arrive = 0;
};

// One request on its way through the device BYTES pipeline.
struct NativeExecutor::DeviceJob {
  Request req;  // results already describe the answer; the kernels fill the regions
  std::shared_ptr<Model> model;
  std::string request_bytes;  // handed back when the device refuses the request
  Completion done;            // gRPC caller ...
  HttpCompletion http_done;   // ... or HTTP caller (exactly one of the two is set)
  int op = 0;
  uint32_t count = 0;
  int batch = 1;
  int n_in = 0, n_out = 0;
  uint64_t in_ptrs[2] = {0, 0}, in_bytes[2] = {0, 0}, out_ptrs[2] = {0, 0}, out_caps[2] = {0, 0};
  std::vector<std::shared_ptr<DeviceRegion>> leases;
  uint64_t t0 = 0, t1 = 0;
};

NativeExecutor::Region::~Region() {
  if (map != nullptr) munmap(map, map_size);
}

NativeExecutor::NativeExecutor() = default;

NativeExecutor::~NativeExecutor() {
  DisableGraphModel(std::string());
  DisableDeviceBytes();
}

Error NativeExecutor::SetModel(const ModelSpec& spec) {
  auto m = std::make_shared<Model>();
  m->spec = spec;
  if (spec.kernel == "add_sub") {
    if (spec.inputs.size() != 2 || spec.outputs.size() != 2) return Error("add_sub needs two inputs and two outputs");
    m->kernel = kAddSub;
  } else if (spec.kernel == "identity") {
    if (spec.inputs.size() != 1 || spec.outputs.size() != 1) return Error("identity needs one input and one output");
    m->kernel = kIdentity;
  } else if (spec.kernel == "bytes_add_sub" || spec.kernel == "bytes_identity") {
    const size_t arity = spec.kernel == "bytes_add_sub" ? 2 : 1;
    if (spec.inputs.size() != arity || spec.outputs.size() != arity) return Error(spec.kernel + ": wrong number of tensors");
    for (const auto& t : spec.inputs)
      if (t.datatype != "BYTES") return Error(spec.kernel + " takes BYTES tensors");
    for (const auto& t : spec.outputs)
      if (t.datatype != "BYTES") return Error(spec.kernel + " produces BYTES tensors");
    m->kernel = arity == 2 ? kBytesAddSub : kBytesIdentity;
    std::unique_lock<std::shared_mutex> lk(mu_);
    models_[spec.name] = std::move(m);
    return Error::Success();
  } else if (spec.kernel == "sequence" || spec.kernel == "dyna_sequence") {
    if (spec.inputs.size() != 1 || spec.outputs.size() != 1 || spec.inputs[0].datatype != "INT32" ||
        spec.outputs[0].datatype != "INT32" || spec.inputs[0].dims != std::vector<int64_t>{1} ||
        spec.outputs[0].dims != std::vector<int64_t>{1} || spec.max_batch != 1)
      return Error(spec.kernel + " needs one INT32 [1] input and output and a batch limit of 1");
    m->kernel = spec.kernel == "sequence" ? kSequence : kDynaSequence;
    m->itemsize = 4;
    std::unique_lock<std::shared_mutex> lk(mu_);
    models_[spec.name] = std::move(m);
    return Error::Success();
  } else if (spec.kernel == "graph") {
    // a GPU model served from captured graphs: one fixed-size input, one fixed-size output, a batch dimension
    if (spec.max_batch < 1 || spec.inputs.size() != 1 || spec.outputs.size() != 1 ||
        ItemSize(spec.inputs[0].datatype) == 0 || ItemSize(spec.outputs[0].datatype) == 0)
      return Error("graph needs one fixed-type input, one fixed-type output and a batch dimension");
    for (const auto* t : {&spec.inputs[0], &spec.outputs[0]})
      for (int64_t d : t->dims)
        if (d < 1) return Error("graph needs fixed tensor shapes");
    m->kernel = kGraph;
    m->itemsize = ItemSize(spec.inputs[0].datatype);
    std::unique_lock<std::shared_mutex> lk(mu_);
    models_[spec.name] = std::move(m);
    return Error::Success();
  } else if (spec.kernel == "repeat") {
    // decoupled: IN INT32[-1], DELAY UINT32[-1], WAIT UINT32[1] -> one response (OUT INT32[1], IDX UINT32[1]) per element
    if (spec.max_batch != 0 || spec.inputs.size() != 3 || spec.outputs.size() != 2) return Error("repeat needs IN, DELAY, WAIT and OUT, IDX");
    m->kernel = kRepeat;
    m->itemsize = 4;
    std::unique_lock<std::shared_mutex> lk(mu_);
    models_[spec.name] = std::move(m);
    return Error::Success();
  } else {
    return Error("unknown native kernel '" + spec.kernel + "'");
  }
  const std::string& dt = spec.inputs[0].datatype;
  m->itemsize = ItemSize(dt);
  if (m->itemsize == 0) return Error("datatype '" + dt + "' has no native kernel");
  for (const auto& t : spec.inputs)
    if (t.datatype != dt || t.dims != spec.inputs[0].dims) return Error("native kernels need equally typed and shaped tensors");
  for (const auto& t : spec.outputs)
    if (t.datatype != dt || t.dims != spec.inputs[0].dims) return Error("native kernels need equally typed and shaped tensors");
  if (m->kernel == kAddSub && (dt == "BOOL" || dt == "FP16" || dt == "BF16")) return Error("add_sub has no " + dt + " form");
  std::unique_lock<std::shared_mutex> lk(mu_);
  models_[spec.name] = std::move(m);
  return Error::Success();
}

void NativeExecutor::RemoveModel(const std::string& name) {
  DisableGraphModel(name);  // no-op for every other kernel
  std::unique_lock<std::shared_mutex> lk(mu_);
  models_.erase(name);
}

Error NativeExecutor::RegisterRegion(const std::string& name, const std::string& key, uint64_t offset, uint64_t size) {
  const int fd = shm_open(key.c_str(), O_RDWR, 0);
  if (fd < 0) return Error("unable to open shared memory object '" + key + "'");
  struct stat st {};
  if (fstat(fd, &st) != 0 || offset + size < offset || static_cast<uint64_t>(st.st_size) < offset + size) {
    close(fd);
    return Error("shared memory object '" + key + "' is smaller than offset + byte size");
  }
  void* p = mmap(nullptr, offset + size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return Error("unable to map shared memory object '" + key + "'");
  auto r = std::make_shared<Region>();
  r->map = p;
  r->map_size = offset + size;
  r->base = static_cast<uint8_t*>(p) + offset;
  r->size = size;
  std::unique_lock<std::shared_mutex> lk(mu_);
  regions_[name] = std::move(r);
  return Error::Success();
}

void NativeExecutor::UnregisterRegion(const std::string& name) {
  std::unique_lock<std::shared_mutex> lk(mu_);
  if (name.empty()) regions_.clear();
  else regions_.erase(name);
}

std::shared_ptr<NativeExecutor::Model> NativeExecutor::FindModel(const std::string& name, const std::string& version) {
  std::shared_lock<std::shared_mutex> lk(mu_);
  auto it = models_.find(name);
  if (it == models_.end()) return nullptr;
  if (!version.empty() && version != it->second->spec.version) return nullptr;
  return it->second;
}

bool NativeExecutor::Model::Step(uint64_t id, bool start, bool end) {
  std::lock_guard<std::mutex> lk(seq_mu);
  if (start) {
    if (!end) open_sequences.insert(id);
    else open_sequences.erase(id);
    return true;
  }
  auto it = open_sequences.find(id);
  if (it == open_sequences.end()) return false;  // the first request of a sequence carries the start flag
  if (end) open_sequences.erase(it);
  return true;
}

bool NativeExecutor::SequenceStep(const std::string& name, uint64_t id, bool start, bool end) {
  std::shared_ptr<Model> m = FindModel(name, std::string());
  if (!m || (m->kernel != kSequence && m->kernel != kDynaSequence) || id == 0) return false;
  return m->Step(id, start, end);
}

long NativeExecutor::OpenSequences(const std::string& name) {
  std::shared_ptr<Model> m = FindModel(name, std::string());
  if (!m) return -1;
  std::lock_guard<std::mutex> lk(m->seq_mu);
  return static_cast<long>(m->open_sequences.size());
}

bool NativeExecutor::TakeStats(const std::string& name, Stats* out) {
  std::shared_ptr<Model> m;
  {
    std::shared_lock<std::shared_mutex> lk(mu_);
    auto it = models_.find(name);
    if (it == models_.end()) return false;
    m = it->second;
  }
  std::lock_guard<std::mutex> lk(m->stats_mu);
  if (m->stats.requests == 0) return false;
  *out = std::move(m->stats);
  m->stats = Stats();
  return true;
}

// Validates the decoded request against the model exactly as far as needed to be sure the general
// path would accept it too, then runs the kernel.  False = not ours (nothing has been written).
bool NativeExecutor::Run(Request* req, const std::shared_ptr<Model>& model) {
  if (model->kernel == kGraph) return false;  // only through the deferred entry points
  const ModelSpec& spec = model->spec;
  const uint64_t t0 = NowNs();
  const bool stateful = model->kernel == kSequence || model->kernel == kDynaSequence;
  if (model->kernel != kAddSub && model->kernel != kIdentity && !stateful) return false;
  if (stateful) {
    if (!req->has_sequence || req->sequence_id == 0) return false;  // the general path words the refusal
    // the last request of a dyna sequence adds the id as a 32-bit integer: larger ids are the general path's to refuse
    if (model->kernel == kDynaSequence && req->sequence_end && req->sequence_id > 0x7fffffffull) return false;
  } else if (req->has_sequence) {
    return false;
  }
  if (req->inputs.size() != spec.inputs.size()) return false;
  const Request::Input* ordered[2] = {nullptr, nullptr};
  int64_t batch = 0;
  size_t count = 0;
  for (auto& in : req->inputs) {
    size_t slot = spec.inputs.size();
    for (size_t k = 0; k < spec.inputs.size(); ++k)
      if (spec.inputs[k].name == in.name) slot = k;
    if (slot == spec.inputs.size() || ordered[slot] != nullptr) return false;
    const TensorSpec& ts = spec.inputs[slot];
    if (in.datatype != ts.datatype) return false;
    const size_t lead = spec.max_batch > 0 ? 1 : 0;
    if (in.shape.size() != ts.dims.size() + lead) return false;
    size_t n = 1;
    for (size_t d = 0; d < in.shape.size(); ++d) {
      const int64_t v = in.shape[d];
      if (v < 0 || v > (int64_t(1) << 32)) return false;
      if (d >= lead && ts.dims[d - lead] != -1 && ts.dims[d - lead] != v) return false;
      n *= static_cast<size_t>(v);
      if (n > (size_t(1) << 34)) return false;
    }
    if (lead) {
      if (in.shape[0] < 1 || in.shape[0] > spec.max_batch) return false;
      batch = in.shape[0];
    }
    if (ordered[0] != nullptr && in.shape != ordered[0]->shape) return false;
    if (in.shm.set) {
      std::shared_ptr<Region> region;
      {
        std::shared_lock<std::shared_mutex> lk(mu_);
        auto it = regions_.find(in.shm.region);
        if (it != regions_.end()) region = it->second;
      }
      if (!region || in.shm.offset < 0 || in.shm.size < 0 ||
          static_cast<uint64_t>(in.shm.offset) + static_cast<uint64_t>(in.shm.size) > region->size)
        return false;
      in.data = region->base + in.shm.offset;
      in.size = static_cast<size_t>(in.shm.size);
      req->held.push_back(std::move(region));
    } else if (!in.has_data) {
      return false;
    }
    if (in.size != n * model->itemsize) return false;
    count = n;
    ordered[slot] = &in;
  }
  if (spec.inputs.size() == 2 && ordered[0]->shape != ordered[1]->shape) return false;
  // outputs: the requested ones, or all of them
  std::vector<Request::Output> wanted = req->outputs;
  if (wanted.empty())
    for (const auto& t : spec.outputs) {
      Request::Output o;
      o.name = t.name;
      wanted.push_back(o);
    }
  const size_t nbytes = count * model->itemsize;
  req->results.clear();
  req->results.reserve(wanted.size());
  std::vector<uint8_t*> dest(spec.outputs.size(), nullptr);
  for (const auto& o : wanted) {
    size_t slot = spec.outputs.size();
    for (size_t k = 0; k < spec.outputs.size(); ++k)
      if (spec.outputs[k].name == o.name) slot = k;
    if (slot == spec.outputs.size() || dest[slot] != nullptr) return false;
    Result r;
    r.name = o.name;
    r.datatype = spec.outputs[slot].datatype;
    r.shape = ordered[0]->shape;
    if (o.shm.set) {
      std::shared_ptr<Region> region;
      {
        std::shared_lock<std::shared_mutex> lk(mu_);
        auto it = regions_.find(o.shm.region);
        if (it != regions_.end()) region = it->second;
      }
      if (!region || o.shm.offset < 0 || o.shm.size < 0 || static_cast<uint64_t>(o.shm.size) < nbytes ||
          static_cast<uint64_t>(o.shm.offset) + nbytes > region->size)
        return false;
      r.shm = o.shm;
      r.shm_bytes = nbytes;
      req->results.push_back(std::move(r));
      dest[slot] = region->base + o.shm.offset;
      req->held.push_back(std::move(region));
    } else {
      r.bytes.resize(nbytes);
      req->results.push_back(std::move(r));
      dest[slot] = reinterpret_cast<uint8_t*>(&req->results.back().bytes[0]);
    }
  }
  // `results` does not grow any more, but the pointers into its strings were taken while it did:
  // strings of this size live on the heap, and reserve() above keeps the vector in place
  const uint64_t t1 = NowNs();
  if (stateful) {
    // everything has been validated: from here on the request is answered, so the table may move
    if (!model->Step(req->sequence_id, req->sequence_start, req->sequence_end)) return false;
    if (dest[0] != nullptr) {
      uint32_t v;
      std::memcpy(&v, ordered[0]->data, 4);
      if (req->sequence_start) v += 1u;
      if (model->kernel == kDynaSequence && req->sequence_end) v += static_cast<uint32_t>(req->sequence_id);
      std::memcpy(dest[0], &v, 4);
    }
  } else if (model->kernel == kAddSub) {
    if (!AddSub(spec.inputs[0].datatype, ordered[0]->data, ordered[1]->data, dest[0], dest[1], count)) return false;
  } else {
    if (dest[0] != nullptr) std::memcpy(dest[0], ordered[0]->data, nbytes);
  }
  const uint64_t t2 = NowNs();
  req->version_used = spec.version;
  {
    const int b = static_cast<int>(batch > 0 ? batch : 1);
    std::lock_guard<std::mutex> lk(model->stats_mu);
    Stats& s = model->stats;
    ++s.requests;
    s.inferences += static_cast<uint64_t>(b);
    s.queue_ns += t0 - req->t_arrive;
    s.last_ms = static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::milliseconds>(
                                          std::chrono::system_clock::now().time_since_epoch()).count());
    BatchStats& bs = s.by_batch[b];
    ++bs.count;
    bs.input_ns += t1 - t0;
    bs.infer_ns += t2 - t1;
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// gRPC
// ---------------------------------------------------------------------------------------------
namespace {

// map<string, InferParameter> entry -> shm reference fields; false on a key this path does not know.
bool ShmParam(const uint8_t* d, size_t n, ShmRef* shm, bool* other) {
  pb::Reader e(d, n);
  std::string key;
  const uint8_t* vd = nullptr;
  size_t vn = 0;
  uint32_t f, wt;
  while (e.Next(&f, &wt)) {
    if (f == 1 && wt == pb::kLen) {
      if (!e.String(&key)) return false;
    } else if (f == 2 && wt == pb::kLen) {
      if (!e.Len(&vd, &vn)) return false;
    } else if (!e.Skip(wt)) {
      return false;
    }
  }
  if (e.Failed()) return false;
  std::string sval;
  int64_t ival = 0;
  bool has_s = false, has_i = false;
  pb::Reader p(vd, vn);
  while (p.Next(&f, &wt)) {
    uint64_t u;
    if (f == 2 && wt == pb::kVarint) {
      if (!p.Varint(&u)) return false;
      ival = static_cast<int64_t>(u);
      has_i = true;
    } else if (f == 3 && wt == pb::kLen) {
      if (!p.String(&sval)) return false;
      has_s = true;
    } else {
      *other = true;
      if (!p.Skip(wt)) return false;
    }
  }
  if (p.Failed()) return false;
  if (key == "shared_memory_region" && has_s) {
    shm->set = true;
    shm->region = sval;
  } else if (key == "shared_memory_byte_size" && has_i) {
    shm->size = ival;
  } else if (key == "shared_memory_offset" && has_i) {
    shm->offset = ival;
  } else {
    *other = true;
  }
  return true;
}

}  // namespace

// Decodes a ModelInferRequest.  `stream`: the request came over ModelStreamInfer, where the one request
// parameter a streaming client commonly sets (triton_enable_empty_final_response) is understood.
bool NativeExecutor::ParseGrpc(const std::string& request, bool stream, Request* out, std::shared_ptr<Model>* model_out) {
  Request& req = *out;
  req.t_arrive = NowNs();
  std::vector<std::pair<const uint8_t*, size_t>> raws;
  pb::Reader r(request);
  uint32_t f, wt;
  bool other = false;
  while (r.Next(&f, &wt)) {
    if (f == 1 && wt == pb::kLen) {
      if (!r.String(&req.model)) return false;
      // cheap early out: most servers hold many models without a native kernel
      if (!FindModel(req.model, std::string())) return false;
    } else if (f == 2 && wt == pb::kLen) {
      if (!r.String(&req.version)) return false;
    } else if (f == 3 && wt == pb::kLen) {
      if (!r.String(&req.id)) return false;
    } else if (f == 4 && wt == pb::kLen) {
      // map<string, InferParameter> entry.  Understood: the sequence controls (numeric correlation id, start and
      // end flags) and, on streams, the final-response switch; any other key sends the request down the general path.
      const uint8_t* d;
      size_t n;
      if (!r.Len(&d, &n)) return false;
      pb::Reader e(d, n);
      std::string key;
      const uint8_t* vd = nullptr;
      size_t vn = 0;
      uint32_t ef, ewt;
      while (e.Next(&ef, &ewt)) {
        if (ef == 1 && ewt == pb::kLen) {
          if (!e.String(&key)) return false;
        } else if (ef == 2 && ewt == pb::kLen) {
          if (!e.Len(&vd, &vn)) return false;
        } else if (!e.Skip(ewt)) {
          return false;
        }
      }
      if (e.Failed()) return false;
      // InferParameter: bool_param = 1, int64_param = 2, uint64_param = 5 (strings and doubles are not ours)
      bool is_bool = false, is_int = false, flag = false;
      uint64_t number = 0;
      pb::Reader p(vd, vn);
      while (p.Next(&ef, &ewt)) {
        uint64_t u;
        if (ewt != pb::kVarint || !p.Varint(&u)) return false;
        if (ef == 1) {
          is_bool = true;
          flag = u != 0;
        } else if (ef == 2 || ef == 5) {
          is_int = true;
          number = u;
        } else {
          return false;
        }
      }
      if (p.Failed() || is_bool == is_int) return false;  // exactly one value (an absent one reads as neither)
      if (key == "triton_enable_empty_final_response" && is_bool && stream) {
        req.want_final = flag;
      } else if (key == "sequence_id" && is_int) {
        if (number >> 63) return false;  // a negative int64: the general path refuses it
        req.sequence_id = number;
        req.has_sequence = true;
      } else if (key == "sequence_start" && is_bool) {
        req.sequence_start = flag;
        req.has_sequence = true;
      } else if (key == "sequence_end" && is_bool) {
        req.sequence_end = flag;
        req.has_sequence = true;
      } else {
        return false;
      }
    } else if (f == 5 && wt == pb::kLen) {
      const uint8_t* d;
      size_t n;
      if (!r.Len(&d, &n)) return false;
      Request::Input in;
      pb::Reader t(d, n);
      uint32_t tf, twt;
      while (t.Next(&tf, &twt)) {
        if (tf == 1 && twt == pb::kLen) {
          if (!t.String(&in.name)) return false;
        } else if (tf == 2 && twt == pb::kLen) {
          if (!t.String(&in.datatype)) return false;
        } else if (tf == 3 && twt == pb::kLen) {  // packed int64
          const uint8_t* sd;
          size_t sn;
          if (!t.Len(&sd, &sn)) return false;
          pb::Reader sr(sd, sn);
          while (!sr.Done()) {
            uint64_t v;
            if (!sr.Varint(&v)) return false;
            in.shape.push_back(static_cast<int64_t>(v));
          }
        } else if (tf == 3 && twt == pb::kVarint) {
          uint64_t v;
          if (!t.Varint(&v)) return false;
          in.shape.push_back(static_cast<int64_t>(v));
        } else if (tf == 4 && twt == pb::kLen) {
          const uint8_t* pd;
          size_t pn;
          if (!t.Len(&pd, &pn) || !ShmParam(pd, pn, &in.shm, &other)) return false;
        } else {
          return false;  // typed contents and anything newer: general path
        }
      }
      if (t.Failed() || other) return false;
      req.inputs.push_back(std::move(in));
    } else if (f == 6 && wt == pb::kLen) {
      const uint8_t* d;
      size_t n;
      if (!r.Len(&d, &n)) return false;
      Request::Output o;
      pb::Reader t(d, n);
      uint32_t tf, twt;
      while (t.Next(&tf, &twt)) {
        if (tf == 1 && twt == pb::kLen) {
          if (!t.String(&o.name)) return false;
        } else if (tf == 2 && twt == pb::kLen) {
          const uint8_t* pd;
          size_t pn;
          if (!t.Len(&pd, &pn) || !ShmParam(pd, pn, &o.shm, &other)) return false;
        } else {
          return false;
        }
      }
      if (t.Failed() || other) return false;
      req.outputs.push_back(std::move(o));
    } else if (f == 7 && wt == pb::kLen) {
      const uint8_t* d;
      size_t n;
      if (!r.Len(&d, &n)) return false;
      raws.emplace_back(d, n);
    } else {
      return false;
    }
  }
  if (r.Failed() || req.model.empty()) return false;
  std::shared_ptr<Model> model = FindModel(req.model, req.version);
  if (!model) return false;
  // raw contents belong to the inputs that are not in shared memory, in order
  size_t next_raw = 0;
  for (auto& in : req.inputs) {
    if (in.shm.set) continue;
    if (next_raw >= raws.size()) return false;
    in.data = raws[next_raw].first;
    in.size = raws[next_raw].second;
    in.has_data = true;
    ++next_raw;
  }
  if (next_raw != raws.size()) return false;
  *model_out = std::move(model);
  return true;
}

namespace {

void BoolParam(pb::Writer* w, uint32_t field, const char* key, bool value) {
  std::string entry, body;
  pb::Writer ew(&entry), vw(&body);
  vw.Bool(1, value);
  ew.String(1, key);
  ew.Message(2, body);
  w->Message(field, entry);
}

}  // namespace

// ModelInferResponse of the results in `req`; `final_flag`: -1 none, 0 / 1 = triton_final_response parameter.
void NativeExecutor::BuildGrpcResponse(const Request& req, const std::vector<Result>& results, int final_flag,
                                       std::string* response) {
  response->clear();
  response->reserve(256);
  pb::Writer w(response);
  w.String(1, req.model);
  w.String(2, req.version_used);
  if (!req.id.empty()) w.String(3, req.id);
  if (final_flag >= 0) BoolParam(&w, 4, "triton_final_response", final_flag != 0);
  std::string scratch, entry, value;
  for (const auto& res : results) {
    scratch.clear();
    pb::Writer t(&scratch);
    t.String(1, res.name);
    t.String(2, res.datatype);
    size_t packed = 0;
    for (int64_t v : res.shape) packed += pb::Writer::VarintSize(static_cast<uint64_t>(v));
    t.LenHeader(3, packed);
    for (int64_t v : res.shape) t.Varint(static_cast<uint64_t>(v));
    if (res.shm.set) {
      auto param = [&](const char* key, const std::string* s, int64_t i) {
        entry.clear();
        value.clear();
        pb::Writer ew(&entry), vw(&value);
        if (s != nullptr) vw.String(3, *s);
        else vw.Int(2, i);
        ew.String(1, key);
        ew.Message(2, value);
        t.Message(4, entry);
      };
      param("shared_memory_region", &res.shm.region, 0);
      param("shared_memory_byte_size", nullptr, res.shm.size);
      if (res.shm.offset) param("shared_memory_offset", nullptr, res.shm.offset);
    }
    w.Message(5, scratch);
  }
  for (const auto& res : results)
    if (!res.shm.set) w.Bytes(6, res.bytes.data(), res.bytes.size());
}

bool NativeExecutor::TryGrpc(const std::string& request, std::string* response) {
  if (!enabled_.load(std::memory_order_relaxed)) return false;
  Request req;
  std::shared_ptr<Model> model;
  if (!ParseGrpc(request, false, &req, &model)) return false;
  if (model->kernel == kRepeat) return false;  // decoupled: streams only (the general path words the refusal)
  if (!Run(&req, model)) {
    declined_.fetch_add(1, std::memory_order_relaxed);
    return false;
  }
  const uint64_t t2 = NowNs();
  BuildGrpcResponse(req, req.results, -1, response);
  {
    int batch = 1;
    if (model->spec.max_batch > 0 && !req.results.empty() && !req.results[0].shape.empty())
      batch = static_cast<int>(req.results[0].shape[0]);
    std::lock_guard<std::mutex> lk(model->stats_mu);
    model->stats.by_batch[batch].output_ns += NowNs() - t2;
  }
  handled_.fetch_add(1, std::memory_order_relaxed);
  return true;
}

// The decoupled kernel: IN[n], DELAY[n] (all zero here), WAIT[1] (zero) -> n responses OUT = [IN[i]], IDX = [i].
bool NativeExecutor::RunRepeat(Request* req, const std::shared_ptr<Model>& model, std::vector<std::string>* responses) {
  const Request::Input *in = nullptr, *delay = nullptr, *wait = nullptr;
  if (req->has_sequence) return false;
  for (const auto& t : req->inputs) {
    const Request::Input** slot = t.name == "IN" ? &in : t.name == "DELAY" ? &delay : t.name == "WAIT" ? &wait : nullptr;
    if (slot == nullptr || *slot != nullptr || t.shm.set || !t.has_data || t.shape.size() != 1 || t.shape[0] < 0) return false;
    *slot = &t;
  }
  if (in == nullptr || delay == nullptr || wait == nullptr) return false;
  if (in->datatype != "INT32" || delay->datatype != "UINT32" || wait->datatype != "UINT32") return false;
  const size_t n = static_cast<size_t>(in->shape[0]);
  if (n > (1u << 20) || delay->shape[0] != in->shape[0] || wait->shape[0] != 1) return false;
  if (in->size != 4 * n || delay->size != 4 * n || wait->size != 4) return false;
  for (size_t i = 0; i < delay->size; ++i)
    if (delay->data[i] != 0) return false;  // a response that has to wait belongs on a thread that may sleep
  for (size_t i = 0; i < 4; ++i)
    if (wait->data[i] != 0) return false;
  bool want_out = req->outputs.empty(), want_idx = req->outputs.empty();
  for (const auto& o : req->outputs) {
    if (o.shm.set) return false;
    if (o.name == "OUT" && !want_out) want_out = true;
    else if (o.name == "IDX" && !want_idx) want_idx = true;
    else return false;
  }
  req->version_used = model->spec.version;
  const uint64_t t0 = NowNs();
  responses->clear();
  responses->reserve(n + 1);
  std::vector<Result> results;
  std::string inner;
  for (size_t i = 0; i < n; ++i) {
    results.clear();
    if (want_out) {
      Result r;
      r.name = "OUT";
      r.datatype = "INT32";
      r.shape = {1};
      r.bytes.assign(reinterpret_cast<const char*>(in->data) + 4 * i, 4);
      results.push_back(std::move(r));
    }
    if (want_idx) {
      Result r;
      r.name = "IDX";
      r.datatype = "UINT32";
      r.shape = {1};
      const uint32_t idx = static_cast<uint32_t>(i);
      r.bytes.assign(reinterpret_cast<const char*>(&idx), 4);
      results.push_back(std::move(r));
    }
    BuildGrpcResponse(*req, results, req->want_final ? 0 : -1, &inner);
    std::string wrapped;
    pb::Writer w(&wrapped);
    w.Message(2, inner);  // ModelStreamInferResponse.infer_response
    responses->push_back(std::move(wrapped));
  }
  if (req->want_final) {
    results.clear();
    BuildGrpcResponse(*req, results, 1, &inner);
    std::string wrapped;
    pb::Writer w(&wrapped);
    w.Message(2, inner);
    responses->push_back(std::move(wrapped));
  }
  {
    std::lock_guard<std::mutex> lk(model->stats_mu);
    Stats& s = model->stats;
    ++s.requests;
    ++s.inferences;
    s.last_ms = static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::milliseconds>(
                                          std::chrono::system_clock::now().time_since_epoch()).count());
    BatchStats& bs = s.by_batch[1];
    ++bs.count;
    bs.infer_ns += NowNs() - t0;
    s.queue_ns += t0 - req->t_arrive;
  }
  return true;
}

bool NativeExecutor::TryGrpcStream(const std::string& request, std::vector<std::string>* responses) {
  if (!enabled_.load(std::memory_order_relaxed)) return false;
  Request req;
  std::shared_ptr<Model> model;
  if (!ParseGrpc(request, true, &req, &model)) return false;
  if (model->kernel == kRepeat) {
    if (!RunRepeat(&req, model, responses)) {
      declined_.fetch_add(1, std::memory_order_relaxed);
      return false;
    }
    handled_.fetch_add(1, std::memory_order_relaxed);
    return true;
  }
  if (!Run(&req, model)) {
    declined_.fetch_add(1, std::memory_order_relaxed);
    return false;
  }
  std::string inner;
  // a one-response model answers a stream with its response marked final when the marker was asked for
  BuildGrpcResponse(req, req.results, req.want_final ? 1 : -1, &inner);
  std::string wrapped;
  pb::Writer w(&wrapped);
  w.Message(2, inner);
  responses->clear();
  responses->push_back(std::move(wrapped));
  handled_.fetch_add(1, std::memory_order_relaxed);
  return true;
}

// ---------------------------------------------------------------------------------------------
// BYTES tensors in device shared memory
// ---------------------------------------------------------------------------------------------
Error NativeExecutor::EnableDeviceBytes(const DevicePipe& pipe) {
  if (pipe.submit == nullptr || pipe.wait == nullptr || pipe.pipe == nullptr || pipe.max_requests <= 0 ||
      pipe.max_count <= 0 || pipe.max_stream_bytes == 0)
    return Error("device BYTES path: incomplete pipeline description");
  DisableDeviceBytes();
  pipe_ = pipe;
  {
    std::lock_guard<std::mutex> lk(q_mu_);
    stop_ = false;
  }
  batch_thread_ = std::thread([this] { DeviceLoop(); });
  pipe_on_.store(true);
  return Error::Success();
}

void NativeExecutor::DisableDeviceBytes() {
  pipe_on_.store(false);  // no new jobs; a request that saw `true` a moment ago is still answered below
  {
    std::lock_guard<std::mutex> lk(q_mu_);
    stop_ = true;
  }
  q_cv_.notify_all();
  if (batch_thread_.joinable()) batch_thread_.join();
}

Error NativeExecutor::RegisterDeviceRegion(const std::string& name, uint64_t address, uint64_t size, int device) {
  if (address == 0 || address + size < address) return Error("device region '" + name + "': invalid address range");
  auto r = std::make_shared<DeviceRegion>();
  r->address = address;
  r->size = size;
  r->device = device;
  std::unique_lock<std::shared_mutex> lk(mu_);
  device_regions_[name] = std::move(r);
  return Error::Success();
}

bool NativeExecutor::UnregisterDeviceRegion(const std::string& name, int timeout_ms) {
  std::vector<std::pair<std::string, std::shared_ptr<DeviceRegion>>> going;
  {
    std::unique_lock<std::shared_mutex> lk(mu_);
    if (name.empty()) {
      for (auto& kv : device_regions_) going.emplace_back(kv.first, kv.second);
      device_regions_.clear();
    } else {
      auto it = device_regions_.find(name);
      if (it != device_regions_.end()) {
        going.emplace_back(it->first, it->second);
        device_regions_.erase(it);
      }
    }
  }
  // leases are only taken under mu_ from the map: nothing new can start on these regions
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms < 0 ? 0 : timeout_ms);
  bool idle = true;
  for (auto& kv : going) {
    while (kv.second->inflight.load(std::memory_order_acquire) > 0) {
      if (std::chrono::steady_clock::now() >= deadline) {
        idle = false;
        break;
      }
      std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
  }
  if (!idle) {
    std::unique_lock<std::shared_mutex> lk(mu_);
    for (auto& kv : going) device_regions_.emplace(kv.first, kv.second);  // still serving requests: stays registered
  }
  return idle;
}

void NativeExecutor::Quiesce() {
  {
    std::unique_lock<std::mutex> lk(q_mu_);
    idle_cv_.wait(lk, [this] { return queue_.empty() && running_ == 0; });
  }
  QuiesceGraphModels();
}

// Validates a request for a BYTES model as far as the general path would before it queues the request
// for the device, and describes the job.  False = not ours.
bool NativeExecutor::PrepareDeviceJob(Request* req, const std::shared_ptr<Model>& model, DeviceJob* job) {
  const ModelSpec& spec = model->spec;
  job->t0 = NowNs();
  const size_t arity = spec.inputs.size();
  if (req->has_sequence) return false;
  if (req->inputs.size() != arity || req->outputs.size() != arity) return false;  // outputs must name their regions
  const Request::Input* ordered[2] = {nullptr, nullptr};
  const size_t lead = spec.max_batch > 0 ? 1 : 0;
  uint64_t count = 0;
  for (const auto& in : req->inputs) {
    size_t slot = arity;
    for (size_t k = 0; k < arity; ++k)
      if (spec.inputs[k].name == in.name) slot = k;
    if (slot == arity || ordered[slot] != nullptr) return false;
    const TensorSpec& ts = spec.inputs[slot];
    if (in.datatype != "BYTES" || !in.shm.set || in.has_data) return false;
    if (in.shape.size() != ts.dims.size() + lead) return false;
    uint64_t n = 1;
    for (size_t d = 0; d < in.shape.size(); ++d) {
      const int64_t v = in.shape[d];
      if (v < 1 || v > pipe_.max_count) return false;
      if (d >= lead && ts.dims[d - lead] != -1 && ts.dims[d - lead] != v) return false;
      n *= static_cast<uint64_t>(v);
      if (n > static_cast<uint64_t>(pipe_.max_count)) return false;
    }
    if (lead && in.shape[0] > spec.max_batch) return false;
    if (count != 0 && (n != count || in.shape != req->inputs[0].shape)) return false;
    count = n;
    ordered[slot] = &in;
  }
  job->count = static_cast<uint32_t>(count);
  job->batch = lead ? static_cast<int>(ordered[0]->shape[0]) : 1;
  job->op = model->kernel == kBytesAddSub ? 0 : 1;
  job->n_in = job->n_out = static_cast<int>(arity);
  const Request::Output* wanted[2] = {nullptr, nullptr};
  for (const auto& o : req->outputs) {
    size_t slot = arity;
    for (size_t k = 0; k < arity; ++k)
      if (spec.outputs[k].name == o.name) slot = k;
    if (slot == arity || wanted[slot] != nullptr || !o.shm.set) return false;
    wanted[slot] = &o;
  }
  {
    std::shared_lock<std::shared_mutex> lk(mu_);
    auto lease = [&](const ShmRef& ref, bool input, uint64_t* address) {
      auto it = device_regions_.find(ref.region);
      if (it == device_regions_.end() || it->second->device != pipe_.device) return false;
      const DeviceRegion& r = *it->second;
      if (ref.offset < 0 || ref.size < (input ? 1 : 0)) return false;
      const uint64_t off = static_cast<uint64_t>(ref.offset), size = static_cast<uint64_t>(ref.size);
      if (off + size < off || off + size > r.size) return false;
      if (input && size > pipe_.max_stream_bytes) return false;
      *address = r.address + off;
      it->second->inflight.fetch_add(1, std::memory_order_acq_rel);
      job->leases.push_back(it->second);
      return true;
    };
    bool ok = true;
    for (size_t k = 0; ok && k < arity; ++k) {
      ok = lease(ordered[k]->shm, true, &job->in_ptrs[k]);
      job->in_bytes[k] = static_cast<uint64_t>(ordered[k]->shm.size);
    }
    for (size_t k = 0; ok && k < arity; ++k) {
      ok = lease(wanted[k]->shm, false, &job->out_ptrs[k]);
      job->out_caps[k] = static_cast<uint64_t>(wanted[k]->shm.size);
    }
    if (!ok) {
      for (auto& r : job->leases) r->inflight.fetch_sub(1, std::memory_order_acq_rel);
      job->leases.clear();
      return false;
    }
  }
  req->results.clear();
  for (const auto& o : req->outputs) {  // in the order the request names them
    Result r;
    r.name = o.name;
    r.datatype = "BYTES";
    r.shape = ordered[0]->shape;
    r.shm = o.shm;
    req->results.push_back(std::move(r));
  }
  req->version_used = spec.version;
  job->t1 = NowNs();
  return true;
}

void NativeExecutor::FinishDeviceJob(DeviceJob* job, bool ok) {
  std::string payload;
  HeaderList headers;
  std::vector<std::string> parts;
  if (ok) {
    if (job->http_done) BuildHttpResponse(job->req, &job->req.results, &headers, &parts);
    else BuildGrpcResponse(job->req, job->req.results, -1, &payload);
    const uint64_t now = NowNs();
    std::lock_guard<std::mutex> lk(job->model->stats_mu);
    Stats& s = job->model->stats;
    ++s.requests;
    s.inferences += static_cast<uint64_t>(job->batch);
    s.queue_ns += job->t0 - job->req.t_arrive;
    s.last_ms = static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::milliseconds>(
                                          std::chrono::system_clock::now().time_since_epoch()).count());
    BatchStats& bs = s.by_batch[job->batch];
    ++bs.count;
    bs.input_ns += job->t1 - job->t0;
    bs.infer_ns += now - job->t1;
    handled_.fetch_add(1, std::memory_order_relaxed);
  } else {
    payload = std::move(job->request_bytes);
    declined_.fetch_add(1, std::memory_order_relaxed);
  }
  // the regions are free again before the caller hears about it: it may unregister them right away
  for (auto& r : job->leases) r->inflight.fetch_sub(1, std::memory_order_acq_rel);
  job->leases.clear();
  if (job->http_done) {
    if (!ok) parts.assign(1, std::move(payload));  // the request body, for the general path
    job->http_done(ok, std::move(headers), std::move(parts));
  } else {
    job->done(ok, std::move(payload));
  }
}

void NativeExecutor::DeviceLoop() {
  const size_t limit = static_cast<size_t>(pipe_.max_requests);
  std::vector<std::unique_ptr<DeviceJob>> batch;
  std::vector<uint64_t> in_ptrs, in_bytes, out_ptrs, out_caps, produced(2 * limit);
  std::vector<uint32_t> codes(limit);
  for (;;) {
    batch.clear();
    {
      std::unique_lock<std::mutex> lk(q_mu_);
      q_cv_.wait(lk, [this] { return stop_ || !queue_.empty(); });
      if (queue_.empty()) return;  // stopping, and everything has been answered
      // one launch group = requests of one operation and element count, oldest first
      const int op = queue_.front()->op;
      const uint32_t count = queue_.front()->count;
      for (auto it = queue_.begin(); it != queue_.end() && batch.size() < limit;) {
        if ((*it)->op == op && (*it)->count == count) {
          batch.push_back(std::move(*it));
          it = queue_.erase(it);
        } else {
          ++it;
        }
      }
      running_ += batch.size();
    }
    in_ptrs.clear();
    in_bytes.clear();
    out_ptrs.clear();
    out_caps.clear();
    for (const auto& job : batch) {
      for (int k = 0; k < job->n_in; ++k) {
        in_ptrs.push_back(job->in_ptrs[k]);
        in_bytes.push_back(job->in_bytes[k]);
      }
      for (int k = 0; k < job->n_out; ++k) {
        out_ptrs.push_back(job->out_ptrs[k]);
        out_caps.push_back(job->out_caps[k]);
      }
    }
    const int n = static_cast<int>(batch.size());
    int rc = pipe_.submit(pipe_.pipe, batch[0]->op, n, static_cast<int>(batch[0]->count), in_ptrs.data(), in_bytes.data(),
                          out_ptrs.data(), out_caps.data(), pipe_.stream);
    if (rc == 0) rc = pipe_.wait(pipe_.pipe, n, codes.data(), produced.data());
    for (int i = 0; i < n; ++i) FinishDeviceJob(batch[static_cast<size_t>(i)].get(), rc == 0 && codes[static_cast<size_t>(i)] == 0);
    {
      std::lock_guard<std::mutex> lk(q_mu_);
      running_ -= batch.size();
      if (queue_.empty() && running_ == 0) idle_cv_.notify_all();
    }
  }
}

NativeExecutor::Outcome NativeExecutor::TryGrpcAsync(std::string* request, std::string* response, Completion done) {
  if (!enabled_.load(std::memory_order_relaxed)) return Outcome::kDeclined;
  Request req;
  std::shared_ptr<Model> model;
  if (!ParseGrpc(*request, false, &req, &model)) return Outcome::kDeclined;
  if (model->kernel == kRepeat) return Outcome::kDeclined;
  if (model->kernel == kGraph) return SubmitGraph(&req, model, request, std::move(done), HttpCompletion());
  if (model->kernel == kBytesAddSub || model->kernel == kBytesIdentity) {
    auto job = std::make_unique<DeviceJob>();
    if (!pipe_on_.load(std::memory_order_acquire) || !PrepareDeviceJob(&req, model, job.get())) {
      declined_.fetch_add(1, std::memory_order_relaxed);
      return Outcome::kDeclined;
    }
    req.inputs.clear();  // pointed into the request bytes, which move now
    job->req = std::move(req);
    job->model = std::move(model);
    job->request_bytes = std::move(*request);
    job->done = std::move(done);
    {
      std::lock_guard<std::mutex> lk(q_mu_);
      if (stop_) {  // the batch thread is going away: hand the request back untouched
        *request = std::move(job->request_bytes);
        for (auto& r : job->leases) r->inflight.fetch_sub(1, std::memory_order_acq_rel);
        return Outcome::kDeclined;
      }
      queue_.push_back(std::move(job));
    }
    q_cv_.notify_one();
    return Outcome::kDeferred;
  }
  if (!Run(&req, model)) {
    declined_.fetch_add(1, std::memory_order_relaxed);
    return Outcome::kDeclined;
  }
  const uint64_t t2 = NowNs();
  BuildGrpcResponse(req, req.results, -1, response);
  {
    int batch = 1;
    if (model->spec.max_batch > 0 && !req.results.empty() && !req.results[0].shape.empty())
      batch = static_cast<int>(req.results[0].shape[0]);
    std::lock_guard<std::mutex> lk(model->stats_mu);
    model->stats.by_batch[batch].output_ns += NowNs() - t2;
  }
  handled_.fetch_add(1, std::memory_order_relaxed);
  return Outcome::kAnswered;
}

// ---------------------------------------------------------------------------------------------
// GPU models served from captured graphs
// ---------------------------------------------------------------------------------------------
struct NativeExecutor::GraphJob {
  Request req;  // results already describe the answer; the scatter copy fills the region
  std::shared_ptr<Model> model;
  std::string request_bytes;  // handed back when the launch fails
  Completion done;            // gRPC caller ...
  HttpCompletion http_done;   // ... or HTTP caller
  int batch = 1;
  uint64_t in_ptr = 0, out_ptr = 0;
  std::vector<std::shared_ptr<DeviceRegion>> leases;
  // tensors inside the messages: a piece of the runner's host arena (input bytes first, output rows behind)
  std::shared_ptr<GraphRunner> runner;
  uint64_t host_ptr = 0, host_bytes = 0, host_out = 0, host_out_bytes = 0;
  uint8_t* sys_out = nullptr;  // output window in a system shared-memory region (held by req.held): filled from host_out
  uint64_t stage_bytes = 0;    // input bytes the staging thread moves to the device arena ...
  uint64_t stage_src = 0;      // ... from here (the request's piece of the host arena)
  uint64_t t_queued = 0;
  uint64_t gap_ns = ~0ull >> 2;  // since the previous arrival for this model
};

class NativeExecutor::GraphRunner {
 public:
  GraphRunner(NativeExecutor* owner, const GraphPipe& pipe) : owner_(owner), pipe_(pipe) {
    for (size_t i = 0; i < pipe_.streams.size(); ++i) {
      instances_.push_back(std::make_unique<Instance>());
      instances_[i]->stream = pipe_.streams[i];
    }
    for (const auto& g : pipe_.graphs) instances_[static_cast<size_t>(g.instance)]->graphs.push_back(g);
    for (auto& inst : instances_)
      std::sort(inst->graphs.begin(), inst->graphs.end(), [](const GraphPipe::Graph& a, const GraphPipe::Graph& b) { return a.batch < b.batch; });
    cap_ = pipe_.max_inflight > 0 ? pipe_.max_inflight : 2 * static_cast<int>(instances_.size());
    if (pipe_.host_arena != 0 && pipe_.host_chunk_bytes > 0)
      chunk_used_.assign(static_cast<size_t>(pipe_.host_arena_bytes / pipe_.host_chunk_bytes), false);
    batch_thread_ = std::thread([this] { BatchLoop(); });
    if (pipe_.device_arena != 0 && pipe_.copy_stream != nullptr && pipe_.host_arena != 0)
      stage_thread_ = std::thread([this] { StageLoop(); });
    for (size_t i = 0; i < instances_.size(); ++i)
      instances_[i]->completer = std::thread([this, i] { CompleteLoop(static_cast<int>(i)); });
  }
  ~GraphRunner() { Stop(); }
  const GraphPipe& pipe() const { return pipe_; }

  // False: the runner is stopping; `*job` is untouched.
  bool Submit(std::unique_ptr<GraphJob>* job) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (stop_ || stage_stop_) return false;
      const uint64_t now = NowNs();
      (*job)->t_queued = now;
      (*job)->gap_ns = now - last_arrival_ns_;
      last_arrival_ns_ = now;
      if ((*job)->stage_bytes != 0 && stage_thread_.joinable()) {
        images_staging_ += (*job)->batch;
        stage_queue_.push_back(std::move(*job));
        stage_cv_.notify_one();
        return true;
      }
      queue_.push_back(std::move(*job));
    }
    cv_.notify_all();
    return true;
  }

  // Answers everything queued or running, then joins the threads.
  void Stop() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (stop_ && !batch_thread_.joinable()) return;
      stage_stop_ = true;  // first the staging thread: what it holds still reaches the batch queue
    }
    stage_cv_.notify_all();
    if (stage_thread_.joinable()) stage_thread_.join();
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    if (batch_thread_.joinable()) batch_thread_.join();
    {
      std::lock_guard<std::mutex> lk(mu_);
      completers_stop_ = true;
    }
    for (auto& inst : instances_) inst->cv.notify_all();
    for (auto& inst : instances_)
      if (inst->completer.joinable()) inst->completer.join();
    for (auto& inst : instances_) {
      for (void* e : inst->free_events) pipe_.event_destroy(e);
      inst->free_events.clear();
    }
    if (stage_event_ != nullptr) pipe_.event_destroy(stage_event_);
    stage_event_ = nullptr;
  }

  void WaitIdle() {
    std::unique_lock<std::mutex> lk(mu_);
    idle_cv_.wait(lk, [this] { return queue_.empty() && running_ == 0 && stage_queue_.empty() && staging_ == 0; });
  }

  void Counters(GraphCounters* out) {
    std::lock_guard<std::mutex> lk(mu_);
    *out = counters_;
    std::lock_guard<std::mutex> alk(arena_mu_);
    out->wire_requests = wire_requests_;
    out->arena_full = arena_full_;
  }

  // A piece of the host arena of at least `bytes` (whole chunks, first fit); 0 when nothing fits.
  uint64_t AllocHost(uint64_t bytes) {
    const uint64_t unit = pipe_.host_chunk_bytes;
    const size_t want = static_cast<size_t>((bytes + unit - 1) / unit);
    std::lock_guard<std::mutex> lk(arena_mu_);
    const size_t n = chunk_used_.size();
    if (want == 0 || want > n) return 0;
    // next fit: start behind the previous allocation, requests are of similar size and short lived
    for (size_t tried = 0, at = next_chunk_; tried < n; ++tried, at = (at + 1) % n) {
      if (at + want > n) continue;
      size_t k = 0;
      while (k < want && !chunk_used_[at + k]) ++k;
      if (k == want) {
        for (k = 0; k < want; ++k) chunk_used_[at + k] = true;
        next_chunk_ = (at + want) % n;
        ++wire_requests_;
        return pipe_.host_arena + static_cast<uint64_t>(at) * unit;
      }
    }
    ++arena_full_;
    return 0;
  }
  void FreeHost(uint64_t ptr, uint64_t bytes) {
    const uint64_t unit = pipe_.host_chunk_bytes;
    const size_t at = static_cast<size_t>((ptr - pipe_.host_arena) / unit), want = static_cast<size_t>((bytes + unit - 1) / unit);
    std::lock_guard<std::mutex> lk(arena_mu_);
    for (size_t k = 0; k < want && at + k < chunk_used_.size(); ++k) chunk_used_[at + k] = false;
  }

 private:
  struct Launched {
    void* event = nullptr;
    std::vector<std::unique_ptr<GraphJob>> jobs;
    int total = 0;
    uint64_t t_exec = 0, t_enq = 0;
  };
  struct Instance {
    std::vector<GraphPipe::Graph> graphs;  // ascending batch
    void* stream = nullptr;
    int busy = 0;  // batches enqueued, not completed
    std::deque<Launched> launched;
    std::condition_variable cv;
    std::vector<void*> free_events;
    std::thread completer;
  };
  using Clock = std::chrono::steady_clock;

  int MinBusy() const {
    int m = instances_[0]->busy;
    for (const auto& inst : instances_) m = std::min(m, inst->busy);
    return m;
  }

  // Input bytes of requests that arrived inside messages (or in system shared memory) go from the host arena
  // to the device arena here, everything that arrived since the previous round in one copy-kernel launch on
  // the copy stream; the requests join the batch queue once their bytes are on the device.
  void StageLoop() {
    if (pipe_.set_device != nullptr) pipe_.set_device(pipe_.device);
    std::vector<const void*> srcs;
    std::vector<void*> dsts;
    std::vector<uint64_t> bytes;
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      stage_cv_.wait(lk, [this] { return stage_stop_ || !stage_queue_.empty(); });
      if (stage_queue_.empty()) return;
      std::vector<std::unique_ptr<GraphJob>> round;
      while (!stage_queue_.empty()) {
        round.push_back(std::move(stage_queue_.front()));
        stage_queue_.pop_front();
      }
      staging_ += round.size();
      long long round_images = 0;
      for (const auto& job : round) round_images += job->batch;
      lk.unlock();
      srcs.clear();
      dsts.clear();
      bytes.clear();
      for (auto& job : round) {
        srcs.push_back(reinterpret_cast<const void*>(static_cast<uintptr_t>(job->stage_src)));
        dsts.push_back(reinterpret_cast<void*>(static_cast<uintptr_t>(pipe_.device_arena + (job->host_ptr - pipe_.host_arena))));
        bytes.push_back(job->stage_bytes);
      }
      int rc = 0;
      if (stage_event_ == nullptr) rc = pipe_.event_create(&stage_event_);
      owner_->EnterLaunch();
      if (rc == 0 && pipe_.memcpy_async != nullptr) {
        for (size_t i = 0; i < round.size() && rc == 0; ++i)
          rc = pipe_.memcpy_async(dsts[i], srcs[i], bytes[i], 1 /* host to device */, pipe_.copy_stream);
      } else if (rc == 0) {
        rc = pipe_.batched_copy(srcs.data(), dsts.data(), bytes.data(), static_cast<int>(round.size()), pipe_.copy_stream);
      }
      if (rc == 0) rc = pipe_.event_record(stage_event_, pipe_.copy_stream);
      owner_->LeaveLaunch();
      if (rc == 0) rc = pipe_.event_sync(stage_event_);
      // the stream is idle here: every 16th round a synchronisation of the stream itself, which is what makes
      // the runtime let go of its per-copy bookkeeping
      if (rc == 0 && pipe_.stream_sync != nullptr && (++stage_rounds_ & 15) == 0) rc = pipe_.stream_sync(pipe_.copy_stream);
      if (rc != 0) {  // the requests take the general path
        for (auto& job : round) owner_->FinishGraphJob(job.get(), false, 0, 0, 0, false);
        lk.lock();
        staging_ -= round.size();
        images_staging_ -= round_images;
        if (queue_.empty() && running_ == 0 && stage_queue_.empty() && staging_ == 0) idle_cv_.notify_all();
        continue;
      }
      lk.lock();
      staging_ -= round.size();
      images_staging_ -= round_images;
      for (auto& job : round) {
        job->in_ptr = pipe_.device_arena + (job->host_ptr - pipe_.host_arena);
        queue_.push_back(std::move(job));
      }
      cv_.notify_all();
    }
  }

  // The serving policy of the dynamic batcher (server/batcher.py: _next_group), on one lock:
  // everything already queued joins the batch; while the GPU has no free slot waiting costs nothing, so
  // arrivals keep being absorbed; with a slot the batch goes at once unless the arrivals look like a burst
  // (then up to burst_window) or every instance still has work (then up to fill_window, looked at again
  // every 200 us); with few images around the batch is capped so that every instance gets one.
  void BatchLoop() {
    if (pipe_.set_device != nullptr) pipe_.set_device(pipe_.device);
    const auto us = [](int v) { return std::chrono::microseconds(v); };
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      cv_.wait(lk, [this] { return stop_ || !queue_.empty(); });
      if (queue_.empty()) return;  // stopping, and everything has been launched
      std::vector<std::unique_ptr<GraphJob>> group;
      group.push_back(std::move(queue_.front()));
      queue_.pop_front();
      int total = group[0]->batch;
      const auto t_first = Clock::now();
      auto deadline = t_first;
      bool have_slot = false;
      int cap = pipe_.max_batch;
      if (pipe_.adaptive_cap) {
        // everything the server holds: on the GPU, queued here, and on its way through the staging copy (left
        // out, a load of large requests looks like a low-concurrency one as soon as its batches are small -
        // and the cap then keeps them small)
        const double seen = static_cast<double>(images_on_gpu_ + images_staging_) +
                            static_cast<double>(queue_.size() + 1) * group[0]->batch;
        // the largest count of the recent past: it gives way by a tenth per 300 ms, not per batch - a load of large
        // requests holds few images at the server between its bursts, and a peak that forgot the bursts within
        // 25 ms let the cap flap on and off under it (64 in flight on the wire: 63k with the cap flapping, 70k without)
        if (seen >= outstanding_peak_) {
          outstanding_peak_ = seen;
          peak_time_ = t_first;
        } else if (t_first - peak_time_ > std::chrono::milliseconds(300)) {
          outstanding_peak_ = std::max(seen, outstanding_peak_ * 0.9);
          peak_time_ = t_first;
        }
        if (outstanding_peak_ <= pipe_.cap_limit * pipe_.max_batch) {
          const int step = std::max(1, group[0]->batch);
          const double per = outstanding_peak_ / (pipe_.cap_depth * static_cast<double>(instances_.size()));
          int c = static_cast<int>((static_cast<long long>(per + step - 1e-9) / step)) * step;
          cap = std::min(pipe_.max_batch, std::max(step, c));
        }
      }
      while (total < cap) {
        if (!queue_.empty()) {
          if (total + queue_.front()->batch > cap) break;  // stays first in line for the next batch
          total += queue_.front()->batch;
          group.push_back(std::move(queue_.front()));
          queue_.pop_front();
          continue;
        }
        if (!have_slot && inflight_ < cap_) {
          ++inflight_;
          have_slot = true;
        }
        if (!have_slot) {
          // GPU full: wait for a slot, absorbing arrivals meanwhile
          // (not woken by stop_: a predicate that stays true would spin here with the lock held while the
          // completers need it to free the slot)
          WaitFor(cv_, lk, us(500), [this] { return !queue_.empty() || inflight_ < cap_; });
          continue;
        }
        const uint64_t now_ns = NowNs();
        if (group.back()->gap_ns < static_cast<uint64_t>(pipe_.burst_gap_us) * 1000ull ||
            (group.size() < last_done_requests_ && now_ns - last_done_ns_ < static_cast<uint64_t>(pipe_.burst_gap_us) * 1000ull))
          deadline = std::max(deadline, t_first + us(pipe_.burst_window_us));
        const bool filling = pipe_.fill_window_us > 0 && MinBusy() >= 1;
        const auto limit = filling ? std::max(deadline, t_first + us(pipe_.fill_window_us)) : deadline;
        const auto now = Clock::now();
        if (now >= limit || stop_) break;
        auto wait = limit - now;
        if (filling && wait > us(200)) wait = us(200);
        const bool woke = WaitFor(cv_, lk, wait, [this] { return stop_ || !queue_.empty(); });
        if (!woke && !filling) break;
      }
      while (!have_slot) {
        if (inflight_ < cap_) {
          ++inflight_;
          have_slot = true;
          break;
        }
        WaitFor(cv_, lk, us(500), [this] { return inflight_ < cap_; });
      }
      running_ += group.size();
      // two idle instances: a batch of several requests runs faster as two halves side by side
      std::vector<std::unique_ptr<GraphJob>> second;
      int second_total = 0;
      if (instances_.size() >= 2 && group.size() >= 2 && total >= pipe_.split_min_batch && inflight_ < cap_) {
        bool idle = true;
        for (const auto& inst : instances_) idle = idle && inst->busy == 0;
        if (idle) {
          int acc = 0;
          size_t keep = 0;
          while (keep < group.size() && !(acc >= (total + 1) / 2 && keep > 0)) acc += group[keep++]->batch;
          if (keep < group.size()) {
            ++inflight_;
            for (size_t i = keep; i < group.size(); ++i) second.push_back(std::move(group[i]));
            group.resize(keep);
            second_total = total - acc;
            total = acc;
          }
        }
      }
      Launch(&lk, std::move(group), total);
      if (!second.empty()) Launch(&lk, std::move(second), second_total);
    }
  }

  // Called with the lock held; releases it around the HIP calls.
  void Launch(std::unique_lock<std::mutex>* lk, std::vector<std::unique_ptr<GraphJob>>&& group, int total) {
    int inst_index = 0;
    for (size_t i = 1; i < instances_.size(); ++i)
      if (instances_[i]->busy < instances_[static_cast<size_t>(inst_index)]->busy) inst_index = static_cast<int>(i);
    Instance& inst = *instances_[static_cast<size_t>(inst_index)];
    ++inst.busy;
    void* event = nullptr;
    if (!inst.free_events.empty()) {
      event = inst.free_events.back();
      inst.free_events.pop_back();
    }
    lk->unlock();
    owner_->EnterLaunch();
    const uint64_t t_exec = NowNs();
    const GraphPipe::Graph* graph = nullptr;
    for (const auto& g : inst.graphs)
      if (g.batch >= total) {
        graph = &g;
        break;
      }
    int rc = graph != nullptr ? 0 : -1;
    if (rc == 0 && event == nullptr) rc = pipe_.event_create(&event);
    if (rc == 0) {
      const size_t n = group.size();
      srcs_.resize(n);
      dsts_.resize(n);
      bytes_.resize(n);
      uint64_t row = 0;
      for (size_t i = 0; i < n; ++i) {
        srcs_[i] = reinterpret_cast<const void*>(static_cast<uintptr_t>(group[i]->in_ptr));
        dsts_[i] = reinterpret_cast<void*>(static_cast<uintptr_t>(graph->in_ptr + row * pipe_.in_row_bytes));
        bytes_[i] = static_cast<uint64_t>(group[i]->batch) * pipe_.in_row_bytes;
        row += static_cast<uint64_t>(group[i]->batch);
      }
      rc = pipe_.batched_copy(srcs_.data(), dsts_.data(), bytes_.data(), static_cast<int>(n), inst.stream);
      if (rc == 0) rc = pipe_.graph_launch(graph->exec, inst.stream);
      if (rc == 0) {
        row = 0;
        for (size_t i = 0; i < n; ++i) {
          srcs_[i] = reinterpret_cast<const void*>(static_cast<uintptr_t>(graph->out_ptr + row * pipe_.out_row_bytes));
          dsts_[i] = reinterpret_cast<void*>(static_cast<uintptr_t>(group[i]->out_ptr));
          bytes_[i] = static_cast<uint64_t>(group[i]->batch) * pipe_.out_row_bytes;
          row += static_cast<uint64_t>(group[i]->batch);
        }
        rc = pipe_.batched_copy(srcs_.data(), dsts_.data(), bytes_.data(), static_cast<int>(n), inst.stream);
      }
      if (rc == 0) rc = pipe_.event_record(event, inst.stream);
    }
    owner_->LeaveLaunch();
    const uint64_t t_enq = NowNs();
    if (rc != 0) {
      // nothing (or not everything) was enqueued: whatever was runs to its end before the regions go back,
      // then the requests take the general path
      if (event != nullptr && pipe_.event_record(event, inst.stream) == 0) pipe_.event_sync(event);
      for (auto& job : group) owner_->FinishGraphJob(job.get(), false, 0, 0, 0, false);
      lk->lock();
      if (event != nullptr) inst.free_events.push_back(event);
      --inst.busy;
      --inflight_;
      running_ -= group.size();
      if (queue_.empty() && running_ == 0 && stage_queue_.empty() && staging_ == 0) idle_cv_.notify_all();
      return;
    }
    lk->lock();
    images_on_gpu_ += total;
    ++counters_.batches;
    counters_.requests += group.size();
    counters_.images += static_cast<uint64_t>(total);
    counters_.padded_images += static_cast<uint64_t>(graph->batch - total);
    Launched l;
    l.event = event;
    l.jobs = std::move(group);
    l.total = total;
    l.t_exec = t_exec;
    l.t_enq = t_enq;
    inst.launched.push_back(std::move(l));
    inst.cv.notify_one();
  }

  void CompleteLoop(int index) {
    if (pipe_.set_device != nullptr) pipe_.set_device(pipe_.device);
    Instance& inst = *instances_[static_cast<size_t>(index)];
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      inst.cv.wait(lk, [&] { return completers_stop_ || !inst.launched.empty(); });
      if (inst.launched.empty()) return;
      Launched l = std::move(inst.launched.front());
      inst.launched.pop_front();
      lk.unlock();
      const int rc = pipe_.event_sync(l.event);
      const uint64_t t_done = NowNs();
      lk.lock();
      --inst.busy;
      --inflight_;
      images_on_gpu_ -= l.total;
      last_done_ns_ = t_done;
      last_done_requests_ = l.jobs.size();
      inst.free_events.push_back(l.event);
      lk.unlock();
      cv_.notify_all();  // a slot is free
      bool first = true;
      for (auto& job : l.jobs) {
        owner_->FinishGraphJob(job.get(), rc == 0, l.t_exec, l.t_enq, t_done, first);
        first = false;
      }
      lk.lock();
      running_ -= l.jobs.size();
      if (queue_.empty() && running_ == 0 && stage_queue_.empty() && staging_ == 0) idle_cv_.notify_all();
    }
  }

  NativeExecutor* owner_;
  GraphPipe pipe_;
  std::vector<std::unique_ptr<Instance>> instances_;
  std::mutex mu_;
  std::condition_variable cv_, idle_cv_;
  std::deque<std::unique_ptr<GraphJob>> queue_;
  bool stop_ = false, completers_stop_ = false;
  int inflight_ = 0, cap_ = 2;
  long long images_on_gpu_ = 0;
  long long images_staging_ = 0;  // in the staging queue or being copied
  uint64_t stage_rounds_ = 0;     // staging thread only
  double outstanding_peak_ = 0;
  Clock::time_point peak_time_{};
  uint64_t last_arrival_ns_ = 0, last_done_ns_ = 0;
  size_t last_done_requests_ = 1, running_ = 0;
  GraphCounters counters_;
  std::thread batch_thread_;
  std::thread stage_thread_;
  std::condition_variable stage_cv_;
  std::deque<std::unique_ptr<GraphJob>> stage_queue_;
  size_t staging_ = 0;
  bool stage_stop_ = false;
  void* stage_event_ = nullptr;
  std::mutex arena_mu_;
  std::vector<bool> chunk_used_;
  size_t next_chunk_ = 0;
  uint64_t wire_requests_ = 0, arena_full_ = 0;
  std::vector<const void*> srcs_;  // scratch of the batch thread
  std::vector<void*> dsts_;
  std::vector<uint64_t> bytes_;
};

std::shared_ptr<NativeExecutor::GraphRunner> NativeExecutor::FindRunner(const std::string& model) {
  std::shared_lock<std::shared_mutex> lk(mu_);
  auto it = runners_.find(model);
  return it == runners_.end() ? nullptr : it->second;
}

Error NativeExecutor::EnableGraphModel(const std::string& model, const GraphPipe& pipe) {
  if (pipe.graph_launch == nullptr || pipe.event_create == nullptr || pipe.event_record == nullptr ||
      pipe.event_sync == nullptr || pipe.event_destroy == nullptr || pipe.batched_copy == nullptr)
    return Error("graph model '" + model + "': incomplete set of entry points");
  if (pipe.streams.empty() || pipe.graphs.empty() || pipe.in_row_bytes == 0 || pipe.out_row_bytes == 0 || pipe.max_batch < 1)
    return Error("graph model '" + model + "': incomplete description");
  std::vector<int> largest(pipe.streams.size(), 0);
  for (const auto& g : pipe.graphs) {
    if (g.instance < 0 || static_cast<size_t>(g.instance) >= pipe.streams.size() || g.batch < 1 || g.exec == nullptr ||
        g.in_ptr == 0 || g.out_ptr == 0)
      return Error("graph model '" + model + "': invalid graph entry");
    largest[static_cast<size_t>(g.instance)] = std::max(largest[static_cast<size_t>(g.instance)], g.batch);
  }
  for (int b : largest)
    if (b < pipe.max_batch) return Error("graph model '" + model + "': every instance needs a graph for the largest batch");
  {
    std::shared_ptr<Model> m = FindModel(model, std::string());
    if (!m || m->kernel != kGraph) return Error("graph model '" + model + "' is not registered with the graph kernel");
  }
  DisableGraphModel(model);
  auto runner = std::make_shared<GraphRunner>(this, pipe);
  std::unique_lock<std::shared_mutex> lk(mu_);
  runners_[model] = std::move(runner);
  return Error::Success();
}

void NativeExecutor::DisableGraphModel(const std::string& model) {
  std::vector<std::shared_ptr<GraphRunner>> going;
  {
    std::unique_lock<std::shared_mutex> lk(mu_);
    if (model.empty()) {
      for (auto& kv : runners_) going.push_back(kv.second);
      runners_.clear();
    } else {
      auto it = runners_.find(model);
      if (it != runners_.end()) {
        going.push_back(it->second);
        runners_.erase(it);
      }
    }
  }
  for (auto& r : going) r->Stop();  // requests that found the runner a moment ago are answered or handed back
}

void NativeExecutor::BeginCapture() {
  std::unique_lock<std::mutex> lk(gate_mu_);
  ++capturing_;
  gate_cv_.wait(lk, [this] { return launching_ == 0; });
}

void NativeExecutor::EndCapture() {
  {
    std::lock_guard<std::mutex> lk(gate_mu_);
    if (capturing_ > 0) --capturing_;
  }
  gate_cv_.notify_all();
}

void NativeExecutor::EnterLaunch() {
  std::unique_lock<std::mutex> lk(gate_mu_);
  gate_cv_.wait(lk, [this] { return capturing_ == 0; });
  ++launching_;
}

void NativeExecutor::LeaveLaunch() {
  {
    std::lock_guard<std::mutex> lk(gate_mu_);
    --launching_;
  }
  gate_cv_.notify_all();
}

void NativeExecutor::QuiesceGraphModels() {
  std::vector<std::shared_ptr<GraphRunner>> all;
  {
    std::shared_lock<std::shared_mutex> lk(mu_);
    for (auto& kv : runners_) all.push_back(kv.second);
  }
  for (auto& r : all) r->WaitIdle();
}

bool NativeExecutor::GraphStats(const std::string& model, GraphCounters* out) {
  std::shared_ptr<GraphRunner> r = FindRunner(model);
  if (!r) return false;
  r->Counters(out);
  return true;
}

// Validates a request for a graph model as far as the general path would before it queues the request,
// and takes the leases on its regions.  False = not ours (the general path serves it, or words the error).
bool NativeExecutor::PrepareGraphJob(Request* req, const std::shared_ptr<Model>& model,
                                     const std::shared_ptr<GraphRunner>& runner, GraphJob* job) {
  const ModelSpec& spec = model->spec;
  const GraphPipe& pipe = runner->pipe();
  if (req->has_sequence || req->inputs.size() != 1 || req->outputs.size() > 1) return false;
  const Request::Input& in = req->inputs[0];
  const TensorSpec& ts = spec.inputs[0];
  if (in.name != ts.name || in.datatype != ts.datatype) return false;
  // either side in a registered device region, or in host memory - inside the message or in a system
  // shared-memory region - and then through the host arena
  const bool out_named = !req->outputs.empty();
  if (in.shm.set == in.has_data) return false;
  if (out_named && req->outputs[0].name != spec.outputs[0].name) return false;
  std::shared_ptr<Region> in_sys, out_sys;
  {
    std::shared_lock<std::shared_mutex> lk(mu_);
    if (in.shm.set && device_regions_.find(in.shm.region) == device_regions_.end()) {
      auto it = regions_.find(in.shm.region);
      if (it != regions_.end()) in_sys = it->second;
    }
    if (out_named && req->outputs[0].shm.set && device_regions_.find(req->outputs[0].shm.region) == device_regions_.end()) {
      auto it = regions_.find(req->outputs[0].shm.region);
      if (it != regions_.end()) out_sys = it->second;
    }
  }
  const bool in_wire = !in.shm.set || in_sys != nullptr;
  const bool out_wire = !out_named || !req->outputs[0].shm.set || out_sys != nullptr;
  if (out_named && !req->outputs[0].shm.set && !req->outputs[0].binary) return false;  // HTTP: rows as JSON numbers
  if ((in_wire || out_wire) && pipe.host_arena == 0) return false;
  if (in.shape.size() != ts.dims.size() + 1) return false;
  for (size_t d = 0; d < ts.dims.size(); ++d)
    if (in.shape[d + 1] != ts.dims[d]) return false;
  const int64_t batch = in.shape[0];
  if (batch < 1 || batch > spec.max_batch || batch > pipe.max_batch) return false;
  const uint64_t in_need = static_cast<uint64_t>(batch) * pipe.in_row_bytes;
  const uint64_t out_need = static_cast<uint64_t>(batch) * pipe.out_row_bytes;
  const uint8_t* in_bytes = in.data;
  if (in_sys) {
    if (in.shm.offset < 0 || static_cast<uint64_t>(in.shm.size) != in_need ||
        static_cast<uint64_t>(in.shm.offset) + in_need > in_sys->size)
      return false;
    in_bytes = in_sys->base + in.shm.offset;
  } else if (in_wire && in.size != in_need) {
    return false;  // the general path words the error
  }
  if (out_sys) {
    const ShmRef& ref = req->outputs[0].shm;
    if (ref.offset < 0 || ref.size < 0 || static_cast<uint64_t>(ref.size) < out_need ||
        static_cast<uint64_t>(ref.offset) + static_cast<uint64_t>(ref.size) > out_sys->size)
      return false;
  }
  const ShmRef none;
  const ShmRef& in_shm = in.shm;
  const ShmRef& out_shm = out_named ? req->outputs[0].shm : none;
  if (!in_wire && (in_shm.offset < 0 || in_shm.size < 0 || static_cast<uint64_t>(in_shm.size) < in_need)) return false;
  if (!out_wire && (out_shm.offset < 0 || out_shm.size < 0 || static_cast<uint64_t>(out_shm.size) < out_need)) return false;
  {
    std::shared_lock<std::shared_mutex> lk(mu_);
    auto lease = [&](const ShmRef& ref, uint64_t need, uint64_t* address) {
      auto it = device_regions_.find(ref.region);
      if (it == device_regions_.end() || it->second->device != pipe.device) return false;
      const DeviceRegion& r = *it->second;
      const uint64_t off = static_cast<uint64_t>(ref.offset), size = static_cast<uint64_t>(ref.size);
      if (off + size < off || off + size > r.size || off + need > r.size) return false;
      if ((r.address + off) % 16 != 0) return false;  // odd offsets: the general path copies through an aligned buffer
      *address = r.address + off;
      it->second->inflight.fetch_add(1, std::memory_order_acq_rel);
      job->leases.push_back(it->second);
      return true;
    };
    if ((!in_wire && !lease(in_shm, in_need, &job->in_ptr)) || (!out_wire && !lease(out_shm, out_need, &job->out_ptr))) {
      for (auto& r : job->leases) r->inflight.fetch_sub(1, std::memory_order_acq_rel);
      job->leases.clear();
      return false;
    }
  }
  if (in_wire || out_wire) {
    const uint64_t in_part = in_wire ? (in_need + 255) / 256 * 256 : 0;
    job->host_bytes = in_part + (out_wire ? out_need : 0);
    job->host_ptr = runner->AllocHost(job->host_bytes);
    if (job->host_ptr == 0) {  // every piece is taken: the general path has its own staging
      for (auto& r : job->leases) r->inflight.fetch_sub(1, std::memory_order_acq_rel);
      job->leases.clear();
      return false;
    }
    job->runner = runner;
    if (in_wire) {
      std::memcpy(reinterpret_cast<void*>(static_cast<uintptr_t>(job->host_ptr)), in_bytes, in_need);
      job->in_ptr = job->host_ptr;  // the staging thread, when there is one, points it at the device arena
      job->stage_bytes = in_need;
      job->stage_src = job->host_ptr;
    }
    if (out_wire) {
      job->host_out = job->host_ptr + in_part;
      job->host_out_bytes = out_need;
      job->out_ptr = job->host_out;
    }
    if (out_sys) {  // stays mapped until the answer is built, whatever happens to the registration
      job->sys_out = out_sys->base + req->outputs[0].shm.offset;
      req->held.push_back(out_sys);
    }
  }
  job->batch = static_cast<int>(batch);
  req->results.clear();
  Result r;
  r.name = spec.outputs[0].name;
  r.datatype = spec.outputs[0].datatype;
  r.shape.push_back(batch);
  for (int64_t d : spec.outputs[0].dims) r.shape.push_back(d);
  if (!out_wire || out_sys) r.shm = out_shm;
  req->results.push_back(std::move(r));
  req->version_used = spec.version;
  return true;
}

NativeExecutor::Outcome NativeExecutor::SubmitGraph(Request* req, const std::shared_ptr<Model>& model, std::string* bytes,
                                                    Completion done, HttpCompletion http_done) {
  std::shared_ptr<GraphRunner> runner = FindRunner(model->spec.name);
  auto job = std::make_unique<GraphJob>();
  if (!runner || !PrepareGraphJob(req, model, runner, job.get())) {
    declined_.fetch_add(1, std::memory_order_relaxed);
    return Outcome::kDeclined;
  }
  req->inputs.clear();  // pointed into the request bytes, which move now
  job->req = std::move(*req);
  job->model = model;
  job->request_bytes = std::move(*bytes);
  job->done = std::move(done);
  job->http_done = std::move(http_done);
  if (!runner->Submit(&job)) {  // stopping: hand the request back untouched
    *bytes = std::move(job->request_bytes);
    if (job->host_ptr != 0) runner->FreeHost(job->host_ptr, job->host_bytes);
    for (auto& r : job->leases) r->inflight.fetch_sub(1, std::memory_order_acq_rel);
    declined_.fetch_add(1, std::memory_order_relaxed);
    return Outcome::kDeclined;
  }
  return Outcome::kDeferred;
}

// One request of a launched batch: builds the answer, counts it, frees the regions, tells the caller.
void NativeExecutor::FinishGraphJob(GraphJob* job, bool ok, uint64_t t_exec, uint64_t t_enq, uint64_t t_done, bool first) {
  std::string payload;
  HeaderList headers;
  std::vector<std::string> parts;
  if (ok) {
    if (job->sys_out != nullptr)  // output rows belong in a system shared-memory region
      std::memcpy(job->sys_out, reinterpret_cast<const void*>(static_cast<uintptr_t>(job->host_out)),
                  static_cast<size_t>(job->host_out_bytes));
    else if (job->host_out != 0)  // ... or travel inside the response
      job->req.results[0].bytes.assign(reinterpret_cast<const char*>(static_cast<uintptr_t>(job->host_out)),
                                       static_cast<size_t>(job->host_out_bytes));
    if (job->http_done) BuildHttpResponse(job->req, &job->req.results, &headers, &parts);
    else BuildGrpcResponse(job->req, job->req.results, -1, &payload);
    const uint64_t now = NowNs();
    std::lock_guard<std::mutex> lk(job->model->stats_mu);
    Stats& s = job->model->stats;
    if (first) ++s.requests;  // executions: one per launched batch
    s.inferences += static_cast<uint64_t>(job->batch);
    s.queue_ns += t_exec - job->req.t_arrive;
    s.last_ms = static_cast<uint64_t>(std::chrono::duration_cast<std::chrono::milliseconds>(
                                          std::chrono::system_clock::now().time_since_epoch()).count());
    BatchStats& bs = s.by_batch[job->batch];
    ++bs.count;
    bs.input_ns += t_enq - t_exec;
    bs.infer_ns += t_done - t_enq;
    bs.output_ns += now - t_done;
    handled_.fetch_add(1, std::memory_order_relaxed);
  } else {
    payload = std::move(job->request_bytes);
    declined_.fetch_add(1, std::memory_order_relaxed);
  }
  // the regions are free again before the caller hears about it: it may unregister them right away
  for (auto& r : job->leases) r->inflight.fetch_sub(1, std::memory_order_acq_rel);
  job->leases.clear();
  if (job->host_ptr != 0) {
    job->runner->FreeHost(job->host_ptr, job->host_bytes);
    job->host_ptr = 0;
    job->runner.reset();
  }
  if (job->http_done) {
    if (!ok) parts.assign(1, std::move(payload));
    job->http_done(ok, std::move(headers), std::move(parts));
  } else {
    job->done(ok, std::move(payload));
  }
}

// ---------------------------------------------------------------------------------------------
// HTTP
// ---------------------------------------------------------------------------------------------
namespace {

// "parameters" object of an input / output; false when it holds a key this path does not know.
bool HttpParams(const json::Value* params, ShmRef* shm, int64_t* binary_size, bool* has_binary_size, bool* binary_flag,
                bool* has_binary_flag) {
  if (params == nullptr) return true;
  if (!params->IsObject()) return false;
  for (const auto& kv : params->Members()) {
    const std::string& k = kv.first;
    const json::Value& v = kv.second;
    if (k == "shared_memory_region" && v.IsString()) {
      shm->set = true;
      shm->region = v.AsString();
    } else if (k == "shared_memory_byte_size" && v.IsNumber()) {
      shm->size = v.AsInt();
    } else if (k == "shared_memory_offset" && v.IsNumber()) {
      shm->offset = v.AsInt();
    } else if (k == "binary_data_size" && v.IsNumber() && binary_size != nullptr) {
      *binary_size = v.AsInt();
      *has_binary_size = true;
    } else if (k == "binary_data" && v.IsBool() && binary_flag != nullptr) {
      *binary_flag = v.AsBool();
      *has_binary_flag = true;
    } else {
      return false;
    }
  }
  return true;
}

}  // namespace

// Decodes the JSON header (+ binary tensors) of an HTTP inference request.
bool NativeExecutor::ParseHttp(const std::string& model_name, const std::string& version, const std::string& body,
                               int64_t header_length, Request* out, std::shared_ptr<Model>* model_out) {
  std::shared_ptr<Model> model = FindModel(model_name, version);
  if (!model) return false;
  Request& req = *out;
  req.t_arrive = NowNs();
  req.model = model_name;
  req.version = version;
  size_t json_len = body.size();
  if (header_length >= 0) {
    if (static_cast<uint64_t>(header_length) > body.size()) return false;
    json_len = static_cast<size_t>(header_length);
  }
  json::Value doc;
  if (!json::Parse(body.data(), json_len, &doc).IsOk() || !doc.IsObject()) return false;
  bool default_binary = false;
  const json::Value* inputs = nullptr;
  const json::Value* outputs = nullptr;
  for (const auto& kv : doc.Members()) {
    if (kv.first == "id" && kv.second.IsString()) {
      req.id = kv.second.AsString();
    } else if (kv.first == "inputs" && kv.second.IsArray()) {
      inputs = &kv.second;
    } else if (kv.first == "outputs" && kv.second.IsArray()) {
      outputs = &kv.second;
    } else if (kv.first == "parameters" && kv.second.IsObject()) {
      for (const auto& p : kv.second.Members()) {
        if (p.first == "binary_data_output" && p.second.IsBool()) {
          default_binary = p.second.AsBool();
        } else if (p.first == "sequence_id" && p.second.IsNumber() && p.second.AsInt() >= 0 &&
                   static_cast<double>(p.second.AsInt()) == p.second.AsDouble()) {
          req.sequence_id = static_cast<uint64_t>(p.second.AsInt());
          req.has_sequence = true;
        } else if (p.first == "sequence_start" && p.second.IsBool()) {
          req.sequence_start = p.second.AsBool();
          req.has_sequence = true;
        } else if (p.first == "sequence_end" && p.second.IsBool()) {
          req.sequence_end = p.second.AsBool();
          req.has_sequence = true;
        } else {
          return false;
        }
      }
    } else {
      return false;
    }
  }
  if (inputs == nullptr) return false;
  size_t cursor = json_len;
  for (const auto& item : inputs->Items()) {
    if (!item.IsObject()) return false;
    Request::Input in;
    const json::Value* params = nullptr;
    for (const auto& kv : item.Members()) {
      if (kv.first == "name" && kv.second.IsString()) in.name = kv.second.AsString();
      else if (kv.first == "datatype" && kv.second.IsString()) in.datatype = kv.second.AsString();
      else if (kv.first == "shape" && kv.second.IsArray()) {
        for (const auto& d : kv.second.Items()) {
          if (!d.IsNumber()) return false;
          in.shape.push_back(d.AsInt());
        }
      } else if (kv.first == "parameters") params = &kv.second;
      else return false;  // "data" in JSON form and anything else: general path
    }
    int64_t bsize = 0;
    bool has_bsize = false;
    if (!HttpParams(params, &in.shm, &bsize, &has_bsize, nullptr, nullptr)) return false;
    if (!in.shm.set) {
      if (!has_bsize || bsize < 0 || cursor + static_cast<uint64_t>(bsize) > body.size()) return false;
      in.data = reinterpret_cast<const uint8_t*>(body.data()) + cursor;
      in.size = static_cast<size_t>(bsize);
      in.has_data = true;
      cursor += static_cast<size_t>(bsize);
    } else if (has_bsize) {
      return false;
    }
    req.inputs.push_back(std::move(in));
  }
  if (cursor != body.size()) return false;
  if (outputs != nullptr) {
    for (const auto& item : outputs->Items()) {
      if (!item.IsObject()) return false;
      Request::Output out;
      out.binary = default_binary;
      const json::Value* params = nullptr;
      for (const auto& kv : item.Members()) {
        if (kv.first == "name" && kv.second.IsString()) out.name = kv.second.AsString();
        else if (kv.first == "parameters") params = &kv.second;
        else return false;
      }
      bool flag = false, has_flag = false;
      if (!HttpParams(params, &out.shm, nullptr, nullptr, &flag, &has_flag)) return false;
      if (has_flag) out.binary = flag;
      if (!out.shm.set && !out.binary) return false;  // JSON "data" answers: general path
      req.outputs.push_back(std::move(out));
    }
  } else if (!default_binary) {
    return false;
  }
  *model_out = std::move(model);
  return true;
}

// JSON header (+ one part per output that is not in shared memory) of the results.
void NativeExecutor::BuildHttpResponse(const Request& req, std::vector<Result>* results, HeaderList* headers,
                                       std::vector<std::string>* parts) {
  std::string head;
  head.reserve(384);
  head += "{\"model_name\":";
  json::detail::WriteEscaped(req.model, &head);
  head += ",\"model_version\":";
  json::detail::WriteEscaped(req.version_used, &head);
  if (!req.id.empty()) {
    head += ",\"id\":";
    json::detail::WriteEscaped(req.id, &head);
  }
  head += ",\"outputs\":[";
  bool first = true, any_blob = false;
  for (const auto& res : (*results)) {
    if (!first) head += ',';
    first = false;
    head += "{\"name\":";
    json::detail::WriteEscaped(res.name, &head);
    head += ",\"datatype\":\"" + res.datatype + "\",\"shape\":[";
    for (size_t i = 0; i < res.shape.size(); ++i) {
      if (i) head += ',';
      head += std::to_string(res.shape[i]);
    }
    head += "],\"parameters\":{";
    if (res.shm.set) {
      head += "\"shared_memory_region\":";
      json::detail::WriteEscaped(res.shm.region, &head);
      head += ",\"shared_memory_byte_size\":" + std::to_string(res.shm.size);
      if (res.shm.offset) head += ",\"shared_memory_offset\":" + std::to_string(res.shm.offset);
    } else {
      head += "\"binary_data_size\":" + std::to_string(res.bytes.size());
      any_blob = true;
    }
    head += "}}";
  }
  head += "]}";
  headers->clear();
  parts->clear();
  if (any_blob) {
    headers->emplace_back("Inference-Header-Content-Length", std::to_string(head.size()));
    headers->emplace_back("Content-Type", "application/octet-stream");
  }
  parts->push_back(std::move(head));
  for (auto& res : (*results))
    if (!res.shm.set) parts->push_back(std::move(res.bytes));
}

bool NativeExecutor::TryHttp(const std::string& model_name, const std::string& version, const std::string& body,
                             int64_t header_length, HeaderList* headers, std::vector<std::string>* parts) {
  if (!enabled_.load(std::memory_order_relaxed)) return false;
  Request req;
  std::shared_ptr<Model> model;
  if (!ParseHttp(model_name, version, body, header_length, &req, &model)) return false;
  if (!Run(&req, model)) {
    declined_.fetch_add(1, std::memory_order_relaxed);
    return false;
  }
  const uint64_t t2 = NowNs();
  BuildHttpResponse(req, &req.results, headers, parts);
  {
    int batch = 1;
    if (model->spec.max_batch > 0 && !req.results.empty() && !req.results[0].shape.empty())
      batch = static_cast<int>(req.results[0].shape[0]);
    std::lock_guard<std::mutex> lk(model->stats_mu);
    model->stats.by_batch[batch].output_ns += NowNs() - t2;
  }
  handled_.fetch_add(1, std::memory_order_relaxed);
  return true;
}


bool NativeExecutor::Defers(const std::string& model) {
  std::shared_ptr<Model> m = FindModel(model, std::string());
  if (!m) return false;
  if (m->kernel == kGraph) return FindRunner(model) != nullptr;
  return (m->kernel == kBytesAddSub || m->kernel == kBytesIdentity) && pipe_on_.load(std::memory_order_acquire);
}

NativeExecutor::Outcome NativeExecutor::TryHttpAsync(const std::string& model_name, const std::string& version,
                                                     std::string* body, int64_t header_length, HeaderList* headers,
                                                     std::vector<std::string>* parts, HttpCompletion done) {
  if (!enabled_.load(std::memory_order_relaxed)) return Outcome::kDeclined;
  Request req;
  std::shared_ptr<Model> model;
  if (!ParseHttp(model_name, version, *body, header_length, &req, &model)) return Outcome::kDeclined;
  if (model->kernel == kGraph) return SubmitGraph(&req, model, body, Completion(), std::move(done));
  if (model->kernel == kBytesAddSub || model->kernel == kBytesIdentity) {
    auto job = std::make_unique<DeviceJob>();
    if (!pipe_on_.load(std::memory_order_acquire) || !PrepareDeviceJob(&req, model, job.get())) {
      declined_.fetch_add(1, std::memory_order_relaxed);
      return Outcome::kDeclined;
    }
    req.inputs.clear();  // pointed into the body, which moves now
    job->req = std::move(req);
    job->model = std::move(model);
    job->request_bytes = std::move(*body);
    job->http_done = std::move(done);
    {
      std::lock_guard<std::mutex> lk(q_mu_);
      if (stop_) {
        *body = std::move(job->request_bytes);
        for (auto& r : job->leases) r->inflight.fetch_sub(1, std::memory_order_acq_rel);
        return Outcome::kDeclined;
      }
      queue_.push_back(std::move(job));
    }
    q_cv_.notify_one();
    return Outcome::kDeferred;
  }
  if (!Run(&req, model)) {
    declined_.fetch_add(1, std::memory_order_relaxed);
    return Outcome::kDeclined;
  }
  const uint64_t t2 = NowNs();
  BuildHttpResponse(req, &req.results, headers, parts);
  {
    int batch = 1;
    if (model->spec.max_batch > 0 && !req.results.empty() && !req.results[0].shape.empty())
      batch = static_cast<int>(req.results[0].shape[0]);
    std::lock_guard<std::mutex> lk(model->stats_mu);
    model->stats.by_batch[batch].output_ns += NowNs() - t2;
  }
  handled_.fetch_add(1, std::memory_order_relaxed);
  return Outcome::kAnswered;
}

}  // namespace server
}  // namespace tca
