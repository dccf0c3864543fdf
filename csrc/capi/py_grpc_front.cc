// Python binding of the native gRPC serving front (`_tca_pyext.GrpcFront`).
//
//   front = GrpcFront(host, port, threads, on_unary, on_stream_message,
//                     on_stream_end, on_cancel)
//   port = front.start()
//   front.reply(call, bytes)            # unary success: message + OK trailers
//   front.send(call, bytes)             # one message of a streaming answer
//   front.finish(call, status, message) # trailers (errors, end of a stream)
//   front.stop()
//
// The callbacks run with the GIL held - on the reactor thread when there is one
// reactor, else on one funnel thread that serves all reactors (gil_funnel.h):
//   on_unary(call, method, request_bytes), on_stream_message(call, method,
//   request_bytes), on_stream_end(call), on_cancel(call).
// A large ModelInfer request (tensors on the wire) is not copied into Python
// objects: on_unary(call, method, head_bytes, blob, spans) gets the request
// WITHOUT its raw_input_contents entries as `head_bytes`, the whole message as
// `blob` (a buffer-protocol object that owns the bytes) and `spans`, the
// (offset, length) of every raw_input_contents entry inside the blob, in order.
// They must not block: answers are produced later through reply / send /
// finish from whichever thread completes the request.
#define PY_SSIZE_T_CLEAN
#include <Python.h>

#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <unordered_set>
#include <utility>
#include <vector>

#include "gil_funnel.h"
#include "tca/grpc_server.h"
#include "tca/native_executor.h"

std::shared_ptr<tca::server::NativeExecutor> tca_native_from_py(PyObject* obj);  // py_native_executor.cc

namespace {

struct PyGrpcFront {
  PyObject_HEAD
  tca::server::GrpcServerFront* front;
  PyObject* on_unary;
  PyObject* on_stream_message;
  PyObject* on_stream_end;
  PyObject* on_cancel;
  tca::capi::GilFunnel* funnel;  // null with one reactor: that thread calls Python itself
  std::shared_ptr<tca::server::NativeExecutor>* native;  // null: every request goes to Python
  // Streams one of whose messages went to Python: from then on all their messages do.  A message answered
  // here is on the wire at once, one handed to Python answers later - answering a later message natively
  // would let its response overtake the earlier one's on the same stream.
  struct PythonStreams {
    std::mutex mu;
    std::unordered_set<uint64_t> calls;
  };
  PythonStreams* python_streams;
};

// Owns one received message and exposes it through the buffer protocol, so numpy
// arrays can alias tensor bytes without a copy made under the GIL.
struct PyBlob {
  PyObject_HEAD
  std::string* data;
};

int BlobGetBuffer(PyObject* self_obj, Py_buffer* view, int flags) {
  auto* self = reinterpret_cast<PyBlob*>(self_obj);
  return PyBuffer_FillInfo(view, self_obj, &(*self->data)[0], static_cast<Py_ssize_t>(self->data->size()),
                           /*readonly=*/0, flags);
}

void BlobDealloc(PyObject* self_obj) {
  auto* self = reinterpret_cast<PyBlob*>(self_obj);
  delete self->data;
  Py_TYPE(self_obj)->tp_free(self_obj);
}

Py_ssize_t BlobLength(PyObject* self_obj) {
  return static_cast<Py_ssize_t>(reinterpret_cast<PyBlob*>(self_obj)->data->size());
}

PyBufferProcs kBlobBuffer = {BlobGetBuffer, nullptr};
PySequenceMethods kBlobSequence = {BlobLength, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};

#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Wmissing-field-initializers"
PyTypeObject kBlobType = {PyVarObject_HEAD_INIT(nullptr, 0)};
#pragma GCC diagnostic pop

constexpr size_t kZeroCopyThreshold = 128 * 1024;  // below this a plain bytes copy is cheaper than the bookkeeping
constexpr uint32_t kRawInputContentsField = 7;     // ModelInferRequest.raw_input_contents

// Splits a serialized ModelInferRequest into the message without its
// raw_input_contents entries (`head`) and the (offset, length) of every such
// entry.  False when the bytes are not a well-formed protobuf message (the caller
// then hands the whole message to Python, whose parser reports the error).
bool SplitRawContents(const std::string& msg, std::string* head, std::vector<std::pair<size_t, size_t>>* spans) {
  const auto* p = reinterpret_cast<const uint8_t*>(msg.data());
  const size_t n = msg.size();
  size_t i = 0;
  auto varint = [&](uint64_t* out) {
    uint64_t v = 0;
    for (int shift = 0; shift < 64 && i < n; shift += 7) {
      const uint8_t b = p[i++];
      v |= static_cast<uint64_t>(b & 0x7F) << shift;
      if (!(b & 0x80)) {
        *out = v;
        return true;
      }
    }
    return false;
  };
  while (i < n) {
    const size_t start = i;
    uint64_t tag = 0, len = 0;
    if (!varint(&tag)) return false;
    const uint32_t field = static_cast<uint32_t>(tag >> 3);
    switch (tag & 7) {
      case 0:
        if (!varint(&len)) return false;
        break;
      case 1:
        if (n - i < 8) return false;
        i += 8;
        break;
      case 5:
        if (n - i < 4) return false;
        i += 4;
        break;
      case 2:
        if (!varint(&len) || len > n - i) return false;
        if (field == kRawInputContentsField) {
          spans->emplace_back(i, static_cast<size_t>(len));
          i += static_cast<size_t>(len);
          continue;  // not part of the head
        }
        i += static_cast<size_t>(len);
        break;
      default:
        return false;
    }
    head->append(msg, start, i - start);
  }
  return true;
}

}  // namespace

// A Blob owning `data` (new reference), or nullptr with a Python error set.  Shared with the HTTP front.
PyObject* tca_new_blob(std::string&& data) {
  auto* blob = PyObject_New(PyBlob, &kBlobType);
  if (blob == nullptr) return nullptr;
  blob->data = new std::string(std::move(data));
  return reinterpret_cast<PyObject*>(blob);
}

namespace {

// Calls fn(*args) with the GIL held.
void CallPython(PyObject* fn, const char* fmt, ...) {
  if (fn == nullptr || fn == Py_None) return;
  va_list args;
  va_start(args, fmt);
  PyObject* packed = Py_VaBuildValue(fmt, args);
  va_end(args);
  if (packed != nullptr) {
    PyObject* result = PyObject_CallObject(fn, packed);
    Py_DECREF(packed);
    if (result == nullptr) PyErr_WriteUnraisable(fn);
    else Py_DECREF(result);
  } else {
    PyErr_WriteUnraisable(fn);
  }
}

// Runs `fn` with the GIL held: on the funnel thread when there is one, else here.
template <typename Fn>
void WithGil(PyGrpcFront* self, Fn&& fn) {
  if (self->funnel != nullptr) {
    self->funnel->Run(std::forward<Fn>(fn));
    return;
  }
  PyGILState_STATE gil = PyGILState_Ensure();
  fn();
  PyGILState_Release(gil);
}

// A unary request -> on_unary, with the GIL held.
void DeliverUnary(PyGrpcFront* self, uint64_t call, const std::string& method, std::string& request);

void DeliverUnary(PyGrpcFront* self, uint64_t call, const std::string& method, std::string& request) {
  if (request.size() >= kZeroCopyThreshold && method == "ModelInfer") {
    std::string head;
    std::vector<std::pair<size_t, size_t>> spans;
    if (SplitRawContents(request, &head, &spans) && !spans.empty()) {
      auto* blob = PyObject_New(PyBlob, &kBlobType);
      PyObject* list = blob ? PyTuple_New(static_cast<Py_ssize_t>(spans.size())) : nullptr;
      if (blob != nullptr && list != nullptr) {
        blob->data = new std::string(std::move(request));
        for (size_t k = 0; k < spans.size(); ++k)
          PyTuple_SET_ITEM(list, static_cast<Py_ssize_t>(k),
                           Py_BuildValue("(nn)", static_cast<Py_ssize_t>(spans[k].first),
                                         static_cast<Py_ssize_t>(spans[k].second)));
        PyObject* result = PyObject_CallFunction(self->on_unary, "Ksy#OO", static_cast<unsigned long long>(call),
                                                 method.c_str(), head.data(), static_cast<Py_ssize_t>(head.size()),
                                                 reinterpret_cast<PyObject*>(blob), list);
        if (result == nullptr) PyErr_WriteUnraisable(self->on_unary);
        else Py_DECREF(result);
      } else {
        PyErr_Clear();
      }
      Py_XDECREF(list);
      if (blob != nullptr) {
        if (list == nullptr) blob->data = new std::string();  // allocation failed before ownership moved
        Py_DECREF(reinterpret_cast<PyObject*>(blob));
      }
      return;
    }
  }
  CallPython(self->on_unary, "(Ksy#)", static_cast<unsigned long long>(call), method.c_str(), request.data(),
             static_cast<Py_ssize_t>(request.size()));
}

int FrontInit(PyObject* self_obj, PyObject* args, PyObject* kwargs) {
  auto* self = reinterpret_cast<PyGrpcFront*>(self_obj);
  const char* host = nullptr;
  int port = 0, threads = 2;
  PyObject *on_unary, *on_msg, *on_end, *on_cancel;
  // optional TLS: server certificate chain + key (PEM files); client_ca makes client certificates mandatory
  const char *tls_cert = nullptr, *tls_key = nullptr, *tls_client_ca = nullptr;
  PyObject* native = nullptr;
  static const char* kwlist[] = {"host", "port", "threads", "on_unary", "on_stream_message", "on_stream_end",
                                 "on_cancel", "tls_cert", "tls_key", "tls_client_ca", "native", nullptr};
  if (!PyArg_ParseTupleAndKeywords(args, kwargs, "siiOOOO|zzzO", const_cast<char**>(kwlist), &host, &port, &threads,
                                   &on_unary, &on_msg, &on_end, &on_cancel, &tls_cert, &tls_key, &tls_client_ca, &native))
    return -1;
  if (auto ex = tca_native_from_py(native)) self->native = new std::shared_ptr<tca::server::NativeExecutor>(std::move(ex));
  std::shared_ptr<tca::net::TlsContext> tls;
  if (tls_cert != nullptr || tls_key != nullptr) {
    tca::net::TlsOptions opt;
    opt.server = true;
    opt.cert_file = tls_cert ? tls_cert : "";
    opt.key_file = tls_key ? tls_key : "";
    opt.ca_file = tls_client_ca ? tls_client_ca : "";
    opt.verify_peer = tls_client_ca != nullptr;
    opt.alpn = {"h2"};
    const tca::Error e = tca::net::TlsContext::Create(opt, &tls);
    if (!e.IsOk()) {
      PyErr_SetString(PyExc_ValueError, e.Message().c_str());
      return -1;
    }
  }
  self->python_streams = new PyGrpcFront::PythonStreams();
  Py_INCREF(on_unary);
  Py_INCREF(on_msg);
  Py_INCREF(on_end);
  Py_INCREF(on_cancel);
  self->on_unary = on_unary;
  self->on_stream_message = on_msg;
  self->on_stream_end = on_end;
  self->on_cancel = on_cancel;
  tca::server::GrpcServerFront::Handlers h;
  self->funnel = threads > 1 ? new tca::capi::GilFunnel() : nullptr;
  h.on_unary = [self](uint64_t call, const std::string& method, std::string&& request) {
    if (self->native != nullptr && method == "ModelInfer") {
      using Outcome = tca::server::NativeExecutor::Outcome;
      std::string response;
      const Outcome outcome = (*self->native)->TryGrpcAsync(&request, &response, [self, call](bool ok, std::string&& payload) {
        if (ok) {  // answered by the executor's batch thread
          self->front->Reply(call, std::move(payload));
          return;
        }
        // the device refused it: the general path takes the request and words the error
        auto req = std::make_shared<std::string>(std::move(payload));
        WithGil(self, [self, call, req] { DeliverUnary(self, call, "ModelInfer", *req); });
      });
      if (outcome == Outcome::kAnswered) {  // on this reactor thread, no interpreter
        self->front->Reply(call, std::move(response));
        return;
      }
      if (outcome == Outcome::kDeferred) return;
    }
    auto req = std::make_shared<std::string>(std::move(request));
    WithGil(self, [self, call, method, req] { DeliverUnary(self, call, method, *req); });
  };
  h.on_stream_message = [self](uint64_t call, const std::string& method, std::string&& request) {
    if (self->native != nullptr && method == "ModelStreamInfer") {
      bool python_owned;
      {
        std::lock_guard<std::mutex> lk(self->python_streams->mu);
        python_owned = self->python_streams->calls.count(call) != 0;
      }
      std::vector<std::string> responses;
      if (!python_owned && (*self->native)->TryGrpcStream(request, &responses)) {  // answered here, in order
        for (auto& m : responses) self->front->Send(call, std::move(m));
        return;
      }
      if (!python_owned) {
        std::lock_guard<std::mutex> lk(self->python_streams->mu);
        self->python_streams->calls.insert(call);
      }
    }
    auto req = std::make_shared<std::string>(std::move(request));
    WithGil(self, [self, call, method, req] {
      CallPython(self->on_stream_message, "(Ksy#)", static_cast<unsigned long long>(call), method.c_str(), req->data(),
                 static_cast<Py_ssize_t>(req->size()));
    });
  };
  h.on_stream_end = [self](uint64_t call) {
    {
      std::lock_guard<std::mutex> lk(self->python_streams->mu);
      self->python_streams->calls.erase(call);
    }
    WithGil(self, [self, call] { CallPython(self->on_stream_end, "(K)", static_cast<unsigned long long>(call)); });
  };
  h.on_cancel = [self](uint64_t call) {
    {
      std::lock_guard<std::mutex> lk(self->python_streams->mu);
      self->python_streams->calls.erase(call);
    }
    WithGil(self, [self, call] { CallPython(self->on_cancel, "(K)", static_cast<unsigned long long>(call)); });
  };
  self->front = new tca::server::GrpcServerFront(host, port, threads, {"ModelStreamInfer"}, std::move(h));
  if (tls) self->front->SetTls(std::move(tls));
  return 0;
}

void StopFront(PyGrpcFront* self) {
  if (self->front == nullptr) return;
  tca::server::GrpcServerFront* front = self->front;
  tca::capi::GilFunnel* funnel = self->funnel;
  // reactor / funnel threads may be waiting for the GIL inside a callback: let them finish
  tca::server::NativeExecutor* native = self->native != nullptr ? self->native->get() : nullptr;
  Py_BEGIN_ALLOW_THREADS
  front->Stop();
  if (native != nullptr) native->Quiesce();  // deferred requests answer through this front (and may need the funnel)
  if (funnel != nullptr) funnel->Stop();
  Py_END_ALLOW_THREADS
}

void FrontDealloc(PyObject* self_obj) {
  auto* self = reinterpret_cast<PyGrpcFront*>(self_obj);
  StopFront(self);
  delete self->front;
  self->front = nullptr;
  delete self->funnel;
  self->funnel = nullptr;
  delete self->native;
  self->native = nullptr;
  delete self->python_streams;
  self->python_streams = nullptr;
  Py_XDECREF(self->on_unary);
  Py_XDECREF(self->on_stream_message);
  Py_XDECREF(self->on_stream_end);
  Py_XDECREF(self->on_cancel);
  Py_TYPE(self_obj)->tp_free(self_obj);
}

PyObject* FrontStart(PyObject* self_obj, PyObject*) {
  auto* self = reinterpret_cast<PyGrpcFront*>(self_obj);
  const tca::Error e = self->front->Start();
  if (!e.IsOk()) {
    PyErr_SetString(PyExc_RuntimeError, e.Message().c_str());
    return nullptr;
  }
  return PyLong_FromLong(self->front->port());
}

PyObject* FrontStop(PyObject* self_obj, PyObject*) {
  StopFront(reinterpret_cast<PyGrpcFront*>(self_obj));
  Py_RETURN_NONE;
}

PyObject* FrontReply(PyObject* self_obj, PyObject* args) {
  auto* self = reinterpret_cast<PyGrpcFront*>(self_obj);
  unsigned long long call = 0;
  Py_buffer view;
  if (!PyArg_ParseTuple(args, "Ky*", &call, &view)) return nullptr;
  std::string message(static_cast<const char*>(view.buf), static_cast<size_t>(view.len));
  PyBuffer_Release(&view);
  self->front->Reply(call, std::move(message));
  Py_RETURN_NONE;
}

PyObject* FrontSend(PyObject* self_obj, PyObject* args) {
  auto* self = reinterpret_cast<PyGrpcFront*>(self_obj);
  unsigned long long call = 0;
  Py_buffer view;
  if (!PyArg_ParseTuple(args, "Ky*", &call, &view)) return nullptr;
  std::string message(static_cast<const char*>(view.buf), static_cast<size_t>(view.len));
  PyBuffer_Release(&view);
  self->front->Send(call, std::move(message));
  Py_RETURN_NONE;
}

PyObject* FrontFinish(PyObject* self_obj, PyObject* args) {
  auto* self = reinterpret_cast<PyGrpcFront*>(self_obj);
  unsigned long long call = 0;
  int status = 0;
  const char* message = "";
  if (!PyArg_ParseTuple(args, "Ki|s", &call, &status, &message)) return nullptr;
  self->front->Finish(call, status, message);
  Py_RETURN_NONE;
}

PyObject* FrontStats(PyObject* self_obj, PyObject*) {
  auto* self = reinterpret_cast<PyGrpcFront*>(self_obj);
  return Py_BuildValue("{s:K,s:K}", "connections", static_cast<unsigned long long>(self->front->ConnectionsAccepted()),
                       "calls", static_cast<unsigned long long>(self->front->CallsStarted()));
}

PyMethodDef kFrontMethods[] = {
    {"start", FrontStart, METH_NOARGS, "Bind, start the reactor threads and return the port."},
    {"stop", FrontStop, METH_NOARGS, "Stop the reactor threads and close every connection."},
    {"reply", FrontReply, METH_VARARGS, "reply(call, message): unary success."},
    {"send", FrontSend, METH_VARARGS, "send(call, message): one message of a streaming answer."},
    {"finish", FrontFinish, METH_VARARGS, "finish(call, status[, message]): end the call with a gRPC status."},
    {"stats", FrontStats, METH_NOARGS, "Counters: connections accepted, calls started."},
    {nullptr, nullptr, 0, nullptr}};

#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Wmissing-field-initializers"
PyTypeObject kFrontType = {PyVarObject_HEAD_INIT(nullptr, 0)};
#pragma GCC diagnostic pop

}  // namespace

// Called from the module init of _tca_pyext.
extern "C" int tca_register_grpc_front(PyObject* module) {
  kFrontType.tp_name = "_tca_pyext.GrpcFront";
  kFrontType.tp_basicsize = sizeof(PyGrpcFront);
  kFrontType.tp_flags = Py_TPFLAGS_DEFAULT;
  kFrontType.tp_doc = "Native gRPC serving front (HTTP/2 on epoll reactor threads).";
  kFrontType.tp_new = PyType_GenericNew;
  kFrontType.tp_init = FrontInit;
  kFrontType.tp_dealloc = FrontDealloc;
  kFrontType.tp_methods = kFrontMethods;
  if (PyType_Ready(&kFrontType) < 0) return -1;
  kBlobType.tp_name = "_tca_pyext.Blob";
  kBlobType.tp_basicsize = sizeof(PyBlob);
  kBlobType.tp_flags = Py_TPFLAGS_DEFAULT;
  kBlobType.tp_doc = "One received message; exposes its bytes through the buffer protocol.";
  kBlobType.tp_dealloc = BlobDealloc;
  kBlobType.tp_as_buffer = &kBlobBuffer;
  kBlobType.tp_as_sequence = &kBlobSequence;
  if (PyType_Ready(&kBlobType) < 0) return -1;
  Py_INCREF(&kFrontType);
  if (PyModule_AddObject(module, "GrpcFront", reinterpret_cast<PyObject*>(&kFrontType)) < 0) {
    Py_DECREF(&kFrontType);
    return -1;
  }
  return 0;
}
