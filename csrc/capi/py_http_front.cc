// Python binding of the native HTTP/1.1 serving front (`_tca_pyext.HttpFront`).
//
//   front = HttpFront(host, port, threads, on_request, on_cancel,
//                     tls_cert=None, tls_key=None, tls_client_ca=None)
//   port = front.start()
//   front.respond(call, status, [(name, value), ...], [part, ...])
//   front.stop()
//
// on_request(call, method, target, headers, body) runs on a reactor thread with
// the GIL held and must not block; `headers` is a dict (names lower-cased, a
// repeated field keeps its last value) and `body` is bytes, or - from 128 KiB
// up - a buffer-protocol object that owns the received bytes, so tensors sent
// on the wire are aliased by numpy, not copied under the GIL.
// respond() may be called from any thread; the parts are sent back to back.
#define PY_SSIZE_T_CLEAN
#include <Python.h>

#include <cstdlib>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "gil_funnel.h"
#include "tca/http_server.h"
#include "tca/native_executor.h"

PyObject* tca_new_blob(std::string&& data);  // py_grpc_front.cc
std::shared_ptr<tca::server::NativeExecutor> tca_native_from_py(PyObject* obj);  // py_native_executor.cc

namespace {

struct PyHttpFront {
  PyObject_HEAD
  tca::server::HttpServerFront* front;
  PyObject* on_request;
  PyObject* on_cancel;
  tca::capi::GilFunnel* funnel;  // null with one reactor: that thread calls Python itself
  std::shared_ptr<tca::server::NativeExecutor>* native;  // null: every request goes to Python
};

// "/v2/models/<name>[/versions/<v>]/infer" -> name, version; false for every other target.
bool InferTarget(const std::string& target, std::string* model, std::string* version) {
  static const std::string kPrefix = "/v2/models/";
  static const std::string kSuffix = "/infer";
  if (target.compare(0, kPrefix.size(), kPrefix) != 0 || target.size() <= kPrefix.size() + kSuffix.size() ||
      target.compare(target.size() - kSuffix.size(), kSuffix.size(), kSuffix) != 0 ||
      target.find_first_of("?%") != std::string::npos)
    return false;
  const std::string mid = target.substr(kPrefix.size(), target.size() - kPrefix.size() - kSuffix.size());
  const size_t slash = mid.find('/');
  if (slash == std::string::npos) {
    *model = mid;
    version->clear();
    return !model->empty();
  }
  static const std::string kVersions = "/versions/";
  if (mid.compare(slash, kVersions.size(), kVersions) != 0) return false;
  *model = mid.substr(0, slash);
  *version = mid.substr(slash + kVersions.size());
  return !model->empty() && !version->empty() && version->find('/') == std::string::npos;
}

void DeliverRequest(PyHttpFront* self, uint64_t call, tca::server::HttpServerFront::Request& req);

// True when the native path answered the request, or has queued it and will answer (or hand it on) later.
bool TryNative(PyHttpFront* self, uint64_t call, tca::server::HttpServerFront::Request& req) {
  if (self->native == nullptr || req.method != "POST") return false;
  std::string model, version;
  if (!InferTarget(req.target, &model, &version)) return false;
  int64_t header_length = -1;
  for (const auto& h : req.headers) {
    if (h.first == "content-encoding" || h.first == "accept-encoding") {
      if (h.second.find("gzip") != std::string::npos || h.second.find("deflate") != std::string::npos) return false;
    } else if (h.first == "inference-header-content-length") {
      char* stop = nullptr;
      header_length = std::strtoll(h.second.c_str(), &stop, 10);
      if (h.second.empty() || *stop != '\0' || header_length < 0) return false;
    }
  }
  tca::server::HttpServerFront::HeaderList headers;
  std::vector<std::string> parts;
  if (!(*self->native)->Defers(model)) {  // answered on this reactor thread, or not at all
    if (!(*self->native)->TryHttp(model, version, req.body, header_length, &headers, &parts)) return false;
    self->front->Respond(call, 200, std::move(headers), std::move(parts));
    return true;
  }
  using Outcome = tca::server::NativeExecutor::Outcome;
  // a model whose requests are queued for the device: the completion keeps what the general path needs besides
  // the body, should the device refuse the request
  const Outcome outcome = (*self->native)->TryHttpAsync(
      model, version, &req.body, header_length, &headers, &parts,
      [self, call, method = req.method, target = req.target, request_headers = req.headers](
          bool ok, tca::server::HttpServerFront::HeaderList&& h, std::vector<std::string>&& p) mutable {
        if (ok) {  // answered by the executor's batch thread
          self->front->Respond(call, 200, std::move(h), std::move(p));
          return;
        }
        // the device refused it: the general path takes the request and words the error
        auto shared = std::make_shared<tca::server::HttpServerFront::Request>();
        shared->method = std::move(method);
        shared->target = std::move(target);
        shared->headers = std::move(request_headers);
        if (!p.empty()) shared->body = std::move(p[0]);
        auto deliver = [self, call, shared] { DeliverRequest(self, call, *shared); };
        if (self->funnel != nullptr) {
          self->funnel->Run(deliver);
          return;
        }
        PyGILState_STATE gil = PyGILState_Ensure();
        deliver();
        PyGILState_Release(gil);
      });
  if (outcome == Outcome::kDeferred) return true;
  if (outcome != Outcome::kAnswered) return false;
  self->front->Respond(call, 200, std::move(headers), std::move(parts));
  return true;
}

constexpr size_t kZeroCopyThreshold = 128 * 1024;

// Both run with the GIL held.
void DeliverRequest(PyHttpFront* self, uint64_t call, tca::server::HttpServerFront::Request& req) {
  PyObject* headers = PyDict_New();
  PyObject* body = nullptr;
  if (headers != nullptr) {
    for (const auto& kv : req.headers) {
      PyObject* v = PyUnicode_DecodeLatin1(kv.second.data(), static_cast<Py_ssize_t>(kv.second.size()), nullptr);
      if (v == nullptr) break;
      PyDict_SetItemString(headers, kv.first.c_str(), v);
      Py_DECREF(v);
    }
    body = req.body.size() >= kZeroCopyThreshold
               ? tca_new_blob(std::move(req.body))
               : PyBytes_FromStringAndSize(req.body.data(), static_cast<Py_ssize_t>(req.body.size()));
  }
  if (headers != nullptr && body != nullptr && !PyErr_Occurred()) {
    PyObject* result = PyObject_CallFunction(self->on_request, "KssOO", static_cast<unsigned long long>(call),
                                             req.method.c_str(), req.target.c_str(), headers, body);
    if (result == nullptr) PyErr_WriteUnraisable(self->on_request);
    else Py_DECREF(result);
  } else {
    PyErr_Clear();
    self->front->Respond(call, 500, {}, {std::string("{\"error\":\"out of memory\"}")});
  }
  Py_XDECREF(headers);
  Py_XDECREF(body);
}

void DeliverCancel(PyHttpFront* self, uint64_t call) {
  PyObject* result = PyObject_CallFunction(self->on_cancel, "K", static_cast<unsigned long long>(call));
  if (result == nullptr) PyErr_WriteUnraisable(self->on_cancel);
  else Py_DECREF(result);
}

int FrontInit(PyObject* self_obj, PyObject* args, PyObject* kwargs) {
  auto* self = reinterpret_cast<PyHttpFront*>(self_obj);
  const char* host = nullptr;
  int port = 0, threads = 2;
  PyObject *on_request, *on_cancel;
  const char *tls_cert = nullptr, *tls_key = nullptr, *tls_client_ca = nullptr;
  PyObject* native = nullptr;
  static const char* kwlist[] = {"host", "port", "threads", "on_request", "on_cancel", "tls_cert", "tls_key",
                                 "tls_client_ca", "native", nullptr};
  if (!PyArg_ParseTupleAndKeywords(args, kwargs, "siiOO|zzzO", const_cast<char**>(kwlist), &host, &port, &threads,
                                   &on_request, &on_cancel, &tls_cert, &tls_key, &tls_client_ca, &native))
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
    opt.alpn = {"http/1.1"};
    const tca::Error e = tca::net::TlsContext::Create(opt, &tls);
    if (!e.IsOk()) {
      PyErr_SetString(PyExc_ValueError, e.Message().c_str());
      return -1;
    }
  }
  Py_INCREF(on_request);
  Py_INCREF(on_cancel);
  self->on_request = on_request;
  self->on_cancel = on_cancel;
  tca::server::HttpServerFront::Handlers h;
  self->funnel = threads > 1 ? new tca::capi::GilFunnel() : nullptr;
  h.on_request = [self](uint64_t call, tca::server::HttpServerFront::Request&& req) {
    if (TryNative(self, call, req)) return;  // answered on this reactor thread, no interpreter
    if (self->funnel != nullptr) {
      auto shared = std::make_shared<tca::server::HttpServerFront::Request>(std::move(req));
      self->funnel->Run([self, call, shared] { DeliverRequest(self, call, *shared); });
      return;
    }
    PyGILState_STATE gil = PyGILState_Ensure();
    DeliverRequest(self, call, req);
    PyGILState_Release(gil);
  };
  h.on_cancel = [self](uint64_t call) {
    if (self->on_cancel == nullptr || self->on_cancel == Py_None) return;
    if (self->funnel != nullptr) {
      self->funnel->Run([self, call] { DeliverCancel(self, call); });
      return;
    }
    PyGILState_STATE gil = PyGILState_Ensure();
    DeliverCancel(self, call);
    PyGILState_Release(gil);
  };
  self->front = new tca::server::HttpServerFront(host, port, threads, std::move(h));
  if (tls) self->front->SetTls(std::move(tls));
  return 0;
}

void StopFront(PyHttpFront* self) {
  if (self->front == nullptr) return;
  tca::server::HttpServerFront* front = self->front;
  tca::capi::GilFunnel* funnel = self->funnel;
  // reactor / funnel threads may be waiting for the GIL inside a callback: let them finish
  tca::server::NativeExecutor* native = self->native != nullptr ? self->native->get() : nullptr;
  Py_BEGIN_ALLOW_THREADS
  front->Stop();
  if (native != nullptr) native->Quiesce();  // queued requests answer through this front (and may need the funnel)
  if (funnel != nullptr) funnel->Stop();
  Py_END_ALLOW_THREADS
}

void FrontDealloc(PyObject* self_obj) {
  auto* self = reinterpret_cast<PyHttpFront*>(self_obj);
  StopFront(self);
  delete self->front;
  self->front = nullptr;
  delete self->funnel;
  self->funnel = nullptr;
  delete self->native;
  self->native = nullptr;
  Py_XDECREF(self->on_request);
  Py_XDECREF(self->on_cancel);
  Py_TYPE(self_obj)->tp_free(self_obj);
}

PyObject* FrontStart(PyObject* self_obj, PyObject*) {
  auto* self = reinterpret_cast<PyHttpFront*>(self_obj);
  const tca::Error e = self->front->Start();
  if (!e.IsOk()) {
    PyErr_SetString(PyExc_RuntimeError, e.Message().c_str());
    return nullptr;
  }
  return PyLong_FromLong(self->front->port());
}

PyObject* FrontStop(PyObject* self_obj, PyObject*) {
  StopFront(reinterpret_cast<PyHttpFront*>(self_obj));
  Py_RETURN_NONE;
}

// respond(call, status, headers, parts): headers = sequence of (name, value) or a dict, parts = sequence of buffers
PyObject* FrontRespond(PyObject* self_obj, PyObject* args) {
  auto* self = reinterpret_cast<PyHttpFront*>(self_obj);
  unsigned long long call = 0;
  int status = 200;
  PyObject *headers_obj = nullptr, *parts_obj = nullptr;
  if (!PyArg_ParseTuple(args, "KiOO", &call, &status, &headers_obj, &parts_obj)) return nullptr;
  tca::server::HttpServerFront::HeaderList headers;
  if (PyDict_Check(headers_obj)) {
    PyObject *k, *v;
    Py_ssize_t pos = 0;
    while (PyDict_Next(headers_obj, &pos, &k, &v)) {
      const char* ks = PyUnicode_AsUTF8(k);
      PyObject* vs_obj = PyObject_Str(v);
      const char* vs = vs_obj ? PyUnicode_AsUTF8(vs_obj) : nullptr;
      if (ks == nullptr || vs == nullptr) {
        Py_XDECREF(vs_obj);
        return nullptr;
      }
      headers.emplace_back(ks, vs);
      Py_DECREF(vs_obj);
    }
  } else if (headers_obj != Py_None) {
    PyObject* seq = PySequence_Fast(headers_obj, "headers must be a dict or a sequence of (name, value)");
    if (seq == nullptr) return nullptr;
    for (Py_ssize_t i = 0; i < PySequence_Fast_GET_SIZE(seq); ++i) {
      const char *ks = nullptr, *vs = nullptr;
      if (!PyArg_ParseTuple(PySequence_Fast_GET_ITEM(seq, i), "ss", &ks, &vs)) {
        Py_DECREF(seq);
        return nullptr;
      }
      headers.emplace_back(ks, vs);
    }
    Py_DECREF(seq);
  }
  std::vector<std::string> parts;
  PyObject* seq = PySequence_Fast(parts_obj, "parts must be a sequence of bytes-like objects");
  if (seq == nullptr) return nullptr;
  for (Py_ssize_t i = 0; i < PySequence_Fast_GET_SIZE(seq); ++i) {
    Py_buffer view;
    if (PyObject_GetBuffer(PySequence_Fast_GET_ITEM(seq, i), &view, PyBUF_SIMPLE) < 0) {
      Py_DECREF(seq);
      return nullptr;
    }
    parts.emplace_back(static_cast<const char*>(view.buf), static_cast<size_t>(view.len));
    PyBuffer_Release(&view);
  }
  Py_DECREF(seq);
  self->front->Respond(call, status, std::move(headers), std::move(parts));
  Py_RETURN_NONE;
}

PyObject* FrontStats(PyObject* self_obj, PyObject*) {
  auto* self = reinterpret_cast<PyHttpFront*>(self_obj);
  return Py_BuildValue("{s:K,s:K}", "connections", static_cast<unsigned long long>(self->front->ConnectionsAccepted()),
                       "requests", static_cast<unsigned long long>(self->front->RequestsStarted()));
}

PyMethodDef kFrontMethods[] = {
    {"start", FrontStart, METH_NOARGS, "Bind, start the reactor threads and return the port."},
    {"stop", FrontStop, METH_NOARGS, "Stop the reactor threads and close every connection."},
    {"respond", FrontRespond, METH_VARARGS, "respond(call, status, headers, parts): answer a request."},
    {"stats", FrontStats, METH_NOARGS, "Counters: connections accepted, requests started."},
    {nullptr, nullptr, 0, nullptr}};

#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Wmissing-field-initializers"
PyTypeObject kFrontType = {PyVarObject_HEAD_INIT(nullptr, 0)};
#pragma GCC diagnostic pop

}  // namespace

// Called from the module init of _tca_pyext (after the gRPC front, which owns the Blob type).
extern "C" int tca_register_http_front(PyObject* module) {
  kFrontType.tp_name = "_tca_pyext.HttpFront";
  kFrontType.tp_basicsize = sizeof(PyHttpFront);
  kFrontType.tp_flags = Py_TPFLAGS_DEFAULT;
  kFrontType.tp_doc = "Native HTTP/1.1 serving front (epoll reactor threads).";
  kFrontType.tp_new = PyType_GenericNew;
  kFrontType.tp_init = FrontInit;
  kFrontType.tp_dealloc = FrontDealloc;
  kFrontType.tp_methods = kFrontMethods;
  if (PyType_Ready(&kFrontType) < 0) return -1;
  Py_INCREF(&kFrontType);
  if (PyModule_AddObject(module, "HttpFront", reinterpret_cast<PyObject*>(&kFrontType)) < 0) {
    Py_DECREF(&kFrontType);
    return -1;
  }
  return 0;
}
