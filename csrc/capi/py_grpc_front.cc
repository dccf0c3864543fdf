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
// The callbacks run on the front's reactor threads with the GIL held:
//   on_unary(call, method, request_bytes), on_stream_message(call, method,
//   request_bytes), on_stream_end(call), on_cancel(call).
// They must not block: answers are produced later through reply / send /
// finish from whichever thread completes the request.
#define PY_SSIZE_T_CLEAN
#include <Python.h>

#include <set>
#include <string>

#include "tca/grpc_server.h"

namespace {

struct PyGrpcFront {
  PyObject_HEAD
  tca::server::GrpcServerFront* front;
  PyObject* on_unary;
  PyObject* on_stream_message;
  PyObject* on_stream_end;
  PyObject* on_cancel;
};

void CallPython(PyObject* fn, const char* fmt, ...) {
  if (fn == nullptr || fn == Py_None) return;
  PyGILState_STATE gil = PyGILState_Ensure();
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
  PyGILState_Release(gil);
}

int FrontInit(PyObject* self_obj, PyObject* args, PyObject* kwargs) {
  auto* self = reinterpret_cast<PyGrpcFront*>(self_obj);
  const char* host = nullptr;
  int port = 0, threads = 2;
  PyObject *on_unary, *on_msg, *on_end, *on_cancel;
  // optional TLS: server certificate chain + key (PEM files); client_ca makes client certificates mandatory
  const char *tls_cert = nullptr, *tls_key = nullptr, *tls_client_ca = nullptr;
  static const char* kwlist[] = {"host", "port", "threads", "on_unary", "on_stream_message", "on_stream_end",
                                 "on_cancel", "tls_cert", "tls_key", "tls_client_ca", nullptr};
  if (!PyArg_ParseTupleAndKeywords(args, kwargs, "siiOOOO|zzz", const_cast<char**>(kwlist), &host, &port, &threads,
                                   &on_unary, &on_msg, &on_end, &on_cancel, &tls_cert, &tls_key, &tls_client_ca))
    return -1;
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
  Py_INCREF(on_unary);
  Py_INCREF(on_msg);
  Py_INCREF(on_end);
  Py_INCREF(on_cancel);
  self->on_unary = on_unary;
  self->on_stream_message = on_msg;
  self->on_stream_end = on_end;
  self->on_cancel = on_cancel;
  tca::server::GrpcServerFront::Handlers h;
  h.on_unary = [self](uint64_t call, const std::string& method, std::string&& request) {
    CallPython(self->on_unary, "(Ksy#)", static_cast<unsigned long long>(call), method.c_str(), request.data(),
               static_cast<Py_ssize_t>(request.size()));
  };
  h.on_stream_message = [self](uint64_t call, const std::string& method, std::string&& request) {
    CallPython(self->on_stream_message, "(Ksy#)", static_cast<unsigned long long>(call), method.c_str(),
               request.data(), static_cast<Py_ssize_t>(request.size()));
  };
  h.on_stream_end = [self](uint64_t call) {
    CallPython(self->on_stream_end, "(K)", static_cast<unsigned long long>(call));
  };
  h.on_cancel = [self](uint64_t call) { CallPython(self->on_cancel, "(K)", static_cast<unsigned long long>(call)); };
  self->front = new tca::server::GrpcServerFront(host, port, threads, {"ModelStreamInfer"}, std::move(h));
  if (tls) self->front->SetTls(std::move(tls));
  return 0;
}

void StopFront(PyGrpcFront* self) {
  if (self->front == nullptr) return;
  tca::server::GrpcServerFront* front = self->front;
  // reactor threads may be waiting for the GIL inside a callback: let them finish
  Py_BEGIN_ALLOW_THREADS
  front->Stop();
  Py_END_ALLOW_THREADS
}

void FrontDealloc(PyObject* self_obj) {
  auto* self = reinterpret_cast<PyGrpcFront*>(self_obj);
  StopFront(self);
  delete self->front;
  self->front = nullptr;
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
  Py_INCREF(&kFrontType);
  if (PyModule_AddObject(module, "GrpcFront", reinterpret_cast<PyObject*>(&kFrontType)) < 0) {
    Py_DECREF(&kFrontType);
    return -1;
  }
  return 0;
}
