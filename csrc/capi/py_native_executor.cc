// Python binding of the native execution path (`_tca_pyext.NativeExecutor`).
//
//   ex = NativeExecutor()
//   ex.set_model(name, version, max_batch, [(name, datatype, dims), ...], [(name, datatype, dims), ...], kernel)
//   ex.remove_model(name)
//   ex.register_region(name, key, offset, byte_size) / ex.unregister_region(name)   # "" = all
//   ex.take_stats(name) -> None | {"requests", "inferences", "queue_ns", "last_ms", "by_batch": {b: (count, in, infer, out)}}
//   ex.set_enabled(flag); ex.counters() -> {"handled", "declined"}
//   ex.sequence_step(name, id, start, end) -> bool; ex.open_sequences(name) -> int
//   ex.enable_device_bytes(submit_address, wait_address, pipe, stream, device, max_requests, max_count, max_stream_bytes)
//   ex.disable_device_bytes(); ex.quiesce()
//   ex.register_device_region(name, address, byte_size, device) / ex.unregister_device_region(name, timeout_ms) -> bool
//
// The serving fronts take the object as their `native=` argument and offer it every inference
// request before Python sees it (tca/native_executor.h).
#define PY_SSIZE_T_CLEAN
#include <Python.h>

#include <memory>
#include <string>

#include "tca/native_executor.h"

namespace {

struct PyNativeExecutor {
  PyObject_HEAD
  std::shared_ptr<tca::server::NativeExecutor>* ex;
};

#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Wmissing-field-initializers"
PyTypeObject kType = {PyVarObject_HEAD_INIT(nullptr, 0)};
#pragma GCC diagnostic pop

int Init(PyObject* self_obj, PyObject*, PyObject*) {
  auto* self = reinterpret_cast<PyNativeExecutor*>(self_obj);
  delete self->ex;
  self->ex = new std::shared_ptr<tca::server::NativeExecutor>(new tca::server::NativeExecutor());
  return 0;
}

void Dealloc(PyObject* self_obj) {
  auto* self = reinterpret_cast<PyNativeExecutor*>(self_obj);
  delete self->ex;
  self->ex = nullptr;
  Py_TYPE(self_obj)->tp_free(self_obj);
}

bool ParseTensors(PyObject* list, std::vector<tca::server::NativeExecutor::TensorSpec>* out) {
  PyObject* seq = PySequence_Fast(list, "tensor specs must be a sequence of (name, datatype, dims)");
  if (seq == nullptr) return false;
  for (Py_ssize_t i = 0; i < PySequence_Fast_GET_SIZE(seq); ++i) {
    const char *name = nullptr, *dt = nullptr;
    PyObject* dims = nullptr;
    if (!PyArg_ParseTuple(PySequence_Fast_GET_ITEM(seq, i), "ssO", &name, &dt, &dims)) {
      Py_DECREF(seq);
      return false;
    }
    tca::server::NativeExecutor::TensorSpec t;
    t.name = name;
    t.datatype = dt;
    PyObject* dseq = PySequence_Fast(dims, "dims must be a sequence of integers");
    if (dseq == nullptr) {
      Py_DECREF(seq);
      return false;
    }
    for (Py_ssize_t k = 0; k < PySequence_Fast_GET_SIZE(dseq); ++k) {
      const long long v = PyLong_AsLongLong(PySequence_Fast_GET_ITEM(dseq, k));
      if (v == -1 && PyErr_Occurred()) {
        Py_DECREF(dseq);
        Py_DECREF(seq);
        return false;
      }
      t.dims.push_back(v);
    }
    Py_DECREF(dseq);
    out->push_back(std::move(t));
  }
  Py_DECREF(seq);
  return true;
}

PyObject* SetModel(PyObject* self_obj, PyObject* args) {
  auto* self = reinterpret_cast<PyNativeExecutor*>(self_obj);
  const char *name = nullptr, *version = nullptr, *kernel = nullptr;
  int max_batch = 0;
  PyObject *inputs = nullptr, *outputs = nullptr;
  if (!PyArg_ParseTuple(args, "ssiOOs", &name, &version, &max_batch, &inputs, &outputs, &kernel)) return nullptr;
  tca::server::NativeExecutor::ModelSpec spec;
  spec.name = name;
  spec.version = version;
  spec.max_batch = max_batch;
  spec.kernel = kernel;
  if (!ParseTensors(inputs, &spec.inputs) || !ParseTensors(outputs, &spec.outputs)) return nullptr;
  const tca::Error e = (*self->ex)->SetModel(spec);
  if (!e.IsOk()) {
    PyErr_SetString(PyExc_ValueError, e.Message().c_str());
    return nullptr;
  }
  Py_RETURN_NONE;
}

PyObject* RemoveModel(PyObject* self_obj, PyObject* args) {
  auto* self = reinterpret_cast<PyNativeExecutor*>(self_obj);
  const char* name = nullptr;
  if (!PyArg_ParseTuple(args, "s", &name)) return nullptr;
  (*self->ex)->RemoveModel(name);
  Py_RETURN_NONE;
}

PyObject* RegisterRegion(PyObject* self_obj, PyObject* args) {
  auto* self = reinterpret_cast<PyNativeExecutor*>(self_obj);
  const char *name = nullptr, *key = nullptr;
  unsigned long long offset = 0, size = 0;
  if (!PyArg_ParseTuple(args, "ssKK", &name, &key, &offset, &size)) return nullptr;
  const tca::Error e = (*self->ex)->RegisterRegion(name, key, offset, size);
  if (!e.IsOk()) {
    PyErr_SetString(PyExc_ValueError, e.Message().c_str());
    return nullptr;
  }
  Py_RETURN_NONE;
}

PyObject* UnregisterRegion(PyObject* self_obj, PyObject* args) {
  auto* self = reinterpret_cast<PyNativeExecutor*>(self_obj);
  const char* name = "";
  if (!PyArg_ParseTuple(args, "|s", &name)) return nullptr;
  (*self->ex)->UnregisterRegion(name);
  Py_RETURN_NONE;
}

PyObject* TakeStats(PyObject* self_obj, PyObject* args) {
  auto* self = reinterpret_cast<PyNativeExecutor*>(self_obj);
  const char* name = nullptr;
  if (!PyArg_ParseTuple(args, "s", &name)) return nullptr;
  tca::server::NativeExecutor::Stats st;
  if (!(*self->ex)->TakeStats(name, &st)) Py_RETURN_NONE;
  PyObject* by_batch = PyDict_New();
  if (by_batch == nullptr) return nullptr;
  for (const auto& kv : st.by_batch) {
    PyObject* key = PyLong_FromLong(kv.first);
    PyObject* val = Py_BuildValue("(KKKK)", static_cast<unsigned long long>(kv.second.count),
                                  static_cast<unsigned long long>(kv.second.input_ns),
                                  static_cast<unsigned long long>(kv.second.infer_ns),
                                  static_cast<unsigned long long>(kv.second.output_ns));
    if (key == nullptr || val == nullptr || PyDict_SetItem(by_batch, key, val) < 0) {
      Py_XDECREF(key);
      Py_XDECREF(val);
      Py_DECREF(by_batch);
      return nullptr;
    }
    Py_DECREF(key);
    Py_DECREF(val);
  }
  return Py_BuildValue("{s:K,s:K,s:K,s:K,s:N}", "requests", static_cast<unsigned long long>(st.requests), "inferences",
                       static_cast<unsigned long long>(st.inferences), "queue_ns",
                       static_cast<unsigned long long>(st.queue_ns), "last_ms",
                       static_cast<unsigned long long>(st.last_ms), "by_batch", by_batch);
}

PyObject* SetEnabled(PyObject* self_obj, PyObject* args) {
  auto* self = reinterpret_cast<PyNativeExecutor*>(self_obj);
  int on = 1;
  if (!PyArg_ParseTuple(args, "p", &on)) return nullptr;
  (*self->ex)->SetEnabled(on != 0);
  Py_RETURN_NONE;
}

PyObject* Counters(PyObject* self_obj, PyObject*) {
  auto* self = reinterpret_cast<PyNativeExecutor*>(self_obj);
  return Py_BuildValue("{s:K,s:K}", "handled", static_cast<unsigned long long>((*self->ex)->Handled()), "declined",
                       static_cast<unsigned long long>((*self->ex)->Declined()));
}

PyObject* EnableDeviceBytes(PyObject* self_obj, PyObject* args) {
  auto* self = reinterpret_cast<PyNativeExecutor*>(self_obj);
  unsigned long long submit = 0, wait = 0, pipe = 0, stream = 0, max_stream = 0;
  int device = 0, max_requests = 0, max_count = 0;
  if (!PyArg_ParseTuple(args, "KKKKiiiK", &submit, &wait, &pipe, &stream, &device, &max_requests, &max_count, &max_stream))
    return nullptr;
  tca::server::NativeExecutor::DevicePipe p;
  p.submit = reinterpret_cast<decltype(p.submit)>(static_cast<uintptr_t>(submit));
  p.wait = reinterpret_cast<decltype(p.wait)>(static_cast<uintptr_t>(wait));
  p.pipe = reinterpret_cast<void*>(static_cast<uintptr_t>(pipe));
  p.stream = reinterpret_cast<void*>(static_cast<uintptr_t>(stream));
  p.device = device;
  p.max_requests = max_requests;
  p.max_count = max_count;
  p.max_stream_bytes = max_stream;
  tca::Error e;
  Py_BEGIN_ALLOW_THREADS
  e = (*self->ex)->EnableDeviceBytes(p);
  Py_END_ALLOW_THREADS
  if (!e.IsOk()) {
    PyErr_SetString(PyExc_ValueError, e.Message().c_str());
    return nullptr;
  }
  Py_RETURN_NONE;
}

PyObject* DisableDeviceBytes(PyObject* self_obj, PyObject*) {
  auto* self = reinterpret_cast<PyNativeExecutor*>(self_obj);
  Py_BEGIN_ALLOW_THREADS  // answers in flight may need the interpreter lock to be delivered
  (*self->ex)->DisableDeviceBytes();
  Py_END_ALLOW_THREADS
  Py_RETURN_NONE;
}

PyObject* Quiesce(PyObject* self_obj, PyObject*) {
  auto* self = reinterpret_cast<PyNativeExecutor*>(self_obj);
  Py_BEGIN_ALLOW_THREADS
  (*self->ex)->Quiesce();
  Py_END_ALLOW_THREADS
  Py_RETURN_NONE;
}

// enable_graph_model(name, entry_points, device, streams, graphs, in_row_bytes, out_row_bytes, max_batch,
//                    max_inflight, fill_window_us, burst_gap_us, burst_window_us, split_min_batch, adaptive_cap
//                    [, host_arena, host_arena_bytes, host_chunk_bytes[, device_arena, copy_stream[, memcpy_async[, cap_limit_tenths]]]])
// entry_points: addresses of (graph_launch, event_create, event_record, event_sync, event_destroy, set_device,
// batched_copy[, stream_sync]); streams: one handle per execution instance; graphs: (instance, batch, exec, in_ptr, out_ptr).
PyObject* EnableGraphModel(PyObject* self_obj, PyObject* args) {
  auto* self = reinterpret_cast<PyNativeExecutor*>(self_obj);
  const char* name = nullptr;
  PyObject *entry = nullptr, *streams = nullptr, *graphs = nullptr;
  int device = 0, max_batch = 0, max_inflight = 0, fill = 0, gap = 300, window = 100, split = 16, adaptive = 1;
  unsigned long long in_row = 0, out_row = 0, arena = 0, arena_bytes = 0, chunk_bytes = 0, device_arena = 0, copy_stream = 0,
                     memcpy_fn = 0;
  int cap_limit_tenths = 0;
  if (!PyArg_ParseTuple(args, "sOiOOKKiiiiiii|KKKKKKi", &name, &entry, &device, &streams, &graphs, &in_row, &out_row, &max_batch,
                        &max_inflight, &fill, &gap, &window, &split, &adaptive, &arena, &arena_bytes, &chunk_bytes,
                        &device_arena, &copy_stream, &memcpy_fn, &cap_limit_tenths))
    return nullptr;
  tca::server::NativeExecutor::GraphPipe p;
  unsigned long long fn[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (!PyArg_ParseTuple(entry, "KKKKKKK|K", &fn[0], &fn[1], &fn[2], &fn[3], &fn[4], &fn[5], &fn[6], &fn[7])) return nullptr;
  p.graph_launch = reinterpret_cast<decltype(p.graph_launch)>(static_cast<uintptr_t>(fn[0]));
  p.event_create = reinterpret_cast<decltype(p.event_create)>(static_cast<uintptr_t>(fn[1]));
  p.event_record = reinterpret_cast<decltype(p.event_record)>(static_cast<uintptr_t>(fn[2]));
  p.event_sync = reinterpret_cast<decltype(p.event_sync)>(static_cast<uintptr_t>(fn[3]));
  p.event_destroy = reinterpret_cast<decltype(p.event_destroy)>(static_cast<uintptr_t>(fn[4]));
  p.set_device = reinterpret_cast<decltype(p.set_device)>(static_cast<uintptr_t>(fn[5]));
  p.batched_copy = reinterpret_cast<decltype(p.batched_copy)>(static_cast<uintptr_t>(fn[6]));
  p.stream_sync = reinterpret_cast<decltype(p.stream_sync)>(static_cast<uintptr_t>(fn[7]));
  PyObject* seq = PySequence_Fast(streams, "streams must be a sequence of handles");
  if (seq == nullptr) return nullptr;
  for (Py_ssize_t i = 0; i < PySequence_Fast_GET_SIZE(seq); ++i) {
    const unsigned long long h = PyLong_AsUnsignedLongLong(PySequence_Fast_GET_ITEM(seq, i));
    if (PyErr_Occurred()) {
      Py_DECREF(seq);
      return nullptr;
    }
    p.streams.push_back(reinterpret_cast<void*>(static_cast<uintptr_t>(h)));
  }
  Py_DECREF(seq);
  seq = PySequence_Fast(graphs, "graphs must be a sequence of (instance, batch, exec, in_ptr, out_ptr)");
  if (seq == nullptr) return nullptr;
  for (Py_ssize_t i = 0; i < PySequence_Fast_GET_SIZE(seq); ++i) {
    tca::server::NativeExecutor::GraphPipe::Graph g;
    unsigned long long exec = 0, in_ptr = 0, out_ptr = 0;
    if (!PyArg_ParseTuple(PySequence_Fast_GET_ITEM(seq, i), "iiKKK", &g.instance, &g.batch, &exec, &in_ptr, &out_ptr)) {
      Py_DECREF(seq);
      return nullptr;
    }
    g.exec = reinterpret_cast<void*>(static_cast<uintptr_t>(exec));
    g.in_ptr = in_ptr;
    g.out_ptr = out_ptr;
    p.graphs.push_back(g);
  }
  Py_DECREF(seq);
  p.device = device;
  p.in_row_bytes = in_row;
  p.out_row_bytes = out_row;
  p.max_batch = max_batch;
  p.max_inflight = max_inflight;
  p.fill_window_us = fill;
  p.burst_gap_us = gap;
  p.burst_window_us = window;
  p.split_min_batch = split;
  p.adaptive_cap = adaptive != 0;  // 0: off; 1: on with the default depth; n > 1: depth n / 10 batches per instance
  if (adaptive > 1) p.cap_depth = adaptive / 10.0;
  if (cap_limit_tenths > 0) p.cap_limit = cap_limit_tenths / 10.0;
  if (arena != 0 && arena_bytes != 0 && chunk_bytes != 0) {
    p.host_arena = arena;
    p.host_arena_bytes = arena_bytes;
    p.host_chunk_bytes = chunk_bytes;
    p.device_arena = device_arena;
    p.copy_stream = reinterpret_cast<void*>(static_cast<uintptr_t>(copy_stream));
    p.memcpy_async = reinterpret_cast<decltype(p.memcpy_async)>(static_cast<uintptr_t>(memcpy_fn));
  }
  const std::string model(name);
  tca::Error e;
  Py_BEGIN_ALLOW_THREADS
  e = (*self->ex)->EnableGraphModel(model, p);
  Py_END_ALLOW_THREADS
  if (!e.IsOk()) {
    PyErr_SetString(PyExc_ValueError, e.Message().c_str());
    return nullptr;
  }
  Py_RETURN_NONE;
}

PyObject* DisableGraphModel(PyObject* self_obj, PyObject* args) {
  auto* self = reinterpret_cast<PyNativeExecutor*>(self_obj);
  const char* name = "";
  if (!PyArg_ParseTuple(args, "|s", &name)) return nullptr;
  const std::string model(name);
  Py_BEGIN_ALLOW_THREADS  // answers in flight may need the interpreter lock to be delivered
  (*self->ex)->DisableGraphModel(model);
  Py_END_ALLOW_THREADS
  Py_RETURN_NONE;
}

PyObject* BeginCapture(PyObject* self_obj, PyObject*) {
  auto* self = reinterpret_cast<PyNativeExecutor*>(self_obj);
  Py_BEGIN_ALLOW_THREADS
  (*self->ex)->BeginCapture();
  Py_END_ALLOW_THREADS
  Py_RETURN_NONE;
}

PyObject* EndCapture(PyObject* self_obj, PyObject*) {
  auto* self = reinterpret_cast<PyNativeExecutor*>(self_obj);
  (*self->ex)->EndCapture();
  Py_RETURN_NONE;
}

PyObject* GraphStats(PyObject* self_obj, PyObject* args) {
  auto* self = reinterpret_cast<PyNativeExecutor*>(self_obj);
  const char* name = nullptr;
  if (!PyArg_ParseTuple(args, "s", &name)) return nullptr;
  tca::server::NativeExecutor::GraphCounters c;
  if (!(*self->ex)->GraphStats(name, &c)) Py_RETURN_NONE;
  return Py_BuildValue("{s:K,s:K,s:K,s:K,s:K,s:K}", "batches", static_cast<unsigned long long>(c.batches), "requests",
                       static_cast<unsigned long long>(c.requests), "images", static_cast<unsigned long long>(c.images),
                       "padded_images", static_cast<unsigned long long>(c.padded_images), "wire_requests",
                       static_cast<unsigned long long>(c.wire_requests), "arena_full",
                       static_cast<unsigned long long>(c.arena_full));
}

PyObject* RegisterDeviceRegion(PyObject* self_obj, PyObject* args) {
  auto* self = reinterpret_cast<PyNativeExecutor*>(self_obj);
  const char* name = nullptr;
  unsigned long long address = 0, size = 0;
  int device = 0;
  if (!PyArg_ParseTuple(args, "sKKi", &name, &address, &size, &device)) return nullptr;
  const tca::Error e = (*self->ex)->RegisterDeviceRegion(name, address, size, device);
  if (!e.IsOk()) {
    PyErr_SetString(PyExc_ValueError, e.Message().c_str());
    return nullptr;
  }
  Py_RETURN_NONE;
}

PyObject* UnregisterDeviceRegion(PyObject* self_obj, PyObject* args) {
  auto* self = reinterpret_cast<PyNativeExecutor*>(self_obj);
  const char* name = "";
  int timeout_ms = 10000;
  if (!PyArg_ParseTuple(args, "|si", &name, &timeout_ms)) return nullptr;
  const std::string region(name);
  bool idle = false;
  Py_BEGIN_ALLOW_THREADS
  idle = (*self->ex)->UnregisterDeviceRegion(region, timeout_ms);
  Py_END_ALLOW_THREADS
  return PyBool_FromLong(idle ? 1 : 0);
}

PyObject* SequenceStep(PyObject* self_obj, PyObject* args) {
  auto* self = reinterpret_cast<PyNativeExecutor*>(self_obj);
  const char* name = nullptr;
  unsigned long long id = 0;
  int start = 0, end = 0;
  if (!PyArg_ParseTuple(args, "sKpp", &name, &id, &start, &end)) return nullptr;
  return PyBool_FromLong((*self->ex)->SequenceStep(name, id, start != 0, end != 0) ? 1 : 0);
}

PyObject* OpenSequences(PyObject* self_obj, PyObject* args) {
  auto* self = reinterpret_cast<PyNativeExecutor*>(self_obj);
  const char* name = nullptr;
  if (!PyArg_ParseTuple(args, "s", &name)) return nullptr;
  return PyLong_FromLong((*self->ex)->OpenSequences(name));
}

PyMethodDef kMethods[] = {
    {"set_model", SetModel, METH_VARARGS, "set_model(name, version, max_batch, inputs, outputs, kernel)"},
    {"remove_model", RemoveModel, METH_VARARGS, "remove_model(name)"},
    {"register_region", RegisterRegion, METH_VARARGS, "register_region(name, key, offset, byte_size)"},
    {"unregister_region", UnregisterRegion, METH_VARARGS, "unregister_region([name]): all regions without a name"},
    {"take_stats", TakeStats, METH_VARARGS, "take_stats(name): statistics accumulated since the last call, or None"},
    {"set_enabled", SetEnabled, METH_VARARGS, "set_enabled(flag)"},
    {"counters", Counters, METH_NOARGS, "requests handled natively / looked at and handed on"},
    {"sequence_step", SequenceStep, METH_VARARGS,
     "sequence_step(name, id, start, end) -> False when the sequence was never started (or the model has no sequence kernel)"},
    {"open_sequences", OpenSequences, METH_VARARGS, "open_sequences(name) -> sequences started and not ended (-1: no such model)"},
    {"enable_device_bytes", EnableDeviceBytes, METH_VARARGS,
     "enable_device_bytes(submit_address, wait_address, pipe, stream, device, max_requests, max_count, max_stream_bytes)"},
    {"disable_device_bytes", DisableDeviceBytes, METH_NOARGS, "answers what is queued and stops the batch thread"},
    {"quiesce", Quiesce, METH_NOARGS, "blocks until no deferred request is queued or running"},
    {"enable_graph_model", EnableGraphModel, METH_VARARGS,
     "enable_graph_model(name, entry_points, device, streams, graphs, in_row_bytes, out_row_bytes, max_batch, max_inflight, "
     "fill_window_us, burst_gap_us, burst_window_us, split_min_batch, adaptive_cap[, host_arena, host_arena_bytes, "
     "host_chunk_bytes[, device_arena, copy_stream[, memcpy_async[, cap_limit_tenths]]]]): page-locked memory for requests with tensors inside the messages, and "
     "device memory of the same size plus a stream for moving their inputs ahead of the batches"},
    {"disable_graph_model", DisableGraphModel, METH_VARARGS, "disable_graph_model([name]): answers what is queued and stops the threads"},
    {"begin_capture", BeginCapture, METH_NOARGS,
     "holds the batch threads' graph launches back (after those in progress) until end_capture(): a stream capture may run"},
    {"end_capture", EndCapture, METH_NOARGS, "the capture is over"},
    {"graph_stats", GraphStats, METH_VARARGS, "graph_stats(name) -> batches / requests / images / padded_images, or None"},
    {"register_device_region", RegisterDeviceRegion, METH_VARARGS, "register_device_region(name, address, byte_size, device)"},
    {"unregister_device_region", UnregisterDeviceRegion, METH_VARARGS,
     "unregister_device_region([name[, timeout_ms]]) -> False when requests still use the region"},
    {nullptr, nullptr, 0, nullptr}};

}  // namespace

// The executor behind a Python NativeExecutor object, or null when `obj` is None / another type.
std::shared_ptr<tca::server::NativeExecutor> tca_native_from_py(PyObject* obj) {
  if (obj == nullptr || obj == Py_None || !PyObject_TypeCheck(obj, &kType)) return nullptr;
  auto* self = reinterpret_cast<PyNativeExecutor*>(obj);
  return self->ex ? *self->ex : nullptr;
}

extern "C" int tca_register_native_executor(PyObject* module) {
  kType.tp_name = "_tca_pyext.NativeExecutor";
  kType.tp_basicsize = sizeof(PyNativeExecutor);
  kType.tp_flags = Py_TPFLAGS_DEFAULT;
  kType.tp_doc = "Native execution path for models with a C++ kernel.";
  kType.tp_new = PyType_GenericNew;
  kType.tp_init = Init;
  kType.tp_dealloc = Dealloc;
  kType.tp_methods = kMethods;
  if (PyType_Ready(&kType) < 0) return -1;
  Py_INCREF(&kType);
  if (PyModule_AddObject(module, "NativeExecutor", reinterpret_cast<PyObject*>(&kType)) < 0) {
    Py_DECREF(&kType);
    return -1;
  }
  return 0;
}
