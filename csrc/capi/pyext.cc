// CPython extension `_tca_pyext`: hot paths of the Python codec that cannot be
// expressed as numpy vector operations (variable-length BYTES tensors).
//
//   pack_bytes(seq_of_bytes) -> bytes          K7 on the host
//   unpack_bytes(buffer, count) -> list[bytes] K8 on the host (count < 0: all)
//   GrpcFront                                  native gRPC serving front (py_grpc_front.cc)
//
// The reference does both with a per-element Python loop
// (`tritonclient/utils/__init__.py:227-246,267-276`).
#define PY_SSIZE_T_CLEAN
#include <Python.h>

#include <cstdint>
#include <cstring>

static PyObject* PackBytes(PyObject*, PyObject* arg) {
  PyObject* seq = PySequence_Fast(arg, "pack_bytes expects a sequence of bytes objects");
  if (!seq) return nullptr;
  const Py_ssize_t n = PySequence_Fast_GET_SIZE(seq);
  PyObject** items = PySequence_Fast_ITEMS(seq);
  Py_ssize_t total = 4 * n;
  for (Py_ssize_t i = 0; i < n; ++i) {
    if (!PyBytes_Check(items[i])) {
      Py_DECREF(seq);
      PyErr_SetString(PyExc_TypeError, "pack_bytes: every element must be a bytes object");
      return nullptr;
    }
    const Py_ssize_t len = PyBytes_GET_SIZE(items[i]);
    if (len > static_cast<Py_ssize_t>(UINT32_MAX)) {
      Py_DECREF(seq);
      PyErr_SetString(PyExc_OverflowError, "pack_bytes: element longer than 4 GiB");
      return nullptr;
    }
    total += len;
  }
  PyObject* out = PyBytes_FromStringAndSize(nullptr, total);
  if (!out) { Py_DECREF(seq); return nullptr; }
  uint8_t* p = reinterpret_cast<uint8_t*>(PyBytes_AS_STRING(out));
  for (Py_ssize_t i = 0; i < n; ++i) {
    const uint32_t len = static_cast<uint32_t>(PyBytes_GET_SIZE(items[i]));
    p[0] = static_cast<uint8_t>(len);
    p[1] = static_cast<uint8_t>(len >> 8);
    p[2] = static_cast<uint8_t>(len >> 16);
    p[3] = static_cast<uint8_t>(len >> 24);
    std::memcpy(p + 4, PyBytes_AS_STRING(items[i]), len);
    p += 4 + len;
  }
  Py_DECREF(seq);
  return out;
}

static PyObject* UnpackBytes(PyObject*, PyObject* args) {
  Py_buffer view;
  long long count = -1;
  if (!PyArg_ParseTuple(args, "y*|L", &view, &count)) return nullptr;
  const uint8_t* buf = static_cast<const uint8_t*>(view.buf);
  const size_t size = static_cast<size_t>(view.len);
  PyObject* list = PyList_New(0);
  if (!list) { PyBuffer_Release(&view); return nullptr; }
  size_t pos = 0;
  long long seen = 0;
  while (count < 0 ? pos < size : seen < count) {
    if (pos + 4 > size) goto malformed;
    {
      const uint32_t len = static_cast<uint32_t>(buf[pos]) | (static_cast<uint32_t>(buf[pos + 1]) << 8) |
                           (static_cast<uint32_t>(buf[pos + 2]) << 16) | (static_cast<uint32_t>(buf[pos + 3]) << 24);
      pos += 4;
      if (len > size - pos) goto malformed;
      PyObject* item = PyBytes_FromStringAndSize(reinterpret_cast<const char*>(buf + pos), len);
      if (!item || PyList_Append(list, item) < 0) {
        Py_XDECREF(item);
        Py_DECREF(list);
        PyBuffer_Release(&view);
        return nullptr;
      }
      Py_DECREF(item);
      pos += len;
      ++seen;
    }
  }
  PyBuffer_Release(&view);
  return list;
malformed:
  Py_DECREF(list);
  PyBuffer_Release(&view);
  PyErr_SetString(PyExc_ValueError, "malformed BYTES tensor: element overruns the buffer");
  return nullptr;
}

// ---------------------------------------------------------------------------
// DLPack producer.  make_dlpack_capsule(address, type_code, bits, lanes, shape,
// device_type, device_id, owner) -> "dltensor" capsule describing NON-owned
// memory.  `owner` is kept alive until the consumer runs the deleter.  The
// deleter is native code: a consumer (e.g. torch) may release its tensor while
// the interpreter is finalising, when calling back into Python is not allowed.
// ---------------------------------------------------------------------------
namespace {

struct DLDevice { int32_t device_type; int32_t device_id; };
struct DLDataType { uint8_t code; uint8_t bits; uint16_t lanes; };
struct DLTensor {
  void* data;
  DLDevice device;
  int32_t ndim;
  DLDataType dtype;
  int64_t* shape;
  int64_t* strides;
  uint64_t byte_offset;
};
struct DLManagedTensor {
  DLTensor dl_tensor;
  void* manager_ctx;
  void (*deleter)(DLManagedTensor*);
};

struct Holder {
  DLManagedTensor managed;
  PyObject* owner;
  int64_t shape[8];
};

void ManagedDeleter(DLManagedTensor* m) {
  Holder* h = static_cast<Holder*>(m->manager_ctx);
  if (h->owner != nullptr && Py_IsInitialized()
#if PY_VERSION_HEX >= 0x03070000
      && !_Py_IsFinalizing()
#endif
  ) {
    PyGILState_STATE st = PyGILState_Ensure();
    Py_DECREF(h->owner);
    PyGILState_Release(st);
  }
  delete h;
}

void CapsuleDestructor(PyObject* capsule) {
  // an unconsumed capsule still owns the managed tensor
  if (PyCapsule_IsValid(capsule, "dltensor")) {
    auto* m = static_cast<DLManagedTensor*>(PyCapsule_GetPointer(capsule, "dltensor"));
    if (m && m->deleter) m->deleter(m);
  }
}

}  // namespace

static PyObject* MakeDlpackCapsule(PyObject*, PyObject* args) {
  unsigned long long address;
  int code, bits, lanes, device_type, device_id;
  PyObject* shape;
  PyObject* owner;
  if (!PyArg_ParseTuple(args, "KiiiOiiO", &address, &code, &bits, &lanes, &shape, &device_type, &device_id, &owner))
    return nullptr;
  PyObject* seq = PySequence_Fast(shape, "shape must be a sequence of ints");
  if (!seq) return nullptr;
  const Py_ssize_t ndim = PySequence_Fast_GET_SIZE(seq);
  if (ndim > 8) {
    Py_DECREF(seq);
    PyErr_SetString(PyExc_ValueError, "at most 8 dimensions are supported");
    return nullptr;
  }
  Holder* h = new Holder();
  for (Py_ssize_t i = 0; i < ndim; ++i) {
    h->shape[i] = PyLong_AsLongLong(PySequence_Fast_GET_ITEM(seq, i));
    if (PyErr_Occurred()) { Py_DECREF(seq); delete h; return nullptr; }
  }
  Py_DECREF(seq);
  h->managed.dl_tensor.data = reinterpret_cast<void*>(static_cast<uintptr_t>(address));
  h->managed.dl_tensor.device = DLDevice{device_type, device_id};
  h->managed.dl_tensor.ndim = static_cast<int32_t>(ndim);
  h->managed.dl_tensor.dtype = DLDataType{static_cast<uint8_t>(code), static_cast<uint8_t>(bits), static_cast<uint16_t>(lanes)};
  h->managed.dl_tensor.shape = h->shape;
  h->managed.dl_tensor.strides = nullptr;
  h->managed.dl_tensor.byte_offset = 0;
  h->managed.manager_ctx = h;
  h->managed.deleter = ManagedDeleter;
  h->owner = (owner == Py_None) ? nullptr : owner;
  Py_XINCREF(h->owner);
  PyObject* capsule = PyCapsule_New(&h->managed, "dltensor", CapsuleDestructor);
  if (!capsule) { Py_XDECREF(h->owner); delete h; return nullptr; }
  return capsule;
}

static PyMethodDef kMethods[] = {
    {"make_dlpack_capsule", MakeDlpackCapsule, METH_VARARGS, "Wrap raw memory as a non-owning DLPack capsule."},
    {"pack_bytes", PackBytes, METH_O, "Serialise a sequence of bytes objects into the BYTES wire form."},
    {"unpack_bytes", UnpackBytes, METH_VARARGS, "Split a BYTES wire buffer into a list of bytes objects."},
    {nullptr, nullptr, 0, nullptr}};

static struct PyModuleDef kModule = {PyModuleDef_HEAD_INIT, "_tca_pyext", "native codec helpers", -1, kMethods,
                                     nullptr, nullptr, nullptr, nullptr};

extern "C" int tca_register_grpc_front(PyObject* module);  // py_grpc_front.cc
extern "C" int tca_register_http_front(PyObject* module);  // py_http_front.cc
extern "C" int tca_register_native_executor(PyObject* module);  // py_native_executor.cc

PyMODINIT_FUNC PyInit__tca_pyext(void) {
  PyObject* module = PyModule_Create(&kModule);
  if (module == nullptr) return nullptr;
  if (tca_register_grpc_front(module) < 0 || tca_register_http_front(module) < 0 ||
      tca_register_native_executor(module) < 0) {
    Py_DECREF(module);
    return nullptr;
  }
  return module;
}
