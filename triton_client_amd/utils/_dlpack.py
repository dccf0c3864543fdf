"""Minimal DLPack (v0.x ``DLManagedTensor``) producer/consumer in ctypes.

Covers what the shared-memory modules need (reference:
``tritonclient/utils/_dlpack.py:57-272``):

* ABI mirrors of ``DLDevice`` / ``DLDataType`` / ``DLTensor`` /
  ``DLManagedTensor``;
* :func:`make_capsule` – wrap a raw address (host or MI355X device memory) as a
  *non-owning* ``"dltensor"`` capsule;
* helpers to inspect capsules handed to us by torch / numpy.

Device memory is advertised as ``kDLROCM`` – the type PyTorch-ROCm produces and
accepts.  Keep-alive of the shape array and of the owner object is done with a
module-level registry keyed by the managed-tensor address instead of manual
``Py_IncRef`` bookkeeping.
"""

from __future__ import annotations

import ctypes
import threading

_api = ctypes.pythonapi
_api.PyCapsule_New.restype = ctypes.py_object
_api.PyCapsule_New.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p]
_api.PyCapsule_GetPointer.restype = ctypes.c_void_p
_api.PyCapsule_GetPointer.argtypes = [ctypes.py_object, ctypes.c_char_p]
_api.PyCapsule_IsValid.restype = ctypes.c_int
_api.PyCapsule_IsValid.argtypes = [ctypes.py_object, ctypes.c_char_p]
_api.PyCapsule_GetName.restype = ctypes.c_char_p
_api.PyCapsule_GetName.argtypes = [ctypes.py_object]

c_str_dltensor = b"dltensor"
c_str_used_dltensor = b"used_dltensor"


class DLDeviceType(ctypes.c_int):
    kDLCPU = 1
    kDLCUDA = 2
    kDLCUDAHost = 3
    kDLOpenCL = 4
    kDLVulkan = 7
    kDLMetal = 8
    kDLVPI = 9
    kDLROCM = 10
    kDLROCMHost = 11
    kDLExtDev = 12
    kDLCUDAManaged = 13
    kDLOneAPI = 14
    kDLWebGPU = 15
    kDLHexagon = 16


class DLDevice(ctypes.Structure):
    _fields_ = [("device_type", ctypes.c_int), ("device_id", ctypes.c_int)]


class DLDataTypeCode(ctypes.c_uint8):
    kDLInt = 0
    kDLUInt = 1
    kDLFloat = 2
    kDLOpaquePointer = 3
    kDLBfloat = 4
    kDLComplex = 5
    kDLBool = 6


class DLDataType(ctypes.Structure):
    _fields_ = [
        ("type_code", ctypes.c_uint8),
        ("bits", ctypes.c_uint8),
        ("lanes", ctypes.c_uint16),
    ]


class DLTensor(ctypes.Structure):
    _fields_ = [
        ("data", ctypes.c_void_p),
        ("device", DLDevice),
        ("ndim", ctypes.c_int),
        ("dtype", DLDataType),
        ("shape", ctypes.POINTER(ctypes.c_int64)),
        ("strides", ctypes.POINTER(ctypes.c_int64)),
        ("byte_offset", ctypes.c_uint64),
    ]


_DELETER = ctypes.CFUNCTYPE(None, ctypes.c_void_p)


class DLManagedTensor(ctypes.Structure):
    _fields_ = [
        ("dl_tensor", DLTensor),
        ("manager_ctx", ctypes.c_void_p),
        ("deleter", _DELETER),
    ]


_DTYPE_TABLE = {
    # BOOL is exported as an 8-bit bool: that is what every DLPack consumer
    # (torch, numpy) implements.
    "BOOL": (DLDataTypeCode.kDLBool, 8),
    "INT8": (DLDataTypeCode.kDLInt, 8),
    "INT16": (DLDataTypeCode.kDLInt, 16),
    "INT32": (DLDataTypeCode.kDLInt, 32),
    "INT64": (DLDataTypeCode.kDLInt, 64),
    "UINT8": (DLDataTypeCode.kDLUInt, 8),
    "UINT16": (DLDataTypeCode.kDLUInt, 16),
    "UINT32": (DLDataTypeCode.kDLUInt, 32),
    "UINT64": (DLDataTypeCode.kDLUInt, 64),
    "FP16": (DLDataTypeCode.kDLFloat, 16),
    "FP32": (DLDataTypeCode.kDLFloat, 32),
    "FP64": (DLDataTypeCode.kDLFloat, 64),
    "BF16": (DLDataTypeCode.kDLBfloat, 16),
}


def triton_to_dlpack_dtype(dtype):
    """Protocol datatype string -> :class:`DLDataType` (BYTES unsupported)."""
    if dtype == "BYTES":
        raise ValueError("DLPack cannot describe variable-length BYTES tensors")
    try:
        code, bits = _DTYPE_TABLE[dtype]
    except KeyError:
        raise ValueError(f"unknown datatype '{dtype}' has no DLPack equivalent") from None
    return DLDataType(code, bits, 1)


def is_contiguous_data(ndim, shape, strides):
    """True when (shape, strides) describe a compact row-major layout."""
    if strides is None or not bool(strides):
        return True
    expected = 1
    for axis in range(int(ndim) - 1, -1, -1):
        extent = shape[axis]
        if extent != 1 and strides[axis] != expected:
            return False
        expected *= extent
    return True


def get_byte_size(dtype, ndim, shape):
    """Number of bytes spanned by a compact tensor of ``dtype``/``shape``."""
    count = 1
    for axis in range(int(ndim)):
        count *= shape[axis]
    return (dtype.bits * dtype.lanes + 7) // 8 * count


# --- producer ---------------------------------------------------------------
_live = {}
_live_lock = threading.Lock()


@_DELETER
def _release_managed(address):
    with _live_lock:
        _live.pop(address, None)


@ctypes.CFUNCTYPE(None, ctypes.c_void_p)
def _capsule_destructor(capsule_ptr):
    capsule = ctypes.cast(capsule_ptr, ctypes.py_object)
    # A consumer renames the capsule to "used_dltensor" and then owns the
    # deleter call; only an unconsumed capsule is released here.
    if _api.PyCapsule_IsValid(capsule, c_str_dltensor):
        address = _api.PyCapsule_GetPointer(capsule, c_str_dltensor)
        _release_managed(address)


try:  # native producer: its deleter is safe to run during interpreter shutdown
    from .._lib import _tca_pyext as _ext  # type: ignore
except Exception:  # pragma: no cover
    _ext = None


def make_capsule(address, dl_dtype, shape, device_type, device_id, owner=None):
    """Wrap raw memory as a non-owning ``"dltensor"`` PyCapsule.

    ``owner`` (any Python object) is kept alive until the consumer calls the
    DLPack deleter, which is how a shared-memory region outlives the tensors
    that alias it.
    """
    if _ext is not None and hasattr(_ext, "make_dlpack_capsule"):
        return _ext.make_dlpack_capsule(int(address), int(dl_dtype.type_code), int(dl_dtype.bits),
                                        int(dl_dtype.lanes), [int(s) for s in shape], int(device_type),
                                        int(device_id), owner)
    ndim = len(shape)
    shape_arr = (ctypes.c_int64 * max(ndim, 1))(*shape)
    managed = DLManagedTensor()
    managed.dl_tensor.data = ctypes.c_void_p(address)
    managed.dl_tensor.device = DLDevice(int(device_type), int(device_id))
    managed.dl_tensor.ndim = ndim
    managed.dl_tensor.dtype = dl_dtype
    managed.dl_tensor.shape = ctypes.cast(shape_arr, ctypes.POINTER(ctypes.c_int64))
    managed.dl_tensor.strides = ctypes.POINTER(ctypes.c_int64)()
    managed.dl_tensor.byte_offset = 0
    managed.manager_ctx = None
    managed.deleter = _release_managed
    key = ctypes.addressof(managed)
    with _live_lock:
        _live[key] = (managed, shape_arr, owner)
    return _api.PyCapsule_New(key, c_str_dltensor, _capsule_destructor)


# --- consumer ---------------------------------------------------------------
def get_dlpack_device(obj):
    """``obj.__dlpack_device__()`` as a ``(type, id)`` tuple, or ``None``."""
    if hasattr(obj, "__dlpack_device__"):
        dev = obj.__dlpack_device__()
        return (int(dev[0]), int(dev[1]))
    return None


def get_dlpack_capsule(obj, stream=None):
    """Obtain the ``"dltensor"`` capsule of a DLPack exporter.

    ``stream`` is forwarded only to device exporters (``kDLROCM``/``kDLCUDA``):
    host exporters such as numpy reject the argument.
    """
    if hasattr(obj, "__dlpack__"):
        if not hasattr(obj, "__dlpack_device__"):
            raise TypeError("object defines __dlpack__ but not __dlpack_device__")
        dev_type = int(obj.__dlpack_device__()[0])
        if dev_type in (DLDeviceType.kDLROCM, DLDeviceType.kDLCUDA) and stream is not None:
            return obj.__dlpack__(stream=stream)
        return obj.__dlpack__()
    return obj  # legacy: a capsule was passed directly


def get_managed_tensor(capsule):
    """:class:`DLManagedTensor` view of a not-yet-consumed capsule."""
    address = _api.PyCapsule_GetPointer(capsule, c_str_dltensor)
    return DLManagedTensor.from_address(address)
