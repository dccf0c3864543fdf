"""Loader for the in-tree native libraries.

``core()`` returns the ctypes handle of ``libtca_core.so`` (host side), ``hip()``
that of ``libtca_hip.so`` (hipshm + CDNA4 kernels).  A missing library is
compiled on first use through :mod:`triton_client_amd._build`; a library that
cannot be built raises ``RuntimeError`` – there is no silent pure-Python
substitute for device code.
"""

from __future__ import annotations

import ctypes
import os
import threading
from ctypes import POINTER, c_char_p, c_float, c_int, c_int64, c_uint8, c_uint16, c_uint32, c_uint64, c_void_p

_lock = threading.Lock()
_core = None
_hip = None

LIBDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lib")


def _ensure(target, filename):
    path = os.path.join(LIBDIR, filename)
    if not os.path.exists(path) or os.environ.get("TCA_REBUILD") == "1":
        from . import _build

        getattr(_build, "build_" + target)()
    if not os.path.exists(path):
        raise RuntimeError(f"native library {filename} is missing and could not be built")
    return path


def core():
    global _core
    if _core is not None:
        return _core
    with _lock:
        if _core is not None:
            return _core
        lib = ctypes.CDLL(_ensure("core", "libtca_core.so"))
        lib.tca_strerror.restype = c_char_p
        lib.tca_strerror.argtypes = [c_int]
        lib.tca_shm_create.restype = c_int
        lib.tca_shm_create.argtypes = [c_char_p, c_char_p, c_uint64, POINTER(c_void_p)]
        lib.tca_shm_attach.restype = c_int
        lib.tca_shm_attach.argtypes = [c_char_p, c_char_p, c_uint64, c_uint64, POINTER(c_void_p)]
        lib.tca_shm_write.restype = c_int
        lib.tca_shm_write.argtypes = [c_void_p, c_uint64, c_uint64, c_void_p]
        lib.tca_shm_read.restype = c_int
        lib.tca_shm_read.argtypes = [c_void_p, c_uint64, c_uint64, c_void_p]
        lib.tca_shm_info.restype = c_int
        lib.tca_shm_info.argtypes = [c_void_p, POINTER(c_void_p), POINTER(c_char_p), POINTER(c_char_p),
                                     POINTER(c_uint64), POINTER(c_uint64)]
        lib.tca_shm_destroy.restype = c_int
        lib.tca_shm_destroy.argtypes = [c_void_p, c_int]
        lib.tca_bytes_index.restype = c_int64
        lib.tca_bytes_index.argtypes = [c_void_p, c_uint64, c_int64, POINTER(c_uint64), POINTER(c_uint32), c_uint64]
        lib.tca_fp32_to_bf16.restype = None
        lib.tca_fp32_to_bf16.argtypes = [POINTER(c_float), POINTER(c_uint16), c_uint64]
        lib.tca_bf16_to_fp32.restype = None
        lib.tca_bf16_to_fp32.argtypes = [POINTER(c_uint16), POINTER(c_float), c_uint64]
        lib.tca_base64_encode.restype = c_uint64
        lib.tca_base64_encode.argtypes = [c_void_p, c_uint64, c_char_p, c_uint64]
        lib.tca_base64_decode.restype = c_int64
        lib.tca_base64_decode.argtypes = [c_char_p, c_uint64, c_void_p, c_uint64]
        lib.tca_decode_images.restype = c_int
        lib.tca_decode_images.argtypes = [c_int, POINTER(c_void_p), POINTER(c_uint64), c_int, c_int, c_void_p]
        lib.tca_free_images.restype = None
        lib.tca_free_images.argtypes = [c_int, c_void_p]
        _core = lib
    return _core


def _preload_torch_hip_runtime():
    """Make PyTorch-ROCm and our library share ONE HIP runtime.

    The torch wheel bundles its own ``libamdhip64.so`` (SONAME
    ``libamdhip64.so.7``).  If ``/opt/rocm/lib/libamdhip64.so.7`` were mapped
    first, torch would later map its private copy as well and the process
    would hold two HIP/HSA runtimes - the second one then reports "no
    ROCm-capable device".  Mapping torch's copy first makes our
    ``DT_NEEDED libamdhip64.so.7`` resolve to it by SONAME.
    """
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
            except OSError:
                pass


def hip():
    """ctypes handle of libtca_hip.so with prototypes installed by :mod:`ops._bind`."""
    global _hip
    if _hip is not None:
        return _hip
    with _lock:
        if _hip is not None:
            return _hip
        _preload_torch_hip_runtime()
        lib = ctypes.CDLL(_ensure("hip", "libtca_hip.so"))
        from .ops import _bind

        _bind.install(lib)
        _hip = lib
    return _hip


_hip_held = None


def hip_gil_held():
    """The same library through ``ctypes.PyDLL``: calls keep the GIL.  For native calls
    of a few microseconds (small shared-memory copies) in a process with many busy
    threads, where giving the interpreter away and queueing to get it back costs far
    more than the call itself."""
    global _hip_held
    if _hip_held is not None:
        return _hip_held
    hip()
    with _lock:
        if _hip_held is None:
            lib = ctypes.PyDLL(_ensure("hip", "libtca_hip.so"))
            from .ops import _bind

            _bind.install(lib)
            _hip_held = lib
    return _hip_held


def pyext():
    """The ``_tca_pyext`` CPython extension (built on demand): BYTES codec and
    the native gRPC serving front."""
    try:
        from ._lib import _tca_pyext
    except ImportError:
        from . import _build

        _build.build_pyext()
        from ._lib import _tca_pyext
    if not hasattr(_tca_pyext, "GrpcFront"):
        raise ImportError("_tca_pyext was built without the gRPC front; rebuild: "
                          "python -m triton_client_amd._build pyext")
    return _tca_pyext


def pyext_available():
    try:
        from ._lib import _tca_pyext  # noqa: F401

        return True
    except Exception:
        return False
