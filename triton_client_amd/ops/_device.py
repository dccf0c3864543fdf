"""Small device-memory toolkit for the ops (works with or without torch).

``DeviceBuffer`` / ``PinnedBuffer`` wrap raw HIP allocations, ``Stream`` a HIP
stream.  ``as_pointer`` accepts a torch tensor, a hipshm region, a
``DeviceBuffer`` or a plain integer address, which is how every op in this
package takes its operands.
"""

from __future__ import annotations

import ctypes
from ctypes import byref, c_void_p

import numpy as np

from .. import _native


class HipError(RuntimeError):
    pass


def lib():
    return _native.hip()


def check(rc, what):
    if rc != 0:
        L = lib()
        detail = (L.tca_hip_rt_error() or b"").decode() or (L.tca_hip_last_error() or b"").decode()
        raise HipError(f"{what} failed (code {rc}) {detail}")


class Stream:
    """Owns a non-blocking HIP stream on ``device``."""

    def __init__(self, device=0):
        self.device = device
        h = c_void_p()
        check(lib().tca_hip_stream_create(int(device), byref(h)), "hipStreamCreate")
        self.handle = h.value

    def synchronize(self):
        check(lib().tca_hip_stream_sync(self.handle), "hipStreamSynchronize")

    def close(self):
        if self.handle:
            lib().tca_hip_stream_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


class DeviceBuffer:
    def __init__(self, nbytes, device=0):
        self.nbytes = int(nbytes)
        self.device = device
        p = c_void_p()
        check(lib().tca_hip_malloc(int(device), self.nbytes, byref(p)), "hipMalloc")
        self.ptr = p.value

    def upload(self, array, stream=None, offset=0):
        """Async H2D of a (preferably pinned) numpy array."""
        a = np.ascontiguousarray(array)
        check(lib().tca_hip_memcpy_async(self.ptr + offset, a.ctypes.data, a.nbytes, 1, stream_handle(stream)),
              "hipMemcpyAsync(H2D)")
        return a  # caller keeps it alive until the stream is synchronised

    def download(self, nbytes=None, offset=0, stream=None):
        n = self.nbytes - offset if nbytes is None else int(nbytes)
        out = np.empty(n, dtype=np.uint8)
        check(lib().tca_hip_memcpy_async(out.ctypes.data, self.ptr + offset, n, 2, stream_handle(stream)),
              "hipMemcpyAsync(D2H)")
        check(lib().tca_hip_stream_sync(stream_handle(stream)), "hipStreamSynchronize")
        return out

    def close(self):
        if self.ptr:
            lib().tca_hip_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PinnedBuffer:
    """Page-locked host memory exposed as a uint8 numpy array (``.array``)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = c_void_p()
        check(lib().tca_hip_host_alloc(self.nbytes, byref(p)), "hipHostMalloc")
        self.ptr = p.value
        self.array = np.ctypeslib.as_array((ctypes.c_uint8 * max(1, self.nbytes)).from_address(self.ptr))

    def close(self):
        if self.ptr:
            self.array = None
            lib().tca_hip_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def stream_handle(stream):
    """HIP stream handle (int) of a ``Stream``, a ``torch.cuda.Stream``, an int or None (null stream)."""
    if stream is None:
        return None
    if isinstance(stream, Stream):
        return stream.handle
    if hasattr(stream, "cuda_stream"):
        return stream.cuda_stream
    return int(stream)


def as_pointer(obj, offset=0):
    """Device address of ``obj`` (+ ``offset`` bytes)."""
    if obj is None:
        return None
    if isinstance(obj, int):
        return obj + offset
    if isinstance(obj, DeviceBuffer):
        return obj.ptr + offset
    if hasattr(obj, "data_ptr"):  # torch tensor
        return obj.data_ptr() + offset
    if hasattr(obj, "address"):  # hipshm region / mapping
        return obj.address + offset
    raise TypeError(f"cannot take a device pointer of {type(obj).__name__}")
