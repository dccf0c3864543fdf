"""POSIX system shared-memory regions (client and server side).

Public API parity with ``tritonclient.utils.shared_memory``
(reference ``utils/shared_memory/__init__.py:93-340``):
``create_shared_memory_region``, ``set_shared_memory_region``,
``get_contents_as_numpy``, ``mapped_shared_memory_regions``,
``destroy_shared_memory_region`` and ``SharedMemoryException``.

Differences by design:

* the handle is a :class:`SharedMemoryRegion` object (not a bare ``c_void_p``)
  that exposes the mapping as a ``memoryview`` – reads are zero-copy numpy
  views of the mapping, exactly like the reference, but bounds are checked;
* ``attach_shared_memory_region`` maps an *existing* key at an offset – this is
  what the in-repo v2 server uses to honour ``register_system_shared_memory``;
* BYTES reads stop after ``prod(shape)`` elements and are validated against the
  region size instead of walking off the end of the mapping.
"""

from __future__ import annotations

import ctypes
from ctypes import byref, c_char_p, c_uint64, c_void_p

import numpy as np

from ... import _native
from .. import deserialize_bytes_tensor  # noqa: F401  (handy re-export)

__all__ = [
    "SharedMemoryException",
    "SharedMemoryRegion",
    "create_shared_memory_region",
    "attach_shared_memory_region",
    "set_shared_memory_region",
    "get_contents_as_numpy",
    "mapped_shared_memory_regions",
    "destroy_shared_memory_region",
]


class SharedMemoryException(Exception):
    """Failure of a shared-memory operation.

    ``err`` is either a message or a native error code (negative int).
    """

    def __init__(self, err):
        if isinstance(err, str):
            self._msg = err
        else:
            code = getattr(err, "value", err)
            self._msg = _native.core().tca_strerror(int(code)).decode()
        super().__init__(self._msg)

    def __str__(self):
        return self._msg


def _check(code):
    if code != 0:
        raise SharedMemoryException(code)


def _b(text):
    return text if isinstance(text, bytes) else str(text).encode("utf-8")


class SharedMemoryRegion:
    """A mapped shm object.  Created via the module level functions."""

    def __init__(self, native_handle, creator):
        self._h = native_handle
        self._creator = creator
        addr, name, key = c_void_p(), c_char_p(), c_char_p()
        off, size = c_uint64(), c_uint64()
        _check(_native.core().tca_shm_info(self._h, byref(addr), byref(name), byref(key), byref(off), byref(size)))
        self._addr = addr.value or 0
        self._name = name.value.decode()
        self._key = key.value.decode()
        self._offset = off.value
        self._size = size.value
        if self._size:
            self._carray = (ctypes.c_ubyte * self._size).from_address(self._addr)
            self._view = memoryview(self._carray).cast("B")
        else:
            self._carray = None
            self._view = memoryview(b"")

    # -- introspection --
    @property
    def name(self):
        return self._name

    @property
    def key(self):
        return self._key

    @property
    def byte_size(self):
        return self._size

    @property
    def offset(self):
        return self._offset

    @property
    def address(self):
        return self._addr

    def buffer(self):
        """Writable ``memoryview`` over the whole mapping."""
        if self._h is None:
            raise SharedMemoryException("shared memory region has been destroyed")
        return self._view

    def _close(self, unlink):
        if self._h is None:
            return
        h, self._h = self._h, None
        try:
            self._view.release()
        except BufferError:
            # numpy views handed out by get_contents_as_numpy still alias the
            # mapping; the caller asked for destruction, so drop ours anyway.
            pass
        self._view = None
        self._carray = None
        _check(_native.core().tca_shm_destroy(h, 1 if unlink else 0))

    def __repr__(self):
        return f"SharedMemoryRegion(name={self._name!r}, key={self._key!r}, byte_size={self._size})"


_mapped_keys = []


def create_shared_memory_region(triton_shm_name, shm_key, byte_size):
    """Create (or open) the shm object ``shm_key``, size it and map it."""
    h = c_void_p()
    _check(_native.core().tca_shm_create(_b(triton_shm_name), _b(shm_key), int(byte_size), byref(h)))
    region = SharedMemoryRegion(h, creator=True)
    _mapped_keys.append(region.key)
    return region


def attach_shared_memory_region(triton_shm_name, shm_key, byte_size, offset=0):
    """Map ``byte_size`` bytes at ``offset`` of an existing shm object."""
    h = c_void_p()
    _check(_native.core().tca_shm_attach(_b(triton_shm_name), _b(shm_key), int(offset), int(byte_size), byref(h)))
    return SharedMemoryRegion(h, creator=False)


def set_shared_memory_region(shm_handle, input_values, offset=0):
    """Copy numpy arrays back-to-back into the region starting at ``offset``.

    BYTES tensors must already be serialised with
    :func:`triton_client_amd.utils.serialize_byte_tensor` (object carrier).
    """
    if not isinstance(input_values, (list, tuple)):
        raise SharedMemoryException("input_values must be specified as a list/tuple of numpy arrays")
    for value in input_values:
        if not isinstance(value, np.ndarray):
            raise SharedMemoryException("each element of input_values must be a numpy array")
    view = shm_handle.buffer()
    cursor = int(offset)
    for value in input_values:
        if value.dtype == np.object_:
            blob = value.item() if value.size else b""
            if not isinstance(blob, (bytes, bytearray, memoryview)):
                raise SharedMemoryException(
                    "object arrays must be serialized with serialize_byte_tensor before being placed in shared memory"
                )
            src = memoryview(blob).cast("B")
        else:
            src = memoryview(np.ascontiguousarray(value)).cast("B")
        n = src.nbytes
        if cursor < 0 or cursor + n > shm_handle.byte_size:
            raise SharedMemoryException(-7)
        view[cursor : cursor + n] = src
        cursor += n


def get_contents_as_numpy(shm_handle, datatype, shape, offset=0):
    """View (fixed-size dtypes) or decode (BYTES) the region as an ndarray."""
    view = shm_handle.buffer()
    start = int(offset)
    count = int(np.prod(shape)) if len(shape) else 1
    dt = np.dtype(datatype)
    if dt != np.object_ and dt.type != np.bytes_:
        need = count * dt.itemsize
        if start + need > shm_handle.byte_size:
            raise SharedMemoryException(
                "The size of the shared memory region is insufficient to provide numpy array with requested size"
            )
        if need == 0:
            return np.empty(shape, dtype=dt)
        return np.frombuffer(view, dtype=dt, count=count, offset=start).reshape(shape)
    from .. import _fast

    window = view[start:]
    try:
        if _fast is not None:
            items = _fast.unpack_bytes(window, count)
        else:
            import struct

            items, pos = [], 0
            for _ in range(count):
                (n,) = struct.unpack_from("<I", window, pos)
                pos += 4
                if pos + n > len(window):
                    raise ValueError
                items.append(bytes(window[pos : pos + n]))
                pos += n
    except (ValueError, Exception) as exc:  # struct.error subclasses Exception
        if isinstance(exc, SharedMemoryException):
            raise
        raise SharedMemoryException(-8) from None
    out = np.empty(count, dtype=np.object_)
    for i, item in enumerate(items):
        out[i] = item
    return out.reshape(shape)


def mapped_shared_memory_regions():
    """Keys of the regions created by this process and not yet destroyed."""
    return _mapped_keys


def destroy_shared_memory_region(shm_handle):
    """Unmap the region; the creator also unlinks the shm object."""
    if shm_handle._creator:
        try:
            _mapped_keys.remove(shm_handle.key)
        except ValueError:
            pass
    shm_handle._close(unlink=shm_handle._creator)
