"""hipshm – device shared-memory regions on HIP IPC for MI355X.

Drop-in for ``tritonclient.utils.cuda_shared_memory`` (reference
``utils/cuda_shared_memory/__init__.py:107-429``): same function names and
semantics, but implemented natively (``csrc/hip/hipshm.cc``) on
``hipMalloc`` / ``hipIpcGetMemHandle`` / ``hipIpcOpenMemHandle`` with pinned
double-buffered staging.  ``import tritonclient.utils.cuda_shared_memory``
resolves to this module.

Client side
    ``create_shared_memory_region``, ``get_raw_handle``,
    ``set_shared_memory_region`` (numpy), ``set_shared_memory_region_from_dlpack``
    (torch-ROCm / numpy / any DLPack exporter), ``get_contents_as_numpy``
    (copies only the requested range), ``as_shared_memory_tensor`` (zero-copy
    ``kDLROCM`` view), ``allocated_shared_memory_regions``,
    ``destroy_shared_memory_region``.
Server side
    ``open_shared_memory_region`` / ``close_shared_memory_region`` (the opening
    process must differ from the exporting one – a HIP IPC rule),
    ``mapping_as_torch``, ``read_mapping``, ``write_mapping``.

Setting ``TCA_HIPSHM_FAKE=1`` switches to host-backed regions with the same
API for machines without a GPU (protocol tests); it is never chosen silently.
"""

from __future__ import annotations

import base64
import ctypes
from ctypes import byref, c_int, c_uint8, c_uint64, c_void_p

import numpy as np

from ... import _native
from .. import _dlpack, triton_to_np_dtype
from .._shared_memory_tensor import SharedMemoryTensor

__all__ = [
    "HipSharedMemoryException",
    "CudaSharedMemoryException",
    "HipSharedMemoryRegion",
    "CudaSharedMemoryRegion",
    "create_shared_memory_region",
    "get_raw_handle",
    "set_shared_memory_region",
    "get_contents_as_numpy",
    "set_shared_memory_region_from_dlpack",
    "as_shared_memory_tensor",
    "allocated_shared_memory_regions",
    "destroy_shared_memory_region",
    "open_shared_memory_region",
    "close_shared_memory_region",
    "mapping_as_torch",
    "read_mapping",
    "write_mapping",
    "device_memory_info",
]


class HipSharedMemoryException(Exception):
    """Failure of a device shared-memory operation."""

    def __init__(self, msg):
        self._msg = msg
        super().__init__(msg)

    def __str__(self):
        return self._msg


# name used by code written against the reference module
CudaSharedMemoryException = HipSharedMemoryException


def _lib():
    return _native.hip()


# Copies up to this many bytes run with the GIL held: they take ~10 microseconds, less
# than handing the interpreter to another thread and queueing to get it back (measured in
# the serving process: BYTES tensors in device regions at concurrency 16).
_SMALL_COPY_BYTES = 64 << 10


def _copy_lib(nbytes):
    return _native.hip_gil_held() if nbytes <= _SMALL_COPY_BYTES else _native.hip()


def _check(rc, what):
    if rc != 0:
        detail = _lib().tca_hip_last_error().decode("utf-8", "replace")
        raise HipSharedMemoryException(f"{what}: {detail}" if detail else what)


class HipSharedMemoryRegion:
    """Handle of a region (owned allocation or opened IPC mapping)."""

    def __init__(self, native, opened=False):
        self._h = native
        self._opened = opened
        ptr, dev, size, host = c_void_p(), c_int(), c_uint64(), c_int()
        raw = (c_uint8 * 64)()
        _check(_lib().tca_hipshm_info(native, byref(ptr), raw, byref(dev), byref(size), byref(host)),
               "unable to query device shared memory region")
        self._base_addr = ptr.value or 0
        self._raw = bytes(raw)
        self._device_id = dev.value
        self._byte_size = size.value
        self._host_backed = bool(host.value)

    @property
    def byte_size(self):
        return self._byte_size

    @property
    def device_id(self):
        return self._device_id

    @property
    def address(self):
        return self._base_addr

    @property
    def host_backed(self):
        return self._host_backed

    def _native(self):
        if self._h is None:
            raise HipSharedMemoryException("device shared memory region has been destroyed")
        return self._h

    def _dlpack_device_id(self):
        return -1 if self._host_backed else self._device_id

    def __repr__(self):
        return (f"HipSharedMemoryRegion(device={self._device_id}, byte_size={self._byte_size}, "
                f"host_backed={self._host_backed})")


CudaSharedMemoryRegion = HipSharedMemoryRegion

_allocated = []


def device_memory_info(device_id=0):
    """``(free_bytes, total_bytes)`` of a GPU (288 GB HBM3E on MI355X)."""
    free, total = c_uint64(), c_uint64()
    _check(_lib().tca_hip_mem_info(int(device_id), byref(free), byref(total)), "unable to query device memory")
    return free.value, total.value


def create_shared_memory_region(triton_shm_name, byte_size, device_id, fine_grained=False):
    """Allocate ``byte_size`` bytes on GPU ``device_id`` and export an IPC handle.

    ``fine_grained=True`` (not in the reference API) asks for zero-filled memory that stays
    coherent between GPUs while kernels run: for flags that one GPU sets and another polls."""
    h = c_void_p()
    name = triton_shm_name.encode("utf-8") if isinstance(triton_shm_name, str) else bytes(triton_shm_name)
    rc = _lib().tca_hipshm_create_ex(name, int(byte_size), int(device_id), 1 if fine_grained else 0, byref(h))
    _check(rc, "unable to create device shared memory handle")
    region = HipSharedMemoryRegion(h)
    region._name = triton_shm_name
    _allocated.append(region)
    return region


def get_raw_handle(shm_handle):
    """Base64 of the 64-byte HIP IPC handle (what ``register_cuda_shared_memory`` takes)."""
    return base64.b64encode(shm_handle._raw)


def _as_buffer(value):
    """``(address, nbytes, keepalive)`` of a numpy array / serialized BYTES carrier."""
    if value.dtype == np.object_:
        blob = value.item() if value.size else b""
        if not isinstance(blob, (bytes, bytearray)):
            raise HipSharedMemoryException(
                "object arrays must be serialized with serialize_byte_tensor before being placed in shared memory")
        buf = (ctypes.c_char * len(blob)).from_buffer_copy(blob) if len(blob) else None
        return (ctypes.addressof(buf) if buf is not None else 0), len(blob), buf
    arr = np.ascontiguousarray(value)
    return arr.ctypes.data, arr.nbytes, arr


def set_shared_memory_region(shm_handle, input_values, offset=0):
    """Copy numpy arrays back-to-back into the region (host -> device)."""
    if not isinstance(input_values, (list, tuple)):
        raise HipSharedMemoryException("input_values must be specified as a list/tuple of numpy arrays")
    for value in input_values:
        if not isinstance(value, np.ndarray):
            raise HipSharedMemoryException("input_values must be specified as a list/tuple of numpy arrays")
    cursor = int(offset)
    for value in input_values:
        addr, nbytes, keep = _as_buffer(value)
        _check(_lib().tca_hipshm_write(shm_handle._native(), cursor, addr, nbytes),
               "unable to set values in device shared memory")
        cursor += nbytes
        del keep


def _read(shm_handle, offset, nbytes):
    out = np.empty(nbytes, dtype=np.uint8)
    _check(_copy_lib(nbytes).tca_hipshm_read(shm_handle._native(), int(offset), out.ctypes.data, int(nbytes)),
           "failed to read device shared memory results")
    return out


def get_contents_as_numpy(shm_handle, datatype, shape, offset=0):
    """Device -> host copy of just the bytes the requested tensor occupies."""
    dt = np.dtype(datatype)
    count = int(np.prod(shape)) if len(shape) else 1
    if dt != np.object_ and dt.type != np.bytes_:
        need = count * dt.itemsize
        if offset + need > shm_handle.byte_size:
            raise HipSharedMemoryException(
                "The size of the shared memory region is insufficient to provide numpy array with requested size")
        if need == 0:
            return np.empty(shape, dtype=dt)
        return _read(shm_handle, offset, need).view(dt).reshape(shape)
    # BYTES: element sizes are unknown up front
    try:
        items = _read_bytes_elements(shm_handle, count, int(offset))
    except ValueError:
        raise HipSharedMemoryException("malformed BYTES tensor in device shared memory region") from None
    out = np.empty(count, dtype=np.object_)
    for i, item in enumerate(items):
        out[i] = item
    return out.reshape(shape)


_INDEX_MAGIC = 0x58444942  # "BIDX": index trailer written behind BYTES results by the server's device path
_WALK_CHUNK = 4096


def _read_bytes_elements(shm_handle, count, offset):
    """``count`` BYTES elements that start at ``offset``, copying only what they occupy.

    Fast path: a server that produced the tensor on the GPU leaves an index in the last bytes of
    the region ({magic, count, payload starts, end of stream}).  If it is there AND agrees with
    the length prefixes of the stream, the stream is fetched in one copy of exactly its size.
    Otherwise the prefixes are walked over reads of a few KiB, growing as needed - never the whole
    tail of the region (the reference copies the entire region first:
    ``tritonclient/utils/cuda_shared_memory/__init__.py:265-276``)."""
    if count == 0:
        return []
    size = shm_handle.byte_size - offset
    need = 12 + 4 * count
    if size >= need + 3:
        at = ((shm_handle.address + shm_handle.byte_size - need) & ~3) - shm_handle.address
        if at >= offset + 4 * count:
            t = _read(shm_handle, at, need).view(np.uint32)
            if int(t[0]) == _INDEX_MAGIC and int(t[1]) == count:
                starts = t[2: 2 + count].astype(np.int64)
                end = int(t[2 + count])
                if int(starts[0]) == 4 and end <= at - offset and bool(np.all(np.diff(starts) >= 4)) and \
                        int(starts[-1]) <= end:
                    blob = _read(shm_handle, offset, end)
                    nxt = np.append(starts[1:] - 4, end)
                    lens = nxt - starts
                    idx = starts - 4
                    prefix = (blob[idx].astype(np.int64) | (blob[idx + 1].astype(np.int64) << 8)
                              | (blob[idx + 2].astype(np.int64) << 16) | (blob[idx + 3].astype(np.int64) << 24))
                    if bool(np.all(prefix == lens)):  # the index describes THIS stream
                        raw = blob.tobytes()
                        return [raw[int(s0): int(s0 + n)] for s0, n in zip(starts, lens)]
    # walk the prefixes over incremental reads
    have = np.empty(0, dtype=np.uint8)
    items, pos = [], 0
    chunk = _WALK_CHUNK
    while len(items) < count:
        if pos + 4 > len(have) or (pos + 4 <= len(have) and pos + 4 + int.from_bytes(have[pos: pos + 4].tobytes(),
                                                                                     "little") > len(have)):
            want = pos + 4 if pos + 4 > len(have) else pos + 4 + int.from_bytes(have[pos: pos + 4].tobytes(), "little")
            if want > size:
                raise ValueError("stream runs past the region")
            upto = min(size, max(want, len(have) + chunk))
            more = _read(shm_handle, offset + len(have), upto - len(have))
            have = np.concatenate([have, more]) if len(have) else more
            chunk *= 2
            continue
        n = int.from_bytes(have[pos: pos + 4].tobytes(), "little")
        items.append(have[pos + 4: pos + 4 + n].tobytes())
        pos += 4 + n
    return items


_DEVICE_TYPES = (_dlpack.DLDeviceType.kDLROCM, _dlpack.DLDeviceType.kDLCUDA)
_HOST_TYPES = (_dlpack.DLDeviceType.kDLCPU, _dlpack.DLDeviceType.kDLROCMHost, _dlpack.DLDeviceType.kDLCUDAHost)


def set_shared_memory_region_from_dlpack(shm_handle, input_values, offset=0):
    """Copy DLPack tensors (torch-ROCm on any GPU, numpy, ...) into the region.

    Device sources are copied device-to-device (over xGMI when the source sits
    on a peer GPU); host sources go through the pinned staging ring.
    """
    cursor = int(offset)
    for value in input_values:
        device = _dlpack.get_dlpack_device(value)
        if device is not None and device[0] not in _DEVICE_TYPES + _HOST_TYPES:
            raise HipSharedMemoryException("DLPack device type {} is not supported".format(device[0]))
        # order the producer's work before our copy: torch hands out the capsule
        # after synchronising its current stream with the legacy default stream
        if device is not None and device[0] in _DEVICE_TYPES:
            try:
                capsule = value.__dlpack__(stream=0)
            except TypeError:
                capsule = value.__dlpack__()
        else:
            capsule = _dlpack.get_dlpack_capsule(value)
        managed = _dlpack.get_managed_tensor(capsule)
        t = managed.dl_tensor
        if not _dlpack.is_contiguous_data(t.ndim, t.shape, t.strides):
            raise HipSharedMemoryException(
                "DLPack tensor is not contiguous. Only contiguous DLPack tensors that are stored in C-Order are "
                "supported.")
        nbytes = _dlpack.get_byte_size(t.dtype, t.ndim, t.shape)
        src = (t.data or 0) + t.byte_offset
        src_device = t.device.device_id if t.device.device_type in _DEVICE_TYPES else -1
        if src_device >= 0 and shm_handle.host_backed:
            raise HipSharedMemoryException("host-backed (fake) regions cannot ingest device tensors")
        _check(_lib().tca_hipshm_copy_from(shm_handle._native(), cursor, src, nbytes, src_device),
               "unable to set values in device shared memory")
        cursor += nbytes
        del capsule


def set_shared_memory_region_bf16(shm_handle, input_values, offset=0, round_nearest=False):
    """fp32 tensors -> BF16 wire form (2 bytes per element, the upper half of the fp32 bits; the
    reference truncates: ``tritonclient/utils/__init__.py:310-320``) back-to-back in the region.

    fp32 tensors that live on a GPU (anything with ``__dlpack__``, e.g. torch-ROCm) are converted
    by the GPU kernel straight into the region - no host copy, no intermediate tensor; host arrays
    are truncated on the host and copied (half the bytes cross the bus).  Not in the reference API,
    which has no device-side producer.  Returns the number of bytes written."""
    from .. import serialize_bf16_tensor

    cursor = int(offset)
    for value in input_values:
        device = _dlpack.get_dlpack_device(value) if hasattr(value, "__dlpack__") else None
        if device is not None and device[0] in _DEVICE_TYPES and not shm_handle.host_backed:
            try:
                capsule = value.__dlpack__(stream=0)
            except TypeError:
                capsule = value.__dlpack__()
            t = _dlpack.get_managed_tensor(capsule).dl_tensor
            if not _dlpack.is_contiguous_data(t.ndim, t.shape, t.strides):
                raise HipSharedMemoryException("DLPack tensor is not contiguous")
            if (int(t.dtype.type_code), int(t.dtype.bits), int(t.dtype.lanes)) != (2, 32, 1):  # kDLFloat
                raise HipSharedMemoryException("BF16 regions are filled from fp32 tensors")
            if t.device.device_id != shm_handle.device_id:
                raise HipSharedMemoryException("the tensor and the region sit on different GPUs")
            count = 1
            for i in range(t.ndim):
                count *= t.shape[i]
            if cursor + 2 * count > shm_handle.byte_size:
                raise HipSharedMemoryException("the region is too small for the tensor")
            if cursor % 2:
                raise HipSharedMemoryException("BF16 tensors start at even offsets")
            lib = _lib()
            rc = lib.tca_fp32_to_bf16_launch((t.data or 0) + t.byte_offset, shm_handle.address + cursor, count,
                                             1 if round_nearest else 0, None)
            _check(rc, "fp32 -> bf16 conversion failed")
            _check(lib.tca_hip_device_synchronize(shm_handle.device_id), "unable to synchronize the device")
            cursor += 2 * count
            del capsule
            continue
        arr = np.ascontiguousarray(np.asarray(value if not hasattr(value, "cpu") else value.cpu()), dtype=np.float32)
        if round_nearest:
            bits = arr.view(np.uint32).astype(np.uint64)
            packed = (((bits + 0x7FFF + ((bits >> 16) & 1)) >> 16) & 0xFFFF).astype(np.uint16)
            nan = np.isnan(arr)
            if nan.any():
                packed[nan] = (arr.view(np.uint32)[nan] >> 16).astype(np.uint16) | 0x40
        else:
            blob = serialize_bf16_tensor(arr)
            packed = np.frombuffer(blob.item() if blob.size else b"", dtype=np.uint16)
        set_shared_memory_region(shm_handle, [packed], offset=cursor)
        cursor += packed.nbytes
    return cursor - int(offset)


def get_contents_as_numpy_bf16(shm_handle, shape, offset=0):
    """A BF16 tensor of ``shape`` in the region as a float32 array: two bytes per element cross the
    bus and are widened on the host."""
    count = int(np.prod(shape)) if len(shape) else 1
    if offset + 2 * count > shm_handle.byte_size:
        raise HipSharedMemoryException(
            "The size of the shared memory region is insufficient to provide numpy array with requested size")
    half = _read(shm_handle, offset, 2 * count).view(np.uint16)
    return (half.astype(np.uint32) << 16).view(np.float32).reshape(shape)


def as_shared_memory_tensor(shm_handle, datatype, shape, offset=0):
    """Zero-copy DLPack view (``kDLROCM``) of the region, e.g. for ``torch.from_dlpack``."""
    return SharedMemoryTensor(datatype, shape, shm_handle._base_addr, offset, shm_handle._byte_size - offset,
                              shm_handle._dlpack_device_id(), owner=shm_handle)


def allocated_shared_memory_regions():
    """Regions created by this process and not yet destroyed."""
    return _allocated


def destroy_shared_memory_region(shm_handle):
    """Free the device memory (unregister it from the server first)."""
    try:
        _allocated.remove(shm_handle)
    except ValueError:
        pass
    h, shm_handle._h = shm_handle._h, None
    if h is not None:
        _check(_lib().tca_hipshm_destroy(h), "unable to free device shared memory")


# ---------------------------------------------------------------- server side
def open_shared_memory_region(name, raw_handle, byte_size, device_id):
    """Map a region exported by another process from its 64 raw handle bytes."""
    if len(raw_handle) != 64:
        raise HipSharedMemoryException(f"a HIP IPC handle is 64 bytes, got {len(raw_handle)}")
    h = c_void_p()
    rc = _lib().tca_hipshm_open(name.encode("utf-8"), bytes(raw_handle), int(byte_size), int(device_id), byref(h))
    _check(rc, "unable to open device shared memory handle")
    return HipSharedMemoryRegion(h, opened=True)


def close_shared_memory_region(mapping):
    h, mapping._h = mapping._h, None
    if h is not None:
        _check(_lib().tca_hipshm_destroy(h), "unable to close device shared memory mapping")


def mapping_as_torch(mapping, datatype, shape, offset=0):
    """``torch`` tensor aliasing the mapping (device tensor, or CPU tensor for
    host-backed regions)."""
    import torch

    smt = SharedMemoryTensor(datatype, shape, mapping._base_addr, offset, mapping._byte_size - offset,
                             mapping._dlpack_device_id(), owner=mapping)
    return torch.from_dlpack(smt)


def read_mapping(mapping, offset, nbytes):
    """Bytes ``[offset, offset+nbytes)`` of the mapping as a host buffer."""
    return _read(mapping, offset, nbytes).data


def read_mappings(items):
    """``[(mapping, offset, nbytes), ...]`` -> list of host buffers, moved in one
    native call (all copies enqueued, one wait): the inputs of one request."""
    n = len(items)
    if n == 1:
        return [read_mapping(*items[0])]
    outs = [np.empty(int(nbytes), dtype=np.uint8) for _, _, nbytes in items]
    handles = (c_void_p * n)(*[m._native() for m, _, _ in items])
    offsets = (c_uint64 * n)(*[int(off) for _, off, _ in items])
    dsts = (c_void_p * n)(*[o.ctypes.data for o in outs])
    sizes = (c_uint64 * n)(*[o.nbytes for o in outs])
    _check(_copy_lib(sum(o.nbytes for o in outs)).tca_hipshm_read_many(n, handles, offsets, dsts, sizes),
           "failed to read device shared memory")
    return [o.data for o in outs]


def write_mappings(items):
    """``[(mapping, offset, blob), ...]``: the outputs of one request in one native call."""
    n = len(items)
    if n == 1:
        return write_mapping(*items[0])
    bufs = [np.frombuffer(blob, dtype=np.uint8) for _, _, blob in items]
    handles = (c_void_p * n)(*[m._native() for m, _, _ in items])
    offsets = (c_uint64 * n)(*[int(off) for _, off, _ in items])
    srcs = (c_void_p * n)(*[b.ctypes.data for b in bufs])
    sizes = (c_uint64 * n)(*[b.nbytes for b in bufs])
    _check(_copy_lib(sum(b.nbytes for b in bufs)).tca_hipshm_write_many(n, handles, offsets, srcs, sizes),
           "unable to write device shared memory")


def write_mapping(mapping, offset, blob):
    buf = np.frombuffer(blob, dtype=np.uint8)
    _check(_copy_lib(buf.nbytes).tca_hipshm_write(mapping._native(), int(offset), buf.ctypes.data, buf.nbytes),
           "unable to write device shared memory")
