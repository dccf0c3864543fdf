"""Tensor codec and error types shared by every transport.

Behavioural parity target: ``src/python/library/tritonclient/utils/__init__.py``
(reference lines 36-348): same public names, same wire formats

* BYTES  : row-major concatenation of ``<u32 little-endian length><payload>``
* BF16   : upper two bytes of the IEEE fp32 pattern (truncation, no rounding)

The implementation is table driven and vectorised (numpy views instead of a
per-element ``struct`` loop); the BYTES packer/unpacker additionally uses the
native ``_tca_pyext`` CPython extension when it has been built.
"""

from __future__ import annotations

import struct as _struct

import numpy as np

from ._shared_memory_tensor import SharedMemoryTensor  # noqa: F401  (re-export)

__all__ = [
    "raise_error",
    "serialized_byte_size",
    "InferenceServerException",
    "np_to_triton_dtype",
    "triton_to_np_dtype",
    "serialize_byte_tensor",
    "deserialize_bytes_tensor",
    "serialize_bf16_tensor",
    "deserialize_bf16_tensor",
    "triton_dtype_byte_size",
    "SharedMemoryTensor",
]


class InferenceServerException(Exception):
    """Raised for every non-success outcome of a client call.

    Parameters
    ----------
    msg : str
        Human readable description.
    status : str, optional
        Transport level status (HTTP code or ``StatusCode.X``).
    debug_details : str, optional
        Extra payload useful when debugging (raw body, grpc debug string).
    """

    def __init__(self, msg, status=None, debug_details=None):
        super().__init__(msg)
        self._msg = msg
        self._status = status
        self._debug_details = debug_details

    def __str__(self):
        text = self._msg if self._msg is not None else super().__str__()
        if self._status is not None:
            text = "[" + self._status + "] " + text
        return text

    def message(self):
        """The message given at construction (may be ``None``)."""
        return self._msg

    def status(self):
        """The status code string (may be ``None``)."""
        return self._status

    def debug_details(self):
        """Extra details for debugging (may be ``None``)."""
        return self._debug_details


def raise_error(msg):
    """Raise :class:`InferenceServerException` carrying ``msg`` (no chaining)."""
    raise InferenceServerException(msg=msg) from None


# ---------------------------------------------------------------------------
# dtype tables
# ---------------------------------------------------------------------------
_TRITON_TO_NP = {
    "BOOL": bool,
    "INT8": np.int8,
    "INT16": np.int16,
    "INT32": np.int32,
    "INT64": np.int64,
    "UINT8": np.uint8,
    "UINT16": np.uint16,
    "UINT32": np.uint32,
    "UINT64": np.uint64,
    "FP16": np.float16,
    "FP32": np.float32,
    "FP64": np.float64,
    "BYTES": np.object_,
    # numpy has no bfloat16; the host-side carrier is float32
    "BF16": np.float32,
}

_NP_TO_TRITON = {
    np.dtype(v): k for k, v in _TRITON_TO_NP.items() if k not in ("BYTES", "BF16")
}

_FIXED_SIZE = {
    "BOOL": 1,
    "INT8": 1,
    "INT16": 2,
    "INT32": 4,
    "INT64": 8,
    "UINT8": 1,
    "UINT16": 2,
    "UINT32": 4,
    "UINT64": 8,
    "FP16": 2,
    "FP32": 4,
    "FP64": 8,
    "BF16": 2,
}


def np_to_triton_dtype(np_dtype):
    """Map a numpy dtype (or scalar type) to the protocol datatype string."""
    try:
        dt = np.dtype(np_dtype)
    except TypeError:
        return None
    name = _NP_TO_TRITON.get(dt)
    if name is not None:
        return name
    if dt == np.object_ or dt.type == np.bytes_:
        return "BYTES"
    return None


def triton_to_np_dtype(dtype):
    """Map a protocol datatype string to the numpy type used to carry it."""
    return _TRITON_TO_NP.get(dtype)


def triton_dtype_byte_size(dtype):
    """Bytes per element on the wire, or ``None`` for variable-size BYTES."""
    return _FIXED_SIZE.get(dtype)


# ---------------------------------------------------------------------------
# BYTES
# ---------------------------------------------------------------------------
try:  # optional accelerator (built by ``build_native.py``)
    from .._lib import _tca_pyext as _fast  # type: ignore
except Exception:  # pragma: no cover - pure python fallback
    _fast = None


def _as_bytes_items(tensor):
    """Row-major list of ``bytes`` payloads for an object/bytes_ ndarray."""
    flat = np.ascontiguousarray(tensor).reshape(-1) if tensor.flags.c_contiguous else tensor.flatten(order="C")
    if tensor.dtype == np.object_:
        out = []
        for item in flat.tolist():
            if type(item) is bytes:
                out.append(item)
            elif isinstance(item, (bytearray, memoryview, np.bytes_)):
                out.append(bytes(item))
            else:
                out.append(str(item).encode("utf-8"))
        return out
    # fixed-width np.bytes_ : numpy strips trailing NULs on item access
    return flat.tolist()


def serialized_byte_size(tensor_value):
    """Sum of payload lengths of an object tensor (length prefixes excluded)."""
    if tensor_value.dtype != np.object_:
        raise_error("The tensor_value dtype must be np.object_")
    if tensor_value.size == 0:
        return 0
    total = 0
    flat = tensor_value.reshape(-1) if tensor_value.flags.c_contiguous else tensor_value.flatten()
    for item in flat.tolist():
        total += len(item)
    return total


def _pack_items(items):
    if _fast is not None:
        return _fast.pack_bytes(items)
    lens = np.fromiter((len(b) for b in items), dtype="<u4", count=len(items))
    prefix = lens.tobytes()
    parts = [None] * (2 * len(items))
    parts[0::2] = [prefix[i : i + 4] for i in range(0, 4 * len(items), 4)]
    parts[1::2] = items
    return b"".join(parts)


def serialize_byte_tensor(input_tensor):
    """Pack a BYTES tensor into its wire form.

    Returns a 0-d ``object`` ndarray whose single item is the ``bytes`` blob
    (empty 1-d object array for an empty tensor) – the same carrier the
    reference returns (``utils/__init__.py:193-246``) so that
    ``.item()`` / ``.size`` based callers keep working.
    """
    if input_tensor.size == 0:
        return np.empty([0], dtype=np.object_)
    if input_tensor.dtype != np.object_ and input_tensor.dtype.type != np.bytes_:
        raise_error("cannot serialize bytes tensor: invalid datatype")
    blob = _pack_items(_as_bytes_items(input_tensor))
    carrier = np.empty((), dtype=np.object_)
    carrier[()] = blob
    return carrier


def deserialize_bytes_tensor(encoded_tensor):
    """Inverse of :func:`serialize_byte_tensor`: 1-d object array of ``bytes``."""
    buf = encoded_tensor
    if isinstance(buf, np.ndarray):
        buf = buf.tobytes()
    if _fast is not None:
        items = _fast.unpack_bytes(buf, -1)
    else:
        view = memoryview(buf).cast("B") if not isinstance(buf, (bytes, bytearray)) else buf
        items = []
        pos, end = 0, len(view)
        unpack = _struct.unpack_from
        while pos < end:
            (n,) = unpack("<I", view, pos)
            pos += 4
            if pos + n > end:
                raise_error("malformed BYTES tensor: element overruns the buffer")
            items.append(bytes(view[pos : pos + n]))
            pos += n
    out = np.empty(len(items), dtype=np.object_)
    for i, b in enumerate(items):
        out[i] = b
    return out


# ---------------------------------------------------------------------------
# BF16
# ---------------------------------------------------------------------------
def serialize_bf16_tensor(input_tensor):
    """fp32 ndarray -> BF16 wire bytes (keeps the upper 16 bits of each value).

    Carrier convention identical to :func:`serialize_byte_tensor`.
    Reference behaviour: ``utils/__init__.py:279-320`` (truncation).
    """
    if input_tensor.size == 0:
        return np.empty([0], dtype=np.object_)
    if input_tensor.dtype != np.float32:
        raise_error("cannot serialize bf16 tensor: invalid datatype")
    bits = np.ascontiguousarray(input_tensor).reshape(-1).view("<u4")
    blob = (bits >> 16).astype("<u2").tobytes()
    carrier = np.empty((), dtype=np.object_)
    carrier[()] = blob
    return carrier


def deserialize_bf16_tensor(encoded_tensor):
    """BF16 wire bytes -> 1-d fp32 ndarray (lower mantissa bits zero)."""
    half = np.frombuffer(encoded_tensor, dtype="<u2")
    return (half.astype("<u4") << 16).view(np.float32)
