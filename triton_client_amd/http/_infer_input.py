"""Input tensor description for the HTTP/REST transport.

API parity: ``tritonclient/http/_infer_input.py:38-272`` (constructor,
``name/datatype/shape/set_shape/set_data_from_numpy/set_shared_memory`` and the
validation messages).  Binary payloads are kept as zero-copy ``memoryview``s of
the caller's array where the dtype allows it; the transport gathers them with
``sendmsg`` instead of concatenating.
"""

import numpy as np

from ..utils import (
    np_to_triton_dtype,
    raise_error,
    serialize_bf16_tensor,
    serialize_byte_tensor,
    triton_to_np_dtype,
)

_SHM_KEYS = ("shared_memory_region", "shared_memory_byte_size", "shared_memory_offset")


def _check_array(datatype, shape, tensor):
    if not isinstance(tensor, np.ndarray):
        raise_error("input_tensor must be a numpy array")
    if datatype == "BF16":
        expected = triton_to_np_dtype(datatype)
        if tensor.dtype != expected:
            raise_error(
                "got unexpected datatype {} from numpy array, expected {} for BF16 type".format(tensor.dtype, expected)
            )
    else:
        got = np_to_triton_dtype(tensor.dtype)
        if got != datatype:
            raise_error("got unexpected datatype {} from numpy array, expected {}".format(got, datatype))
    if len(shape) != tensor.ndim or any(int(a) != int(b) for a, b in zip(shape, tensor.shape)):
        raise_error(
            "got unexpected numpy array shape [{}], expected [{}]".format(
                str(tensor.shape)[1:-1], str(list(shape))[1:-1]
            )
        )


def _utf8_items(tensor):
    """Row-major python strings of a BYTES tensor for JSON transport."""
    out = []
    current = None
    try:
        for current in tensor.flatten(order="C").tolist():
            if isinstance(current, (bytes, np.bytes_)):
                out.append(bytes(current).decode("utf-8"))
            else:
                out.append(str(current))
    except UnicodeDecodeError:
        raise_error(
            f'Failed to encode "{current}" using UTF-8. Please use binary_data=True, if'
            " you want to pass a byte array."
        )
    return out


class InferInput:
    """Describes one input tensor of an inference request.

    Parameters
    ----------
    name : str
    shape : list of int
    datatype : str
        Protocol datatype name, e.g. ``"FP16"``.
    """

    def __init__(self, name, shape, datatype):
        self._name = name
        self._shape = shape
        self._datatype = datatype
        self._parameters = {}
        self._data = None
        self._raw_data = None

    def name(self):
        return self._name

    def datatype(self):
        return self._datatype

    def shape(self):
        return self._shape

    def set_shape(self, shape):
        self._shape = shape
        return self

    def set_data_from_numpy(self, input_tensor, binary_data=True):
        """Attach tensor data.

        ``binary_data=True`` places the raw bytes after the JSON header;
        ``False`` embeds the values in the JSON (not possible for BF16).
        """
        _check_array(self._datatype, self._shape, input_tensor)
        for key in _SHM_KEYS:
            self._parameters.pop(key, None)

        if not binary_data:
            self._parameters.pop("binary_data_size", None)
            self._raw_data = None
            if self._datatype == "BF16":
                raise_error("BF16 inputs must be sent as binary data over HTTP. Please set binary_data=True")
            if self._datatype == "BYTES":
                self._data = _utf8_items(input_tensor) if input_tensor.size else []
            else:
                self._data = input_tensor.flatten(order="C").tolist()
            return self

        self._data = None
        if self._datatype == "BYTES":
            packed = serialize_byte_tensor(input_tensor)
            self._raw_data = packed.item() if packed.size else b""
        elif self._datatype == "BF16":
            packed = serialize_bf16_tensor(input_tensor)
            self._raw_data = packed.item() if packed.size else b""
        else:
            # snapshot semantics (the caller may reuse its array): one copy here,
            # none later – the transport sends this buffer in place.
            self._raw_data = input_tensor.tobytes()
        self._parameters["binary_data_size"] = len(self._raw_data)
        return self

    def set_shared_memory(self, region_name, byte_size, offset=0):
        """Read the tensor from a registered (system or device) shm region."""
        self._data = None
        self._raw_data = None
        self._parameters.pop("binary_data_size", None)
        self._parameters["shared_memory_region"] = region_name
        self._parameters["shared_memory_byte_size"] = byte_size
        if offset != 0:
            self._parameters["shared_memory_offset"] = offset
        else:
            self._parameters.pop("shared_memory_offset", None)
        return self

    # -- transport hooks --
    def _get_binary_data(self):
        return self._raw_data

    def _get_tensor(self):
        tensor = {"name": self._name, "shape": list(self._shape), "datatype": self._datatype}
        if self._parameters:
            tensor["parameters"] = self._parameters
        if self._raw_data is None and "shared_memory_region" not in self._parameters and self._data is not None:
            tensor["data"] = self._data
        return tensor
