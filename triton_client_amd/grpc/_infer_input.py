"""Input tensor description for the gRPC transport.

API parity: ``tritonclient/grpc/_infer_input.py`` (``set_data_from_numpy`` has no
``binary_data`` switch: data always travels in ``raw_input_contents``).
"""

import numpy as np

from ..utils import (
    np_to_triton_dtype,
    raise_error,
    serialize_bf16_tensor,
    serialize_byte_tensor,
    triton_to_np_dtype,
)
from . import service_pb2

_SHM_KEYS = ("shared_memory_region", "shared_memory_byte_size", "shared_memory_offset")


class InferInput:
    """Describes one input tensor of a gRPC inference request."""

    def __init__(self, name, shape, datatype):
        self._input = service_pb2.ModelInferRequest.InferInputTensor()
        self._input.name = name
        self._input.shape.extend(shape)
        self._input.datatype = datatype
        self._raw_content = None

    def name(self):
        return self._input.name

    def datatype(self):
        return self._input.datatype

    def shape(self):
        return list(self._input.shape)

    def set_shape(self, shape):
        del self._input.shape[:]
        self._input.shape.extend(shape)
        return self

    def set_data_from_numpy(self, input_tensor):
        """Attach tensor data (sent as one ``raw_input_contents`` entry)."""
        if not isinstance(input_tensor, np.ndarray):
            raise_error("input_tensor must be a numpy array")
        datatype = self._input.datatype
        if datatype == "BF16":
            expected = triton_to_np_dtype(datatype)
            if input_tensor.dtype != expected:
                raise_error(
                    "got unexpected datatype {} from numpy array, expected {} for BF16 type".format(
                        input_tensor.dtype, expected
                    )
                )
        else:
            got = np_to_triton_dtype(input_tensor.dtype)
            if got != datatype:
                raise_error("got unexpected datatype {} from numpy array, expected {}".format(got, datatype))
        shape = self._input.shape
        if len(shape) != input_tensor.ndim or any(a != b for a, b in zip(shape, input_tensor.shape)):
            raise_error(
                "got unexpected numpy array shape [{}], expected [{}]".format(
                    str(input_tensor.shape)[1:-1], str(list(shape))[1:-1]
                )
            )
        for key in _SHM_KEYS:
            if key in self._input.parameters:
                del self._input.parameters[key]
        self._input.ClearField("contents")
        if datatype == "BYTES":
            packed = serialize_byte_tensor(input_tensor)
            self._raw_content = packed.item() if packed.size else b""
        elif datatype == "BF16":
            packed = serialize_bf16_tensor(input_tensor)
            self._raw_content = packed.item() if packed.size else b""
        else:
            self._raw_content = input_tensor.tobytes()
        return self

    def set_shared_memory(self, region_name, byte_size, offset=0):
        """Read the tensor from a registered (system or device) shm region."""
        self._input.ClearField("contents")
        self._raw_content = None
        params = self._input.parameters
        params["shared_memory_region"].string_param = region_name
        params["shared_memory_byte_size"].int64_param = byte_size
        if offset != 0:
            params["shared_memory_offset"].int64_param = offset
        else:
            if "shared_memory_offset" in params:
                del params["shared_memory_offset"]
        return self

    def _get_tensor(self):
        return self._input

    def _get_content(self):
        return self._raw_content
