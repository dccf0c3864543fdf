"""Requested-output description for the gRPC transport
(API parity: ``tritonclient/grpc/_requested_output.py``)."""

from ..utils import raise_error
from . import service_pb2


class InferRequestedOutput:
    def __init__(self, name, class_count=0):
        self._output = service_pb2.ModelInferRequest.InferRequestedOutputTensor()
        self._output.name = name
        if class_count != 0:
            self._output.parameters["classification"].int64_param = class_count

    def name(self):
        return self._output.name

    def set_shared_memory(self, region_name, byte_size, offset=0):
        """Have the server write this output into a registered shm region."""
        params = self._output.parameters
        if "classification" in params:
            raise_error("shared memory can't be set on classification output")
        params["shared_memory_region"].string_param = region_name
        params["shared_memory_byte_size"].int64_param = byte_size
        if offset != 0:
            params["shared_memory_offset"].int64_param = offset
        else:
            if "shared_memory_offset" in params:
                del params["shared_memory_offset"]

    def unset_shared_memory(self):
        for key in ("shared_memory_region", "shared_memory_byte_size", "shared_memory_offset"):
            if key in self._output.parameters:
                del self._output.parameters[key]

    def _get_tensor(self):
        return self._output
