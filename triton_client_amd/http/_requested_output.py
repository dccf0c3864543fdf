"""Requested-output description for the HTTP/REST transport.

API parity: ``tritonclient/http/_requested_output.py:31-117``.
"""

from ..utils import raise_error


class InferRequestedOutput:
    """Selects an output and how it travels back.

    Parameters
    ----------
    name : str
    binary_data : bool
        Raw bytes after the JSON header (default) or values inside the JSON.
    class_count : int
        When non-zero ask for the top-``class_count`` classification strings.
    """

    def __init__(self, name, binary_data=True, class_count=0):
        self._name = name
        self._binary = binary_data
        self._parameters = {}
        if class_count != 0:
            self._parameters["classification"] = class_count
        self._parameters["binary_data"] = binary_data

    def name(self):
        return self._name

    def set_shared_memory(self, region_name, byte_size, offset=0):
        """Have the server write this output into a registered shm region."""
        if "classification" in self._parameters:
            raise_error("shared memory can't be set on classification output")
        if self._binary:
            self._parameters["binary_data"] = False
        self._parameters["shared_memory_region"] = region_name
        self._parameters["shared_memory_byte_size"] = byte_size
        if offset != 0:
            self._parameters["shared_memory_offset"] = offset
        else:
            self._parameters.pop("shared_memory_offset", None)

    def unset_shared_memory(self):
        """Undo :meth:`set_shared_memory`."""
        self._parameters["binary_data"] = self._binary
        for key in ("shared_memory_region", "shared_memory_byte_size", "shared_memory_offset"):
            self._parameters.pop(key, None)

    def _get_tensor(self):
        tensor = {"name": self._name}
        if self._parameters:
            tensor["parameters"] = self._parameters
        return tensor
